// dp_update.hip - SDE/ODE perturbation + conformer update for every pose of the batch, device resident (gfx950).
//
// Replaces the per-sample Python loop of the reference:
//   sampling.py:223-254          tr/rot/tor perturbations from scores and noise
//   diffusion_utils.py:23-79     modify_conformer (rigid motion about the centroid, torsions, Kabsch re-alignment)
//   torsion.py:64-109            modify_conformer_torsion_angles (sequential over rotatable bonds, float64 rotation
//                                matrices applied to float32 coordinates, skip theta == 0)
//   geometry.py:6-85, 88-136     axis-angle -> matrix (quaternion route, float32), Kabsch
// The ligand normal buffer [N,33] is handled as the reference handles it: reinterpreted as [11,N,3], i.e. flat slot s
// is attached to atom (s mod N).
// One workgroup per pose; atoms / normal slots spread over 256 threads; bonds processed sequentially with barriers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "dp_dev.h"

namespace dp {

struct Mat3d { double m[3][3]; };

// scipy Rotation.from_rotvec(rv).as_matrix() in double
__device__ inline Mat3d rotvec_to_matrix(double x, double y, double z) {
  const double angle = sqrt(x * x + y * y + z * z);
  double scale;
  if (angle <= 1e-3) {
    const double a2 = angle * angle;
    scale = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
  } else {
    scale = sin(angle / 2.0) / angle;
  }
  const double qx = scale * x, qy = scale * y, qz = scale * z, qw = cos(angle / 2.0);
  const double x2 = qx * qx, y2 = qy * qy, z2 = qz * qz, w2 = qw * qw;
  const double xy = qx * qy, zw = qz * qw, xz = qx * qz, yw = qy * qw, yz = qy * qz, xw = qx * qw;
  Mat3d R;
  R.m[0][0] = x2 - y2 - z2 + w2; R.m[1][0] = 2 * (xy + zw);     R.m[2][0] = 2 * (xz - yw);
  R.m[0][1] = 2 * (xy - zw);     R.m[1][1] = -x2 + y2 - z2 + w2; R.m[2][1] = 2 * (yz + xw);
  R.m[0][2] = 2 * (xz + yw);     R.m[1][2] = 2 * (yz - xw);     R.m[2][2] = -x2 - y2 + z2 + w2;
  return R;
}

// geometry.py axis_angle_to_matrix (float32 throughout)
__device__ inline void axis_angle_to_matrix_f32(float ax, float ay, float az, float R[3][3]) {
  const float angle = sqrtf(ax * ax + ay * ay + az * az);
  const float half = 0.5f * angle;
  float s;
  if (fabsf(angle) < 1e-6f) s = 0.5f - (angle * angle) / 48.f;
  else s = sinf(half) / angle;
  const float r = cosf(half), i = ax * s, j = ay * s, k = az * s;
  const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  R[0][0] = 1 - two_s * (j * j + k * k); R[0][1] = two_s * (i * j - k * r);     R[0][2] = two_s * (i * k + j * r);
  R[1][0] = two_s * (i * j + k * r);     R[1][1] = 1 - two_s * (i * i + k * k); R[1][2] = two_s * (j * k - i * r);
  R[2][0] = two_s * (i * k - j * r);     R[2][1] = two_s * (j * k + i * r);     R[2][2] = 1 - two_s * (i * i + j * j);
}

// Optimal rotation for H = sum (a - ca)(b - cb)^T via cyclic Jacobi on H^T H (double). Returns false if degenerate/non-finite.
__device__ inline bool kabsch_rotation(const double H[3][3], double R[3][3]) {
  double A[3][3], V[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += H[k][i] * H[k][j];
      A[i][j] = s;
      V[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 24; ++sweep) {
    // converged: rotations by angles < 1e-22 leave every entry that matters unchanged in double precision.  (Without the relative test the
    // loop never stops early - rounding keeps the off-diagonal at ~1e-30 of the diagonal - and thread 0 spends 24 sweeps = ~64 rotations of
    // dependent fp64 divides and square roots where 3-4 sweeps converge: half of this kernel's time.)
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300 || off <= 1e-22 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {   // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {   // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  // sort eigenpairs descending
  int idx[3] = {0, 1, 2};
  double ev[3] = {A[0][0], A[1][1], A[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (ev[idx[j]] < ev[idx[j + 1]]) { const int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
  double Vs[3][3], U[3][3];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][idx[c]];
  // make V right-handed
  {
    const double det = Vs[0][0] * (Vs[1][1] * Vs[2][2] - Vs[1][2] * Vs[2][1]) - Vs[0][1] * (Vs[1][0] * Vs[2][2] - Vs[1][2] * Vs[2][0]) +
                       Vs[0][2] * (Vs[1][0] * Vs[2][1] - Vs[1][1] * Vs[2][0]);
    if (det < 0) for (int r = 0; r < 3; ++r) Vs[r][2] = -Vs[r][2];
  }
  const double s0 = sqrt(fmax(ev[idx[0]], 0.0));
  if (!(s0 > 0.0) || !isfinite(s0)) return false;
  int rank = 0;
  for (int c = 0; c < 3; ++c) {
    const double sc = sqrt(fmax(ev[idx[c]], 0.0));
    if (sc > 1e-10 * s0) {
      for (int r = 0; r < 3; ++r) U[r][c] = (H[r][0] * Vs[0][c] + H[r][1] * Vs[1][c] + H[r][2] * Vs[2][c]) / sc;
      rank = c + 1;
    } else break;
  }
  if (rank < 2) return false;
  // re-orthonormalise u0,u1 (guards against loss of orthogonality for close singular values)
  {
    double n0 = sqrt(U[0][0] * U[0][0] + U[1][0] * U[1][0] + U[2][0] * U[2][0]);
    for (int r = 0; r < 3; ++r) U[r][0] /= n0;
    const double d = U[0][0] * U[0][1] + U[1][0] * U[1][1] + U[2][0] * U[2][1];
    for (int r = 0; r < 3; ++r) U[r][1] -= d * U[r][0];
    double n1 = sqrt(U[0][1] * U[0][1] + U[1][1] * U[1][1] + U[2][1] * U[2][1]);
    for (int r = 0; r < 3; ++r) U[r][1] /= n1;
  }
  // third left vector from the cross product: det(U) = +1, det(V) = +1  => R = V U^T is a proper rotation.
  // This is the reference's reflection fix (geometry.py:129-132): the smallest singular direction absorbs the sign.
  U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
  U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
  U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      R[i][j] = Vs[i][0] * U[j][0] + Vs[i][1] * U[j][1] + Vs[i][2] * U[j][2];
      if (!isfinite(R[i][j])) return false;
    }
  return true;
}

// sum of v[c] over the 256 threads, every thread gets the result.  The pairing is the classic LDS tree (i += i + s for s = 128, 64, ..., 1), so
// the rounding of every sum is the same as that tree's; the last six levels run inside wave 0 on cross-lane moves instead of LDS + barrier
// (3 workgroup barriers per call instead of 9).  Whole workgroup must call it.
template <int NV>
__device__ inline void tree_sum(double (&v)[NV], double* s_red, int tid) {
#pragma unroll
  for (int c = 0; c < NV; ++c) s_red[tid * NV + c] = v[c];
  __syncthreads();
  if (tid < 128) {
#pragma unroll
    for (int c = 0; c < NV; ++c) s_red[tid * NV + c] += s_red[(tid + 128) * NV + c];
  }
  __syncthreads();
  if (tid < 64) {
    double x[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) x[c] = s_red[tid * NV + c] + s_red[(tid + 64) * NV + c];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
      for (int c = 0; c < NV; ++c) x[c] += __shfl_down(x[c], s, 64);      // lane i < s: x[i] + x[i + s], both from the level above
    }
    if (tid == 0) {
#pragma unroll
      for (int c = 0; c < NV; ++c) s_red[256 * NV - NV + c] = x[c];       // (slot 255: nobody reads it below level 128)
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NV; ++c) v[c] = s_red[256 * NV - NV + c];
  __syncthreads();                                   // s_red may be rewritten right away
}

// modify_conformer_torsion_angles (torsion.py:64-109) on LDS-resident coordinates; whole workgroup must call it.
// The bonds' atoms and the mask rows are staged in LDS first (`s_uv`: u | v << 16 per bond; `s_mask`: T x N bytes when they fit TMASK_CAP,
// else read in place): the loop over the bonds is sequential - every bond sees the positions the previous ones left - so a global load per
// bond and per mask row was a full memory latency, twice, on every bond's critical path.
constexpr int TMASK_CAP = 8192;
__device__ inline void stage_torsions(const DpBatch& b, int p, int l0, int N, int t0, int T, int* s_uv, uint8_t* s_mask) {
  const int tid = threadIdx.x;
  for (int i = tid; i < T; i += 256) s_uv[i] = (b.tor_u[t0 + i] - l0) | ((b.tor_v[t0 + i] - l0) << 16);
  if (T * N <= TMASK_CAP) {
    const uint8_t* mask = b.mask_rotate + b.mask_ptr[p];
    for (int i = tid; i < T * N; i += 256) s_mask[i] = mask[i];
  }
}
__device__ inline void apply_torsions(const DpBatch& b, int p, int l0, int N, int t0, int T, int NS3, float* s_pos, float* s_pts,
                                      const float* s_theta, const int* s_uv, const uint8_t* s_mask) {
  const int tid = threadIdx.x;
  (void)l0; (void)t0;
  const bool lds_mask = T * N <= TMASK_CAP;
  const uint8_t* gmask = b.mask_rotate + b.mask_ptr[p];
  for (int kb = 0; kb < T; ++kb) {
    const float theta = s_theta[kb];
    if (theta == 0.f) continue;                      // uniform across the workgroup
    const int uv = s_uv[kb];
    const int u = uv & 0xFFFF, v = uv >> 16;
    const float pv[3] = {s_pos[v * 3], s_pos[v * 3 + 1], s_pos[v * 3 + 2]};
    float rv[3] = {s_pos[u * 3] - pv[0], s_pos[u * 3 + 1] - pv[1], s_pos[u * 3 + 2] - pv[2]};
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rv[0], rv[0]), __fmul_rn(rv[1], rv[1])), __fmul_rn(rv[2], rv[2])));
    for (int c = 0; c < 3; ++c) rv[c] = __fdiv_rn(__fmul_rn(rv[c], theta), nrm);
    const Mat3d Rk = rotvec_to_matrix((double)rv[0], (double)rv[1], (double)rv[2]);
    __syncthreads();                                 // everyone has read pos[u], pos[v] before anyone writes
    const uint8_t* mk = (lds_mask ? s_mask : gmask) + (size_t)kb * N;
    for (int a = tid; a < N; a += 256)
      if (mk[a]) {
        const float d[3] = {s_pos[a * 3] - pv[0], s_pos[a * 3 + 1] - pv[1], s_pos[a * 3 + 2] - pv[2]};
        for (int c = 0; c < 3; ++c)
          s_pos[a * 3 + c] = (float)(((double)d[0] * Rk.m[c][0] + (double)d[1] * Rk.m[c][1] + (double)d[2] * Rk.m[c][2]) + (double)pv[c]);
      }
    for (int s = tid; s < NS3; s += 256)
      if (mk[s % N]) {
        const float d[3] = {s_pts[s * 3] - pv[0], s_pts[s * 3 + 1] - pv[1], s_pts[s * 3 + 2] - pv[2]};
        for (int c = 0; c < 3; ++c)
          s_pts[s * 3 + c] = (float)(((double)d[0] * Rk.m[c][0] + (double)d[1] * Rk.m[c][1] + (double)d[2] * Rk.m[c][2]) + (double)pv[c]);
      }
    __syncthreads();
  }
}

// randomize_position (sampling.py:16-63): random torsions (no re-alignment), random rotation about the centroid with the
// centroid REMOVED (the ligand lands on the pharmacophore centroid = origin), then a N(0, tr_sigma_max) translation.
// Random numbers are drawn by the host exactly like the reference does and passed in.
__global__ __launch_bounds__(256) void randomize_kernel(DpBatch b, const float* __restrict__ theta, const float* __restrict__ Rm,
                                                        const float* __restrict__ trv, int no_torsion) {
  __shared__ float s_pos[MAXN_L * 3];
  __shared__ float s_pts[MAXN_L * 33];
  __shared__ float s_theta[MAXT];
  __shared__ float s_c[3];
  __shared__ int s_uv[MAXT];
  __shared__ uint8_t s_mask[TMASK_CAP];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], N = b.lig_ptr[p + 1] - l0;
  const int t0 = b.tor_ptr[p], T = no_torsion ? 0 : b.tor_ptr[p + 1] - t0;
  const int NS3 = 11 * N;
  for (int i = tid; i < N * 3; i += 256) s_pos[i] = b.pos[(size_t)l0 * 3 + i];
  for (int i = tid; i < T; i += 256) s_theta[i] = theta[t0 + i];
  stage_torsions(b, p, l0, N, t0, T, s_uv, s_mask);
  __syncthreads();
  for (int s = tid; s < NS3; s += 256)
    for (int c = 0; c < 3; ++c) s_pts[s * 3 + c] = b.norm[(size_t)l0 * 33 + s * 3 + c] + s_pos[(s % N) * 3 + c];
  __syncthreads();
  if (T > 0) apply_torsions(b, p, l0, N, t0, T, NS3, s_pos, s_pts, s_theta, s_uv, s_mask);
  __syncthreads();
  if (tid == 0) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = 0; i < N; ++i) { sx += s_pos[i * 3]; sy += s_pos[i * 3 + 1]; sz += s_pos[i * 3 + 2]; }
    s_c[0] = sx / (float)N; s_c[1] = sy / (float)N; s_c[2] = sz / (float)N;
  }
  __syncthreads();
  const float* R = Rm + (size_t)p * 9;
  for (int s = tid; s < NS3; s += 256) {
    const float q[3] = {s_pts[s * 3] - s_c[0], s_pts[s * 3 + 1] - s_c[1], s_pts[s * 3 + 2] - s_c[2]};
    const int a = s % N;
    const float w[3] = {s_pos[a * 3] - s_c[0], s_pos[a * 3 + 1] - s_c[1], s_pos[a * 3 + 2] - s_c[2]};
    for (int c = 0; c < 3; ++c) {
      const float pt = q[0] * R[c * 3] + q[1] * R[c * 3 + 1] + q[2] * R[c * 3 + 2];
      const float pa = w[0] * R[c * 3] + w[1] * R[c * 3 + 1] + w[2] * R[c * 3 + 2];
      b.norm[(size_t)l0 * 33 + s * 3 + c] = pt - pa;
    }
  }
  for (int a = tid; a < N; a += 256) {
    const float w[3] = {s_pos[a * 3] - s_c[0], s_pos[a * 3 + 1] - s_c[1], s_pos[a * 3 + 2] - s_c[2]};
    for (int c = 0; c < 3; ++c)
      b.pos[(size_t)(l0 + a) * 3 + c] = (w[0] * R[c * 3] + w[1] * R[c * 3 + 1] + w[2] * R[c * 3 + 2]) + (trv ? trv[p * 3 + c] : 0.f);
  }
}

__global__ __launch_bounds__(256) void conformer_update_kernel(DpBatch b, DpStepParams sp, const float* __restrict__ tr,
                                                               const float* __restrict__ rot, const float* __restrict__ tor,
                                                               const float* __restrict__ z_tr, const float* __restrict__ z_rot,
                                                               const float* __restrict__ z_tor, int32_t* __restrict__ status) {
  __shared__ float s_pos[MAXN_L * 3];        // current (rigid -> flexible) positions
  __shared__ float s_rig[MAXN_L * 3];        // rigid positions (Kabsch target)
  __shared__ float s_pts[MAXN_L * 33];       // normal "points", flat slot s <-> atom s % N
  __shared__ float s_theta[MAXT];
  __shared__ double s_R[3][3], s_t[3];
  __shared__ float s_Rf[3][3], s_c[3], s_trp[3];
  __shared__ double s_red[256 * 9];
  __shared__ int s_flag, s_sbad;
  __shared__ int s_uv[MAXT];
  __shared__ uint8_t s_mask[TMASK_CAP];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], N = b.lig_ptr[p + 1] - l0;
  const int t0 = b.tor_ptr[p], T = sp.no_torsion ? 0 : b.tor_ptr[p + 1] - t0;
  const int NS3 = 11 * N;
  for (int i = tid; i < N * 3; i += 256) s_pos[i] = b.pos[(size_t)l0 * 3 + i];
  for (int i = tid; i < T; i += 256) s_theta[i] = sp.tor_c1 * tor[t0 + i] + (z_tor ? sp.tor_c2 * z_tor[t0 + i] : 0.f);
  stage_torsions(b, p, l0, N, t0, T, s_uv, s_mask);
  if (tid == 0) {
    s_flag = 0;
    // a non-finite SCORE (status bit 2): not a failure of this update - the score model left the arithmetic it was run in
    int bad = 0;
    for (int c = 0; c < 3; ++c) bad |= (!isfinite(tr[p * 3 + c]) || !isfinite(rot[p * 3 + c])) ? 1 : 0;
    s_sbad = bad;
  }
  __syncthreads();
  for (int i = tid; i < T; i += 256)
    if (!isfinite(tor[t0 + i])) s_sbad = 1;          // (racing writers store the same value)
  if (tid == 0) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = 0; i < N; ++i) { sx += s_pos[i * 3]; sy += s_pos[i * 3 + 1]; sz += s_pos[i * 3 + 2]; }
    s_c[0] = sx / (float)N; s_c[1] = sy / (float)N; s_c[2] = sz / (float)N;
    float rp[3];
    for (int c = 0; c < 3; ++c) {
      s_trp[c] = sp.tr_c1 * tr[p * 3 + c] + (z_tr ? sp.tr_c2 * z_tr[p * 3 + c] : 0.f);
      rp[c] = sp.rot_c1 * rot[p * 3 + c] + (z_rot ? sp.rot_c2 * z_rot[p * 3 + c] : 0.f);
    }
    float R[3][3];
    axis_angle_to_matrix_f32(rp[0], rp[1], rp[2], R);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s_Rf[i][j] = R[i][j];
  }
  __syncthreads();
  // normal points = vec + pos[s % N]; rigid transform of points and atoms: (x - c) R^T + tr + c
  for (int s = tid; s < NS3; s += 256) {
    const int a = s % N;
    float q[3];
    for (int c = 0; c < 3; ++c) q[c] = (b.norm[(size_t)l0 * 33 + s * 3 + c] + s_pos[a * 3 + c]) - s_c[c];
    for (int c = 0; c < 3; ++c)
      s_pts[s * 3 + c] = ((q[0] * s_Rf[c][0] + q[1] * s_Rf[c][1] + q[2] * s_Rf[c][2]) + s_trp[c]) + s_c[c];
  }
  __syncthreads();
  for (int a = tid; a < N; a += 256) {
    float q[3];
    for (int c = 0; c < 3; ++c) q[c] = s_pos[a * 3 + c] - s_c[c];
    for (int c = 0; c < 3; ++c) {
      const float v = ((q[0] * s_Rf[c][0] + q[1] * s_Rf[c][1] + q[2] * s_Rf[c][2]) + s_trp[c]) + s_c[c];
      s_rig[a * 3 + c] = v;
    }
  }
  __syncthreads();
  for (int i = tid; i < N * 3; i += 256) s_pos[i] = s_rig[i];
  __syncthreads();
  if (T > 0) {
    apply_torsions(b, p, l0, N, t0, T, NS3, s_pos, s_pts, s_theta, s_uv, s_mask);
    // Kabsch: align flexible (s_pos) onto rigid (s_rig)
    {
      double acc[6] = {0, 0, 0, 0, 0, 0};
      for (int a = tid; a < N; a += 256)
        for (int c = 0; c < 3; ++c) { acc[c] += s_pos[a * 3 + c]; acc[3 + c] += s_rig[a * 3 + c]; }
      // centroid reduction (fixed tree order -> deterministic): both centroids in one tree
      tree_sum<6>(acc, s_red, tid);
      const double cA[3] = {acc[0] / N, acc[1] / N, acc[2] / N};
      const double cB[3] = {acc[3] / N, acc[4] / N, acc[5] / N};
      double H[3][3];
      {
        double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = tid; a < N; a += 256)
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) h[i * 3 + j] += ((double)s_pos[a * 3 + i] - cA[i]) * ((double)s_rig[a * 3 + j] - cB[j]);
        tree_sum<9>(h, s_red, tid);
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) H[i][j] = h[i * 3 + j];
      }
      if (tid == 0) {
        double R[3][3];
        const bool ok = kabsch_rotation(H, R);
        if (!ok) {
          s_flag = 1;
          for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = i == j ? 1.0 : 0.0;
        }
        for (int i = 0; i < 3; ++i) {
          for (int j = 0; j < 3; ++j) s_R[i][j] = R[i][j];
          s_t[i] = -(R[i][0] * cA[0] + R[i][1] * cA[1] + R[i][2] * cA[2]) + cB[i];     // t = -R cA + cB
        }
      }
      __syncthreads();
      for (int a = tid; a < N; a += 256) {
        const double x = s_pos[a * 3], y = s_pos[a * 3 + 1], z = s_pos[a * 3 + 2];
        for (int c = 0; c < 3; ++c) s_rig[a * 3 + c] = (float)(x * s_R[c][0] + y * s_R[c][1] + z * s_R[c][2] + s_t[c]);   // aligned positions
      }
      __syncthreads();
      for (int s = tid; s < NS3; s += 256) {
        const double x = s_pts[s * 3], y = s_pts[s * 3 + 1], z = s_pts[s * 3 + 2];
        const int a = s % N;
        for (int c = 0; c < 3; ++c) {
          const float q = (float)(x * s_R[c][0] + y * s_R[c][1] + z * s_R[c][2] + s_t[c]);
          b.norm[(size_t)l0 * 33 + s * 3 + c] = q - s_rig[a * 3 + c];
        }
      }
      for (int i = tid; i < N * 3; i += 256) {
        const float v = s_rig[i];
        b.pos[(size_t)l0 * 3 + i] = v;
        if (!isfinite(v)) s_flag = 1;
      }
    }
  } else {
    for (int s = tid; s < NS3; s += 256)
      for (int c = 0; c < 3; ++c) b.norm[(size_t)l0 * 33 + s * 3 + c] = s_pts[s * 3 + c] - s_rig[(s % N) * 3 + c];
    for (int i = tid; i < N * 3; i += 256) {
      const float v = s_rig[i];
      b.pos[(size_t)l0 * 3 + i] = v;
      if (!isfinite(v)) s_flag = 1;
    }
  }
  __syncthreads();
  // sticky across the steps of a run (the caller zeroes it); the FIRST cause is kept: bit 2 = the scores of this step were non-finite while the pose was
  // still fine (with the two-limb fp16 kernels: a value beyond fp16's range - run the batch again with DpWork.node_f16 = 0, edge_bf16x6 = 0; with the
  // exact kernels: the model itself diverged), bit 0 = finite scores, but the update produced a non-finite pose (degenerate Kabsch alignment)
  if (tid == 0 && status) {
    const int cur = status[p];
    const int add = (cur & 5) ? 0 : (s_sbad ? 4 : (s_flag ? 1 : 0));
    if (add) status[p] = cur | add;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Gaussian-volume pharmacophore-fit SURROGATE for ranking candidate poses on the device.
// NOT AncPhore (closed binary, absent): same Gaussian overlap kernel and constants as the reference's own type/count similarity
// (`get_perfect_similarity`, src/inference.py:273-295: vol = 7.999999999 (pi/(a1+a2))^1.5 exp(-a1 a2 r^2/(a1+a2))), evaluated
// on the actual geometry.  Per pose:
//     V    = sum over features m (non-EX) of w_m * max over ligand atoms n with phorefp[n, type_m] of G(a_m, a_m, |r_nm|^2)
//     Vref = sum over features m of w_m * G(a_m, a_m, 0)
//     X    = sum over exclusion spheres m, atoms n of exp(-a_m |r_nm|^2 / 2)
//     score = V / Vref - ex_lambda * X / N_l          (-1 when the model has no weighted feature)
// One workgroup per pose, one thread per pharmacophore node, fixed-order reduction: reruns are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fit_overlap_kernel(DpBatch b, const float* pos_in, float ex_lambda, float* score) {
  __shared__ float s_pos[MAXN_L * 3];
  __shared__ float s_v[MAXN_P], s_ref[MAXN_P], s_x[MAXN_P];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], nl = b.lig_ptr[p + 1] - l0, p0 = b.ph_ptr[p], np_ = b.ph_ptr[p + 1] - p0;
  const float* pos = pos_in ? pos_in : b.pos;
  for (int i = tid; i < nl * 3; i += 256) s_pos[i] = pos[(size_t)l0 * 3 + i];
  __syncthreads();
  float v = 0.f, ref = 0.f, x = 0.f;
  if (tid < np_) {
    const int m = p0 + tid, type = b.ph_type[m];
    const float alpha = b.ph_scalar[m * 2], w = b.ph_scalar[m * 2 + 1];
    const float px = b.ph_pos[m * 3], py = b.ph_pos[m * 3 + 1], pz = b.ph_pos[m * 3 + 2];
    if (type != 10) {
      const float g0 = 7.999999999f * powf(3.14159265358979323846f / (2.f * alpha), 1.5f);
      float best = 0.f;
      for (int n = 0; n < nl; ++n) {
        if (b.phorefp[(size_t)(l0 + n) * 11 + type] != 0.f) {
          const float dx = s_pos[n * 3] - px, dy = s_pos[n * 3 + 1] - py, dz = s_pos[n * 3 + 2] - pz;
          best = fmaxf(best, expf(-0.5f * alpha * (dx * dx + dy * dy + dz * dz)));
        }
      }
      v = w * g0 * best;
      ref = w * g0;
    } else {
      for (int n = 0; n < nl; ++n) {
        const float dx = s_pos[n * 3] - px, dy = s_pos[n * 3 + 1] - py, dz = s_pos[n * 3 + 2] - pz;
        x += expf(-0.5f * alpha * (dx * dx + dy * dy + dz * dz));
      }
    }
  }
  s_v[tid] = v;
  s_ref[tid] = ref;
  s_x[tid] = x;
  __syncthreads();
  if (tid == 0) {
    float V = 0.f, R = 0.f, X = 0.f;
    for (int m = 0; m < np_; ++m) {
      V += s_v[m];
      R += s_ref[m];
      X += s_x[m];
    }
    score[p] = (R > 0.f ? V / R : -1.f) - ex_lambda * X / (float)(nl > 0 ? nl : 1);
  }
}

}  // namespace dp

extern "C" {

int dp_conformer_update(const DpBatch* b, const DpStepParams* sp, const float* tr, const float* rot, const float* tor,
                        const float* z_tr, const float* z_rot, const float* z_tor, int32_t* status, void* stream) {
  if (!b || !sp || !tr || !rot || b->B <= 0) return DP_E_BADARG;
  if (b->max_nl > dp::MAXN_L || b->max_t > dp::MAXT) return DP_E_TOOLARGE;
  if (!sp->no_torsion && b->n_tor > 0 && !tor) return DP_E_BADARG;
  hipLaunchKernelGGL(dp::conformer_update_kernel, dim3(b->B), dim3(256), 0, (hipStream_t)stream, *b, *sp, tr, rot, tor, z_tr, z_rot,
                     z_tor, status);
  return (int)hipGetLastError();
}

int dp_randomize_position(const DpBatch* b, const float* theta, const float* rot_mats, const float* tr, int32_t no_torsion, void* stream) {
  if (!b || b->B <= 0 || !rot_mats) return DP_E_BADARG;
  if (b->max_nl > dp::MAXN_L || b->max_t > dp::MAXT) return DP_E_TOOLARGE;
  if (!no_torsion && b->n_tor > 0 && !theta) return DP_E_BADARG;
  hipLaunchKernelGGL(dp::randomize_kernel, dim3(b->B), dim3(256), 0, (hipStream_t)stream, *b, theta, rot_mats, tr, (int)no_torsion);
  return (int)hipGetLastError();
}

int dp_sample_step(const DpBatch* b, const DpWeights* w, DpWork* k, const DpStepParams* sp, float* tr, float* rot, float* tor,
                   const float* z_tr, const float* z_rot, const float* z_tor, void* stream) {
  int rc = dp_forward(b, w, k, tr, rot, tor, stream);
  if (rc) return rc;
  return dp_conformer_update(b, sp, tr, rot, tor, z_tr, z_rot, z_tor, k ? k->status : nullptr, stream);
}


int dp_fit_overlap(const DpBatch* b, const float* pos, float ex_lambda, float* score, void* stream) {
  if (!b || b->B <= 0 || !score) return DP_E_BADARG;
  if (b->max_nl > dp::MAXN_L || b->max_np > dp::MAXN_P) return DP_E_TOOLARGE;
  hipLaunchKernelGGL(dp::fit_overlap_kernel, dim3(b->B), dim3(256), 0, (hipStream_t)stream, *b, pos, ex_lambda, score);
  return (int)hipGetLastError();
}

}  // extern "C"

// sizeof() of every ABI struct, for the host binding's self-check
extern "C" int dp_struct_sizes(int32_t* out, int32_t n) {
  const int32_t v[7] = {(int32_t)sizeof(DpConvArgs), (int32_t)sizeof(DpBatch), (int32_t)sizeof(DpWeights), (int32_t)sizeof(DpWork),
                        (int32_t)sizeof(DpStepParams), (int32_t)sizeof(DpMlp), (int32_t)sizeof(DpSchedule)};
  for (int i = 0; i < n && i < 7; ++i) out[i] = v[i];
  return 7;
}
