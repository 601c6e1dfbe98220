// dp_rng.hip - on-device random draws of the sampler (`dp_sample(..., noise_or_null, seed, ...)`).
//
// The reference draws its noise on the host from the global generators, inside the loop: per reverse step and per `batch_size` chunk
// z_tr, z_rot ~ N(0, 1) [b, 3] and z_tor ~ N(0, 1) [sum T] (src/utils/sampling.py:230-243), and before the loop the start randomisation
// of every complex - torsions U(-pi, pi), a uniform random rotation, a N(0, tr_sigma_max) translation (sampling.py:16-63).  A screening
// run over 10^5 ligands x 40 samples x 20 steps would draw and upload S x N x (6 + T) normals from one Python thread.  Here every number
// is a pure function of (seed, pose uid, step, component): Philox-4x32-10 (Salmon et al., SC'11), key = seed, counter =
// (pose uid, step, block, stream), so a pose's trajectory does not depend on which rank, device batch or position inside a batch it is
// processed in.  The pre-drawn tables of DpSchedule stay the parity path (identical noise for the CPU oracle and this library).
//
//   stream 0  step noise, translation / rotation: block 0 -> z_tr[0..2], z_rot[0]; block 1 -> z_rot[1..2]
//   stream 1  step noise, torsions: torsion j of the pose -> block j / 4, lane j % 4
//   stream 2  start randomisation: block 0 -> rotation (4 normals -> unit quaternion, uniform on SO(3) like scipy's Rotation.random),
//             block 1 -> translation (3 normals); the start step is numbered DP_RNG_STEP_START
//   stream 3  start randomisation, torsion angles: torsion j -> block j / 4, lane j % 4 (uniform in (-pi, pi))
// Normals: Box-Muller on the (0, 1) uniforms ((x >> 9) + 0.5) * 2^-23 of a block's four words (two pairs -> four normals).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffphore_amd.h"

namespace dpr {

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

// Philox-4x32, 10 rounds
__host__ __device__ inline U4 philox4x32(U4 ctr, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(M0, ctr.x), lo0 = M0 * ctr.x, hi1 = mulhi32(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = U4{hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
    k0 += W0;
    k1 += W1;
  }
  return ctr;
}

// uniform in the OPEN interval (0, 1) from the top 23 bits of a word: (k + 1/2) 2^-23, k < 2^23 - every value is exactly representable in fp32
// (24 significand bits hold 2 k + 1), the smallest is 2^-24 and the largest 1 - 2^-24, so log() never sees 0 and the angle never reaches 2 pi.
// (ABI 8 used the top 24 bits, (k + 1/2) 2^-24: for k >= 2^23 the sum rounds to even - a half-ulp bias on half of the draws, and 1.0 for k = 2^24 - 1.)
__device__ inline float unit_open(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

struct F4 { float v[4]; };

__device__ inline F4 normals4(uint64_t seed, uint32_t uid, uint32_t step, uint32_t block, uint32_t stream) {
  const U4 r = philox4x32(U4{uid, step, block, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
  F4 o;
  const float r0 = sqrtf(-2.0f * logf(unit_open(r.x))), r1 = sqrtf(-2.0f * logf(unit_open(r.z)));
  float s0, c0, s1, c1;
  sincosf(6.28318530717958647692f * unit_open(r.y), &s0, &c0);
  sincosf(6.28318530717958647692f * unit_open(r.w), &s1, &c1);
  o.v[0] = r0 * c0; o.v[1] = r0 * s0; o.v[2] = r1 * c1; o.v[3] = r1 * s1;
  return o;
}

__device__ inline uint32_t uid_of(const int32_t* pose_uid, int32_t uid_base, int p) { return (uint32_t)(pose_uid ? pose_uid[p] : uid_base + p); }

// one thread per (pose, block): blocks 0 / 1 = translation + rotation, blocks 2.. = the pose's torsions in fours
__global__ void step_noise_kernel(DpBatch b, uint64_t seed, int32_t step, const int32_t* __restrict__ pose_uid, int32_t uid_base,
                                  float* __restrict__ z_tr, float* __restrict__ z_rot, float* __restrict__ z_tor, int blocks_per_pose) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.B * blocks_per_pose) return;
  const int p = i / blocks_per_pose, blk = i % blocks_per_pose;
  const uint32_t uid = uid_of(pose_uid, uid_base, p);
  if (blk < 2) {
    const F4 n = normals4(seed, uid, (uint32_t)step, (uint32_t)blk, 0u);
    if (blk == 0) {
      if (z_tr) { z_tr[p * 3] = n.v[0]; z_tr[p * 3 + 1] = n.v[1]; z_tr[p * 3 + 2] = n.v[2]; }
      if (z_rot) z_rot[p * 3] = n.v[3];
    } else if (z_rot) {
      z_rot[p * 3 + 1] = n.v[0]; z_rot[p * 3 + 2] = n.v[1];
    }
    return;
  }
  if (!z_tor) return;
  const int t0 = b.tor_ptr[p], T = b.tor_ptr[p + 1] - t0, j0 = 4 * (blk - 2);
  if (j0 >= T) return;
  const F4 n = normals4(seed, uid, (uint32_t)step, (uint32_t)(blk - 2), 1u);
  for (int l = 0; l < 4 && j0 + l < T; ++l) z_tor[t0 + j0 + l] = n.v[l];
}

__global__ void start_draw_kernel(DpBatch b, uint64_t seed, const int32_t* __restrict__ pose_uid, int32_t uid_base, float tr_sigma_max,
                                  float* __restrict__ theta, float* __restrict__ rot_mats, float* __restrict__ tr, int blocks_per_pose) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.B * blocks_per_pose) return;
  const int p = i / blocks_per_pose, blk = i % blocks_per_pose;
  const uint32_t uid = uid_of(pose_uid, uid_base, p);
  if (blk == 0) {
    if (!rot_mats) return;
    // unit quaternion from four normals (scipy Rotation.random: normalised Gaussian quaternion), then its rotation matrix
    const F4 n = normals4(seed, uid, DP_RNG_STEP_START, 0u, 2u);
    const float inv = rsqrtf(n.v[0] * n.v[0] + n.v[1] * n.v[1] + n.v[2] * n.v[2] + n.v[3] * n.v[3]);
    const float x = n.v[0] * inv, y = n.v[1] * inv, z = n.v[2] * inv, w = n.v[3] * inv;
    float* m = rot_mats + (size_t)p * 9;
    m[0] = 1.f - 2.f * (y * y + z * z); m[1] = 2.f * (x * y - z * w);       m[2] = 2.f * (x * z + y * w);
    m[3] = 2.f * (x * y + z * w);       m[4] = 1.f - 2.f * (x * x + z * z); m[5] = 2.f * (y * z - x * w);
    m[6] = 2.f * (x * z - y * w);       m[7] = 2.f * (y * z + x * w);       m[8] = 1.f - 2.f * (x * x + y * y);
    return;
  }
  if (blk == 1) {
    if (!tr) return;
    const F4 n = normals4(seed, uid, DP_RNG_STEP_START, 1u, 2u);
    for (int c = 0; c < 3; ++c) tr[p * 3 + c] = tr_sigma_max * n.v[c];
    return;
  }
  if (!theta) return;
  const int t0 = b.tor_ptr[p], T = b.tor_ptr[p + 1] - t0, j0 = 4 * (blk - 2);
  if (j0 >= T) return;
  const U4 r = philox4x32(U4{uid, DP_RNG_STEP_START, (uint32_t)(blk - 2), 3u}, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
  for (int l = 0; l < 4 && j0 + l < T; ++l) theta[t0 + j0 + l] = (2.0f * unit_open(w[l]) - 1.0f) * 3.14159265358979323846f;
}

}  // namespace dpr

extern "C" {

int dp_noise_normals(uint64_t seed, int32_t step, const DpBatch* b, const int32_t* pose_uid, int32_t uid_base, float* z_tr, float* z_rot,
                     float* z_tor, void* stream) {
  if (!b || b->B <= 0 || step < 0 || (uint32_t)step == DP_RNG_STEP_START) return DP_E_BADARG;
  if (!z_tr && !z_rot && !z_tor) return DP_OK;
  const int bpp = 2 + (z_tor ? (b->max_t + 3) / 4 : 0);
  const long long n = (long long)b->B * bpp;
  hipLaunchKernelGGL(dpr::step_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *b, seed, step, pose_uid,
                     uid_base, z_tr, z_rot, z_tor, bpp);
  return (int)hipGetLastError();
}

int dp_randomize_draws(uint64_t seed, const DpBatch* b, const int32_t* pose_uid, int32_t uid_base, float tr_sigma_max, float* theta,
                       float* rot_mats, float* tr, void* stream) {
  if (!b || b->B <= 0) return DP_E_BADARG;
  if (!theta && !rot_mats && !tr) return DP_OK;
  const int bpp = 2 + (theta ? (b->max_t + 3) / 4 : 0);
  const long long n = (long long)b->B * bpp;
  hipLaunchKernelGGL(dpr::start_draw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *b, seed, pose_uid, uid_base,
                     tr_sigma_max, theta, rot_mats, tr, bpp);
  return (int)hipGetLastError();
}

// host evaluation of the generator (tests: known-answer vectors of Philox-4x32-10, and the exact words a device draw is made of)
int dp_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  if (!ctr || !key || !out) return DP_E_BADARG;
  const dpr::U4 r = dpr::philox4x32(dpr::U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
  return DP_OK;
}

}  // extern "C"
