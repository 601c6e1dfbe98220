/* dp_host_forward.c - a host written in plain C against include/diffphore_amd.h: no Python, no torch, no engine.py.
 *
 * Reads a dump of one packed batch (the DpBatch arrays), the parameter blob and the model scalars, then
 *     dp_weights_bind -> dp_work_bytes -> hipMalloc -> dp_work_bind -> dp_work_init -> dp_prepare -> dp_forward
 * and writes tr / rot / tor.  tests/test_c_host_gpu.py compares them bit for bit with the Python host's result.
 *
 * With a third argument ("guard") it goes on like a sampler: dp_conformer_update with the scores, then the status words.  Bit 2 (non-finite scores
 * while the pose was fine: with the default two-limb fp16 kernels that is how a value beyond fp16's range surfaces, include/diffphore_amd.h) makes it
 * restore the start state, switch the workspace to the exact-fp32 kernels (DpWork.node_f16 = 0, edge_bf16x6 = 0) and run the forward again - what
 * diffphore_amd/sampling.py does with a flagged batch; bit 0 alone (a degenerate Kabsch alignment) does not.  It prints "replayed=0|1".
 *
 * Dump format: records of { char name[64]; int32 dtype (0 f32, 1 i32, 2 u8); int64 count; data }, names = DpBatch field names,
 * plus "dims" (i32[16]: B n_lig n_ph n_bond n_tor n_pp n_x ll_cap n_llp max_nl max_np max_t no_tail_nl no_tail_np n_xs ph_shared), "blob" (f32), "scalars" (f32[12]),
 * "precision" (i32[1]), "conv_algo" (i32[1]).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "diffphore_amd.h"

#define CHECK(x) do { int rc_ = (int)(x); if (rc_ != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, rc_, __LINE__); return 2; } } while (0)

typedef struct Rec { char name[64]; int32_t dtype; int64_t count; void* host; void* dev; } Rec;

static Rec recs[64];
static int nrec = 0;

static Rec* find(const char* name) {
  for (int i = 0; i < nrec; ++i)
    if (strcmp(recs[i].name, name) == 0) return &recs[i];
  fprintf(stderr, "record %s missing\n", name);
  exit(3);
}

int main(int argc, char** argv) {
  if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s dump.bin out.bin [guard]\n", argv[0]); return 1; }
  const int guard = argc == 4 && strcmp(argv[3], "guard") == 0;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("dump"); return 1; }
  while (nrec < 64) {
    Rec* r = &recs[nrec];
    if (fread(r->name, 1, 64, f) != 64) break;
    if (fread(&r->dtype, 4, 1, f) != 1 || fread(&r->count, 8, 1, f) != 1) return 1;
    const size_t esz = r->dtype == 2 ? 1 : 4, bytes = (size_t)r->count * esz;
    r->host = malloc(bytes ? bytes : 1);
    if (bytes && fread(r->host, 1, bytes, f) != bytes) return 1;
    CHECK(hipMalloc(&r->dev, bytes ? bytes : 16));
    if (bytes) CHECK(hipMemcpy(r->dev, r->host, bytes, hipMemcpyHostToDevice));
    ++nrec;
  }
  fclose(f);

  const int32_t* d = (const int32_t*)find("dims")->host;
  DpBatch b;
  memset(&b, 0, sizeof(b));
  b.B = d[0]; b.n_lig = d[1]; b.n_ph = d[2]; b.n_bond = d[3]; b.n_tor = d[4]; b.n_pp = d[5]; b.n_x = d[6]; b.ll_cap = d[7];
  b.n_llp = d[8]; b.max_nl = d[9]; b.max_np = d[10]; b.max_t = d[11];
  b.no_tail_nl = d[12]; b.no_tail_np = d[13];      /* hints: which node-side instantiation dp_forward may launch */
  b.n_xs = d[14];                                  /* the cross pairs with type agreement (the normals convs' listed pairs) */
  b.ph_shared = d[15];                             /* every pose on one pharmacophore (dp_pack_query's finding) */
#define I32(field) b.field = (const int32_t*)find(#field)->dev
#define F32(field) b.field = (const float*)find(#field)->dev
  I32(lig_ptr); I32(ph_ptr); I32(bond_ptr); I32(tor_ptr); I32(pp_ptr); I32(x_ptr); I32(ll_ptr); I32(mask_ptr); I32(pose_ptr); I32(llp_ptr);
  I32(lig_x); I32(bond_src); I32(bond_dst); I32(bond_type); I32(tor_u); I32(tor_v);
  b.mask_rotate = (const uint8_t*)find("mask_rotate")->dev;
  F32(phorefp); F32(norm_angle1); F32(norm_angle2); F32(ph_pos); F32(ph_norm); I32(ph_x); F32(ph_scalar); I32(ph_type);
  I32(pp_src); I32(pp_dst); I32(pp_rowptr); I32(pp_rowend);
  I32(xs_src); I32(xs_nbr); I32(xs_feat); I32(xs_rowptr); I32(xt_src); I32(xt_nbr); I32(xt_feat); I32(xt_rowptr);
  b.pos = (float*)find("pos")->dev;
  b.norm = (float*)find("norm")->dev;
  F32(t);

  const int32_t precision = *(const int32_t*)find("precision")->host;
  Rec* blob = find("blob");
  if (blob->count != dp_weights_blob_floats(precision)) { fprintf(stderr, "blob size %lld != library layout %lld\n", (long long)blob->count, (long long)dp_weights_blob_floats(precision)); return 4; }
  DpModelScalars sc;
  memcpy(&sc, find("scalars")->host, sizeof(sc));
  DpWeights w;
  CHECK(dp_weights_bind((const float*)blob->dev, precision, &sc, &w));

  const size_t wb = dp_work_bytes(&b);
  if (wb == 0) { fprintf(stderr, "dp_work_bytes rejected the batch\n"); return 4; }
  void* work = NULL;
  CHECK(hipMalloc(&work, wb));
  DpWork k;
  float *tr, *rot, *tor;
  CHECK(dp_work_bind(&b, work, wb, &k, &tr, &rot, &tor));
  k.conv_algo = *(const int32_t*)find("conv_algo")->host;
  {                                                /* this host wrote the times itself: it can state that every pose carries the same ones */
    const float* th = (const float*)find("t")->host;
    int uni = 1;
    for (int i = 1; i < b.B; ++i) uni &= th[3 * i] == th[0] && th[3 * i + 1] == th[1] && th[3 * i + 2] == th[2];
    k.t_uniform = uni;
  }
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  CHECK(dp_work_init(&b, &k, st));
  CHECK(dp_prepare(&b, &w, &k, st));
  CHECK(dp_forward(&b, &w, &k, tr, rot, tor, st));
  CHECK(hipStreamSynchronize(st));
  int replayed = 0;
  if (guard) {
    DpStepParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.tr_c1 = 0.05f; sp.rot_c1 = 0.02f; sp.tor_c1 = 0.03f;           /* a plain ODE-like step: the perturbation is c1 * score */
    CHECK(dp_conformer_update(&b, &sp, tr, rot, tor, NULL, NULL, NULL, k.status, st));
    int32_t* status = (int32_t*)malloc(sizeof(int32_t) * (size_t)b.B);
    CHECK(hipMemcpyAsync(status, k.status, sizeof(int32_t) * (size_t)b.B, hipMemcpyDeviceToHost, st));
    CHECK(hipStreamSynchronize(st));
    int over = 0, kabsch = 0;
    for (int i = 0; i < b.B; ++i) { over |= status[i] & 4; kabsch |= status[i] & 1; }
    /* the start state back (the update moved the poses), whatever happens next */
    CHECK(hipMemcpy(b.pos, find("pos")->host, sizeof(float) * 3 * (size_t)b.n_lig, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b.norm, find("norm")->host, sizeof(float) * 33 * (size_t)b.n_lig, hipMemcpyHostToDevice));
    if (over) {
      k.node_f16 = 0; k.edge_bf16x6 = 0;                              /* exact fp32 MFMA everywhere */
      CHECK(dp_work_init(&b, &k, st));
      CHECK(dp_prepare(&b, &w, &k, st));
      CHECK(dp_forward(&b, &w, &k, tr, rot, tor, st));
      CHECK(hipStreamSynchronize(st));
      replayed = 1;
    }
    printf("status: over_range=%d kabsch=%d replayed=%d\n", over ? 1 : 0, kabsch ? 1 : 0, replayed);
    free(status);
  }

  const int nt = b.n_tor > 0 ? b.n_tor : 0;
  float* out = (float*)malloc(sizeof(float) * (size_t)(6 * b.B + nt + 1));
  CHECK(hipMemcpy(out, tr, sizeof(float) * 3 * b.B, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(out + 3 * b.B, rot, sizeof(float) * 3 * b.B, hipMemcpyDeviceToHost));
  if (nt) CHECK(hipMemcpy(out + 6 * b.B, tor, sizeof(float) * nt, hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror("out"); return 1; }
  fwrite(out, sizeof(float), (size_t)(6 * b.B + nt), o);
  fclose(o);
  printf("dp_host_forward: B=%d n_tor=%d work=%zu bytes ok\n", b.B, nt, wb);
  return 0;
}
