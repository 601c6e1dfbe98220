#!/usr/bin/env python3
"""Known-answer vectors of the on-device random stream (dp_noise_normals / dp_randomize_draws, csrc/dp_rng.hip) -> rng_stream.npz.

A pose's noise is documented as a pure function of (seed, pose uid, step, component); users who pin a seed rely on that function not
changing between versions.  Run on an MI355X box (`python tests/golden/make_rng_golden.py`); the fixture holds
    * the Philox-4x32-10 words of every (uid, step, block, stream) counter used below (host entry point dp_philox4x32: exact integers,
      re-checked on CPU by tests/test_abi_cpu.py), and
    * the device's normals / uniform angles / rotation matrices for those counters (re-checked on the GPU to 2e-6: the device's logf /
      sincosf are not bit-specified).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffphore_amd import _lib, synthetic  # noqa: E402
from diffphore_amd.engine import PackedBatch, _declare, _stream  # noqa: E402

SEED = 20250321
UIDS = [0, 1, 77, 2 ** 31 - 1]
STEPS = [0, 5, 19]


def words(seed, uid, step, block, stream):
    L = _lib.lib()
    U4, U2 = C.c_uint32 * 4, C.c_uint32 * 2
    L.dp_philox4x32.argtypes = [U4, U2, U4]
    out = U4()
    assert L.dp_philox4x32(U4(uid & 0xFFFFFFFF, step & 0xFFFFFFFF, block, stream), U2(seed & 0xFFFFFFFF, seed >> 32), out) == 0
    return list(out)


def main():
    gs = synthetic.synthetic_pairs(4242, len(UIDS), 16, 7, 9)           # 7 torsions per pose: blocks 0 and 1 of the torsion streams
    assert all(int(g["ligand"].edge_mask.sum()) == 7 for g in gs)
    pb = PackedBatch(gs, "cuda")
    L = _declare()
    uid = torch.tensor(UIDS, dtype=torch.int32, device="cuda")
    B, nT = pb.B, int(pb.tor_ptr[-1])
    out = dict(seed=np.uint64(SEED), uids=np.asarray(UIDS, dtype=np.int64), steps=np.asarray(STEPS, dtype=np.int64), n_tor=np.int64(7))
    for st in STEPS:
        ztr, zrot, ztor = torch.empty(B, 3, device="cuda"), torch.empty(B, 3, device="cuda"), torch.empty(nT, device="cuda")
        _lib.check(L.dp_noise_normals(SEED, st, C.byref(pb.struct), uid.data_ptr(), 0, ztr.data_ptr(), zrot.data_ptr(), ztor.data_ptr(),
                                      _stream()), "dp_noise_normals")
        torch.cuda.synchronize()
        out[f"z_tr_{st}"], out[f"z_rot_{st}"], out[f"z_tor_{st}"] = ztr.cpu().numpy(), zrot.cpu().numpy(), ztor.cpu().numpy().reshape(B, 7)
        out[f"words_{st}"] = np.asarray([[words(SEED, u, st, blk, strm) for (blk, strm) in ((0, 0), (1, 0), (0, 1), (1, 1))] for u in UIDS],
                                        dtype=np.uint32)
    th, rm, tr = torch.empty(nT, device="cuda"), torch.empty(B, 3, 3, device="cuda"), torch.empty(B, 3, device="cuda")
    _lib.check(L.dp_randomize_draws(SEED, C.byref(pb.struct), uid.data_ptr(), 0, 5.0, th.data_ptr(), rm.data_ptr(), tr.data_ptr(), _stream()),
               "dp_randomize_draws")
    torch.cuda.synchronize()
    out["theta"], out["rot_mats"], out["tr0"] = th.cpu().numpy().reshape(B, 7), rm.cpu().numpy(), tr.cpu().numpy()
    out["words_start"] = np.asarray([[words(SEED, u, 0xFFFFFFFF, blk, strm) for (blk, strm) in ((0, 2), (1, 2), (0, 3), (1, 3))] for u in UIDS],
                                    dtype=np.uint32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rng_stream.npz")
    np.savez(path, **out)
    print("wrote", path, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
