#!/usr/bin/env python3
"""Where the host stage of one sampling_phore call goes (the `pack_and_noise_s` of bench.py's e2e line), piece by piece, on the box it
runs on: collate (native / numpy), upload, workspace, dp_prepare, noise.  usage: python tools/host_stage_timing.py [poses=512] [samples_per_complex=1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffphore_amd import config, synthetic, weights  # noqa: E402
from diffphore_amd.engine import DeviceModel, Engine, PackedBatch, Workspace  # noqa: E402


def tm(f, n=5, sync=True):
    f()
    if sync:
        torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    gs = synthetic.synthetic_pairs(2, N // per, 30, 8, 32)
    poses = [gs[i // per].sample_copy() for i in range(N)]
    model = DeviceModel(weights.init_state_dict(), config.ModelConfig(), "cuda")
    print(f"{N} poses, {per} sample(s) per complex")
    print("  collate, numpy            %.2f ms" % tm(lambda: PackedBatch(poses, "cuda", upload=False, host_pack="numpy"), sync=False))
    print("  collate, native (cached)  %.2f ms" % tm(lambda: PackedBatch(poses, "cuda", upload=False, host_pack="native"), sync=False))
    pbs = [PackedBatch(poses, "cuda", upload=False) for _ in range(6)]
    it = iter(pbs)
    print("  upload (one block)        %.2f ms" % tm(lambda: next(it).upload(), n=5))
    pb = pbs[0]
    print("  Workspace (alloc + init)  %.2f ms" % tm(lambda: Workspace(pb)))
    print("  Engine (+ dp_prepare)     %.2f ms" % tm(lambda: Engine(model, pb)))
    from diffphore_amd.sampling import _draw_reference_order
    print("  host noise, 20 steps      %.2f ms" % tm(lambda: [_draw_reference_order(N, pb.nt, 20) for _ in range(20)], sync=False))
    fresh = [gs[i // per].sample_copy() for i in range(N)]
    for g in fresh:
        try:
            object.__delattr__(g, "_dp_view")
        except AttributeError:
            pass
    t = time.perf_counter()
    PackedBatch(fresh, "cuda", upload=False)
    print("  collate, native, first time (views built) %.2f ms" % ((time.perf_counter() - t) * 1e3))


if __name__ == "__main__":
    main()
