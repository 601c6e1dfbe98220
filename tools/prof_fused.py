#!/usr/bin/env python3
"""Cycle-stamp table of the FUSED layer-3 ligand-side launch (ligand-ligand + phore->ligand + phore->ligand(normals)) at BASELINE configs[1]:
where a workgroup's wave 0 spends its time, per conv of the launch.  Needs the profiling build of the kind-3 kernels:

    tools/build_variant.sh prof32 "-DDP_PROF" dp_conv_node_k3.hip          (fp32 node-side kernel)
    tools/build_variant.sh prof "-DDP_PROF" dp_conv_node_bf16_k3.hip       (bf16 node-side kernel; run with --precision bf16)
    DIFFPHORE_LIB=build_abl/libprof32.so python tools/prof_fused.py [--precision f32|bf16] [--poses 512]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffphore_amd import _lib, config, synthetic, weights  # noqa: E402
from diffphore_amd.engine import DeviceModel, Engine, PackedBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--poses", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    gs = synthetic.synthetic_pairs(2, a.poses, 30, 8, 32)
    model = DeviceModel(weights.init_state_dict(), config.ModelConfig(), "cuda", precision=a.precision)
    pb = PackedBatch(gs, "cuda")
    eng = Engine(model, pb)
    pb.set_time(0.5, 0.5, 0.5)
    L = _lib.lib()
    buf = (C.c_ulonglong * 16)()
    eng.forward()
    torch.cuda.synchronize()
    L.dp_debug_prof_node(None, 1)
    for _ in range(a.steps):
        eng.forward()
    torch.cuda.synchronize()
    L.dp_debug_prof_node(buf, 0)
    v = np.asarray(list(buf), dtype=np.float64)
    tot = v.sum()
    print(f"layer-3 ligand-side launch, {a.poses} poses, precision {a.precision}: share of wave 0's cycles (sum over workgroups and {a.steps} launches)")
    if a.precision == "f32":
        names = ["staging (operand requests, SH factors -> LDS, barriers)", "GEMM1 (radial layer 1)", "GEMM-A (+ z arithmetic, operand fetch)", "GEMM-B (+ W2 / T fetch)"]
        for c, cn in enumerate(("ligand-ligand", "phore->ligand", "phore->ligand (normals)")):
            sub = v[4 * c:4 * c + 4]
            print(f"  {cn:26s} {100 * sub.sum() / tot:5.1f} %   " + "   ".join(f"{n.split(' (')[0]} {100 * x / tot:4.1f}" for n, x in zip(names, sub)))
        print(f"  {'k-partial reduce + epilogue':26s} {100 * v[12] / tot:5.1f} %")
    else:
        for n, x in zip(("staging", "GEMM1", "GEMM-A", "GEMM-B", "reduce + epilogue"), v[:5]):
            print(f"  {n:20s} {100 * x / tot:5.1f} %")


if __name__ == "__main__":
    main()
