#!/usr/bin/env python3
"""Micro-benchmark of the fused TP conv kernel alone (dense cross edge set, cfg2 shape by default)."""
import argparse
import ctypes as C
import json

import torch

from diffphore_amd import _lib, packing, weights

ALG_BYTES = {0: 2516, 1: 4636, 2: 6756, 3: 9236}   # 4*(numel + D_in + 9) per edge-conv
FLOPS = {0: 79200 + 2000, 1: 139200 + 4200, 2: 199200 + 6400, 3: 271200 + 8400}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--nl", type=int, default=30)
    ap.add_argument("--np", type=int, default=32)
    ap.add_argument("--kind", type=int, default=3)
    ap.add_argument("--slices", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="p2l")
    ap.add_argument("--prec", default="f32")
    ap.add_argument("--phase-split", type=int, default=0)
    ap.add_argument("--algo", default="edge", choices=["edge", "node"])
    ap.add_argument("--path-split", type=int, default=0, help="node-side: two workgroups per node group, half of the out slots each")
    ap.add_argument("--no-tail", type=int, default=0, help="DpConvArgs.no_tail: 1 = the lean node-side instantiation (no pose ends on a pass of <= 16 slots)")
    ap.add_argument("--prof", action="store_true", help="print the cycle-stamp breakdown (library built with -DDP_PROF)")
    ap.add_argument("--lib", default=None, help="alternative libdiffphore_hip.so (ablation builds)")
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = a.lib
    dims = _lib.conv_dims(a.kind)
    sd = weights.init_state_dict(1)
    f16 = a.prec == "f16x3" and a.algo == "node"    # the fp32-grade two-limb fp16 node-side kernel (DP_PREC_F16X3): weights of an fp32 model
    f16e = a.prec == "f16x3" and a.algo == "edge"   # the same arithmetic on the edge-side kernel (what an fp32 model's phore-phore / torsion convs run)
    pk = packing.pack_conv(sd, f"encoder.phore_to_lig_conv_layers.{a.kind}", a.kind, _lib.PREC_F32 if a.prec == "f16x3" else _lib.PRECISIONS[a.prec],
                           node_side=a.algo == "node" and a.prec == "f32", node_bf16=a.algo == "node" and a.prec == "bf16", node_f16=f16, x6=f16e)
    if f16:
        pk["w1n"], pk["w2n"] = pk["w1h"], pk["w2h"]
    if f16e:
        pk["w2p"], pk["bn_scale"] = pk["w2x"], pk["bnx"]
    B, NL, NP = a.B, a.nl, a.np
    dev = "cuda"
    lig_ptr = torch.arange(B + 1, dtype=torch.int32) * NL
    ph_ptr = torch.arange(B + 1, dtype=torch.int32) * NP
    x_ptr = torch.arange(B + 1, dtype=torch.int32) * (NL * NP)
    E = B * NL * NP
    hl = torch.randn(B * NL, dims["d_in"], device=dev)
    hp = torch.randn(B * NP, dims["d_in"], device=dev)
    emb = torch.randn(E, 20, device=dev)
    sh = torch.randn(E, 9, device=dev)
    n_out = B * NL if a.mode == "p2l" else B * NP
    out = torch.zeros(n_out, dims["d_out"], device=dev)
    keep = {k: v.to(dev) for k, v in pk.items()}
    keep.update(lig_ptr=lig_ptr.to(dev), ph_ptr=ph_ptr.to(dev), x_ptr=x_ptr.to(dev))
    args = _lib.DpConvArgs()
    args.kind, args.B, args.slices = a.kind, B, a.slices
    args.phase_split = a.phase_split
    args.precision = _lib.PREC_F16X3 if a.prec == "f16x3" else _lib.PRECISIONS[a.prec]
    args.max_scatter = NL if a.mode == "p2l" else NP
    args.mode = _lib.EDGES_P2L if a.mode == "p2l" else _lib.EDGES_L2P
    args.scat_ptr = keep["lig_ptr"].data_ptr() if a.mode == "p2l" else keep["ph_ptr"].data_ptr()
    args.lig_ptr, args.ph_ptr, args.x_ptr = keep["lig_ptr"].data_ptr(), keep["ph_ptr"].data_ptr(), keep["x_ptr"].data_ptr()
    args.gather_tab = (hp if a.mode == "p2l" else hl).data_ptr()
    args.gather_stride = dims["d_in"]
    args.attr1_tab, args.attr1_stride = hl.data_ptr(), dims["d_in"]
    args.attr2_tab, args.attr2_stride = hp.data_ptr(), dims["d_in"]
    args.edge_emb, args.edge_sh, args.sh_stride = emb.data_ptr(), sh.data_ptr(), 9
    if a.prec == "bf16":
        emb_bf = torch.zeros((E + 1) * 32, dtype=torch.int16, device=dev)
        _lib.check(_lib.lib().dp_edge_emb_bf16(emb.data_ptr(), E, emb_bf.data_ptr(), torch.cuda.current_stream().cuda_stream), "dp_edge_emb_bf16")
        args.edge_emb_bf, args.emb_flag_row = emb_bf.data_ptr(), E
    if f16:
        emb_h = torch.zeros((E + 1) * 64, dtype=torch.int16, device=dev)
        _lib.check(_lib.lib().dp_edge_emb_f16(emb.data_ptr(), E, emb_h.data_ptr(), torch.cuda.current_stream().cuda_stream), "dp_edge_emb_f16")
        args.edge_emb_bf, args.emb_flag_row = emb_h.data_ptr(), E
    args.w1p, args.w2p = keep["w1p"].data_ptr(), keep["w2p"].data_ptr()
    if a.algo == "node":
        nb = _lib.lib().dp_conv_node_max_nodes()
        args.algo, args.w1n, args.w2n = _lib.ALGO_NODE, keep["w1n"].data_ptr(), keep["w2n"].data_ptr()
        args.slices = (args.max_scatter + nb - 1) // nb
        args.path_split = a.path_split
        args.no_tail = a.no_tail
    args.bn_scale, args.bn_shift = keep["bn_scale"].data_ptr(), keep["bn_shift"].data_ptr()
    args.out, args.out_stride, args.accumulate = out.data_ptr(), dims["d_out"], 0
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(L.dp_tp_conv(C.byref(args), st), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        L.dp_tp_conv(C.byref(args), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    if a.prof and a.algo == "node":
        buf = (C.c_ulonglong * 16)()
        L.dp_debug_prof_node(None, 1)
        L.dp_tp_conv(C.byref(args), st)
        torch.cuda.synchronize()
        L.dp_debug_prof_node(buf, 0)
        tot = sum(buf[:11]) or 1
        names = ["staging: 2nd barrier (or whole staging)", "GEMM1", "GEMM-A (+operand fetch)", "GEMM-B", "reduce (or whole epilogue)",
                 "pass top: barrier", "pass top: operand requests", "staging loop", "epilogue: operand requests + barrier", "AccStore", "epilogue: 2nd barrier"]
        for n, v in zip(names, buf[:11]):
            if v == 0:
                continue
            print(f"  {n:24s} {v:14d} ticks  {100.0 * v / tot:5.1f} %")
    elif a.prof:
        buf = (C.c_ulonglong * 16)()
        L.dp_debug_prof(None, 1)
        L.dp_tp_conv(C.byref(args), st)
        torch.cuda.synchronize()
        L.dp_debug_prof(buf, 0)
        tot = sum(buf[:8]) or 1
        names = ["gemm1+split", "operands+sync", "ring prologue", "tiles ph0", "segsum ph0", "tiles ph1", "segsum ph1", "epilogue"]
        for n, v in zip(names, buf[:8]):
            print(f"  {n:14s} {v:14d} ticks  {100.0 * v / tot:5.1f} %")
    print(json.dumps(dict(kind=a.kind, prec=a.prec, algo=a.algo, mode=a.mode, B=B, edges=E, ms=ms, ns_per_edge=ms * 1e6 / E,
                          alg_GBps=E * ALG_BYTES[a.kind] / ms / 1e6, TFLOPs=E * FLOPS[a.kind] / ms / 1e9)))


if __name__ == "__main__":
    main()
