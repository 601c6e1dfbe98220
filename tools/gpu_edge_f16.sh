#!/bin/bash
export PYTHONPATH=$PWD
python -m pytest tests/test_conv_f16_gpu.py -q -s -k edge_side 2>&1 | grep -E "edge-side|bf16x6|passed|failed|Error" | tail -20
for prec in bf16x6 f16x3e; do for kind in 2 1 0; do
python - <<PY
import ctypes as C, torch, json
from diffphore_amd import _lib, packing, weights
# phore-phore-like CSR is awkward to build here: time the dense cross edge set through the edge-side kernel instead (same kernel, 491 520 edges)
kind, prec = $kind, "$prec"
dims = _lib.conv_dims(kind); sd = weights.init_state_dict(1)
x6 = prec == "f16x3e"
pk = packing.pack_conv(sd, f"encoder.phore_to_lig_conv_layers.{kind}", kind, _lib.PREC_F32 if x6 else _lib.PREC_BF16X6, x6=x6)
B, NL, NP = 512, 30, 32; dev = "cuda"
E = B * NL * NP
a = _lib.DpConvArgs()
t = dict(lig_ptr=torch.arange(B + 1, dtype=torch.int32) * NL, ph_ptr=torch.arange(B + 1, dtype=torch.int32) * NP, x_ptr=torch.arange(B + 1, dtype=torch.int32) * (NL * NP))
t = {k: v.to(dev) for k, v in t.items()}
hl, hp = torch.randn(B * NL, dims["d_in"], device=dev), torch.randn(B * NP, dims["d_in"], device=dev)
emb, sh = torch.randn(E, 20, device=dev), torch.randn(E, 9, device=dev)
out = torch.zeros(B * NL, dims["d_out"], device=dev)
keep = {k: v.to(dev) for k, v in pk.items()}
a.kind, a.B, a.slices, a.mode, a.max_scatter = kind, B, 1, _lib.EDGES_P2L, NL
a.precision = _lib.PREC_F16X3 if x6 else _lib.PREC_BF16X6
a.scat_ptr = t["lig_ptr"].data_ptr(); a.lig_ptr = t["lig_ptr"].data_ptr(); a.ph_ptr = t["ph_ptr"].data_ptr(); a.x_ptr = t["x_ptr"].data_ptr()
a.gather_tab, a.gather_stride = hp.data_ptr(), dims["d_in"]; a.attr1_tab, a.attr1_stride = hl.data_ptr(), dims["d_in"]; a.attr2_tab, a.attr2_stride = hp.data_ptr(), dims["d_in"]
a.edge_emb, a.edge_sh, a.sh_stride = emb.data_ptr(), sh.data_ptr(), 9
a.w1p = keep["w1p"].data_ptr(); a.w2p = (keep["w2x"] if x6 else keep["w2p"]).data_ptr()
a.bn_scale = (keep["bnx"] if x6 else keep["bn_scale"]).data_ptr(); a.bn_shift = keep["bn_shift"].data_ptr()
a.out, a.out_stride = out.data_ptr(), dims["d_out"]
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
for _ in range(3): _lib.check(L.dp_tp_conv(C.byref(a), st), "conv")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): L.dp_tp_conv(C.byref(a), st)
e1.record(); torch.cuda.synchronize()
print(prec, "kind", kind, "ms", round(e0.elapsed_time(e1) / 30, 4))
PY
done; done
