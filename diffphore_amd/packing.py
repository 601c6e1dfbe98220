"""Host-side weight re-ordering for the HIP kernels (runs once at model load; not on the hot path)."""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib
from . import config as C

_OUT_IRREPS = {0: C.SEQ[1], 1: C.SEQ[2], 2: C.SEQ[3], 3: C.SEQ[3], 4: C.FINAL_OUT, 5: C.TOR_OUT}


def bn_affine(sd: Dict[str, torch.Tensor], prefix: str, irreps, eps: float = 1e-5):
    """e3nn BatchNorm(eval) as a per-channel affine map: out = x*scale + shift."""
    rm, rv = sd[prefix + ".running_mean"].double(), sd[prefix + ".running_var"].double()
    w, b = sd[prefix + ".weight"].double(), sd[prefix + ".bias"].double()
    scale, shift = [], []
    im = iv = 0
    for mul, l, p in irreps:
        d = 2 * l + 1
        s = (rv[iv:iv + mul] + eps).pow(-0.5) * w[iv:iv + mul]
        iv += mul
        if l == 0 and p == 1:
            sh = b[im:im + mul] - rm[im:im + mul] * s
            im += mul
        else:
            sh = torch.zeros(mul, dtype=torch.float64)
        scale.append(s.repeat_interleave(d))
        shift.append(sh.repeat_interleave(d))
    return torch.cat(scale).float().contiguous(), torch.cat(shift).float().contiguous()


def pack_conv(sd: Dict[str, torch.Tensor], prefix: str, kind: int, precision: int = 0, node_side: bool = False, x6: bool = False,
              node_bf16: bool = False, node_f16: bool = False) -> Dict[str, torch.Tensor]:
    """fc.0 / fc.3 / batch_norm of one TensorProductConvLayer -> MFMA operand order (host tensors).  `node_side` adds the layout of
    the node-side kernel (dp_conv_node_impl.h: layer convs over the dense cross edge sets, fp32); `x6` adds the second layer packed for the
    edge-side kernel at DP_PREC_F16X3 (`w2x` + `bnx`: what the edge-side convs of an fp32 model run on, DpConvW.w2x / bn_scale_x); `node_bf16` adds the bf16 MFMA fragments of the
    DP_PREC_BF16 node-side kernel (dp_conv_node_bf16_impl.h) under the same names `w1n` / `w2n`; `node_f16` adds the two-limb fp16 fragments of
    the fp32-grade DP_PREC_F16X3 node-side kernel (dp_conv_node_f16_impl.h) as `w1h` / `w2h` (DpConvW.w1h / w2h)."""
    L = _lib.lib()
    dims = _lib.conv_dims(kind)
    fc0_w = sd[prefix + ".fc.0.weight"].float().contiguous()
    fc0_b = sd[prefix + ".fc.0.bias"].float().contiguous()
    fc3_w = sd[prefix + ".fc.3.weight"].float().contiguous()
    fc3_b = sd[prefix + ".fc.3.bias"].float().contiguous()
    assert tuple(fc0_w.shape) == (dims["hidden"], dims["n_attr"]), (prefix, fc0_w.shape, dims)
    assert tuple(fc3_w.shape) == (dims["numel"], dims["hidden"]), (prefix, fc3_w.shape, dims)
    w1p = torch.empty(L.dp_conv_w1p_floats(kind), dtype=torch.float32)
    w2p = torch.empty(L.dp_conv_w2p_floats(kind, precision), dtype=torch.float32)
    _lib.check(L.dp_pack_conv_weights(kind, precision, fc0_w.data_ptr(), fc0_b.data_ptr(), fc3_w.data_ptr(), fc3_b.data_ptr(),
                                      w1p.data_ptr(), w2p.data_ptr()), "dp_pack_conv_weights")
    scale, shift = bn_affine(sd, prefix + ".batch_norm", _OUT_IRREPS[kind])
    out = dict(w1p=w1p, w2p=w2p, bn_scale=scale, bn_shift=shift)
    if node_side:
        w1n = torch.empty(L.dp_conv_w1n_floats(kind), dtype=torch.float32)
        w2n = torch.empty(L.dp_conv_w2n_floats(kind), dtype=torch.float32)
        # the node-side kernel carries the BatchNorm scale inside its packed second layer (see include/diffphore_amd.h)
        _lib.check(L.dp_pack_conv_weights_node(kind, fc0_w.data_ptr(), fc0_b.data_ptr(), fc3_w.data_ptr(), fc3_b.data_ptr(),
                                               scale.data_ptr(), w1n.data_ptr(), w2n.data_ptr()), "dp_pack_conv_weights_node")
        out.update(w1n=w1n, w2n=w2n)
    if node_bf16:
        assert not node_side, "one node-side layout per packed conv"
        w1b = torch.empty(L.dp_conv_w1b_floats(kind), dtype=torch.float32)
        w2b = torch.empty(L.dp_conv_w2b_floats(kind), dtype=torch.float32)
        _lib.check(L.dp_pack_conv_weights_node_bf16(kind, fc0_w.data_ptr(), fc0_b.data_ptr(), fc3_w.data_ptr(), fc3_b.data_ptr(),
                                                    scale.data_ptr(), w1b.data_ptr(), w2b.data_ptr()), "dp_pack_conv_weights_node_bf16")
        out.update(w1n=w1b, w2n=w2b)
    if node_f16:
        w1h = torch.empty(L.dp_conv_w1h_floats(kind), dtype=torch.float32)
        w2h = torch.empty(L.dp_conv_w2h_floats(kind), dtype=torch.float32)
        _lib.check(L.dp_pack_conv_weights_node_f16(kind, fc0_w.data_ptr(), fc0_b.data_ptr(), fc3_w.data_ptr(), fc3_b.data_ptr(),
                                                   scale.data_ptr(), w1h.data_ptr(), w2h.data_ptr()), "dp_pack_conv_weights_node_f16")
        out.update(w1h=w1h, w2h=w2h)
    if x6:
        # the second layer once more for the EDGE-side kernel at DP_PREC_F16X3 (two fp16 limbs of W2 * 2^w; w = float[0] of the image's last quad)
        # and the BatchNorm scale that undoes the 2^w (the contraction, the segmented mean and the scale are linear in W2)
        w1_ = torch.empty(L.dp_conv_w1p_floats(kind), dtype=torch.float32)
        w2x = torch.empty(L.dp_conv_w2p_floats(kind, _lib.PREC_F16X3), dtype=torch.float32)
        _lib.check(L.dp_pack_conv_weights(kind, _lib.PREC_F16X3, fc0_w.data_ptr(), fc0_b.data_ptr(), fc3_w.data_ptr(), fc3_b.data_ptr(),
                                          w1_.data_ptr(), w2x.data_ptr()), "dp_pack_conv_weights (f16x3)")
        assert torch.equal(w1_, w1p)                       # the first layer's packing does not depend on the precision
        wexp = float(w2x[-4])
        assert wexp == int(wexp) and abs(wexp) <= 40, wexp
        out["w2x"] = w2x
        out["bnx"] = (scale.double() * 2.0 ** -wexp).float().contiguous()
    return out
