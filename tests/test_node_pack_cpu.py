"""CPU check of the node-side conv's host half (dp_pack_conv_weights_node) and of the re-association it relies on.

The kernel (dp_conv_node.hip) computes   T[n][k][u,m] = sum_{e in n} hid[e][k] z_p[e][u,m]   and then
out[n][w,m] = sum_{k,u} W2_p[(u,w)][k] T[n][k][u,m]   from weights packed in MFMA-fragment order.  Here the packed arrays are
un-permuted with the same index formulas the kernel uses (lane = 16 g + c; GEMM1 K-step s <-> attr (s/5)*20 + 5g + s%5; GEMM-B step
4i + rr <-> in-channel NU*ut + i, hidden row 16q + 4g + rr) and the result is held to the oracle's TensorProductConvLayer in fp64.
No GPU, no compute call into the library - only the host packer runs."""
import numpy as np
import pytest
import torch

from diffphore_amd import _lib, config, packing, weights

# (ztype, mul_in, mul_out, in_off, out_off, w_off, fan_in) per kind: dp_spec.h
Z000, Z011, Z101, Z110, Z111, Z121 = range(6)
PATHS = {
    0: [(Z000, 20, 20, 0, 0, 0, 20), (Z011, 20, 10, 0, 20, 400, 20)],
    1: [(Z000, 20, 20, 0, 0, 0, 30), (Z011, 20, 10, 0, 20, 400, 40), (Z101, 10, 10, 20, 20, 600, 40), (Z110, 10, 20, 20, 0, 700, 30),
        (Z111, 10, 10, 20, 50, 900, 10), (Z121, 10, 10, 20, 20, 1000, 40)],
    2: [(Z000, 20, 20, 0, 0, 0, 30), (Z011, 20, 10, 0, 20, 400, 50), (Z101, 10, 10, 20, 20, 600, 50), (Z110, 10, 20, 20, 0, 700, 30),
        (Z111, 10, 10, 20, 50, 900, 30), (Z121, 10, 10, 20, 20, 1000, 50), (Z101, 10, 10, 50, 50, 1100, 30), (Z111, 10, 10, 50, 20, 1200, 50),
        (Z110, 10, 20, 50, 80, 1300, 10), (Z121, 10, 10, 50, 50, 1500, 30)],
    3: [(Z000, 20, 20, 0, 0, 0, 30), (Z011, 20, 10, 0, 20, 400, 50), (Z101, 10, 10, 20, 20, 600, 50), (Z110, 10, 20, 20, 0, 700, 30),
        (Z111, 10, 10, 20, 50, 900, 50), (Z121, 10, 10, 20, 20, 1000, 50), (Z101, 10, 10, 50, 50, 1100, 50), (Z111, 10, 10, 50, 20, 1200, 50),
        (Z110, 10, 20, 50, 80, 1300, 30), (Z121, 10, 10, 50, 50, 1500, 50), (Z000, 20, 20, 80, 80, 1600, 30), (Z011, 20, 10, 80, 50, 2000, 50)],
}
D_OUT = {0: 50, 1: 80, 2: 100, 3: 100}


def _oracle_conv(sd, prefix, kind, x_gather, attr, sh, scatter, n_out):
    from oracle import e3
    from oracle.model import OracleScoreModel, scatter_mean
    m = OracleScoreModel(sd, config.ModelConfig(), dtype=torch.float64, tables=(np.ones(1000), np.ones(5001)))
    ir_in, ir_out = m.seq[min(kind, 3)], m.seq[min(kind + 1, 3)]
    w = m.mlp(prefix + ".fc", attr.double())
    tp = e3.fctp_apply(x_gather.double(), sh.double(), w, ir_in, m.sh_irreps, ir_out)
    out = scatter_mean(tp, scatter, n_out)
    bn = prefix + ".batch_norm"
    return e3.batchnorm_eval(out, ir_out, m.sd[bn + ".running_mean"], m.sd[bn + ".running_var"], m.sd[bn + ".weight"], m.sd[bn + ".bias"])


def _z(zt, x, sh):
    """z[e][u, m] of one path: x [E, mul_in, d_in] gathered input, sh [E, 9]; the kernel's term tables (term_of in dp_conv_node.hip)."""
    s1 = sh[:, 1:4]
    c = sh[:, 6] / np.sqrt(3.0)
    M = np.zeros((sh.shape[0], 3, 3))
    M[:, 0, 0], M[:, 1, 1], M[:, 2, 2] = -c - sh[:, 8], 2 * c, -c + sh[:, 8]
    M[:, 0, 1] = M[:, 1, 0] = sh[:, 5]
    M[:, 0, 2] = M[:, 2, 0] = sh[:, 4]
    M[:, 1, 2] = M[:, 2, 1] = sh[:, 7]
    if zt == Z000:
        return x[:, :, :1]
    if zt == Z011:
        return x[:, :, :1] * s1[:, None, :]
    if zt == Z101:
        return x
    if zt == Z110:
        return (x * s1[:, None, :]).sum(-1, keepdims=True)
    if zt == Z111:
        return np.cross(x, s1[:, None, :])
    return np.einsum("eui,eim->eum", x, M)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_node_side_packing_reproduces_the_conv(kind):
    L = _lib.lib()
    sd = weights.init_state_dict(11)
    prefix = f"encoder.phore_to_lig_conv_layers.{kind}"
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, node_side=True)
    w1n = pk["w1n"].numpy().astype(np.float64).reshape(4, 16, 64)
    w2n = pk["w2n"].numpy().astype(np.float64)
    assert w2n.size == L.dp_conv_w2n_floats(kind)
    dims = _lib.conv_dims(kind)
    g = torch.Generator().manual_seed(5 + kind)
    NL, NP = 3, 7                                                    # 3 scatter nodes (ligand atoms), 7 edges each (phore nodes)
    hl, hp = torch.randn(NL, dims["d_in"], generator=g), torch.randn(NP, dims["d_in"], generator=g)
    E = NL * NP
    emb, sh = torch.randn(E, 20, generator=g), torch.randn(E, 9, generator=g)
    sh[:, 0] = 1.0
    src, dst = torch.arange(NL).repeat_interleave(NP), torch.arange(NP).repeat(NL)
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, NL).numpy()

    # GEMM1 from the packed first layer: hid[e][k], k = 16 q + (lane & 15)
    A = attr.numpy().astype(np.float64)
    hid = np.zeros((E, 64))
    for q in range(4):
        for lane in range(64):
            gq, k = lane >> 4, 16 * q + (lane & 15)
            for s in range(16):
                a_val = A[:, (s // 5) * 20 + 5 * gq + (s % 5)] if s < 15 else (1.0 if gq == 0 else 0.0)
                hid[:, k] += a_val * w1n[q, s, lane]
    hid = np.maximum(hid, 0.0)
    assert np.allclose(hid[:, 60], 1.0) and np.all(hid[:, 61:] == 0.0)
    xg, shn = hp[dst].numpy().astype(np.float64), sh.numpy().astype(np.float64)

    out = np.zeros((NL, D_OUT[kind]))
    off = 0
    for (zt, mul_in, mul_out, in_off, out_off, w_off, fan_in) in PATHS[kind]:
        d_in = 1 if zt in (Z000, Z011) else 3
        d_out = 1 if zt in (Z000, Z110) else 3
        x = xg[:, in_off:in_off + mul_in * d_in].reshape(E, mul_in, d_in)
        z = _z(zt, x, shn)                                            # [E, mul_in, d_out]
        Tn = np.einsum("nek,neum->nkum", hid.reshape(NL, NP, 64), z.reshape(NL, NP, mul_in, d_out))
        nu = 5 if d_out == 3 else 10
        steps, MT = 4 * nu, (mul_out + 15) // 16
        for ut in range(mul_in // nu):
            for mt in range(MT):
                blk = w2n[off:off + 4 * 64 * steps].reshape(4, 64, steps)
                off += 4 * 64 * steps
                for q in range(4):
                    for lane in range(64):
                        gq, w = lane >> 4, 16 * mt + (lane & 15)
                        if w >= mul_out:
                            assert np.all(blk[q, lane] == 0.0)
                            continue
                        for s in range(steps):
                            u, k = nu * ut + (s >> 2), 16 * q + 4 * gq + (s & 3)
                            out[:, out_off + w * d_out:out_off + (w + 1) * d_out] += blk[q, lane, s] * Tn[:, k, u, :]
    assert off == w2n.size
    out /= NP                                                          # scatter mean (the kernel applies it to hid: same linear map)
    out = out + pk["bn_shift"].numpy().astype(np.float64)              # the BatchNorm scale is folded into the packed second layer
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 1e-6, err


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
def test_bf16_node_packing_and_model_vs_oracle(kind, mode):
    """DP_PREC_BF16: the packed bf16 fragments (dp_pack_conv_weights_node_bf16) un-permuted with the kernel's index formulas and the conv
    evaluated with bf16 roundings where the kernel rounds (tests/bf16_model.py) stay within bf16-grade distance of the fp64 oracle:
    <= 2e-2 of the output's largest magnitude (measured 2-6e-3: ~2^-9 per operand, averaged over 32-edge / 640-term sums)."""
    from bf16_model import node_conv_bf16
    L = _lib.lib()
    sd = weights.init_state_dict(11)
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_conv_layers"
    prefix = f"encoder.{fam}.{kind}"
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_BF16, node_bf16=True)
    assert pk["w1n"].numel() == L.dp_conv_w1b_floats(kind) and pk["w2n"].numel() == L.dp_conv_w2b_floats(kind)
    dims = _lib.conv_dims(kind)
    g = torch.Generator().manual_seed(50 + kind)
    NL, NP = 5, 37
    hl, hp = torch.randn(NL, dims["d_in"], generator=g), torch.randn(NP, dims["d_in"], generator=g)
    E = NL * NP
    emb, sh = torch.randn(E, 20, generator=g), torch.randn(E, 9, generator=g)
    sh[:, 0] = 1.0
    src, dst = torch.arange(NL).repeat_interleave(NP), torch.arange(NP).repeat(NL)
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    if mode == "p2l":
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, NL).numpy()
        got = node_conv_bf16(kind, pk["w1n"].numpy(), pk["w2n"].numpy(), pk["bn_shift"].numpy(), emb.numpy(), hl[src, :20].numpy(), hp[dst, :20].numpy(),
                             hp[dst].numpy(), sh.numpy(), src.numpy(), NL, s_img=1)
    else:
        ref = _oracle_conv(sd, prefix, kind, hl[src], attr, sh, dst, NP).numpy()
        got = node_conv_bf16(kind, pk["w1n"].numpy(), pk["w2n"].numpy(), pk["bn_shift"].numpy(), emb.numpy(), hp[dst, :20].numpy(), hl[src, :20].numpy(),
                             hl[src].numpy(), sh.numpy(), dst.numpy(), NP, s_img=2)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"kind {kind} {mode}: bf16 model vs fp64 oracle {err:.2e}")
    assert err < 2e-2, err


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
def test_f16_node_packing_vs_oracle(kind, mode):
    """DP_PREC_F16X3: the packed two-limb fp16 fragments (dp_pack_conv_weights_node_f16), un-permuted with the kernel's index formulas
    (tests/f16_model.py), reproduce the fp64 oracle's conv to 1e-6 (the limbs carry >= 21 bits: hi = rne16(v), lo = rne16(v - hi)); the header
    exponent w puts the largest |W2 * 2^w| into (1024, 2048]; the "edge does not exist" row of the first layer is -60000 in units of 16 W1."""
    from f16_model import FLAG_W, node_conv_f16_images, unpack_w1h, unpack_w2h
    L = _lib.lib()
    sd = dict(weights.init_state_dict(11))
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_conv_layers"
    prefix = f"encoder.{fam}.{kind}"
    sd[prefix + ".fc.3.weight"] = sd[prefix + ".fc.3.weight"] * (3.0 if mode == "p2l" else 0.02)       # two different exponents
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, node_f16=True)
    assert pk["w1h"].numel() == L.dp_conv_w1h_floats(kind) and pk["w2h"].numel() == L.dp_conv_w2h_floats(kind)
    W1, l1 = unpack_w1h(pk["w1h"].numpy())
    assert np.all(W1[0, :, 20] == FLAG_W) and np.all(l1[1][0, :, 20] == 0.0)
    # the host's fp16 conversion rounds to nearest: the hi limb is within half an fp16 ulp of the value, the lo limb within half an ulp of the rest
    def half_ulp(x):
        return np.spacing(np.abs(x).astype(np.float16)).astype(np.float64) / 2
    v = l1[0] + l1[1]
    assert np.all(np.abs(v - l1[0]) <= half_ulp(l1[0])) and np.mean(l1[0] == v.astype(np.float32).astype(np.float16).astype(np.float64)) > 0.999
    ref1 = 16.0 * sd[prefix + ".fc.0.weight"].double().numpy()[:, :20]
    assert np.all(np.abs(W1[0, :60, :20] - ref1) <= half_ulp(l1[1][0, :60, :20]) + 2.0 ** -25)
    wexp, W2, l2 = unpack_w2h(kind, pk["w2h"].numpy())
    mx = max(np.abs(w).max() for w in W2) * 2.0 ** wexp
    assert 1024.0 < mx <= 2048.0, (wexp, mx)
    for limbs in l2:
        assert np.all(np.abs(limbs[1]) <= half_ulp(limbs[0]))
    dims = _lib.conv_dims(kind)
    g = torch.Generator().manual_seed(50 + kind)
    NL, NP = 5, 37
    hl, hp = torch.randn(NL, dims["d_in"], generator=g), torch.randn(NP, dims["d_in"], generator=g)
    E = NL * NP
    emb, sh = torch.randn(E, 20, generator=g), torch.randn(E, 9, generator=g)
    sh[:, 0] = 1.0
    src, dst = torch.arange(NL).repeat_interleave(NP), torch.arange(NP).repeat(NL)
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    if mode == "p2l":
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, NL).numpy()
        got = node_conv_f16_images(kind, pk["w1h"].numpy(), pk["w2h"].numpy(), pk["bn_shift"].numpy(), emb.numpy(), hl[src, :20].numpy(),
                                   hp[dst, :20].numpy(), hp[dst].numpy(), sh.numpy(), src.numpy(), NL, s_img=1)
    else:
        ref = _oracle_conv(sd, prefix, kind, hl[src], attr, sh, dst, NP).numpy()
        got = node_conv_f16_images(kind, pk["w1h"].numpy(), pk["w2h"].numpy(), pk["bn_shift"].numpy(), emb.numpy(), hp[dst, :20].numpy(),
                                   hl[src, :20].numpy(), hl[src].numpy(), sh.numpy(), dst.numpy(), NP, s_img=2)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"kind {kind} {mode}: unpacked f16x3 images vs fp64 oracle {err:.2e} (w = {wexp})")
    assert err < 2e-6, err
