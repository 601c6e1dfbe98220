"""What the DP_PREC_BF16 node-side conv (diffphore_amd/csrc/dp_conv_node_bf16_impl.h) is SPECIFIED to compute, restated in numpy float64 with
bf16 roundings (round to nearest even) at exactly the points where the kernel rounds:

    emb, attr_scatter[:20], attr_gather[:20], W1 (+ bias)      -> bf16     (GEMM1 operands; fp32 accumulate)
    hid = relu(pre) / degree                                   -> bf16     (GEMM-A's A operand)
    z_p[e][u,m] = sum_t x_t * s_t  (fp32)                      -> bf16     (GEMM-A's B operand)
    T_p[n][k][u,m] = sum_e hid[e][k] z_p[e][u,m]  (fp32 acc)    -> bf16     (GEMM-B's B operand)
    W2 * path coefficient * CG norm * BatchNorm scale          -> bf16     (GEMM-B's A operand; fp32 accumulate)
    out = sum over convs + BatchNorm shifts (+ residual)          fp32

Test infrastructure: the GPU kernel is held to this tightly (it may differ by fp32 summation order and by values that sit on a bf16
rounding boundary), and this model is held to the fp64 oracle at the bf16-grade tolerance the tests state.  Path tables: dp_spec.h."""
import numpy as np

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
CG = {Z000: 1.0, Z011: 3 ** -0.5, Z101: 3 ** -0.5, Z110: 3 ** -0.5, Z111: 6 ** -0.5, Z121: 10 ** -0.5}


def bf16(x):
    """round-to-nearest-even to bf16, returned as float64"""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(a.shape)


def bf16_bits_to_f64(h):
    return (np.asarray(h, dtype=np.uint32) << 16).view(np.float32).astype(np.float64)


def z_of(zt, x, sh):
    """z[e][u, m] of one path in float32 arithmetic: x [E, mul_in, d_in] gathered input, sh [E, 9] (term tables of dp_conv_node_impl.h)"""
    x, sh = x.astype(np.float32), sh.astype(np.float32)
    s1 = sh[:, 1:4]
    c = np.float32(0.57735026918962576451) * sh[:, 6]
    M = np.zeros((sh.shape[0], 3, 3), dtype=np.float32)
    M[:, 0, 0], M[:, 1, 1], M[:, 2, 2] = -c - sh[:, 8], np.float32(2) * c, -c + sh[:, 8]
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


def unpack_w1b(w1b_floats):
    """[3 images][k 64][feature 32] float64 from the packed bf16 image (dp_pack_conv_weights_node_bf16: lane = 16 g + k % 16 holds features 8g..8g+7)"""
    h = np.asarray(w1b_floats, dtype=np.float32).view(np.uint16).reshape(3, 4, 64, 8)
    out = np.zeros((3, 64, 32))
    for q in range(4):
        for lane in range(64):
            g, k = lane >> 4, 16 * q + (lane & 15)
            out[:, k, 8 * g:8 * g + 8] = bf16_bits_to_f64(h[:, q, lane, :])
    return out


def unpack_w2b(kind, w2b_floats):
    """per path: W2[u][w][k 64] float64 (path coefficient, CG norm, BatchNorm scale folded, bf16-rounded) from the packed GEMM-B fragments
    [path][supertile][chunk][M tile][q][lane][8]: lane = 16 g + w % 16, element j <-> in-channel 10 st + 2 ch + (j >> 2), k = 16 q + 4 g + (j & 3)"""
    h = np.asarray(w2b_floats, dtype=np.float32).view(np.uint16)
    off, res = 0, []
    for (zt, mul_in, mul_out, *_rest) in PATHS[kind]:
        MT = (mul_out + 15) // 16
        W = np.zeros((mul_in, 16 * MT, 64))
        for st in range(mul_in // 10):
            for ch in range(5):
                for mt in range(MT):
                    blk = bf16_bits_to_f64(h[off:off + 4 * 64 * 8]).reshape(4, 64, 8)
                    off += 4 * 64 * 8
                    for q in range(4):
                        for lane in range(64):
                            g, w = lane >> 4, 16 * mt + (lane & 15)
                            for j in range(8):
                                W[10 * st + 2 * ch + (j >> 2), w, 16 * q + 4 * g + (j & 3)] = blk[q, lane, j]
        assert np.all(W[:, mul_out:, :] == 0.0) and np.all(W[:, :, 61:] == 0.0)
        res.append(W[:, :mul_out, :])
    assert off == h.size, (off, h.size)
    return res


def node_conv_bf16(kind, w1b_floats, w2b_floats, bn_shift, emb, attr_s, attr_g, x_gather, sh, scat, n_out, s_img=1, extra_hid=None, slot=None):
    """One conv: per-edge arrays emb [E,20], attr_s [E,20] (scatter node's scalars), attr_g [E,20], x_gather [E,D_in], sh [E,9]; scat [E] scatter
    node of each edge.  `s_img`: which first-layer block the SCATTER node's scalars feed (1 = attr1: phore->ligand and ligand-ligand, 2 = attr2:
    ligand->phore).  `extra_hid`: (emb2 [E,20], present [E]) - the pair mode's second (bond) edge on the same pair: its hidden vector adds.
    `slot` [E]: the edge's slot among its node's edges (dense modes: index of the gather node in its pose): the kernel works in passes of 32
    slots and rounds T once per pass (default: one pass).
    Returns out [n_out, D_out] float64 WITHOUT residual (mean and BatchNorm folded as in the kernel, shift added)."""
    W1 = unpack_w1b(w1b_floats)                        # [img][k][f]
    W2 = unpack_w2b(kind, w2b_floats)
    E = emb.shape[0]
    deg = np.bincount(scat, minlength=n_out).astype(np.float64)
    if extra_hid is not None:
        deg = deg + np.bincount(scat[extra_hid[1]], minlength=n_out)
    inv = (1.0 / np.maximum(deg, 1.0)).astype(np.float32).astype(np.float64)
    g_img = 3 - s_img
    us = bf16(attr_s) @ W1[s_img, :, :20].T + W1[s_img, :, 20][None, :]   # the scatter tile feeds the bias slot
    ug = bf16(attr_g) @ W1[g_img, :, :20].T
    pre = bf16(emb) @ W1[0, :, :20].T + us.astype(np.float32).astype(np.float64) + ug.astype(np.float32).astype(np.float64)
    hid = np.maximum(pre, 0.0)
    if extra_hid is not None:
        pre2 = bf16(extra_hid[0]) @ W1[0, :, :20].T + us.astype(np.float32).astype(np.float64) + ug.astype(np.float32).astype(np.float64)
        hid = hid + np.where(extra_hid[1][:, None], np.maximum(pre2, 0.0), 0.0)
    hid = bf16(hid.astype(np.float32).astype(np.float64) * inv[scat][:, None])
    out = np.zeros((n_out, D_OUT[kind]))
    for p, (zt, mul_in, mul_out, in_off, out_off, _w_off, _fan) in enumerate(PATHS[kind]):
        d_in = 1 if zt in (Z000, Z011) else 3
        d_out = 1 if zt in (Z000, Z110) else 3
        x = x_gather[:, in_off:in_off + mul_in * d_in].reshape(E, mul_in, d_in)
        z = bf16(z_of(zt, x, sh))                                         # [E, mul_in, d_out]
        o = np.zeros((n_out, mul_out, d_out))
        passes = np.zeros(E, dtype=np.int64) if slot is None else np.asarray(slot) // 32
        for et in range(int(passes.max()) + 1):
            sel = passes == et
            T = np.zeros((n_out, 64, mul_in, d_out))
            np.add.at(T, scat[sel], hid[sel][:, :, None, None] * z[sel][:, None, :, :])
            o += np.einsum("uwk,nkum->nwm", W2[p], bf16(T))               # [n, mul_out, d_out]
        out[:, out_off:out_off + mul_out * d_out] += o.reshape(n_out, mul_out * d_out)
    return out + np.asarray(bn_shift, dtype=np.float64)[None, :]
