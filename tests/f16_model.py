"""Un-permutes the two-limb fp16 images of the DP_PREC_F16X3 node-side conv (dp_pack_conv_weights_node_f16) with the index formulas the kernel
uses (diffphore_amd/csrc/dp_conv_node_f16_impl.h) and evaluates the conv from them in float64.  Test infrastructure: tests/test_node_pack_cpu.py
holds the packed images to the fp64 oracle this way (layout, scales and exponent header, no GPU); the kernel itself is held to the oracle by
tests/test_conv_f16_gpu.py."""
import numpy as np

from bf16_model import D_OUT, PATHS, Z000, Z011, Z110, z_of

W1_SCALE = 16.0
FLAG_W = -60000.0
HDR_QUADS = 4


def _h(bits):
    return np.asarray(bits, dtype=np.uint16).view(np.float16).astype(np.float64)


def unpack_w1h(w1h_floats):
    """([img 3][k 64][feature 32] float64 = hi + lo, the two limb arrays) from the packed image [img][limb][q][lane][8]"""
    h = np.asarray(w1h_floats, dtype=np.float32).view(np.uint16).reshape(3, 2, 4, 64, 8)
    limbs = np.zeros((2, 3, 64, 32))
    for q in range(4):
        for lane in range(64):
            g, k = lane >> 4, 16 * q + (lane & 15)
            limbs[:, :, k, 8 * g:8 * g + 8] = np.stack([_h(h[:, lb, q, lane, :]) for lb in range(2)])
    return limbs[0] + limbs[1], limbs


def unpack_w2h(kind, w2h_floats):
    """(w exponent, per path W2[u][w][k 64] float64 = (hi + lo) * 2^-w, per path limb arrays) from [header][path][supertile][limb][chunk][M tile][q][lane][8]"""
    f = np.asarray(w2h_floats, dtype=np.float32)
    wexp = int(f[0])
    assert float(wexp) == f[0] and np.all(f[1:4 * HDR_QUADS] == 0.0)
    h = f.view(np.uint16)
    off, res, limbs_all = HDR_QUADS * 8, [], []
    for (zt, mul_in, mul_out, *_rest) in PATHS[kind]:
        MT = (mul_out + 15) // 16
        W = np.zeros((2, mul_in, 16 * MT, 64))
        for st in range(mul_in // 10):
            for lb in range(2):
                for ch in range(5):
                    for mt in range(MT):
                        blk = _h(h[off:off + 4 * 64 * 8]).reshape(4, 64, 8)
                        off += 4 * 64 * 8
                        for q in range(4):
                            for lane in range(64):
                                g, w = lane >> 4, 16 * mt + (lane & 15)
                                for j in range(8):
                                    W[lb, 10 * st + 2 * ch + (j >> 2), w, 16 * q + 4 * g + (j & 3)] = blk[q, lane, j]
        assert np.all(W[:, :, mul_out:, :] == 0.0) and np.all(W[:, :, :, 61:] == 0.0)
        limbs_all.append(W[:, :, :mul_out, :])
        res.append((W[0] + W[1])[:, :mul_out, :] * 2.0 ** -wexp)
    assert off == h.size, (off, h.size)
    return wexp, res, limbs_all


def node_conv_f16_images(kind, w1h_floats, w2h_floats, bn_shift, emb, attr_s, attr_g, x_gather, sh, scat, n_out, s_img=1):
    """the conv evaluated in float64 from the UNPACKED images (activations unrounded): out [n_out, D_out] without residual"""
    W1, _ = unpack_w1h(w1h_floats)
    _, W2, _ = unpack_w2h(kind, w2h_floats)
    deg = np.bincount(scat, minlength=n_out).astype(np.float64)
    inv = 1.0 / np.maximum(deg, 1.0)
    g_img = 3 - s_img
    pre = emb.astype(np.float64) @ W1[0, :, :20].T + attr_s.astype(np.float64) @ W1[s_img, :, :20].T + W1[s_img, :, 20][None, :] \
        + attr_g.astype(np.float64) @ W1[g_img, :, :20].T
    hid = np.maximum(pre, 0.0) / W1_SCALE * inv[scat][:, None]
    out = np.zeros((n_out, D_OUT[kind]))
    E = emb.shape[0]
    for p, (zt, mul_in, mul_out, in_off, out_off, _w_off, _fan) in enumerate(PATHS[kind]):
        d_in = 1 if zt in (Z000, Z011) else 3
        d_out = 1 if zt in (Z000, Z110) else 3
        x = x_gather[:, in_off:in_off + mul_in * d_in].reshape(E, mul_in, d_in)
        z = z_of(zt, x, sh).astype(np.float64)
        T = np.zeros((n_out, 64, mul_in, d_out))
        np.add.at(T, scat, hid[:, :, None, None] * z[:, None, :, :])
        out[:, out_off:out_off + mul_out * d_out] += np.einsum("uwk,nkum->nwm", W2[p], T).reshape(n_out, mul_out * d_out)
    return out + np.asarray(bn_shift, dtype=np.float64)[None, :]
