"""CPU-only checks of the C-ABI shared library: it loads without a GPU, exports every symbol include/diffphore_amd.h
declares, and the ctypes struct mirrors have the sizes the C compiler gives them.  No compute calls."""
import ctypes as C
import os
import re

import pytest

from diffphore_amd import _lib, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "diffphore_amd.h")).read()
    names = set(re.findall(r"^\s*(?:int|size_t|int64_t)\s+(dp_\w+)\s*\(", hdr, flags=re.M))
    assert {"dp_tp_conv", "dp_tp_conv_node", "dp_forward", "dp_sample_step", "dp_sample", "dp_sample_graph_create", "dp_sample_graph_launch",
            "dp_sample_graph_destroy", "dp_conformer_update", "dp_prepare", "dp_randomize_position", "dp_fit_overlap",
            "dp_pack_conv_weights", "dp_pack_conv_weights_node", "dp_pack_bn", "dp_work_bytes", "dp_work_bind", "dp_work_init",
            "dp_weights_segments", "dp_weights_blob_floats", "dp_weights_bind", "dp_resolve_conv_algo", "dp_abi_version",
            "dp_struct_sizes", "dp_noise_normals", "dp_randomize_draws", "dp_philox4x32", "dp_forward_launch_of", "dp_tp_conv_node_fused", "dp_tp_conv_node_fused_split",
            "dp_build_info", "dp_pack_conv_weights_node_bf16", "dp_conv_w1b_floats", "dp_conv_w2b_floats", "dp_edge_emb_bf16"} <= names
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in the header but not exported"
    assert "dp_list_reduce" in names
    # ... and nothing else: the C++ launchers between the library's translation units are not part of the ABI (csrc/exports.map)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TtDdBbRrWwVv"}
    assert exported == names, sorted(exported ^ names)


def test_source_hash_ignores_comments_and_layout_but_not_code():
    """build.source_hash is taken over the code only: rewording a comment or re-wrapping a line does not re-key the committed PMC records, any
    token change does; string / character literals are code"""
    from diffphore_amd.csrc.build import _code_only
    base = 'int a = 1; // note\nconst char* s = "// kept /* kept */";\n/* block\n   comment */ char c = \'"\';   int   b;'
    assert _code_only(base) == 'int a = 1; const char* s = "// kept /* kept */"; char c = \'"\'; int b;'
    assert _code_only(base.replace("// note", "// another note").replace("int   b", "int\n b")) == _code_only(base)
    assert _code_only(base.replace("a = 1", "a = 2")) != _code_only(base)
    assert _code_only(base.replace("// kept", "// changed")) != _code_only(base)


def test_committed_traffic_records_match_the_kernel_sources():
    """bench.py quotes `roofline.traffic` only from a PMC record taken on the SAME kernel sources (profiles/conv_hbm_traffic.json, keyed by
    build.source_hash()): a kernel edit without a new tools/gpu_evidence.sh run would make the driver's bench line say
    traffic: null.  Records for the headline workload (config 2, f32) and the bf16 workload (config 5)."""
    import json
    import os
    from diffphore_amd.csrc import build
    recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "conv_hbm_traffic.json")))
    h = build.source_hash()
    mine = {(r["config"], r["precision"]) for r in recs if r["source_hash"] == h}
    assert (2, "f32") in mine and (5, "bf16") in mine, f"no PMC traffic record for kernel sources {h}: re-run tools/gpu_evidence.sh and commit profiles/"


def test_build_identity_and_ablation_builds_are_refused(monkeypatch):
    """dp_build_info: the product build carries no ablation / profiling switch and the hash of the sources it was compiled from (the hash
    bench.py keys PMC profiles with); `_lib.lib()` refuses a library that reports an ablation switch unless DIFFPHORE_ALLOW_ABLATION=1."""
    from diffphore_amd.csrc import build
    info = _lib.build_info()
    assert info["abi_version"] == _lib.DP_ABI_VERSION and info["nabl"] == 0 and info["ablate"] == 0 and info["prof"] == 0
    assert info["source_hash"] == build.source_hash(), "libdiffphore_hip.so is stale: rebuild with python diffphore_amd/csrc/build.py"
    L = _lib.lib()

    class FakeInfo:
        argtypes = restype = None

        def __call__(self, ref):
            L.dp_build_info(ref)
            ref._obj.nabl = 8
            return 0

    class Fake:
        """the real library with a dp_build_info that reports -DDP_NABL=8"""
        dp_build_info = FakeInfo()

        def __getattr__(self, n):
            return getattr(L, n)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(C, "CDLL", lambda path: Fake())
    monkeypatch.delenv("DIFFPHORE_ALLOW_ABLATION", raising=False)
    with pytest.raises(_lib.DiffPhoreLibraryError, match="ablation build"):
        _lib.lib()
    monkeypatch.setenv("DIFFPHORE_ALLOW_ABLATION", "1")
    assert _lib.lib() is not None
    monkeypatch.setattr(_lib, "_lib", None)


def test_workspace_defaults_are_positive_sense():
    """dp_work_bind initialises DpWork.edge_bf16x6 = 1 (ABI 9: 1 = the fp32-grade bf16x6 split on the edge-side convs of an fp32 model, 0 = exact fp32)
    and DpWork.node_f16 = 1 (ABI 10: 1 = the fp32-grade two-limb fp16 split on the node-side convs of an fp32 model, 0 = exact fp32 MFMA)"""
    L = engine._declare()
    b = engine.DpBatch()
    b.B, b.n_lig, b.n_ph, b.n_bond, b.n_tor, b.n_pp, b.n_x, b.ll_cap = 2, 10, 12, 20, 4, 30, 120, 300
    n = L.dp_work_bytes(C.byref(b))
    w = engine.DpWork()
    ptrs = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
    assert L.dp_work_bind(C.byref(b), 1 << 20, n, C.byref(w), *[C.byref(p) for p in ptrs]) == 0
    assert w.edge_bf16x6 == 1 and w.node_f16 == 1 and w.conv_fuse == 0 and w.aux_mode == 0


def test_struct_layouts_match_the_compiler():
    L = _lib.lib()
    out = (C.c_int32 * 7)()
    L.dp_struct_sizes.argtypes = [C.POINTER(C.c_int32), C.c_int32]
    assert L.dp_struct_sizes(out, 7) == 7
    want = [C.sizeof(_lib.DpConvArgs), C.sizeof(engine.DpBatch), C.sizeof(engine.DpWeights), C.sizeof(engine.DpWork),
            C.sizeof(engine.DpStepParams), C.sizeof(engine.DpMlp), C.sizeof(engine.DpSchedule)]
    assert list(out) == want


def test_workspace_and_blob_layout_queries_need_no_gpu():
    """dp_work_bytes / dp_work_bind / dp_weights_segments are pure host arithmetic: callable here, consistent with each other"""
    L = engine._declare()
    b = engine.DpBatch()
    assert L.dp_work_bytes(C.byref(b)) == 0                                   # B == 0 -> rejected
    b.B, b.n_lig, b.n_ph, b.n_bond, b.n_tor, b.n_pp, b.n_x, b.ll_cap, b.n_xs = 4, 120, 128, 250, 32, 900, 3840, 4200, 117
    n1 = L.dp_work_bytes(C.byref(b))
    assert n1 > 4 * (3840 * (20 + 9 + 9 + 1) + 4200 * 29 + 2 * 120 * 100)     # at least the big per-edge / per-node arrays
    w = engine.DpWork()
    ptrs = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
    base = 1 << 20                                                            # a fake (aligned) device address: bind touches no memory
    assert L.dp_work_bind(C.byref(b), base, n1, C.byref(w), *[C.byref(p) for p in ptrs]) == 0
    fields = [getattr(w, n) for n, _ in engine.DpWork._fields_ if n not in ("h_l", "h_p", "prof_start", "prof_stop", "prof_layer",
                                                                            "prof_conv", "prof_events", "conv_algo", "aux_stream", "ev_fork", "ev_join", "conv_fuse", "path_split", "edge_bf16x6", "ll_node_max_atoms", "aux_mode", "node_f16", "xn_sparse", "aux2_stream", "ev_join2", "ev_x2", "xs_place", "t_uniform", "ph_share", "t_row", "tp_row")]
    fields += list(w.h_l) + list(w.h_p) + [p.value for p in ptrs]
    assert all(f is not None and base <= f < base + n1 and f % 256 == 0 for f in fields) and len(set(fields)) == len(fields)
    assert L.dp_work_bind(C.byref(b), base, n1 - 1, C.byref(w), *[C.byref(p) for p in ptrs]) == -3
    assert L.dp_work_bind(C.byref(b), base + 8, n1, C.byref(w), *[C.byref(p) for p in ptrs]) == -1
    b.n_x *= 2
    assert L.dp_work_bytes(C.byref(b)) > n1
    for prec in _lib.PRECISIONS.values():
        n = L.dp_weights_segments(prec, None, 0)
        segs = (engine.DpSegment * n)()
        assert L.dp_weights_segments(prec, segs, n) == n
        end = 0
        for sg in segs:
            assert sg.offset >= end and sg.offset % 4 == 0 and sg.numel > 0
            end = sg.offset + sg.numel
        assert L.dp_weights_blob_floats(prec) >= end
        names = [sg.name.decode() for sg in segs]
        assert len(set(names)) == n and "p2l_conv.3.w2p" in names and ("p2l_conv.3.w2n" in names) == (prec in (_lib.PREC_F32, _lib.PREC_BF16))
        assert ("ph_conv.0.w2x" in names) == ("tor_conv.w2x" in names) == (prec == _lib.PREC_F32) and "ph_conv.3.w2x" not in names
        assert ("ph_conv.0.bnx" in names) == ("tor_conv.bnx" in names) == ("lig_conv.2.bnx" in names) == (prec == _lib.PREC_F32)
        # the node-side families of an fp32 model carry the two-limb fp16 images too (DP_PREC_F16X3), phore-phore (edge-side) does not
        assert ("p2l_conv.3.w2h" in names) == ("lig_conv.0.w1h" in names) == (prec == _lib.PREC_F32) and "ph_conv.0.w2h" not in names


def test_pack_bn_matches_the_python_packer():
    import numpy as np
    from diffphore_amd import config, packing, weights
    L = engine._declare()
    sd = weights.init_state_dict(3)
    irs = {2: config.SEQ[3], 4: config.FINAL_OUT, 5: config.TOR_OUT}
    for kind, prefix in ((2, "encoder.lig_conv_layers.2"), (4, "final_conv"), (5, "tor_bond_conv")):
        want_s, want_b = packing.bn_affine(sd, prefix + ".batch_norm", irs[kind])
        t = {k: sd[f"{prefix}.batch_norm.{k}"].float().contiguous() for k in ("running_mean", "running_var", "weight", "bias")}
        got_s, got_b = np.zeros(want_s.numel(), dtype=np.float32), np.zeros(want_s.numel(), dtype=np.float32)
        L.dp_pack_bn.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 2
        rc = L.dp_pack_bn(kind, t["running_mean"].data_ptr() if t["running_mean"].numel() else None, t["running_var"].data_ptr(),
                          t["weight"].data_ptr(), t["bias"].data_ptr() if t["bias"].numel() else None, 1e-5, got_s.ctypes.data,
                          got_b.ctypes.data)
        assert rc == 0
        np.testing.assert_allclose(got_s, want_s.numpy(), rtol=2e-7)
        np.testing.assert_allclose(got_b, want_b.numpy(), rtol=2e-6, atol=1e-7)


def test_bad_arguments_are_rejected_not_crashed():
    L = _lib.lib()
    assert L.dp_tp_conv(None, None) == -1
    a = _lib.DpConvArgs()
    assert L.dp_tp_conv(C.byref(a), None) == -1
    engine._declare()
    assert L.dp_forward(None, None, None, None, None, None, None) == -1
    assert L.dp_pack_conv_weights(99, 0, None, None, None, None, None, None) == -1


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    from diffphore_amd import config, weights
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DiffPhoreLibraryError):
        engine.DeviceModel(weights.init_state_dict(), config.ModelConfig(), "cpu")
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdiffphore_hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.DiffPhoreLibraryError):
        _lib.lib()


def test_philox_known_answers():
    """Philox-4x32-10 against the known-answer vectors published with Random123 (kat_vectors: philox4x32 10): the generator behind
    dp_noise_normals / dp_randomize_draws, evaluated by the library's host entry point (same code as the device function)."""
    L = _lib.lib()
    U4, U2 = C.c_uint32 * 4, C.c_uint32 * 2
    L.dp_philox4x32.argtypes = [U4, U2, U4]
    L.dp_philox4x32.restype = C.c_int
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        out = U4()
        assert L.dp_philox4x32(U4(*ctr), U2(*key), out) == 0
        assert tuple(out) == want, [hex(x) for x in out]


def test_rng_stream_fixture_words_and_box_muller():
    """tests/golden/rng_stream.npz (made on an MI355X by make_rng_golden.py) pins the seeded noise stream: the Philox words of every
    counter are exact integers - recomputed here through the host entry point - and the device's normals / uniform angles / rotation
    matrices follow from them by the documented construction (Box-Muller on ((x >> 9) + 0.5) * 2^-23, strictly inside (0, 1); csrc/dp_rng.hip), recomputed in
    numpy float64 and compared at 2e-6 (device logf / sincosf are not bit-specified)."""
    import numpy as np
    G = np.load(os.path.join(ROOT, "tests", "golden", "rng_stream.npz"))
    L = _lib.lib()
    U4, U2 = C.c_uint32 * 4, C.c_uint32 * 2
    L.dp_philox4x32.argtypes = [U4, U2, U4]
    seed = int(G["seed"])

    def words(uid, step, block, stream):
        out = U4()
        assert L.dp_philox4x32(U4(uid & 0xFFFFFFFF, step & 0xFFFFFFFF, block, stream), U2(seed & 0xFFFFFFFF, seed >> 32), out) == 0
        return np.asarray(list(out), dtype=np.uint32)

    def unit(w):
        return ((w >> 9).astype(np.float64) + 0.5) / 8388608.0

    def normals(w):
        u = unit(w)
        r0, r1 = np.sqrt(-2 * np.log(u[0])), np.sqrt(-2 * np.log(u[2]))
        return np.asarray([r0 * np.cos(2 * np.pi * u[1]), r0 * np.sin(2 * np.pi * u[1]), r1 * np.cos(2 * np.pi * u[3]), r1 * np.sin(2 * np.pi * u[3])])

    for st in G["steps"]:
        for i, uid in enumerate(G["uids"]):
            W = [words(int(uid), int(st), blk, strm) for (blk, strm) in ((0, 0), (1, 0), (0, 1), (1, 1))]
            assert np.array_equal(np.stack(W), G[f"words_{st}"][i])
            n0, n1, t0, t1 = (normals(w) for w in W)
            np.testing.assert_allclose(G[f"z_tr_{st}"][i], n0[:3], atol=2e-6, rtol=2e-6)
            np.testing.assert_allclose(G[f"z_rot_{st}"][i], [n0[3], n1[0], n1[1]], atol=2e-6, rtol=2e-6)
            np.testing.assert_allclose(G[f"z_tor_{st}"][i], np.concatenate([t0, t1[:3]]), atol=2e-6, rtol=2e-6)
    for i, uid in enumerate(G["uids"]):
        W = [words(int(uid), 0xFFFFFFFF, blk, strm) for (blk, strm) in ((0, 2), (1, 2), (0, 3), (1, 3))]
        assert np.array_equal(np.stack(W), G["words_start"][i])
        q = normals(W[0])
        x, y, z, w = q / np.linalg.norm(q)                  # scalar-last unit quaternion -> rotation matrix
        R = np.asarray([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(G["rot_mats"][i], R, atol=3e-6)
        np.testing.assert_allclose(G["tr0"][i], 5.0 * normals(W[1])[:3], atol=1e-5, rtol=2e-6)
        th = (2 * np.concatenate([unit(W[2]), unit(W[3])[:3]]) - 1) * np.pi
        np.testing.assert_allclose(G["theta"][i], th, atol=2e-6)
