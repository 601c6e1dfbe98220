"""A host in plain C (tests/c_host/dp_host_forward.c) sizes the workspace, binds the parameter blob and runs dp_forward through
include/diffphore_amd.h alone; its scores must equal the Python host's bit for bit (same kernels, same buffers' contents)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from diffphore_amd import _lib, config, synthetic, weights
from diffphore_amd.engine import DeviceModel, Engine, PackedBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "dp_host_forward.c")


def build_c_host(out_dir):
    exe = os.path.join(out_dir, "dp_host_forward")
    lib_dir = os.path.join(ROOT, "diffphore_amd", "csrc")
    cmd = ["gcc", "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L" + lib_dir, "-ldiffphore_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def write_dump(path, pb, model, conv_algo):
    def rec(f, name, arr):
        arr = np.ascontiguousarray(arr)
        code = {np.dtype("float32"): 0, np.dtype("int32"): 1, np.dtype("uint8"): 2}[arr.dtype]
        f.write(name.encode().ljust(64, b"\0"))
        f.write(struct.pack("<iq", code, arr.size))
        f.write(arr.tobytes())
    s = pb.struct
    with open(path, "wb") as f:
        rec(f, "dims", np.asarray([s.B, s.n_lig, s.n_ph, s.n_bond, s.n_tor, s.n_pp, s.n_x, s.ll_cap, s.n_llp, s.max_nl, s.max_np, s.max_t, s.no_tail_nl, s.no_tail_np, s.n_xs, s.ph_shared], dtype=np.int32))
        for k, v in pb.static.items():
            rec(f, k, v.cpu().numpy())
        rec(f, "pos", pb.pos.cpu().numpy())
        rec(f, "norm", pb.norm.cpu().numpy())
        rec(f, "t", pb.t.cpu().numpy())
        rec(f, "blob", model.blob.cpu().numpy())
        rec(f, "scalars", np.frombuffer(bytes(model.scalars), dtype=np.float32))
        rec(f, "precision", np.asarray([_lib.PRECISIONS[model.precision]], dtype=np.int32))
        rec(f, "conv_algo", np.asarray([conv_algo], dtype=np.int32))


def test_header_is_plain_c_and_host_program_links(tmp_path):
    """(no GPU needed) the C ABI header compiles as C11 and every entry point the C host uses resolves at link time"""
    build_c_host(str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["edge", "node"])
def test_c_host_forward_equals_python_host(tmp_path, algo):
    gs = synthetic.synthetic_pairs(71, 5, (9, 34), (0, 7), (6, 40))
    cfg = config.ModelConfig()
    model = DeviceModel(weights.init_state_dict(), cfg, "cuda")
    pb = PackedBatch(gs, "cuda")
    pb.set_time(0.7, 0.7, 0.7)
    eng = Engine(model, pb, conv_algo=algo)
    tr, rot, tor = eng.forward()
    torch.cuda.synchronize()
    want = np.concatenate([tr.cpu().numpy().ravel(), rot.cpu().numpy().ravel(), tor.cpu().numpy().ravel()])
    exe = build_c_host(str(tmp_path))
    dump, out = str(tmp_path / "dump.bin"), str(tmp_path / "out.bin")
    write_dump(dump, pb, model, {"edge": 1, "node": 2}[algo])
    res = subprocess.run([exe, dump, out], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr + res.stdout
    got = np.fromfile(out, dtype=np.float32)
    assert got.shape == want.shape and np.all(np.isfinite(got))
    assert np.array_equal(got, want), float(np.abs(got - want).max())


@pytest.mark.gpu
def test_c_host_range_guard_replays_on_exact_fp32_only_for_status_bit_2(tmp_path):
    """The range guard of the two-limb fp16 kernels as a C caller sees it: dp_conformer_update raises status bit 2 when the SCORES it
    is handed are non-finite while the pose was fine.  On a model whose atom embeddings are scaled by 10 (the limb kernels leave fp16's range, every
    fp32 product stays finite) the C host sees bit 2, restores the start state, switches DpWork.node_f16 / edge_bf16x6 off and ends with the scores
    of the exact-fp32 kernels, bit for bit; on the plain model nothing is flagged and nothing is replayed; a pose that only has a degenerate
    Kabsch alignment (finite scores, bit 0) is not replayed either."""
    gs = synthetic.synthetic_pairs(4242, 6, (12, 20), (0, 4), (8, 16))
    cfg = config.ModelConfig()
    exe = build_c_host(str(tmp_path))
    for scale, expect in ((10.0, "over_range=1 kabsch=0 replayed=1"), (1.0, "over_range=0 kabsch=0 replayed=0")):
        sd = weights.init_state_dict()
        for k in sd:
            if k.startswith("encoder.lig_node_embedding.atom_embedding_list"):
                sd[k] = sd[k] * scale
        model = DeviceModel(sd, cfg, "cuda")
        pb = PackedBatch(gs, "cuda")
        pb.set_time(0.7, 0.7, 0.7)
        ref = Engine(model, pb, conv_algo="node", exact=(scale != 1.0))      # what the C host must end with: exact kernels iff it had to replay
        tr, rot, tor = ref.forward()
        torch.cuda.synchronize()
        want = np.concatenate([tr.cpu().numpy().ravel(), rot.cpu().numpy().ravel(), tor.cpu().numpy().ravel()])
        assert np.all(np.isfinite(want))
        if scale != 1.0:
            limb = Engine(model, pb, conv_algo="node")
            assert limb.limb_kernels() and not all(bool(torch.isfinite(x).all()) for x in limb.forward())      # the limb kernels really overflow here
        dump, out = str(tmp_path / f"dump{scale}.bin"), str(tmp_path / f"out{scale}.bin")
        write_dump(dump, pb, model, 2)
        res = subprocess.run([exe, dump, out, "guard"], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr + res.stdout
        assert expect in res.stdout, res.stdout
        got = np.fromfile(out, dtype=np.float32)
        assert np.array_equal(got, want), float(np.abs(got - want).max())
    # bit 0 alone: finite scores, but a ligand whose atoms coincide - no replay (status through the Python binding of the same entry points)
    from diffphore_amd.engine import DpStepParams
    model = DeviceModel(weights.init_state_dict(), cfg, "cuda")
    pb = PackedBatch(gs, "cuda")
    pb.set_time(0.7, 0.7, 0.7)
    eng = Engine(model, pb, conv_algo="node")
    eng.forward()
    sp = DpStepParams()
    sp.tr_c1, sp.rot_c1, sp.tor_c1 = 0.05, 0.02, 0.03
    eng.update(sp)
    torch.cuda.synchronize()
    assert int(eng.ws.t["status"].abs().sum()) == 0
    eng.ws.tr[1, 0] = float("inf")                                            # a non-finite score for pose 1 ...
    eng.update(sp)
    torch.cuda.synchronize()
    st = eng.ws.t["status"].cpu().tolist()
    assert st[1] == 4 and sum(st) == 4, st                                   # ... is bit 2 for that pose only, not bit 0
    eng.ws.tr[1, 0] = 0.0
    eng.update(sp)                                                            # the pose is non-finite now, the scores are not: the first cause stays
    torch.cuda.synchronize()
    assert eng.ws.t["status"].cpu().tolist() == st
