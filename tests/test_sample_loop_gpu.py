"""dp_sample (the whole reverse-diffusion loop in one C call), its hipGraph replay and dp_forward's optional second stream are
re-arrangements of the SAME launches: final coordinates must be bit-identical to the per-step path (sampling.py:204-255 loop body)."""
import copy
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffphore_amd import config, synthetic, weights  # noqa: E402
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn  # noqa: E402
from diffphore_amd.sampling import sampling_phore  # noqa: E402
from diffphore_amd.score_model import TensorProductScoreModel  # noqa: E402


def _args():
    cfg = config.ModelConfig()
    return types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)


def _run(gs, noise, S, monkeypatch, **env):
    for k in ("DIFFPHORE_WHOLE_LOOP", "DIFFPHORE_GRAPH", "DIFFPHORE_TWO_STREAMS", "DIFFPHORE_CONV_ALGO", "DIFFPHORE_XN_SPARSE", "DIFFPHORE_PATH_SPLIT", "DIFFPHORE_PP_LIST", "DIFFPHORE_PH_SHARE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args = _args()
    m = TensorProductScoreModel(device="cuda").load_state_dict(weights.init_state_dict())
    sched = get_t_schedule(20)[:S + 1]
    out, _ = sampling_phore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, batch_size=4, noise=noise,
                            no_final_step_noise=True)
    return torch.cat([g["ligand"].pos for g in out]).numpy(), torch.cat([g["ligand"].norm for g in out]).numpy()


@pytest.mark.parametrize("algo", ["edge", "node", "node-large"])
def test_whole_loop_graph_and_two_streams_are_bit_identical_to_per_step(monkeypatch, algo):
    """`node-large`: the forms a batch that fills the chip takes (round 6) forced onto this small one - fused launches instead of the conv split, the
    normals convs as s0-only launch + listed pairs, the phore-phore convs as one list launch (already this batch's choice) - through the same drivers"""
    large = dict(DIFFPHORE_XN_SPARSE="1", DIFFPHORE_PATH_SPLIT="0") if algo == "node-large" else {}
    algo = "node" if algo == "node-large" else algo

    def run(gs, noise, S, monkeypatch, **env):        # every arm of this case carries the same switches
        return _run(gs, noise, S, monkeypatch, **dict(large, **env))
    gs = synthetic.synthetic_pairs(91, 7, (8, 30), (0, 6), (5, 34))
    S = 5
    gen = torch.Generator().manual_seed(3)
    nT = sum(int(g["ligand"].edge_mask.sum()) for g in gs)
    noise = dict(tr=torch.randn(S, len(gs), 3, generator=gen), rot=torch.randn(S, len(gs), 3, generator=gen), tor=torch.randn(S, nT, generator=gen))
    base = run(gs, noise, S, monkeypatch, DIFFPHORE_WHOLE_LOOP="0", DIFFPHORE_TWO_STREAMS="0", DIFFPHORE_CONV_ALGO=algo)
    assert np.all(np.isfinite(base[0]))
    variants = {
        "dp_sample eager": dict(DIFFPHORE_GRAPH="0", DIFFPHORE_TWO_STREAMS="0"),
        "dp_sample graph": dict(DIFFPHORE_GRAPH="1", DIFFPHORE_TWO_STREAMS="0"),
        "two streams, per step": dict(DIFFPHORE_WHOLE_LOOP="0", DIFFPHORE_TWO_STREAMS="1"),
        "two streams, graph": dict(DIFFPHORE_GRAPH="1", DIFFPHORE_TWO_STREAMS="1"),
    }
    for name, env in variants.items():
        got = run(gs, noise, S, monkeypatch, DIFFPHORE_CONV_ALGO=algo, **env)
        assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), name


def test_samples_of_one_complex_share_the_layer0_phore_conv_in_every_driver(monkeypatch):
    """the reference's own call shape (inference.py:183-209: N samples of ONE complex): every driver states the uniform times (dp_sample itself, the
    per-step loop through PackedBatch.set_time_row), the batch is on one pharmacophore, the layer-0 phore-phore conv runs once per step - and the
    trajectories are those of the per-pose conv, bit for bit (both are the list launch + dp_list_finish on this shape)"""
    g0 = synthetic.synthetic_pairs(96, 1, 18, 4, 14)[0]
    gs = [g0.sample_copy() for _ in range(8)]
    rng = np.random.default_rng(11)
    for g in gs:
        g["ligand"].pos = g["ligand"].pos + torch.from_numpy(rng.normal(0, 0.5, size=tuple(g["ligand"].pos.shape)).astype(np.float32))
    S = 4
    gen = torch.Generator().manual_seed(4)
    nT = sum(int(g["ligand"].edge_mask.sum()) for g in gs)
    noise = dict(tr=torch.randn(S, len(gs), 3, generator=gen), rot=torch.randn(S, len(gs), 3, generator=gen), tor=torch.randn(S, nT, generator=gen))
    base = _run(gs, noise, S, monkeypatch, DIFFPHORE_WHOLE_LOOP="0", DIFFPHORE_TWO_STREAMS="0", DIFFPHORE_PH_SHARE="0")
    assert np.all(np.isfinite(base[0]))
    for name, env in {"per step": dict(DIFFPHORE_WHOLE_LOOP="0", DIFFPHORE_TWO_STREAMS="0"), "dp_sample eager": dict(DIFFPHORE_GRAPH="0", DIFFPHORE_TWO_STREAMS="0"),
                      "two streams, graph": dict(DIFFPHORE_GRAPH="1", DIFFPHORE_TWO_STREAMS="1"),
                      "large-batch forms": dict(DIFFPHORE_XN_SPARSE="1", DIFFPHORE_PATH_SPLIT="0", DIFFPHORE_GRAPH="1")}.items():
        got = _run(gs, noise, S, monkeypatch, **env)
        if name == "large-batch forms":             # (other launches than `base`: the same poses within the kernels' tolerance)
            assert np.abs(got[0] - base[0]).max() < 1e-4, name
        else:
            assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), name


def test_graph_replay_reproduces_the_run():
    """replaying the captured loop after resetting the state gives the same trajectory (what a screening driver does per batch)"""
    from diffphore_amd.engine import DeviceModel, Engine, PackedBatch
    from diffphore_amd.sampling import _step_params
    args = _args()
    gs = synthetic.synthetic_pairs(92, 4, 14, 3, 12)
    model = DeviceModel(weights.init_state_dict(), config.ModelConfig(), "cuda")
    pb = PackedBatch(gs, "cuda")
    eng = Engine(model, pb)
    S = 4
    sched = get_t_schedule(20)
    sps = [_step_params(t, S, sched, sched, sched, t_to_sigma_fn(args), args, False, False) for t in range(S)]
    T = torch.tensor([[float(sched[t])] * 3 for t in range(S)], device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    ztr, zrot = torch.randn(S, pb.B, 3, device="cuda", generator=g), torch.randn(S, pb.B, 3, device="cuda", generator=g)
    ztor = torch.randn(S, int(pb.tor_ptr[-1]), device="cuda", generator=g)
    pos0, norm0 = pb.pos.clone(), pb.norm.clone()
    eng.sample(sps, T, ztr, zrot, ztor, graph=True)
    a = pb.pos.clone()
    pb.pos.copy_(pos0)
    pb.norm.copy_(norm0)
    eng.replay()
    torch.cuda.synchronize()
    assert torch.equal(pb.pos, a) and not torch.equal(a, pos0)
    assert int(eng.ws.t["status"].abs().sum()) == 0


def test_visualization_recorders_follow_their_poses_across_device_batches():
    """`visualization_list` (one recorder per pose, `.add(coords, part=1, order=step + 2)` after every step, sampling.py:251-254) with more
    poses than `max_poses`: every device batch feeds the recorders of ITS poses - same frames as a single-batch run."""
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sampling_phore
    from diffphore_amd.score_model import TensorProductScoreModel

    class Rec:
        def __init__(self):
            self.frames = []

        def add(self, coords, part, order):
            self.frames.append((order, part, coords.numpy().copy()))

    gs = synthetic.synthetic_pairs(77, 6, (14, 22), (2, 4), (10, 20))
    cfg = config.ModelConfig()
    args = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
    S = 3
    sched = get_t_schedule(20)[:S + 1]
    m = TensorProductScoreModel(device="cuda").load_state_dict(weights.init_state_dict())
    g0 = torch.Generator().manual_seed(5)
    nT = sum(int(g["ligand"].edge_mask.sum()) for g in gs)
    noise = dict(tr=torch.randn(S, 6, 3, generator=g0), rot=torch.randn(S, 6, 3, generator=g0), tor=torch.randn(S, nT, generator=g0))
    runs = []
    for mp in (8192, 4):
        recs = [Rec() for _ in gs]
        out, _ = sampling_phore([g.sample_copy() for g in gs], m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, noise=noise,
                                visualization_list=recs, max_poses=mp)
        assert all(len(r.frames) == S and [f[0] for f in r.frames] == [2, 3, 4] for r in recs)
        # the last frame is the final pose in the original frame
        for r, g in zip(recs, out):
            np.testing.assert_allclose(r.frames[-1][2], (g["ligand"].pos + g.original_center).numpy(), atol=1e-6)
        runs.append(recs)
    for a, b in zip(*runs):
        for fa, fb in zip(a.frames, b.frames):
            assert np.abs(fa[2] - fb[2]).max() < 2e-5
