"""On-device noise (dp_noise_normals / dp_randomize_draws, SURVEY.md 8b `seed` path of dp_sample) and the streaming screen driver
(distributed.screen).

The seeded path is NOT the reference's random stream (that is what `noise=` / the global generators are for); what it must be is
    * statistically right: N(0, 1) normals, U(-pi, pi) torsions, Haar-uniform rotations (scipy KS tests on ~10^5 draws),
    * a pure function of (seed, pose uid, step, component): the same pose gets the same trajectory whatever the batch composition,
      the device-batch size, the order of the stream or the number of ranks - bit for bit on the noise, <= 2e-5 A on trajectories
      (the batch composition changes the workgroup slicing of the edge-side convs, not the noise),
    * built from the Philox words the host entry point returns (tests/test_abi_cpu.py pins those to the published vectors).
"""
import ctypes as C
import os
import socket
import types

import numpy as np
import pytest
import torch
from scipy import stats

pytestmark = pytest.mark.gpu

from diffphore_amd import config, synthetic, weights  # noqa: E402


def _args():
    cfg = config.ModelConfig()
    return types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False, inference_steps=3)


def _rmsd(a, b):
    return float(np.sqrt(((np.asarray(a) - np.asarray(b)) ** 2).sum(-1).mean()))


def _draw(pb, seed, step, uid=None, uid_base=0):
    from diffphore_amd import _lib
    from diffphore_amd.engine import _declare, _stream
    L = _declare()
    B, nT = pb.B, int(pb.tor_ptr[-1])
    ztr, zrot, ztor = torch.empty(B, 3, device="cuda"), torch.empty(B, 3, device="cuda"), torch.empty(max(nT, 1), device="cuda")
    _lib.check(L.dp_noise_normals(seed, step, C.byref(pb.struct), _lib.ptr(uid), uid_base, ztr.data_ptr(), zrot.data_ptr(), ztor.data_ptr(),
                                  _stream()), "dp_noise_normals")
    torch.cuda.synchronize()
    return ztr.cpu().numpy(), zrot.cpu().numpy(), ztor.cpu().numpy()[:nT]


_POOL = []


def _pool():
    """300 small complexes (seconds to build), shared by the statistics tests: sample counts come from steps / seeds, not from poses"""
    if not _POOL:
        _POOL.extend(synthetic.synthetic_pairs(41, 300, (15, 30), (2, 5), 12))
    return _POOL


def test_step_normals_are_standard_normal_and_keyed_by_uid_and_step():
    from diffphore_amd import _lib
    from diffphore_amd.engine import PackedBatch
    gs = _pool()
    pb = PackedBatch(gs, "cuda")
    a = _draw(pb, 1234, 0)
    steps = [a] + [_draw(pb, 1234, s_) for s_ in range(1, 12)]
    allz = np.concatenate([x.reshape(-1) for st in steps for x in st])
    assert allz.size > 25000 and np.isfinite(allz).all()
    print("n =", allz.size, "mean %.4f var %.4f skew %.4f kurt %.4f" % (allz.mean(), allz.var(), stats.skew(allz), stats.kurtosis(allz)))
    assert abs(allz.mean()) < 4 / np.sqrt(allz.size) and abs(allz.var() - 1) < 0.03
    assert stats.kstest(allz, "norm").pvalue > 1e-3
    for t in range(3):                                                      # each table on its own, too
        assert stats.kstest(np.concatenate([st[t].reshape(-1) for st in steps]), "norm").pvalue > 1e-3
    # components of one pose are independent draws (no word is used twice): correlation of z_tr vs z_rot columns ~ 0
    cc = np.corrcoef(np.concatenate([np.concatenate([st[0], st[1]], 1) for st in steps]).T)
    assert np.abs(cc - np.eye(6)).max() < 5 / np.sqrt(len(gs) * len(steps))
    # same (seed, step) -> same numbers; another step or seed -> fresh ones
    b = _draw(pb, 1234, 0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    c, d = steps[1], _draw(pb, 1235, 0)
    assert abs(np.corrcoef(a[0].reshape(-1), c[0].reshape(-1))[0, 1]) < 0.15 and abs(np.corrcoef(a[2], d[2])[0, 1]) < 0.15
    # keyed by the pose uid, not by the position in the batch: a permuted batch with the uids carried along draws the same numbers
    perm = np.random.default_rng(0).permutation(len(gs))
    pb2 = PackedBatch([gs[i] for i in perm], "cuda")
    uid = torch.from_numpy(perm.astype(np.int32)).cuda()
    e = _draw(pb2, 1234, 0, uid=uid)
    assert np.array_equal(e[0], a[0][perm]) and np.array_equal(e[1], a[1][perm])
    tp = pb.tor_ptr
    assert np.array_equal(e[2], np.concatenate([a[2][int(tp[i]):int(tp[i + 1])] for i in perm]))
    # uid_base shifts the default uids
    f = _draw(PackedBatch(gs[100:110], "cuda"), 1234, 0, uid_base=100)
    assert np.array_equal(f[0], a[0][100:110])
    # the first normal of pose 0 from the host-side Philox words (Box-Muller on ((x >> 9) + 0.5) * 2^-23)
    U4, U2 = C.c_uint32 * 4, C.c_uint32 * 2
    L = _lib.lib()
    L.dp_philox4x32.argtypes = [U4, U2, U4]
    out = U4()
    L.dp_philox4x32(U4(0, 0, 0, 0), U2(1234, 0), out)
    u = [((x >> 9) + 0.5) / 8388608.0 for x in out]
    want = [np.sqrt(-2 * np.log(u[0])) * np.cos(2 * np.pi * u[1]), np.sqrt(-2 * np.log(u[0])) * np.sin(2 * np.pi * u[1]),
            np.sqrt(-2 * np.log(u[2])) * np.cos(2 * np.pi * u[3]), np.sqrt(-2 * np.log(u[2])) * np.sin(2 * np.pi * u[3])]
    got = [a[0][0, 0], a[0][0, 1], a[0][0, 2], a[1][0, 0]]
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)


def test_start_randomisation_draws_are_uniform():
    from diffphore_amd import _lib
    from diffphore_amd.engine import PackedBatch, _declare, _stream
    gs = _pool()
    pb = PackedBatch(gs, "cuda")
    B, nT = pb.B, int(pb.tor_ptr[-1])
    L = _declare()
    ths, rms, trs = [], [], []
    for seed in range(77, 87):                                              # 10 seeds x 300 poses
        th, rm, tr = torch.empty(nT, device="cuda"), torch.empty(B, 3, 3, device="cuda"), torch.empty(B, 3, device="cuda")
        _lib.check(L.dp_randomize_draws(seed, C.byref(pb.struct), None, 0, 5.0, th.data_ptr(), rm.data_ptr(), tr.data_ptr(), _stream()), "draws")
        torch.cuda.synchronize()
        ths.append(th.cpu().numpy()), rms.append(rm.cpu().numpy().astype(np.float64)), trs.append(tr.cpu().numpy())
    th, rm, tr = np.concatenate(ths), np.concatenate(rms), np.concatenate(trs)
    assert np.abs(th).max() < np.pi and stats.kstest(th, "uniform", args=(-np.pi, 2 * np.pi)).pvalue > 1e-3
    assert stats.kstest(tr.reshape(-1) / 5.0, "norm").pvalue > 1e-3
    # proper rotations ...
    assert np.abs(rm @ rm.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5 and np.abs(np.linalg.det(rm) - 1).max() < 1e-5
    # ... Haar-uniform: the image of a fixed vector is uniform on the sphere (z ~ U(-1, 1)), and the rotation angle has density
    # (1 - cos w) / pi on [0, pi]
    z = rm[:, 2, 0]
    assert stats.kstest(z, "uniform", args=(-1, 2)).pvalue > 1e-3
    w = np.arccos(np.clip((np.trace(rm, axis1=1, axis2=2) - 1) / 2, -1, 1))
    assert stats.kstest(w, lambda x: (x - np.sin(x)) / np.pi).pvalue > 1e-3


def _model():
    from diffphore_amd.score_model import TensorProductScoreModel
    return TensorProductScoreModel(device="cuda").load_state_dict(weights.init_state_dict())


def test_seeded_sampler_is_independent_of_batching_and_order():
    """sampling_phore(seed=...) with start randomisation on the device: one device batch vs three (max_poses), and a shuffled pose list
    carrying its uids - the same pose ends at the same place."""
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sampling_phore
    gs = synthetic.synthetic_pairs(43, 12, (14, 30), (2, 4), (10, 40))
    args, S = _args(), 3
    sched = get_t_schedule(20)[:S + 1]
    m = _model()
    rz = dict(no_torsion=False, no_random=False, tr_sigma_max=args.tr_sigma_max)

    def run(order, **kw):
        poses = [gs[i].sample_copy() for i in order]
        out, _ = sampling_phore(poses, m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, seed=99, randomize=rz,
                                pose_uid=list(order), **kw)
        return {i: g["ligand"].pos.numpy().copy() for i, g in zip(order, out)}

    base = run(list(range(12)))
    assert all(np.isfinite(v).all() for v in base.values())
    small = run(list(range(12)), max_poses=5)
    shuf = run([7, 3, 11, 0, 5, 9, 1, 2, 10, 4, 8, 6], max_poses=7)
    for other, what in ((small, "three device batches"), (shuf, "shuffled + two device batches")):
        worst = max(_rmsd(base[i], other[i]) for i in range(12))
        print(what, "worst RMSD vs one batch (A):", worst)
        assert worst < 2e-5
    # another seed gives other trajectories
    poses = [gs[i].sample_copy() for i in range(12)]
    out, _ = sampling_phore(poses, m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, seed=100, randomize=rz)
    assert min(_rmsd(base[i], out[i]["ligand"].pos.numpy()) for i in range(12)) > 0.1
    # the step-by-step host loop (keep_update path) consumes the same device draws as the whole-loop call
    os.environ["DIFFPHORE_WHOLE_LOOP"] = "0"
    try:
        step = run(list(range(12)))
    finally:
        os.environ.pop("DIFFPHORE_WHOLE_LOOP")
    assert max(_rmsd(base[i], step[i]) for i in range(12)) < 2e-5


def test_seeded_loop_captured_into_a_hipgraph_equals_the_eager_loop(monkeypatch):
    """DP_NOISE_SEED inside dp_sample_graph_create: the per-step dp_noise_normals launches are captured with their step index baked in;
    a replay draws the same numbers as the eager loop (bit-identical final poses)."""
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sampling_phore
    gs = synthetic.synthetic_pairs(45, 5, (14, 24), (2, 4), (10, 24))
    args, S = _args(), 4
    sched = get_t_schedule(20)[:S + 1]
    m = _model()
    res = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("DIFFPHORE_GRAPH", graph)
        out, _ = sampling_phore([g.sample_copy() for g in gs], m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, seed=2024)
        res[graph] = torch.cat([g["ligand"].pos for g in out])
    assert torch.isfinite(res["0"]).all() and torch.equal(res["0"], res["1"])


def test_seeded_poses_of_one_complex_differ():
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sampling_phore
    g = synthetic.synthetic_pairs(44, 1, 20, 5, 16)[0]
    args, S = _args(), 2
    sched = get_t_schedule(20)[:S + 1]
    poses = [g.sample_copy() for _ in range(6)]
    out, _ = sampling_phore(poses, _model(), S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, seed=5,
                            randomize=dict(no_torsion=False, no_random=False, tr_sigma_max=args.tr_sigma_max), pose_uid=[0, 1, 2, 3, 4, 0])
    p = [x["ligand"].pos.numpy() for x in out]
    assert _rmsd(p[0], p[5]) < 1e-6                      # same uid -> same pose
    assert min(_rmsd(p[i], p[j]) for i in range(5) for j in range(i)) > 0.5


def test_screen_stream_equals_one_call_and_keeps_two_batches_alive():
    """distributed.screen over a lazily built library: equal to sampling_phore on the materialised pose list (same seed / uids), the
    loader is called once per owned unit, and never more than two device batches' worth of units are alive at once."""
    import gc
    import weakref

    from diffphore_amd import distributed as D
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sampling_phore
    lib = synthetic.ScreenLibrary(4, 9)
    args, NS = _args(), 4
    m = _model()
    alive, calls, peak = [], [], [0]

    def load(i):
        g = lib.unit(i)
        calls.append(i)
        alive.append(weakref.ref(g))
        gc.collect()
        peak[0] = max(peak[0], sum(r() is not None for r in alive))
        return g

    res = {}
    mine, coll = D.screen(len(lib), load, m, args, costs=[lib.cost(i) for i in range(len(lib))], samples_per_complex=NS, max_poses=2 * NS,
                          seed=321, on_result=lambda i, arr: res.__setitem__(i, arr))
    assert coll is None and sorted(mine) == list(range(9)) and sorted(calls) == list(range(9)) and sorted(res) == list(range(9))
    print("units alive at once (2 units per device batch):", peak[0])
    assert peak[0] <= 4 + 1                      # current batch + the staged next one (+ the unit being built)
    poses = [lib.unit(i).sample_copy() for i in range(9) for _ in range(NS)]
    sched = get_t_schedule(args.inference_steps)
    out, _ = sampling_phore(poses, m, args.inference_steps, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, seed=321,
                            randomize=dict(no_torsion=False, no_random=False, tr_sigma_max=args.tr_sigma_max),
                            pose_uid=[i * NS + s for i in range(9) for s in range(NS)])
    worst = max(_rmsd(res[i][s], out[i * NS + s]["ligand"].pos.numpy()) for i in range(9) for s in range(NS))
    print("worst RMSD streamed vs one call (A):", worst)
    assert worst < 2e-5 and all(r.shape == (NS, lib.sizes(i)[0], 3) for i, r in res.items())


@pytest.mark.parametrize("kernels", ["f32", "f16x3", "bf16"])
def test_screen_isolates_failing_units(monkeypatch, kernels):
    """(`kernels`: the node-side conv kernels the batch runs - the exact-fp32 ones (these 6-pose batches take the path split), the two-limb fp16
    ones (path split off) or the bf16 model's.  ADVICE r4: the 16-bit kernels gathered the rows behind a pose's last node from the NEXT pose of
    the batch; with a NaN unit next to a healthy one that leaked 0 x NaN into the healthy unit - their gathers are now limited to the pose.)
    ADVICE r3: one bad ligand must not end a library run (the reference wraps every complex in try / except and retries SVD failures,
    src/inference.py:200-219).  A unit whose loader raises, a unit over the kernel limits and a unit whose Kabsch alignment can never
    succeed (non-finite coordinates: flagged on every retry) are reported as failures - exactly once each, with their reasons - while
    every other unit is delivered and equals the run of a clean library (the noise is keyed by seed and pose uid)."""
    from diffphore_amd import distributed as D
    lib = synthetic.ScreenLibrary(4, 8)
    args, NS = _args(), 3
    if kernels == "f16x3":
        monkeypatch.setenv("DIFFPHORE_PATH_SPLIT", "0")
    if kernels == "bf16":
        from diffphore_amd.score_model import TensorProductScoreModel
        m = TensorProductScoreModel(device="cuda", precision="bf16").load_state_dict(weights.init_state_dict())
    else:
        m = _model()
    big = max(range(8), key=lambda i: lib.sizes(i)[0])                # the unit with the most atoms plays "over the limits"
    monkeypatch.setitem(D.LIMITS, "atoms", lib.sizes(big)[0] - 1)
    if any(lib.sizes(i)[0] > D.LIMITS["atoms"] for i in range(8) if i != big):
        pytest.skip("library has several units at the largest size")
    others = [i for i in range(8) if i != big]
    raises, nan_unit = others[0], others[1]

    def load(i):
        if i == raises:
            raise OSError("ligand file is corrupt")
        g = lib.unit(i)
        if i == nan_unit:
            g["ligand"].pos = g["ligand"].pos * float("nan")
        return g

    got, failed, calls = {}, [], []
    mine, coll = D.screen(8, load, m, args, costs=[lib.cost(i) for i in range(8)], samples_per_complex=NS, max_poses=2 * NS, seed=77,
                          on_result=lambda i, arr: (calls.append(i), got.__setitem__(i, arr)), failed=failed, retries=2)
    assert coll is None and sorted(calls) == list(range(8)), calls                      # every unit reported exactly once
    why = dict(failed)
    assert sorted(why) == sorted([raises, nan_unit, big]) and all(got[i] is None for i in why)
    assert "load_unit raised OSError" in why[raises] and "kernel limits" in why[big] and "Kabsch" in why[nan_unit]
    good = [i for i in range(8) if i not in why]
    clean = {}
    monkeypatch.setitem(D.LIMITS, "atoms", 256)
    D.screen(8, lib.unit, m, args, costs=[lib.cost(i) for i in range(8)], samples_per_complex=NS, max_poses=2 * NS, seed=77,
             on_result=lambda i, arr: clean.__setitem__(i, arr))
    worst = max(_rmsd(got[i][s], clean[i][s]) for i in good for s in range(NS))
    print("worst RMSD of the surviving units vs the clean library (A):", worst)
    assert worst < 2e-5
    # collected mode: None entries in processing order
    monkeypatch.setitem(D.LIMITS, "atoms", lib.sizes(big)[0] - 1)
    mine2, coll2 = D.screen(8, load, m, args, costs=[lib.cost(i) for i in range(8)], samples_per_complex=NS, max_poses=2 * NS, seed=77, retries=0)
    assert [c is None for c in coll2] == [i in why for i in mine2]


def test_screen_reruns_range_overflow_units_on_exact_fp32(monkeypatch):
    """An fp32 model whose features leave the two-limb fp16 kernels' range (atom embeddings x 10: the limb kernels go non-finite, every fp32
    product stays finite - tests/test_forward_gpu.py): `screen()` runs the flagged complexes once more with the SAME noise key on the exact-fp32
    kernels before any fresh-key retry - every unit is delivered, and with what a screen on the exact kernels delivers (same seed: bit for
    bit up to the batching, which a pose's result does not depend on beyond fp32 rounding).  With the rerun switched off the units fail."""
    from diffphore_amd import distributed as D
    from diffphore_amd import engine
    from diffphore_amd.score_model import TensorProductScoreModel
    sd = weights.init_state_dict()
    for k in sd:
        if k.startswith("encoder.lig_node_embedding.atom_embedding_list"):
            sd[k] = sd[k] * 10.0
    m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
    units = [synthetic.synthetic_pairs(900 + i, 1, 12 + i, 0, 9 + i)[0] for i in range(5)]
    args, NS = _args(), 2
    args.inference_steps = 1
    made = []
    real = engine.Engine.__init__

    def spy(self, model, pb, conv_algo=None, exact=False):
        made.append((pb.B, bool(exact)))
        real(self, model, pb, conv_algo=conv_algo, exact=exact)
    monkeypatch.setattr(engine.Engine, "__init__", spy)

    def go(**kw):
        got, failed = {}, []
        D.screen(5, lambda i: units[i], m, args, samples_per_complex=NS, max_poses=4 * NS, seed=31, on_result=got.__setitem__, failed=failed, **kw)
        return got, failed
    for k in ("DIFFPHORE_NODE_F16", "DIFFPHORE_EDGE_X6"):
        monkeypatch.setenv(k, "0")
    exact, f0 = go()
    assert not f0 and all(np.isfinite(exact[i]).all() for i in range(5))
    for k in ("DIFFPHORE_NODE_F16", "DIFFPHORE_EDGE_X6"):
        monkeypatch.delenv(k)
    made.clear()
    got, f1 = go()
    assert not f1 and sorted(got) == list(range(5))
    assert [e for _, e in made] == [False, False, True, True], made        # two limb batches (4 + 1 units), then the 5 flagged units again, exact
    worst = max(_rmsd(got[i][s_], exact[i][s_]) / max(1.0, float(np.abs(exact[i][s_]).max())) for i in range(5) for s_ in range(NS))
    print("worst relative deviation from the exact screen:", worst)
    assert worst < 1e-5
    monkeypatch.setenv("DIFFPHORE_EXACT_REPLAY", "0")
    got2, f2 = go(retries=1)
    assert sorted(i for i, _ in f2) == list(range(5)) and all(got2[i] is None for i in range(5))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _screen_worker(rank, world, port, q):
    import torch.distributed as dist
    from diffphore_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lib = synthetic.ScreenLibrary(4, 7)
        built = []

        def load(i):
            built.append(i)
            return lib.unit(i)
        res = {}
        mine, _ = D.screen(len(lib), load, _model(), _args(), costs=[lib.cost(i) for i in range(len(lib))], samples_per_complex=3,
                           max_poses=6, seed=11, on_result=lambda i, arr: res.__setitem__(i, arr))
        assert sorted(built) == sorted(mine)                    # a rank builds its own units only
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(res, gathered, dst=0)
        if rank == 0:
            q.put({k: v for d in gathered for k, v in d.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_screen_two_ranks_equal_one_rank_with_the_same_seed():
    import torch.multiprocessing as mp
    from diffphore_amd import distributed as D
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_screen_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        sharded = q.get(timeout=150)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    lib = synthetic.ScreenLibrary(4, 7)
    res = {}
    D.screen(len(lib), lib.unit, _model(), _args(), samples_per_complex=3, max_poses=8192, seed=11, on_result=lambda i, arr: res.__setitem__(i, arr))
    assert sorted(sharded) == sorted(res) == list(range(7))
    worst = max(_rmsd(sharded[i][s], res[i][s]) for i in range(7) for s in range(3))
    print("worst RMSD two ranks vs one (A):", worst)
    assert worst < 2e-5


def test_device_stream_reproduces_the_committed_fixture(golden_dir):
    """The seeded stream is part of the interface (a pinned seed must keep giving the same poses): the device regenerates
    tests/golden/rng_stream.npz (uids up to 2^31 - 1, steps 0 / 5 / 19, both torsion blocks, the start-randomisation streams)."""
    from diffphore_amd import _lib
    from diffphore_amd.engine import PackedBatch, _declare, _stream
    G = np.load(os.path.join(golden_dir, "rng_stream.npz"))
    gs = synthetic.synthetic_pairs(4242, len(G["uids"]), 16, 7, 9)
    pb = PackedBatch(gs, "cuda")
    uid = torch.from_numpy(G["uids"].astype(np.int32)).cuda()
    seed = int(G["seed"])
    for st in G["steps"]:
        a = _draw(pb, seed, int(st), uid=uid)
        np.testing.assert_allclose(a[0], G[f"z_tr_{st}"], atol=2e-6, rtol=2e-6)
        np.testing.assert_allclose(a[1], G[f"z_rot_{st}"], atol=2e-6, rtol=2e-6)
        np.testing.assert_allclose(a[2].reshape(-1, 7), G[f"z_tor_{st}"], atol=2e-6, rtol=2e-6)
    L = _declare()
    B, nT = pb.B, int(pb.tor_ptr[-1])
    th, rm, tr = torch.empty(nT, device="cuda"), torch.empty(B, 3, 3, device="cuda"), torch.empty(B, 3, device="cuda")
    _lib.check(L.dp_randomize_draws(seed, C.byref(pb.struct), uid.data_ptr(), 0, 5.0, th.data_ptr(), rm.data_ptr(), tr.data_ptr(), _stream()), "draws")
    torch.cuda.synchronize()
    np.testing.assert_allclose(th.cpu().numpy().reshape(B, 7), G["theta"], atol=2e-6)
    np.testing.assert_allclose(rm.cpu().numpy(), G["rot_mats"], atol=3e-6)
    np.testing.assert_allclose(tr.cpu().numpy(), G["tr0"], atol=1e-5, rtol=2e-6)


@pytest.mark.timeout(300)
def test_screening_example_runs(tmp_path):
    """examples/screen_synthetic.py end to end on a tiny library (the documented entry point of the streaming driver)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "screen_synthetic.py"), "--ligands", "6", "--samples", "4", "--steps", "3",
                        "--max-poses", "8", "--out", str(tmp_path)], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = open(os.path.join(str(tmp_path), "screen_rank0.csv")).read().strip().splitlines()
    assert rows[0].startswith("ligand,atoms,best_overlap") and len(rows) == 7 and "pose-steps/s" in r.stdout
