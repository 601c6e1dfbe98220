"""BASELINE configs[3] / configs[4] per-GPU workloads (diffphore_amd/workloads.py) on the device vs the fp64 oracle, and the
multi-GPU contract on ONE GPU: two ranks (gloo rendezvous, both on cuda:0) run their LPT shards of a virtual screen with weights
broadcast from rank 0, and reproduce the unsharded run pose for pose."""
import copy
import os
import socket
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffphore_amd import config, synthetic, weights, workloads  # noqa: E402
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn  # noqa: E402


def _args():
    cfg = config.ModelConfig()
    return types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)


def _rmsd(a, b):
    return float(np.sqrt(((a - b) ** 2).sum(-1).mean()))


def _start_and_noise(poses, keys, S):
    """Start poses and per-step noise that depend only on a pose's global key, not on which rank / batch it lands in."""
    from oracle import conformer
    from scipy.spatial.transform import Rotation
    draws, ztr, zrot, ztor = [], [], [], []
    for g, key in zip(poses, keys):
        rng = np.random.default_rng(key)
        T = int(g["ligand"].edge_mask.sum())
        draws.append(dict(tor=rng.uniform(-np.pi, np.pi, size=T), rot=Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix(),
                          tr=torch.from_numpy(rng.normal(scale=3.0, size=(1, 3)).astype(np.float32))))
        ztr.append(rng.normal(size=(S, 1, 3)).astype(np.float32))
        zrot.append(rng.normal(size=(S, 1, 3)).astype(np.float32))
        ztor.append(rng.normal(size=(S, T)).astype(np.float32))
    conformer.randomize_position(poses, False, False, 5.0, draws=draws)
    for g in poses:
        g["ligand"].norm = g["ligand"].norm.reshape(-1, 33)
    return dict(tr=torch.from_numpy(np.concatenate(ztr, 1)), rot=torch.from_numpy(np.concatenate(zrot, 1)),
                tor=torch.from_numpy(np.concatenate(ztor, 1)))


@pytest.mark.parametrize("cfg_id,precision,tol", [(4, "f32", 1e-4), (5, "bf16x3", 5e-3), (5, "bf16", 6e-3)])
def test_config_workload_matches_oracle(cfg_id, precision, tol, monkeypatch):
    """2 complexes x 40 samples of the configs[3] / configs[4] generators, first steps of their schedules (20 / 100 steps), HIP vs the
    fp64 oracle on a sample of the poses; configs[4] runs BASELINE's 'bf16 mixed precision': `bf16` = plain bf16 operands with fp32
    accumulate (8 significand bits: the bound is that of the arithmetic, not of the implementation - tests/test_conv_bf16_gpu.py holds the
    kernel to its bf16 specification at 1e-3), `bf16x3` = the 2-way split (~2^-17 products)."""
    from diffphore_amd.sampling import sampling_phore
    from diffphore_amd.score_model import TensorProductScoreModel
    from oracle import sampler as osampler
    from oracle.model import OracleScoreModel
    from oracle.tables import load_tables
    poses, mine = workloads.build(cfg_id, 2, 0, 1)
    assert len(poses) == 2 * workloads.SAMPLES and sorted(mine) == [0, 1]
    S_total, S = workloads.steps_of(cfg_id), 3
    assert S_total == (20 if cfg_id == 4 else 100)
    sched = get_t_schedule(S_total)[:S + 1]
    args = _args()
    noise = _start_and_noise(poses, [1000 + i for i in range(len(poses))], S)
    sd = weights.init_state_dict()
    m = TensorProductScoreModel(device="cuda", precision=precision).load_state_dict(sd)
    b, _ = sampling_phore(copy.deepcopy(poses), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, noise=noise)
    assert all(np.isfinite(g["ligand"].pos.numpy()).all() for g in b)
    pick = [0, 1, 39, 40, 41, 79]                          # samples of both complexes
    T = [int(g["ligand"].edge_mask.sum()) for g in poses]
    toff = np.concatenate([[0], np.cumsum(T)])
    sub = dict(tr=noise["tr"][:, pick], rot=noise["rot"][:, pick],
               tor=torch.cat([noise["tor"][:, toff[i]:toff[i + 1]] for i in pick], 1))
    om = OracleScoreModel(sd, config.ModelConfig(), dtype=torch.float64, tables=load_tables())
    a, _ = osampler.sampling_phore([copy.deepcopy(poses[i]) for i in pick], om, S, sched, sched, sched, t_to_sigma_fn(args), args,
                                   batch_size=len(pick), noise=sub)
    r = np.asarray([_rmsd(x["ligand"].pos.numpy(), b[i]["ligand"].pos.numpy()) for x, i in zip(a, pick)])
    print(f"config {cfg_id} ({precision}) RMSD vs fp64 oracle (A):", r)
    assert (r < tol).all(), r


# bounds of the full 100-step configs[4] run below, per precision: (every pose after 20 steps, every pose after 100 steps), Angstrom RMSD from the
# fp64 oracle with the same start poses and noise.  Measured on MI355X (printed by the test, profiles/r5_configs4_100step.txt), max over the 6 poses
# after 20 / 100 steps:   f32 (library default: f16x3 node-side + split edge-side convs) 9.3e-5 / 8.9e-5;  f32-f16x3 (f16x3 node-side convs, exact
# fp32 edge-side) and f32-exact (exact fp32 MFMA everywhere): see the profile;  bf16x3 9.5e-5 / 9.1e-5;  bf16 7.4e-4 / 1.6e-3.
# (One pose of the six sits on a discrete decision that some perturbations flip within the first 20 steps - it then stays at 9e-5 for the other 80 -
# and others do not: both f16x3 arms and the exact arm end at 4-5e-5, the other five poses stay below 5.4e-5 in every arm.)
FULL_RUN_BOUNDS = {"f32": (1.5e-4, 3e-4), "f32-f16x3": (1.5e-4, 3e-4), "f32-exact": (5e-5, 1e-4), "f32-small": (1.5e-4, 3e-4), "f32-pathsplit": (1.5e-4, 3e-4),
                   "bf16x3": (1.5e-4, 3e-4), "bf16": (2.5e-3, 5e-3)}
# environment of the fp32 arms: DIFFPHORE_NODE_F16 (two-limb fp16 node-side kernel), DIFFPHORE_EDGE_X6 (fp32-grade operand split of the edge-side
# convs), DIFFPHORE_PATH_SPLIT.  "f32" = what a LARGE batch of an fp32 model runs by default (f16x3 node-side kernel in fused launches - forced with
# DIFFPHORE_PATH_SPLIT=0 - + f16x3 edge-side), "f32-f16x3" = that node-side kernel beside exact-fp32 edge-side convs, "f32-exact" = fp32 MFMA
# everywhere, "f32-small" = the library's own choice for this 6-pose batch (the f16x3 node-side kernel with its conv split + f16x3 edge-side),
# "f32-pathsplit" = the exact-fp32 path-split / conv-split node-side kernels (the small-batch default until the f16x3 kernel had a conv split;
# DIFFPHORE_PATH_SPLIT=1) + f16x3 edge-side
FP32_ARMS = {"f32": {"DIFFPHORE_PATH_SPLIT": "0"}, "f32-f16x3": {"DIFFPHORE_PATH_SPLIT": "0", "DIFFPHORE_EDGE_X6": "0"},
             "f32-exact": {"DIFFPHORE_NODE_F16": "0", "DIFFPHORE_EDGE_X6": "0"}, "f32-small": {}, "f32-pathsplit": {"DIFFPHORE_PATH_SPLIT": "1"}}


@pytest.mark.parametrize("precision", ["f32", "f32-f16x3", "f32-exact", "f32-small", "f32-pathsplit", "bf16x3", "bf16"])
def test_configs4_full_100_step_schedule_vs_oracle(precision, monkeypatch):
    """BASELINE configs[4] over its WHOLE calibrated schedule (100 reverse steps, /root/reference/src/utils/sampling.py:204-255 accumulates every
    step's error): 6 poses (3 samples of 2 target-fishing complexes), same start poses and pre-drawn noise for the HIP path and the fp64
    oracle, compared every 20 steps.  Prints the RMSD growth; asserts the per-precision bounds of FULL_RUN_BOUNDS at 20 and at 100 steps."""
    from diffphore_amd.sampling import sampling_phore
    from diffphore_amd.score_model import TensorProductScoreModel
    from oracle import sampler as osampler
    from oracle.model import OracleScoreModel
    from oracle.tables import load_tables
    cplx, _mine = workloads.build(5, 2, 0, 1)
    poses = [cplx[i] for i in (0, 1, 2, 40, 41, 42)]
    S_total, SEG = workloads.steps_of(5), 20
    assert S_total == 100
    sched = get_t_schedule(S_total)
    args = _args()
    noise = _start_and_noise(poses, [4000 + i for i in range(len(poses))], S_total)
    sd = weights.init_state_dict()
    for k_, v_ in FP32_ARMS.get(precision, {}).items():          # which kernels the fp32 model's convs run on
        monkeypatch.setenv(k_, v_)
    m = TensorProductScoreModel(device="cuda", precision="f32" if precision in FP32_ARMS else precision).load_state_dict(sd)
    om = OracleScoreModel(sd, config.ModelConfig(), dtype=torch.float64, tables=load_tables())
    # ONE 100-step call each (the schedule's last step is special: dt = t); intermediate coordinates through the reference's own
    # `keep_update` bookkeeping on the HIP side (diffusion_utils.py:71-77: docked_poses per step) and the oracle's pos_trace
    kargs = types.SimpleNamespace(**vars(args), keep_update=True)
    b, _ = sampling_phore(copy.deepcopy(poses), m, S_total, sched, sched, sched, "cuda", t_to_sigma_fn(args), kargs, noise=noise)
    trace = []
    a, _ = osampler.sampling_phore(copy.deepcopy(poses), om, S_total, sched, sched, sched, t_to_sigma_fn(args), args, batch_size=len(poses),
                                   noise=noise, pos_trace=trace)
    assert len(trace) == S_total and all(len(g.docked_poses) == S_total for g in b)
    growth = [[_rmsd(trace[k - 1][i].numpy(), np.asarray(b[i].docked_poses[k - 1])) for i in range(len(poses))] for k in range(SEG, S_total + 1, SEG)]
    final = [_rmsd(x["ligand"].pos.numpy(), y["ligand"].pos.numpy()) for x, y in zip(a, b)]
    assert np.allclose(final, growth[-1], atol=1e-6)
    growth = np.asarray(growth)
    print(f"configs[4] {precision}: RMSD (A) from the fp64 oracle after 20 / 40 / 60 / 80 / 100 steps, per pose:")
    for k, row in enumerate(growth):
        print(f"   after {20 * (k + 1):3d} steps: max {row.max():.2e}  median {np.median(row):.2e}  ", " ".join(f"{v:.1e}" for v in row))
    assert np.isfinite(growth).all()
    b20, b100 = FULL_RUN_BOUNDS[precision]
    assert (growth[0] < b20).all(), (precision, growth[0])
    assert (growth[-1] < b100).all(), (precision, growth[-1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, S, q):
    import torch.distributed as dist
    from diffphore_amd import distributed as D
    from diffphore_amd.sampling import sampling_phore
    from diffphore_amd.score_model import TensorProductScoreModel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        cplx = synthetic.synthetic_pairs(4, 6, (15, 45), (2, 12), 79, seed=synthetic.BASE_SEED + 4, shared_phore=True)
        mine = D.shard_indices([D.pose_cost(g) for g in cplx], world)[rank]
        NS = 3
        poses = [cplx[i].sample_copy() for i in mine for _ in range(NS)]
        keys = [7000 + 10 * i + s for i in mine for s in range(NS)]
        noise = _start_and_noise(poses, keys, S)
        # rank 0 owns the real weights; the others start from a different random model and must receive rank 0's blob
        m = TensorProductScoreModel(device="cuda").load_state_dict(weights.init_state_dict(20250321 if rank == 0 else 99))
        blob = m.device_model.blob.cpu()
        D.broadcast_weights(blob, src=0)
        m.device_model.blob.copy_(blob)
        args = _args()
        sched = get_t_schedule(20)[:S + 1]
        out, _ = sampling_phore(poses, m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, noise=noise)
        allpos = D.gather_positions(keys, [g["ligand"].pos for g in out], 0, dst=0) if False else None
        payload = (keys, [g["ligand"].pos.numpy() for g in out])
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(payload, gathered, dst=0)
        if rank == 0:
            q.put({k: p for ks, ps in gathered for k, p in zip(ks, ps)})
        del allpos
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_equal_the_unsharded_run():
    import torch.multiprocessing as mp
    from diffphore_amd.sampling import sampling_phore
    from diffphore_amd.score_model import TensorProductScoreModel
    S, NS = 3, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    sharded = q.get(timeout=500)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cplx = synthetic.synthetic_pairs(4, 6, (15, 45), (2, 12), 79, seed=synthetic.BASE_SEED + 4, shared_phore=True)
    poses = [cplx[i].sample_copy() for i in range(6) for _ in range(NS)]
    keys = [7000 + 10 * i + s for i in range(6) for s in range(NS)]
    noise = _start_and_noise(poses, keys, S)
    m = TensorProductScoreModel(device="cuda").load_state_dict(weights.init_state_dict())
    args = _args()
    sched = get_t_schedule(20)[:S + 1]
    out, _ = sampling_phore(poses, m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, noise=noise)
    assert sorted(sharded) == sorted(keys)
    worst = max(_rmsd(sharded[k], g["ligand"].pos.numpy()) for k, g in zip(keys, out))
    print("worst RMSD sharded vs unsharded (A):", worst)
    assert worst < 2e-5      # same kernels on the same poses; only the batch composition (workgroup slicing of the edge-side convs) differs


@pytest.mark.timeout(900)
@pytest.mark.parametrize("cfg_id,extra", [(2, ["--poses", "24"]), (4, ["--units", "1"]), (5, ["--units", "1"])])
def test_bench_self_launches_two_ranks(cfg_id, extra):
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself (here both on the one GPU of the box, gloo rendezvous) and
    rank 0 prints the contract's JSON line with n_gpus = 2 and the whole-job pose count."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DIFFPHORE_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--config", str(cfg_id),
                          "--no-cpu-baseline", "--no-alt", "--no-e2e"] + extra, capture_output=True, text=True, timeout=800, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["finite"] and d["config"]["kabsch_flagged_poses"] == 0
    per = {2: 24, 4: 40, 5: 40}[cfg_id]
    assert d["config"]["poses_total"] == 2 * per
    assert f"configs[{cfg_id - 1}]" in d["config"]["workload"]
    assert d["dtype"].startswith("bf16 operands" if cfg_id == 5 else "f32")
    # scaling-readiness fields: per-rank rates, load imbalance of the LPT shards, the one collective's time
    rk = d["config"]["ranks"]
    assert len(rk["pose_steps_per_s"]) == 2 and len(rk["seconds"]) == 2 and sum(rk["poses"]) == d["config"]["poses_total"]
    assert rk["imbalance_max_over_mean"] >= 1.0 and rk["weight_broadcast_ms"] is not None and rk["weight_broadcast_ms"] > 0
    assert d["config"]["library"]["nabl"] == 0 and len(d["config"]["library"]["source_hash"]) == 16
    assert d["config"]["edge_side_convs"] == ("bf16" if cfg_id == 5 else "f16x3")
