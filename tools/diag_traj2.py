#!/usr/bin/env python3
"""Free-running per-step divergence HIP vs fp64 oracle for the 6-step trajectory test case (pose-resolved), both conv formulations."""
import copy, os, sys, types
import numpy as np, torch
torch.set_num_threads(min(32, os.cpu_count() or 1))
from diffphore_amd import config, synthetic, weights
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
from diffphore_amd.sampling import sampling_phore
from diffphore_amd.score_model import TensorProductScoreModel
from oracle import conformer, sampler as osampler
from oracle.model import OracleScoreModel
from oracle.tables import load_tables
from scipy.spatial.transform import Rotation

ode = len(sys.argv) > 1 and sys.argv[1] == "ode"
gs = synthetic.synthetic_pairs(97, 6, 18, 4, 20) + synthetic.synthetic_pairs(98, 2, 25, 0, 14)
rng = np.random.default_rng(9)
draws = [dict(tor=rng.uniform(-np.pi, np.pi, size=int(g["ligand"].edge_mask.sum())),
              rot=Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix(),
              tr=torch.from_numpy(rng.normal(size=(1, 3)).astype(np.float32))) for g in gs]
conformer.randomize_position(gs, False, False, 5.0, draws=draws)
for g in gs:
    g["ligand"].norm = g["ligand"].norm.reshape(-1, 33)
S = 6
sched = get_t_schedule(S)
cfg = config.ModelConfig()
args = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                             "tor_sigma_min", "tor_sigma_max")}, no_torsion=False, keep_update=True)
gen = torch.Generator().manual_seed(607201)
nT = sum(int(g["ligand"].edge_mask.sum()) for g in gs)
noise = dict(tr=torch.randn(S, len(gs), 3, generator=gen), rot=torch.randn(S, len(gs), 3, generator=gen), tor=torch.randn(S, nT, generator=gen))
sd = weights.init_state_dict()
om = OracleScoreModel(sd, cfg, dtype=torch.float64, tables=load_tables())
ptrace, strace = [], []
a, _ = osampler.sampling_phore(copy.deepcopy(gs), om, S, sched, sched, sched, t_to_sigma_fn(args), args, ode=ode, batch_size=8, noise=noise, pos_trace=ptrace, trace=strace)
for algo in ("edge", "node"):
    os.environ["DIFFPHORE_CONV_ALGO"] = algo
    m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
    b, _ = sampling_phore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, ode=ode, batch_size=5, noise=noise)
    for t in range(S):
        r = [float(np.sqrt(((ptrace[t][i].numpy() - np.asarray(y.docked_poses[t])) ** 2).sum(-1).mean())) for i, y in enumerate(b)]
        print(algo, "step", t, " ".join(f"{v:.1e}" for v in r))
    sp_tr = None
    from diffphore_amd.sampling import _step_params
    for t in range(S):
        sp = _step_params(t, S, sched, sched, sched, t_to_sigma_fn(args), args, ode, False)
        tr_o, rot_o, tor_o = strace[t]
        print(algo, "pose0 step", t, "tr upd HIP", np.asarray(b[0].rvs_tr_update[t]).ravel(), "oracle", (sp.tr_c1 * tr_o[0]).numpy(),
              "| rot HIP", np.asarray(b[0].rvs_rot_update[t]).ravel(), "oracle", (sp.rot_c1 * rot_o[0]).numpy(),
              "| tor HIP", np.asarray(b[0].rvs_tor_update[t]).ravel(), "oracle", (sp.tor_c1 * tor_o[:4]).numpy())
