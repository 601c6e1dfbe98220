#!/usr/bin/env python3
"""Per-step divergence of one trajectory test case between the HIP path and the fp64 oracle (teacher-forced: every step starts from the
fp64 oracle's state), to tell a discontinuous decision of the reference algorithm from a kernel error."""
import copy
import sys
import types

import numpy as np
import torch

from diffphore_amd import config, synthetic, weights
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
from diffphore_amd.sampling import sampling_phore
from diffphore_amd.score_model import TensorProductScoreModel
from oracle import conformer, sampler as osampler
from oracle.model import OracleScoreModel
from oracle.tables import load_tables
from scipy.spatial.transform import Rotation


def main():
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
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
    gen = torch.Generator().manual_seed(607201)
    nT = sum(int(g["ligand"].edge_mask.sum()) for g in gs)
    noise = dict(tr=torch.randn(S, len(gs), 3, generator=gen), rot=torch.randn(S, len(gs), 3, generator=gen), tor=torch.randn(S, nT, generator=gen))
    sd = weights.init_state_dict()
    om = OracleScoreModel(sd, cfg, dtype=torch.float64, tables=load_tables())
    om32 = OracleScoreModel(sd, cfg, dtype=torch.float32, tables=load_tables())
    m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
    state = copy.deepcopy(gs)
    rm = lambda a, b: np.asarray([float(np.sqrt(((x["ligand"].pos.numpy() - y["ligand"].pos.numpy()) ** 2).sum(-1).mean())) for x, y in zip(a, b)])
    for t in range(S):
        # one step from the SAME (fp64 oracle) state with each implementation: schedules sliced so that step t is "the first of S-t"
        sub = {k: v[t:t + 1] for k, v in noise.items()}
        kw = dict(ode=ode, noise=sub)
        sch = np.concatenate([sched[t:t + 1], sched[t + 1:t + 2] if t + 1 < S else [0.0]])

        def one(model, fn, dev=None):
            st = copy.deepcopy(state)
            a = (st, model, 1, sch, sch, sch) + ((dev,) if dev else ()) + (t_to_sigma_fn(args), args)
            out, _ = fn(*a, batch_size=8, **kw)
            return out
        # NOTE: a 1-step call treats its step as the last one (dt = t); the same convention is applied to all three implementations
        o64 = one(om, osampler.sampling_phore)
        o32 = one(om32, osampler.sampling_phore)
        hip = one(m, sampling_phore, "cuda")
        print(f"step {t}: HIP vs fp64 {rm(hip, o64)}\n        fp32 oracle vs fp64 {rm(o32, o64)}")
        state = o64


if __name__ == "__main__":
    main()
