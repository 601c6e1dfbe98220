#!/usr/bin/env python3
"""What capturing the sampling loop into a hipGraph costs per batch (dp_sample_graph_create: capture + instantiate), beside the loop itself as a graph
replay and as an eager dp_sample - a single-use graph (one device batch of a screen, one complex of inference.py's loop) pays it every time.
usage: graph_create_cost.py [poses ...]"""
import sys
import time
import types

import numpy as np
import torch

from diffphore_amd import config, synthetic, weights
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
from diffphore_amd.engine import DeviceModel, Engine, PackedBatch
from diffphore_amd.sampling import _step_params, randomize_position

cfg = config.ModelConfig()
margs = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max", "tor_sigma_min", "tor_sigma_max")},
                              no_torsion=False)
model = DeviceModel(weights.init_state_dict(), cfg, "cuda")
S = 20
sched = get_t_schedule(S)
sps = [_step_params(t, S, sched, sched, sched, t_to_sigma_fn(margs), margs, False, False) for t in range(S)]
T = torch.tensor([[float(sched[t])] * 3 for t in range(S)], device="cuda")
c = synthetic.CONFIGS[2]
allg = synthetic.synthetic_pairs(2, 512, c["n_atoms"], c["n_tor"], c["n_phore"])
for B in [int(x) for x in sys.argv[1:]] or [40, 512]:
    gs = [allg[i % len(allg)].sample_copy() for i in range(B)]
    torch.manual_seed(1); np.random.seed(1)
    randomize_position(gs, False, False, cfg.tr_sigma_max)
    pb = PackedBatch(gs, "cuda")
    pos0, norm0 = pb.pos.clone(), pb.norm.clone()
    g = torch.Generator(device="cuda").manual_seed(2)
    ztr, zrot = torch.randn(S, B, 3, device="cuda", generator=g), torch.randn(S, B, 3, device="cuda", generator=g)
    ztor = torch.randn(S, int(pb.tor_ptr[-1]), device="cuda", generator=g)
    eng = Engine(model, pb, conv_algo="node")
    eng.enable_two_streams()
    res = {}
    for rep in range(4):
        for mode in ("eager", "graph"):
            pb.pos.copy_(pos0); pb.norm.copy_(norm0); eng.ws.t["status"].zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.sample(sps, T, ztr, zrot, ztor, graph=(mode == "graph"))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.setdefault(mode, []).append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
        pb.pos.copy_(pos0); pb.norm.copy_(norm0); eng.ws.t["status"].zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.replay()
        torch.cuda.synchronize()
        res.setdefault("replay", []).append((0.0, (time.perf_counter() - t0) * 1e3))
    for mode, v in res.items():
        v = sorted(v[1:], key=lambda x: x[1])
        print(f"poses {B} {mode}: host call {v[len(v) // 2][0]:.2f} ms, until done {v[len(v) // 2][1]:.2f} ms")
