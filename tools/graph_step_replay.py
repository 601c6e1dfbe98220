#!/usr/bin/env python3
"""Timing experiment: the loop as ONE captured step launched S times (same step every time: wrong numbers, right timing) against the S-step graph and the
eager dp_sample - what a device-stepped single-step graph could reach.  usage: graph_step_replay.py [poses ...]"""
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
for B in [int(x) for x in sys.argv[1:]] or [1, 40, 512]:
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
    out = {}
    for mode in ("one-step graph x S", "S-step graph"):
        best_create, best = 1e9, 1e9
        for rep in range(4):
            pb.pos.copy_(pos0); pb.norm.copy_(norm0); eng.ws.t["status"].zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 1 if mode.startswith("one") else S
            eng.sample(sps[:n], T[:n].contiguous(), ztr[:n].contiguous(), zrot[:n].contiguous(), ztor[:n].contiguous(), graph=True)
            for _ in range(S - 1 if n == 1 else 0):
                eng.replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            best_create = min(best_create, (time.perf_counter() - t0) * 1e3)
            pb.pos.copy_(pos0); pb.norm.copy_(norm0); eng.ws.t["status"].zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(S if n == 1 else 1):
                eng.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(f"poses {B} {mode}: create + first run {best_create:.2f} ms, replays alone {best:.2f} ms")
