#!/usr/bin/env python3
"""Host timeline of one distributed.screen call over the configs[3] library (96 units x 40 samples): when each unit is built (load_unit) and when each
finished complex is delivered - shows what the first device batch waits for.  Run from the repo root on a GPU box."""
import os, sys, time, types
import torch
sys.path.insert(0, ".")
from diffphore_amd import distributed as D, synthetic, workloads, weights, config
from diffphore_amd.score_model import TensorProductScoreModel
c = 4
n_lib = 96
lib = synthetic.ScreenLibrary(c, n_lib)
sd = weights.init_state_dict()
m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
cfg = config.ModelConfig()
margs = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max", "tor_sigma_min", "tor_sigma_max")}, no_torsion=False, inference_steps=20)
per_batch = workloads.DEFAULT_UNITS[c] * workloads.SAMPLES
D.screen(2, lib.unit, m, margs, samples_per_complex=2, max_poses=per_batch, seed=1, on_result=lambda i, a: None)
torch.cuda.synchronize()
for rep in range(2):
    ev = []
    t0 = time.perf_counter()
    def load(i):
        t1 = time.perf_counter(); g = lib.unit(i); ev.append(("load", i, t1 - t0, time.perf_counter() - t0)); return g
    def sink(i, arr):
        ev.append(("done", i, time.perf_counter() - t0, 0))
    costs = [lib.cost(i) for i in range(n_lib)]
    tc = time.perf_counter() - t0
    D.screen(n_lib, load, m, margs, costs=costs, samples_per_complex=workloads.SAMPLES, max_poses=per_batch, seed=607201, on_result=sink)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: total {dt:.3f} s, costs {tc*1e3:.1f} ms")
    loads = [e for e in ev if e[0] == "load"]; dones = [e for e in ev if e[0] == "done"]
    print(" first 20 events:", " ".join(f"{e[0][0]}{e[1]}@{e[2]*1e3:.0f}" + (f"-{e[3]*1e3:.0f}" if e[0] == "load" else "") for e in ev[:40]))
    print(" load total", sum(e[3] - e[2] for e in loads), "first done at", dones[0][2], "16th done at", dones[15][2], "last", dones[-1][2])
