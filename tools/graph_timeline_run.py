#!/usr/bin/env python3
"""The sampling loop of N samples of ONE complex as the drop-in runs it (dp_sample captured into a hipGraph, second stream), replayed a few times -
to be run under rocprofv3 --kernel-trace (tools/gpu_graph_timeline.sh), which step_timeline.py then cuts one step out of.
usage: graph_timeline_run.py [poses] [config id: 1 = 20 atoms / 79 nodes, 2 = 30 atoms / 32 nodes]"""
import sys
import time
import types

import numpy as np
import torch

from diffphore_amd import config, synthetic, weights
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
from diffphore_amd.engine import DeviceModel, Engine, PackedBatch
from diffphore_amd.sampling import _step_params, randomize_position

B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cid = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = synthetic.CONFIGS[cid]
cfg = config.ModelConfig()
margs = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max", "tor_sigma_min", "tor_sigma_max")},
                              no_torsion=False)
model = DeviceModel(weights.init_state_dict(), cfg, "cuda")
S = 20
sched = get_t_schedule(S)
sps = [_step_params(t, S, sched, sched, sched, t_to_sigma_fn(margs), margs, False, False) for t in range(S)]
T = torch.tensor([[float(sched[t])] * 3 for t in range(S)], device="cuda")
g0 = synthetic.synthetic_pairs(cid, 1, c["n_atoms"], c["n_tor"], c["n_phore"])[0]
gs = [g0.sample_copy() for _ in range(B)]
torch.manual_seed(1)
np.random.seed(1)
randomize_position(gs, False, False, cfg.tr_sigma_max)
pb = PackedBatch(gs, "cuda")
pos0, norm0 = pb.pos.clone(), pb.norm.clone()
g = torch.Generator(device="cuda").manual_seed(2)
ztr, zrot = torch.randn(S, B, 3, device="cuda", generator=g), torch.randn(S, B, 3, device="cuda", generator=g)
ztor = torch.randn(S, int(pb.tor_ptr[-1]), device="cuda", generator=g)
eng = Engine(model, pb, conv_algo="node")
eng.enable_two_streams()
eng.sample(sps, T, ztr, zrot, ztor, graph=True)
best = 1e9
for rep in range(4):
    pb.pos.copy_(pos0)
    pb.norm.copy_(norm0)
    eng.ws.t["status"].zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.replay()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / S * 1e3)
print(f"poses {B} config {cid}: {best:.4f} ms per step (graph + second stream, best of 4 replays)")
