"""Experiment: the configs[1] batch as K independent sub-batches stepped side by side on K streams (each with its own auxiliary stream), against the
one-batch step bench.py measures.  The serial head (time embedding -> node / edge embeddings) and tail (torsion / centre convs -> heads -> conformer
update) of a step, ~0.3 ms of 3.5 with the chip nearly idle, could run under another sub-batch's conv launches.  python tools/exp_concurrent_batches.py"""
import sys, time, types
import numpy as np
import torch

sys.path.insert(0, ".")
from diffphore_amd import config, weights, workloads           # noqa: E402
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn      # noqa: E402
from diffphore_amd.engine import DeviceModel, Engine, PackedBatch, _HipHandles      # noqa: E402
from diffphore_amd.sampling import _step_params, randomize_position      # noqa: E402

dev = torch.device("cuda", 0)
cfg = config.ModelConfig()
margs = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max", "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
t2s = t_to_sigma_fn(margs)
mdl = DeviceModel(weights.init_state_dict(), cfg, dev, precision="f32")
graphs, _ = workloads.build(2, 512, 0, 1)
np.random.seed(1); torch.manual_seed(1)
randomize_position(graphs, False, False, cfg.tr_sigma_max, device=dev)
S = 20
sched = get_t_schedule(S)
sps = [_step_params(t, S, sched, sched, sched, t2s, margs, False, False) for t in range(S)]
T_dev = torch.tensor([[float(sched[t])] * 3 for t in range(S)], dtype=torch.float32, device=dev)


def run(K, steps=60, warmup=10, stagger=False):
    n = len(graphs) // K
    subs = []
    for k in range(K):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            pb = PackedBatch(graphs[k * n:(k + 1) * n], dev)
            if K > 1:
                _HipHandles._cache.pop(0, None)          # a private auxiliary stream + events per engine
            eng = Engine(mdl, pb)
            gen = torch.Generator(device=dev).manual_seed(5 + k)
            nT = int(pb.tor_ptr[-1])
            zs = [(torch.randn(pb.B, 3, device=dev, generator=gen), torch.randn(pb.B, 3, device=dev, generator=gen), torch.randn(max(nT, 1), device=dev, generator=gen)) for _ in range(S)]
            subs.append((st, pb, eng, zs, pb.pos.clone(), pb.norm.clone(), eng._handles))
    torch.cuda.synchronize()

    def step(i):
        for k, (st, pb, eng, zs, p0, n0, _) in enumerate(subs):
            j = i + (k * S // (2 * K) if stagger else 0)
            t = j % S
            with torch.cuda.stream(st):
                if t == 0:
                    pb.pos.copy_(p0); pb.norm.copy_(n0)
                pb.t.copy_(T_dev[t].expand(pb.B, 3))
                eng.sample_step(sps[t], zs[t][0], zs[t][1], zs[t][2])
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(s[1].pos).all()) for s in subs)
    print(f"K={K} stagger={stagger}: {n * K * steps / el:,.0f} pose-steps/s, {el / steps * 1e3:.3f} ms per step of {n * K} poses, finite={ok}", flush=True)


for K, sg in ((1, False), (2, False), (2, True), (4, False), (1, False), (2, False)):
    run(K, stagger=sg)
