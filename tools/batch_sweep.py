#!/usr/bin/env python3
"""ms per reverse-diffusion step vs batch size (cfg2 shape by default) for the conv formulations and loop drivers:

    steps        per-step enqueue from Python (dp_sample_step), one stream
    loop         dp_sample eager (one C call for the whole loop), one stream
    graph        dp_sample captured into a hipGraph and replayed
    graph+2s     the same with dp_forward's second stream (phore-side convs / torsion head beside the ligand side)

Used to pick DP_NODE_AUTO_MIN_POSES, TWO_STREAM_MAX_POSES and GRAPH_MAX_POSES, and for the small-batch numbers in DESIGN.md 6."""
import argparse
import csv
import sys
import time
import types

import torch

from diffphore_amd import config, synthetic, weights
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
from diffphore_amd.engine import DeviceModel, Engine, PackedBatch
from diffphore_amd.sampling import _step_params, randomize_position


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--batches", default="8,16,40,64,128,256,512,1024,2048")
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--shared-phore", action="store_true", help="every pose a sample of ONE complex (the reference's own call: N samples x one complex)")
    ap.add_argument("--algos", default=None, help="comma list of conv formulations (default: edge,node where both exist)")
    a = ap.parse_args()
    c = synthetic.CONFIGS[a.config]
    cfg = config.ModelConfig()
    margs = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                  "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
    model = DeviceModel(weights.init_state_dict(), cfg, "cuda", precision=a.precision)
    S = 20
    sched = get_t_schedule(S)
    t2s = t_to_sigma_fn(margs)
    sps = [_step_params(t, S, sched, sched, sched, t2s, margs, False, False) for t in range(S)]
    T = torch.tensor([[float(sched[t])] * 3 for t in range(S)], device="cuda")
    rows = []
    bmax = max(int(b) for b in a.batches.split(","))
    allg = synthetic.synthetic_pairs(a.config, min(bmax, 512), c["n_atoms"], c["n_tor"], c["n_phore"])
    for B in (int(b) for b in a.batches.split(",")):
        gs = [allg[0 if a.shared_phore else i % len(allg)].sample_copy() for i in range(B)]
        torch.manual_seed(1)
        import numpy as np
        np.random.seed(1)
        randomize_position(gs, False, False, cfg.tr_sigma_max)
        pb = PackedBatch(gs, "cuda")
        pos0, norm0 = pb.pos.clone(), pb.norm.clone()
        g = torch.Generator(device="cuda").manual_seed(2)
        ztr, zrot = torch.randn(S, B, 3, device="cuda", generator=g), torch.randn(S, B, 3, device="cuda", generator=g)
        ztor = torch.randn(S, int(pb.tor_ptr[-1]), device="cuda", generator=g)
        row = dict(poses=B)
        for algo in (a.algos.split(",") if a.algos else (["edge", "node"] if a.precision in ("f32", "bf16") else ["edge"])):
            for drv in ("steps", "loop", "graph", "graph+2s"):
                eng = Engine(model, pb, conv_algo=algo)
                if drv == "graph+2s":
                    eng.enable_two_streams()
                elif eng._handles is not None:          # auto two-stream for small batches: switch it off for the single-stream drivers
                    w = eng.ws.struct
                    w.aux_stream = w.ev_fork = w.ev_join = None

                def reset():
                    pb.pos.copy_(pos0)
                    pb.norm.copy_(norm0)
                    eng.ws.t["status"].zero_()
                best = 1e9
                for rep in range(a.reps + 1):
                    reset()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if drv == "steps":
                        for t in range(S):
                            pb.set_time_row(T[t])
                            eng.sample_step(sps[t], ztr[t], zrot[t], ztor[t])
                    elif drv == "loop":
                        eng.sample(sps, T, ztr, zrot, ztor, graph=False)
                    else:
                        if rep == 0:
                            eng.sample(sps, T, ztr, zrot, ztor, graph=True)
                        else:
                            eng.replay()
                    torch.cuda.synchronize()
                    if rep > 0:
                        best = min(best, (time.perf_counter() - t0) / S * 1e3)
                row[f"{algo}/{drv}"] = round(best, 4)
                del eng
        print(row, flush=True)
        rows.append(row)
    if a.out:
        with open(a.out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)


if __name__ == "__main__":
    sys.exit(main())
