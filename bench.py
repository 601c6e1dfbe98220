#!/usr/bin/env python3
"""bench.py - DiffPhore reverse-diffusion hot path on MI355X.

One "step" = one reverse-diffusion step (score-model forward + SDE perturbation + conformer update) over the whole
per-GPU batch of poses, inputs resident in HBM.  Workload at N=1: BASELINE.json configs[1] - 512 synthetic
ligand-pharmacophore pairs (30 atoms, 8 torsions, 32 pharmacophore nodes), fp32.  With --gpus N each rank runs its own
512 poses (independent samples: no data-path collective; one RCCL broadcast of the weight blob at start-up), weak scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (fused TP conv, layer-2 instantiation): algorithmic
bytes of SURVEY.md 8d per launch / mean launch duration measured with HIP events recorded inside dp_forward on the
launch stream during the timed steps.  `cpu_baseline` = this repo's CPU oracle (reference-style torch-CPU restatement,
fp32) timed on a bounded sample of the same workload on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffphore_amd import config, synthetic, weights  # noqa: E402
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn  # noqa: E402

ALG_BYTES_PER_EDGE = {0: 2516, 1: 4636, 2: 6756, 3: 9236}      # 4*(numel + D_in + 9) per edge-conv, fp32
ALG_FLOPS_PER_EDGE = {0: 79200 + 2000, 1: 139200 + 4200, 2: 199200 + 6400, 3: 271200 + 8400}   # radial-MLP GEMM + TP
MFMA_F32_PEAK_TFLOPS = 157.3                                      # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
MEASURED_TRAFFIC_BYTES = int((2 * 37757.6 + 6000.0) * 1024)   # profiles/r1_conv_hbm_traffic.csv, layer-2 phore->lig launch
HBM_PEAK_GBPS = 8000.0                                           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
PROF_LAYER, PROF_CONV = 2, 1                                     # layer-2 phore->ligand conv: dense cross edge set


class HipEvents:
    def __init__(self, n):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.ev = []
        for _ in range(2 * n):
            e = C.c_void_p()
            assert self.hip.hipEventCreate(C.byref(e)) == 0
            self.ev.append(e)

    def pair(self, i):
        return self.ev[2 * i], self.ev[2 * i + 1]

    def ms(self, i):
        a, b = self.pair(i)
        self.hip.hipEventSynchronize(b)
        out = C.c_float()
        rc = self.hip.hipEventElapsedTime(C.byref(out), a, b)
        return out.value if rc == 0 else float("nan")


def cpu_baseline(graphs, sd, cfg, args, n_poses, n_steps):
    """Reference-style CPU path (oracle, fp32, torch threads = all host cores) on a bounded sample of the workload."""
    import copy
    from oracle import sampler as osampler
    from oracle.model import OracleScoreModel
    from oracle.tables import load_tables
    torch.set_num_threads(min(os.cpu_count() or 1, 32))   # more threads only add fork/join overhead on these small ops
    om = OracleScoreModel(sd, cfg, dtype=torch.float32, tables=load_tables())
    sample = copy.deepcopy(graphs[:n_poses])
    sched = get_t_schedule(20)[:n_steps + 1]
    t0 = time.perf_counter()
    osampler.sampling_phore(sample, om, n_steps, sched, sched, sched, t_to_sigma_fn(args), args, batch_size=20)
    dt = time.perf_counter() - t0
    return dict(value=n_poses * n_steps / dt, unit="pose-steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n_poses} poses x {n_steps} steps of the same workload, oracle (torch-CPU fp32 restatement), {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--poses", type=int, default=512, help="poses per GPU")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3],
                    help="synthetic.CONFIGS key = BASELINE.json configs index + 1 (default 2 = configs[1], the 512-pair workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x6", "bf16x9", "bf16x3"],
                    help="arithmetic of the radial-MLP contraction; the headline value uses exact fp32 MFMA")
    ap.add_argument("--no-alt", action="store_true", help="skip the informational split-bf16 passes")
    ap.add_argument("--cpu-poses", type=int, default=40)
    ap.add_argument("--cpu-steps", type=int, default=8)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if local >= ndev and os.environ.get("DIFFPHORE_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    local = local % max(ndev, 1)                            # (gloo smoke runs may put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("DIFFPHORE_DIST_BACKEND", "nccl")      # "nccl" = RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from diffphore_amd.engine import DeviceModel, Engine, PackedBatch
    from diffphore_amd.sampling import _step_params
    import types

    cfg = config.ModelConfig()
    margs = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                  "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
    shape = synthetic.CONFIGS[a.config]
    n_pairs = a.poses
    graphs = synthetic.synthetic_pairs(a.config, n_pairs, shape["n_atoms"], shape["n_tor"], shape["n_phore"],
                                       seed=synthetic.BASE_SEED + a.config + 1000 * rank)
    sd = weights.init_state_dict()
    model = DeviceModel(sd, cfg, dev, precision=a.precision)
    if world > 1:
        # the only collective of the job: weights from rank 0 to every GPU over xGMI
        dist.broadcast(model.blob, src=0)
        torch.cuda.synchronize()
    # initial state: randomize_position on the device (same call the sampler's callers make, inference.py:185)
    from diffphore_amd.sampling import randomize_position
    np.random.seed(607201 + rank)
    torch.manual_seed(607201 + rank)
    randomize_position(graphs, False, False, cfg.tr_sigma_max, device=dev)
    pb0 = PackedBatch(graphs, dev)
    pos0, norm0 = pb0.pos.clone(), pb0.norm.clone()
    S_total = 20
    sched = get_t_schedule(S_total)
    t2s = t_to_sigma_fn(margs)
    gen = torch.Generator(device=dev).manual_seed(607201 + rank)
    B, nT = pb0.B, int(pb0.tor_ptr[-1])
    # pre-drawn noise resident in HBM (the reference draws it on the host inside the loop, sampling.py:230-243)
    zs = [(torch.randn(B, 3, device=dev, generator=gen), torch.randn(B, 3, device=dev, generator=gen),
           torch.randn(max(nT, 1), device=dev, generator=gen)) for _ in range(a.warmup + a.steps)]

    def timed_run(mdl):
        pb = pb0
        pb.pos.copy_(pos0)
        pb.norm.copy_(norm0)
        eng = Engine(mdl, pb)
        ev = HipEvents(a.steps)

        def step(i, prof_idx=None):
            t_idx = i % S_total
            sp = _step_params(t_idx, S_total, sched, sched, sched, t2s, margs, False, False)
            pb.set_time(float(sched[t_idx]), float(sched[t_idx]), float(sched[t_idx]))
            w = eng.ws.struct
            if prof_idx is None:
                w.prof_start, w.prof_stop = None, None
            else:
                s0, s1 = ev.pair(prof_idx)
                w.prof_start, w.prof_stop, w.prof_layer, w.prof_conv = s0, s1, PROF_LAYER, PROF_CONV
            z = zs[i]
            eng.sample_step(sp, z[0], z[1], z[2] if nT else None)

        for i in range(a.warmup):
            step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + i, prof_idx=i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            te = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        nbad = int((eng.ws.t["status"] != 0).sum().item())
        fin = bool(torch.isfinite(pb.pos).all().item())
        return el, float(np.nanmean([ev.ms(i) for i in range(a.steps)])), nbad, fin

    elapsed, kern_ms, bad, finite = timed_run(model)
    pb = pb0
    alt = None
    if not a.no_alt and a.precision == "f32":
        # informational passes with the radial GEMM on the bf16 matrix pipe by operand splitting (never the headline `value`)
        alt = []
        for prec, what in (("bf16x9", "3-way bf16 split of both operands, all 9 cross products: every fp32 x fp32 product exact, fp32 accumulate"),
                           ("bf16x6", "same split, the 6 largest cross products (~2^-24 relative); passes the same parity tolerances")):
            alt_model = DeviceModel(sd, cfg, dev, precision=prec)
            if world > 1:
                dist.broadcast(alt_model.blob, src=0)
            el2, k2, bad2, fin2 = timed_run(alt_model)
            alt.append({"precision": f"{prec} ({what})", "value": B * world * a.steps / el2, "ms_per_step": el2 / a.steps * 1e3,
                        "kernel_ms": k2, "finite": fin2, "kabsch_flagged_poses": bad2})
            del alt_model

    n_x = int(pb.x_ptr[-1])
    alg_bytes = ALG_BYTES_PER_EDGE[PROF_LAYER] * n_x
    flops = ALG_FLOPS_PER_EDGE[PROF_LAYER] * n_x
    traffic = MEASURED_TRAFFIC_BYTES if (a.config == 2 and a.poses == 512 and a.precision == "f32") else None
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else float("nan")

    if rank == 0:
        total_poses = B * world
        out = {
            "metric": "poses/sec (samples x steps) on the reverse-diffusion hot path",
            "value": total_poses * a.steps / elapsed,
            "unit": "pose-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.precision == "f32" else a.precision, "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{a.config - 1}]: {B} synthetic lig-phore pairs per GPU "
                                   f"({shape['n_atoms']} atoms, {shape['n_tor']} torsions, {shape['n_phore']} phore nodes), "
                                   f"one reverse-diffusion step per bench step, fp32, random-init weights",
                       "poses_per_gpu": B, "poses_per_sec_at_20_steps": total_poses * a.steps / elapsed / 20.0,
                       "parallelism": f"independent poses sharded over {world} GPU(s), RCCL weight broadcast only",
                       "kabsch_flagged_poses": bad, "finite": finite, "precision": a.precision,
                       "alt_precision_runs": alt},
            # The dominant kernel generates the per-edge TP weights with fp32 MFMA in registers (they never reach HBM), so it is
            # bound by the fp32 matrix pipe; SURVEY.md 8d asks for both views when the radial MLP is fused into the TP kernel.
            "roofline": {"bound": "mfma", "achieved": flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None,
                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / (kern_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if kern_ms > 0 else None,
                         "traffic": traffic,
                         "kernel": "dp::tp_conv_kernel<2> (layer-2 phore->ligand conv, dense cross edges)",
                         "kernel_ms": kern_ms, "edges_per_launch": n_x,
                         "alg_flops_per_edge": ALG_FLOPS_PER_EDGE[PROF_LAYER], "alg_bytes_per_edge": ALG_BYTES_PER_EDGE[PROF_LAYER],
                         "hbm_equivalent": {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                                            "note": "algorithmic bytes of the e3nn operator boundary (per-edge weights counted as "
                                                    "HBM input, SURVEY.md 8d) / launch time; the metric's 'fused TP-scatter HBM GB/s'"},
                         "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, gfx950 "
                                         "correction), profiles/r1_conv_hbm_traffic.csv; null for non-default workloads"},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(graphs, sd, cfg, margs, a.cpu_poses, a.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
