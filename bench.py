#!/usr/bin/env python3
"""bench.py - DiffPhore reverse-diffusion hot path on MI355X.

One "step" = one reverse-diffusion step (score-model forward + SDE perturbation + conformer update) over the whole
per-GPU batch of poses, inputs and pre-drawn noise resident in HBM.  Workload at N=1: BASELINE.json configs[1] - 512
synthetic ligand-pharmacophore pairs (30 atoms, 8 torsions, 32 pharmacophore nodes).  `--config 1..5` selects the other
BASELINE configs (diffphore_amd/workloads.py).  Poses are independent: with --gpus N every rank runs its own shard
(no data-path collective; one RCCL broadcast of the weight blob at start-up), weak scaling.

`python bench.py --gpus N` without a torchrun environment launches N ranks itself (torch.distributed.run, 127.0.0.1).

Timed region: W warm-up steps (untimed), state reset to the randomised start poses, then EXACTLY K steps walking the
schedule 0..S-1 (state reset at every wrap), bracketed by barrier + synchronize; MAX over ranks.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (fused TP conv, the layer-3 ligand-side launch: node-side it
carries ligand-ligand + phore->ligand + phore->ligand(normals) in one launch): algorithmic flops / bytes per launch (summed over
its convs) / mean launch duration measured with HIP events recorded inside dp_forward on the launch stream during the timed steps;
`roofline.step_weighted` = the same ratio over ALL node-side launches of a step (a short extra pass that brackets every launch).
`cpu_baseline` = this repo's CPU oracle (reference-style torch-CPU restatement, fp32) on a bounded sample of the same workload
(rank 0, N=1 only).  `config.edge_side_convs` says what the convs that stay on the edge-side kernel in an fp32 model run ("f16x3" = two fp16
limbs per operand by default, fp32-grade; `value_exact_fp32` / `alt_runs[0]` carry the exact-fp32-everywhere arm).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_EDGE = {0: 2516, 1: 4636, 2: 6756, 3: 9236}      # 4*(numel + D_in + 9) per edge-conv, fp32
ALG_FLOPS_PER_EDGE = {0: 79200 + 2000, 1: 139200 + 4200, 2: 199200 + 6400, 3: 271200 + 8400}   # radial-MLP GEMM + TP
# node-side formulation: per EDGE 2*61*(20 + C) flop (edge part of radial layer 1 + GEMM-A over C = sum_p mul_in*d_out
# columns), per scatter NODE 2*61*(21 + W) flop (its block of radial layer 1 incl. bias + GEMM-B, W = sum_p mul_in*mul_out*d_out), per
# gather NODE 2*61*20 flop (its block of radial layer 1, counted once per launch); no padding, no per-workgroup recomputation counted
NODE_COLS = {0: 80, 1: 180, 2: 280, 3: 360}
NODE_W = {0: 1000, 1: 2100, 2: 3200, 3: 4200}
# MFMAs issued per wave per 32-edge pass: radial layer 1 (32 edge tiles x 6 + three node tiles = 16) + GEMM-A (128 per column tile) + GEMM-B
NODE_MFMA_PER_WAVE = {0: 208 + 6 * 128 + 400, 1: 208 + 13 * 128 + 840, 2: 208 + 20 * 128 + 1280, 3: 208 + 26 * 128 + 1680}
# the bf16-operand kernel (dp_conv_node_bf16_impl.h): GEMM1 32 edge tiles + 3 node tiles, GEMM-A ONE 16x16x32 MFMA per (node, column tile), GEMM-B
# 5 K chunks x M tiles x m components per supertile
NODE_MFMA_PER_WAVE_BF16 = {0: 35 + 16 * 6 + 50, 1: 35 + 16 * 13 + 105, 2: 35 + 16 * 20 + 160, 3: 35 + 16 * 26 + 210}
# the fp32-grade two-limb fp16 kernel (dp_conv_node_f16_impl.h, DP_PREC_F16X3), per wave per 32-edge pass of an 8-NODE group: GEMM1 (16 edge tiles +
# 3 node tiles) x 3 cross products, GEMM-A 3 MFMAs per (node, column tile), GEMM-B 2 W2 limbs x (5 K chunks x M tiles x m components) per supertile
# (the two T limbs ride in the N dimension).  Kind 3: 1 101 = SQ_INSTS_MFMA / waves of profiles/r5_f16_node_pmc.csv
NODE_MFMA_PER_WAVE_F16 = {0: 57 + 24 * 6 + 100, 1: 57 + 24 * 13 + 210, 2: 57 + 24 * 20 + 320, 3: 57 + 24 * 26 + 420}
# A normals conv as its s0-only node-side launch + its listed pairs (DpWork.xn_sparse, round 6): the launch evaluates the paths through the l = 0
# harmonic only (Z000 / Z101: columns, second-layer rows and MFMAs below) on every pair, the other paths on the listed pairs edge-side
NODE_COLS_S0 = {0: 20, 1: 50, 2: 80, 3: 100}
NODE_W_S0 = {0: 400, 1: 700, 2: 1000, 3: 1400}
NODE_MFMA_PER_WAVE_F16_S0 = {0: 57 + 24 * 2 + 40, 1: 57 + 24 * 4 + 70, 2: 57 + 24 * 6 + 100, 3: 57 + 24 * 8 + 140}
NUMEL_NOT_L0 = {0: 200, 1: 600, 2: 1000, 3: 1200}              # radial-MLP outputs of the paths a DP_EDGES_LIST launch evaluates
LIST_TP_FLOPS = {0: 1200, 1: 2800, 2: 4400, 3: 5600}           # 2 * sum over those paths of mul_in * mul_out * (2 l_out + 1)
MFMA_F32_PEAK_TFLOPS = 157.3                                      # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0                                    # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparsity figure)
HBM_PEAK_GBPS = 8000.0                                           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
PREC_OF_CONFIG = {5: "bf16"}                                      # configs[4] is "bf16 mixed precision"; every other config fp32
PROF_LAYER, PROF_CONV = 3, 1                                     # layer-3 phore->ligand conv: layer 3 has no pharmacophore side, its ligand-side launch runs ALONE
                                                                 # (the step's longest launch by wall is layer 2's, beside the layer's other side)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "conv_hbm_traffic.json")


def kernel_source_hash() -> str:
    """The hash of the kernel sources the LOADED library was compiled from (dp_build_info): a PMC traffic figure is only quoted for the
    build it was measured on."""
    from diffphore_amd import _lib
    return _lib.build_info()["source_hash"]


def measured_traffic(config, poses, precision, conv_algo):
    """HBM bytes per launch of the roofline kernel from committed rocprofv3 PMC passes, or None if no profile matches."""
    try:
        recs = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None
    src = kernel_source_hash()
    if isinstance(recs, dict):
        recs = [recs]
    for r in recs:
        if not isinstance(r, dict):
            continue
        if (r.get("source_hash") == src and r.get("config") == config and r.get("poses") == poses and r.get("precision") == precision
                and r.get("conv_algo", "edge") == conv_algo):
            return int(r["traffic_bytes"]), r.get("profile")
    return None, None


class HipEvents:
    def __init__(self, n):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]
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

    def elapsed(self, a, b):
        self.hip.hipEventSynchronize(b)
        out = C.c_float()
        return out.value if self.hip.hipEventElapsedTime(C.byref(out), a, b) == 0 else float("nan")

    def destroy(self):
        for e in self.ev:
            self.hip.hipEventDestroy(e)
        self.ev = []


def cpu_baseline(graphs, sd, cfg, margs, n_poses, n_steps, S_total):
    """Reference-style CPU path (oracle, fp32) on a bounded sample of the workload: warm-up discarded, median of 3."""
    import copy

    import torch
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from oracle import sampler as osampler
    from oracle.model import OracleScoreModel
    from oracle.tables import load_tables
    om = OracleScoreModel(sd, cfg, dtype=torch.float32, tables=load_tables())
    sched = get_t_schedule(S_total)[:n_steps + 1]
    n_poses = min(n_poses, len(graphs))

    def once(np_=None, ns_=None):
        np_, ns_ = np_ or n_poses, ns_ or n_steps
        sample = copy.deepcopy(graphs[:np_])
        t0 = time.perf_counter()
        osampler.sampling_phore(sample, om, ns_, sched[:ns_ + 1], sched[:ns_ + 1], sched[:ns_ + 1], t_to_sigma_fn(margs), margs, batch_size=20)
        return time.perf_counter() - t0

    # BASELINE.md 3 planned os.cpu_count() threads.  Measured on the 256-thread GPU host (round 2, profiles/r2_cpu_threads.md): a
    # 4-pose x 1-step probe takes 0.21 s with 32 torch threads and 175.9 s with 256 (fork/join per small op): the baseline
    # uses min(cpu_count, 32) threads, which is the FASTER setting for the CPU side.
    ncpu = os.cpu_count() or 1
    best = min(ncpu, 32)
    torch.set_num_threads(best)
    once()                                           # warm-up, discarded
    runs = sorted(once() for _ in range(3))
    dt = runs[1]
    return dict(value=n_poses * n_steps / dt, unit="pose-steps/s", cores=best, kind="port",
                sample=(f"{n_poses} poses x {n_steps} steps of the same workload, oracle (torch-CPU fp32 restatement), warm-up discarded, "
                        f"median of 3 = {dt:.2f} s (runs {', '.join(f'{r:.2f}' for r in runs)}); {best} torch threads of {ncpu} logical CPUs "
                        f"(all {ncpu} is ~800x slower on these small ops: profiles/r2_cpu_threads.md)"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks on this node, one per GPU, and relay rank 0's JSON line."""
    import torch
    backend = os.environ.get("DIFFPHORE_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < n and backend == "nccl":
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible (set DIFFPHORE_DIST_BACKEND=gloo to smoke-test several ranks on one GPU)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100 = 5 x the 20-step schedule)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4, 5],
                    help="BASELINE.json configs index + 1 (default 2 = configs[1], the 512-pair workload)")
    ap.add_argument("--poses", type=int, default=None, help="configs 1-3: poses per GPU (default 512/512/256)")
    ap.add_argument("--units", type=int, default=None, help="configs 4/5: ligands / pharmacophores per GPU (x40 samples each; default 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=None, choices=["f32", "bf16x6", "bf16x9", "bf16x3", "bf16"],
                    help="arithmetic of the radial-MLP contraction (default: f32; bf16 = plain bf16 operands with fp32 accumulate for config 5, "
                         "BASELINE's 'bf16 mixed precision')")
    ap.add_argument("--conv-algo", default=None, choices=["edge", "node"], help="fused conv formulation (default: library default)")
    ap.add_argument("--no-alt", action="store_true", help="skip the informational passes at other precisions / formulations")
    ap.add_argument("--no-e2e", action="store_true", help="skip the drop-in sampling_phore() end-to-end pass")
    ap.add_argument("--no-other", action="store_true", help="skip the short passes over the other BASELINE configs (default run only)")
    ap.add_argument("--cpu-poses", type=int, default=20)
    ap.add_argument("--cpu-steps", type=int, default=4)
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    backend = os.environ.get("DIFFPHORE_DIST_BACKEND", "nccl")      # "nccl" = RCCL over xGMI on ROCm
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an AMD GPU (no CPU path exists for the hot path)")
    if local >= ndev and backend == "nccl":
        raise SystemExit(f"LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    local = local % ndev                                             # (gloo smoke runs may put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import types

    from diffphore_amd import config, weights, workloads
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.engine import DeviceModel, Engine, PackedBatch
    from diffphore_amd.sampling import _step_params, randomize_position

    cfg = config.ModelConfig()
    margs = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                  "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
    units = (a.poses if a.config in (1, 2, 3) else a.units) or workloads.DEFAULT_UNITS[a.config]
    precision = a.precision or (PREC_OF_CONFIG.get(a.config, "f32"))
    sd = weights.init_state_dict()
    models = {}

    bcast_ms = {}

    def model_for(prec):
        if prec not in models:
            models[prec] = DeviceModel(sd, cfg, dev, precision=prec)
            if world > 1:
                # the only collective of the job: weights from rank 0 to every GPU over xGMI
                torch.cuda.synchronize()
                t_b = time.perf_counter()
                dist.broadcast(models[prec].blob, src=0)
                torch.cuda.synchronize()
                bcast_ms[prec] = (time.perf_counter() - t_b) * 1e3
        return models[prec]

    class Workload:
        """One BASELINE config on this rank: poses packed on the device, randomised start state, schedule, pre-drawn noise."""

        def __init__(self, config, n_units):
            self.config, self.units = config, n_units
            self.graphs, _mine = workloads.build(config, n_units, rank, world)
            self.S_total = workloads.steps_of(config)
            # initial state: randomize_position on the device (same call the sampler's callers make, inference.py:185)
            np.random.seed(607201 + rank)
            torch.manual_seed(607201 + rank)
            randomize_position(self.graphs, False, False, cfg.tr_sigma_max, device=dev)
            self.pb = PackedBatch(self.graphs, dev)
            self.pos0, self.norm0 = self.pb.pos.clone(), self.pb.norm.clone()
            self.sched = get_t_schedule(self.S_total)
            gen = torch.Generator(device=dev).manual_seed(607201 + rank)
            self.B, self.nT = self.pb.B, int(self.pb.tor_ptr[-1])
            B, nT, S = self.B, self.nT, self.S_total
            # pre-drawn noise resident in HBM (the reference draws it on the host inside the loop, sampling.py:230-243); one set per step
            self.zs = [(torch.randn(B, 3, device=dev, generator=gen), torch.randn(B, 3, device=dev, generator=gen),
                        torch.randn(max(nT, 1), device=dev, generator=gen)) for _ in range(S)]
            self.sps = [_step_params(t, S, self.sched, self.sched, self.sched, t2s, margs, False, False) for t in range(S)]
            self.T_dev = torch.tensor([[float(self.sched[t])] * 3 for t in range(S)], dtype=torch.float32, device=dev)   # diffusion times, resident

        def timed_run(self, mdl, conv_algo, steps, warmup, prof=True, edge_x6=None, table_steps=0, node_f16=None, env=None):
            """-> dict.  `edge_x6`: "0" forces exact fp32 on the edge-side convs of an fp32 model (DIFFPHORE_EDGE_X6), `node_f16`: "0" forces the
            exact-fp32 MFMA kernel on its node-side convs (DIFFPHORE_NODE_F16); `table_steps` > 0 adds an untimed pass that brackets EVERY
            tensor-product launch of the interaction layers (DpWork.prof_events) for the per-launch table."""
            pb, B, S_total, zs, sps, nT = self.pb, self.B, self.S_total, self.zs, self.sps, self.nT
            envs = {"DIFFPHORE_EDGE_X6": edge_x6, "DIFFPHORE_NODE_F16": node_f16}
            envs.update(env or {})
            keep = {k_: os.environ.get(k_) for k_ in envs}
            for k_, v_ in envs.items():
                if v_ is not None:
                    os.environ[k_] = v_
            try:
                eng = Engine(mdl, pb, conv_algo=conv_algo)
            finally:
                for k_, v_ in envs.items():
                    if v_ is not None:
                        if keep[k_] is None:
                            os.environ.pop(k_, None)
                        else:
                            os.environ[k_] = keep[k_]
            ev = HipEvents(steps if prof else 0)

            def reset():
                pb.pos.copy_(self.pos0)
                pb.norm.copy_(self.norm0)
                eng.ws.t["status"].zero_()

            def step(i, prof_idx=None):
                t_idx = i % S_total
                if t_idx == 0:
                    reset()
                pb.set_time_row(self.T_dev[t_idx])
                w = eng.ws.struct
                if prof_idx is None:
                    w.prof_start, w.prof_stop = None, None
                else:
                    s0, s1 = ev.pair(prof_idx)
                    w.prof_start, w.prof_stop, w.prof_layer, w.prof_conv = s0, s1, PROF_LAYER, PROF_CONV
                z = zs[t_idx]
                eng.sample_step(sps[t_idx], z[0], z[1], z[2] if nT else None)

            for i in range(warmup):
                step(i)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i, prof_idx=i if prof else None)
            torch.cuda.synchronize()
            el_own = time.perf_counter() - t0                       # this rank's own time (before it waits for the others)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            rank_s = [el_own]
            if dist is not None:
                te = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                el = float(te.item())
                allr = [torch.zeros(2, device=dev, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(allr, torch.tensor([el_own, float(B)], device=dev, dtype=torch.float64))
                rank_s = [float(x[0].item()) for x in allr]
                self.rank_poses = [int(x[1].item()) for x in allr]
            else:
                self.rank_poses = [B]
            nbad = int((eng.ws.t["status"] != 0).sum().item())
            fin = bool(torch.isfinite(pb.pos).all().item())
            kms = float(np.nanmean([ev.ms(i) for i in range(steps)])) if prof else float("nan")
            ev.destroy()
            # edge counts of the last forward (the radius graphs depend on the poses): per-slot TF/s in profiles/*_conv_slots.csv
            t = eng.ws.t
            self.edge_counts = {"x": int(pb.x_ptr[-1]), "pp": int(pb.pp_ptr[-1]), "center": int(pb.lig_ptr[-1]),
                                "ll": int((t["ll_rowend"] - t["ll_rowptr"]).sum().item()),
                                "tor": int((t["t_rowend"] - t["t_rowptr"])[:max(nT, 0)].sum().item()) if nT else 0}
            self.prof_launch = eng.launch_of(PROF_LAYER, PROF_CONV)      # (first conv, count, node-side) of the launch the events bracket
            table = None
            if table_steps > 0:
                # per-launch time table: every TP launch of the four interaction layers bracketed, one step at a time (not part of `value`)
                launches = {}
                for layer in range(4):
                    for conv in range(6 if layer < 3 else 3):
                        f, n, ns = eng.launch_of(layer, conv)
                        launches[(layer, f)] = (n, ns)
                te_ = HipEvents(24)
                arr = (C.c_void_p * 48)()
                for (layer, f) in launches:
                    e0, e1 = te_.pair(layer * 6 + f)
                    arr[(layer * 6 + f) * 2], arr[(layer * 6 + f) * 2 + 1] = e0, e1
                w = eng.ws.struct
                w.prof_events = C.cast(arr, C.c_void_p)
                acc = {k_: [] for k_ in launches}
                for i in range(table_steps):
                    step(i)
                    torch.cuda.synchronize()
                    for (layer, f) in launches:
                        acc[(layer, f)].append(te_.ms(layer * 6 + f))
                w.prof_events = None
                te_.destroy()
                table = [{"layer": layer, "first_conv": f, "convs": n, "node_side": bool(ns), "kind": int(ns), "ms": float(np.nanmean(acc[(layer, f)]))}
                         for (layer, f), (n, ns) in sorted(launches.items())]
            # what the node-side launches of THIS batch run: the two-limb fp16 kernel has no path-split form (small batches stay on the fp32 kernel)
            nsc = eng.node_side_convs if (eng.conv_algo == "node" and not (eng.node_side_convs == "f16x3" and eng.path_split_on())) else \
                ("f32" if eng.node_side_convs == "f16x3" else eng.node_side_convs)
            return dict(elapsed=el, kernel_ms=kms, bad=nbad, finite=fin, algo=eng.conv_algo, edge_side_convs=eng.edge_side_convs, node_side_convs=nsc,
                        rank_seconds=rank_s, launch_table=table)

    t2s = t_to_sigma_fn(margs)
    wl = Workload(a.config, units)
    graphs, S_total, pb0, B, nT = wl.graphs, wl.S_total, wl.pb, wl.B, wl.nT
    sched = wl.sched
    model = model_for(precision)

    main = wl.timed_run(model, a.conv_algo, a.steps, a.warmup, table_steps=min(a.steps, 10))
    elapsed, kern_ms, bad, finite, algo_used = main["elapsed"], main["kernel_ms"], main["bad"], main["finite"], main["algo"]
    prof_launch, edge_counts = wl.prof_launch, dict(wl.edge_counts)
    rank_poses = list(wl.rank_poses)
    # the same step with everything on ONE stream (DIFFPHORE_TWO_STREAMS=0): a per-launch table whose launches do not overlap - clean per-launch
    # fractions for layers 0-2, whose two sides share the chip in the default run (never `value`)
    single = None
    if a.config == 2 and a.conv_algo is None:
        try:
            single = wl.timed_run(model, a.conv_algo, min(a.steps, 20), 3, prof=False, table_steps=min(a.steps, 10), env={"DIFFPHORE_TWO_STREAMS": "0"})
        except Exception as e:      # noqa: BLE001 - informational
            single = {"error": str(e)[:200]}
    pb = pb0
    totals = torch.tensor([B, int(pb.x_ptr[-1])], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    total_poses = int(totals[0].item())

    alt = None
    if not a.no_alt and a.config == 2 and a.precision is None and a.conv_algo is None:
        # informational passes (never the headline `value`): other precisions / the other conv formulation, same workload
        alt = []
        for prec, algo, x6, f16, what in (
                ("f32", None, "0", "0", "exact fp32 MFMA EVERYWHERE (DIFFPHORE_NODE_F16=0 DIFFPHORE_EDGE_X6=0: node-side convs on v_mfma_f32_16x16x4_f32, the "
                                        "edge-side convs - phore-phore, torsion bonds - on the fp32 pipe too)"),
                ("f32", None, None, "0", "round 4's node-side kernels: node-side convs exact fp32 MFMA (DIFFPHORE_NODE_F16=0), edge-side convs as in the headline"),
                ("f32", "edge" if algo_used != "edge" else "node", None, None, "fp32 model, the other conv formulation"),
                ("bf16x9", None, None, None, "3-way bf16 split of both operands, all 9 cross products: every fp32 x fp32 product exact, fp32 accumulate"),
                ("bf16x6", None, None, None, "same split, the 6 largest cross products (~2^-24 relative); passes the same parity tolerances"),
                ("bf16", None, None, None, "plain bf16 operands, fp32 accumulate (configs[4]'s precision) on the configs[1] batch: bf16-grade, 2^-9 per operand")):
            try:
                r2 = wl.timed_run(model_for(prec), algo, a.steps, a.warmup, edge_x6=x6, node_f16=f16)
            except Exception as e:      # noqa: BLE001 - an informational pass must never sink the headline line
                alt.append({"precision": prec, "conv_algo": algo, "error": str(e)[:200]})
                continue
            alt.append({"precision": f"{prec} ({what})", "conv_algo": r2["algo"], "edge_side_convs": r2["edge_side_convs"], "node_side_convs": r2["node_side_convs"],
                        "value": total_poses * a.steps / r2["elapsed"], "ms_per_step": r2["elapsed"] / a.steps * 1e3, "kernel_ms": r2["kernel_ms"],
                        "finite": r2["finite"], "kabsch_flagged_poses": r2["bad"]})

    e2e = None
    if not a.no_e2e and world == 1:
        # secondary figure: the whole drop-in call sampling_phore(data_list, ...) - host packing of the HeteroData list, noise
        # draw + upload, device loop, write-back over PCIe - on the same poses (never `value`: that is HBM-resident)
        import copy

        from diffphore_amd.sampling import sampling_phore
        from diffphore_amd.score_model import TensorProductScoreModel
        m = TensorProductScoreModel(device=dev, precision=precision)
        m.load_state_dict(sd)
        gl = [g.sample_copy() for g in graphs]
        # warm-up on the WHOLE pose list, two steps: a 512-pose batch launches kernels an 8-pose one never does (fused launches instead of the conv split,
        # the s0-only / list kernels of the normals convs) and the runtime loads a kernel's code at its first launch (~10 ms, seen as 75 instead of 62 ms of loop)
        sampling_phore([g.sample_copy() for g in graphs], m, 2, sched, sched, sched, dev, t2s, margs, batch_size=20)
        torch.cuda.synchronize()
        import gc
        gc.collect()             # a generation-2 collection inside the 0.14 s timed call would be a third of it (seen: 20 -> 110 ms host stage)
        tm = {}
        t0 = time.perf_counter()
        sampling_phore(gl, m, S_total, sched, sched, sched, dev, t2s, margs, batch_size=20, timing=tm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e2e = {"what": "drop-in sampling_phore(data_list) incl. host packing, noise draw/upload and write-back (reference-order normals from the global "
                       "CPU generator: the first 2 steps' before the first launch, the others while the GPU runs those)", "poses": len(gl),
               "steps": S_total, "seconds": dt, "value": len(gl) * S_total / dt, "unit": "pose-steps/s",
               "breakdown_s": {k: round(v, 4) for k, v in tm.items()}}
        # the same call with the noise drawn on the device (seed=): no S x N x (6 + T) normals drawn by one host thread and uploaded
        gl = [g.sample_copy() for g in graphs]
        sampling_phore(gl[:8], m, 2, sched, sched, sched, dev, t2s, margs, batch_size=20, seed=1)     # warm-up (first launch of the RNG kernels)
        torch.cuda.synchronize()
        gc.collect()
        tm2 = {}
        t0 = time.perf_counter()
        sampling_phore(gl, m, S_total, sched, sched, sched, dev, t2s, margs, batch_size=20, timing=tm2, seed=607201)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        e2e["seeded"] = {"what": "same call, noise drawn on the device (dp_noise_normals, seed=)", "seconds": dt2, "value": len(gl) * S_total / dt2,
                         "breakdown_s": {k: round(v, 4) for k, v in tm2.items()}}
        del copy

    e2e_one = None
    if not a.no_e2e and world == 1 and a.config == 2 and a.precision is None and a.conv_algo is None:
        # the reference's OWN call pattern (src/inference.py:184-209): ONE complex, 40 samples, 20 steps - randomize_position + sampling_phore on the
        # 40 copies, wall clock of the two calls (host packing, noise, device loop, write-back), median of 5 after a warm-up; latency-bound
        from diffphore_amd import synthetic
        from diffphore_amd.sampling import sampling_phore
        from diffphore_amd.score_model import TensorProductScoreModel
        m1 = TensorProductScoreModel(device=dev, precision=precision)
        m1.load_state_dict(sd)
        sched20 = get_t_schedule(20)
        e2e_one = []
        for what, (na, nt_, npn) in (("configs[0] shape (20 atoms, 5 torsions, 79 pharmacophore nodes)", (20, 5, 79)),
                                     ("configs[1] shape (30 atoms, 8 torsions, 32 pharmacophore nodes)", (30, 8, 32))):
            try:
                g0 = synthetic.synthetic_pairs(71, 1, na, nt_, npn)[0]
                runs = []
                for rep in range(6):
                    gl = [g0.sample_copy() for _ in range(40)]
                    np.random.seed(11 + rep); torch.manual_seed(11 + rep)
                    tm1 = {}
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    randomize_position(gl, False, False, cfg.tr_sigma_max, device=dev)
                    t1 = time.perf_counter()
                    sampling_phore(gl, m1, 20, sched20, sched20, sched20, dev, t2s, margs, batch_size=40, timing=tm1)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    runs.append((t2 - t0, t1 - t0, tm1.get("loop_s", float("nan")), tm1.get("pack_and_noise_s", float("nan")), tm1.get("write_back_s", float("nan"))))
                runs = sorted(runs[1:])                      # first = warm-up (kernel load, engine caches)
                med = runs[len(runs) // 2]
                e2e_one.append({"shape": what, "samples": 40, "steps": 20, "wall_ms": med[0] * 1e3, "randomize_position_ms": med[1] * 1e3,
                                "device_loop_ms": med[2] * 1e3, "ms_per_step_device": med[2] * 1e3 / 20, "pack_and_noise_ms": med[3] * 1e3,
                                "write_back_ms": med[4] * 1e3, "pose_steps_per_s": 40 * 20 / med[0]})
            except Exception as e:      # noqa: BLE001 - informational
                e2e_one.append({"shape": what, "error": str(e)[:200]})

    e2e_screen = None
    if not a.no_e2e and world == 1 and (a.config in (4, 5) or (a.config == 2 and not a.no_other and a.precision is None and a.conv_algo is None and a.poses is None)):
        # secondary figure for the library workloads (configs[3] virtual screen / configs[4] target fishing): the streaming driver
        # distributed.screen over a lazily generated library 6x the per-GPU bench size - units are built on the host (`load_unit`) while
        # the GPU samples the previous device batch, start randomisation and noise are drawn on the device (seed), every finished
        # complex is handed to a sink and dropped.  Reported: whole-call pose-steps/s, the host time spent building units (hidden behind
        # the device loop except for the first batch) and the process RSS before / after (flat = nothing accumulates).
        import resource

        from diffphore_amd import distributed as D
        from diffphore_amd import synthetic
        from diffphore_amd.score_model import TensorProductScoreModel
        c = a.config if a.config in (4, 5) else 4
        n_lib = 6 * (units if a.config in (4, 5) else workloads.DEFAULT_UNITS[c])     # 6 device batches: the pipeline's fill / drain is a sixth of the run
        lib = synthetic.ScreenLibrary(c, n_lib)
        prec_c = PREC_OF_CONFIG.get(c, "f32")
        m = TensorProductScoreModel(device=dev, precision=prec_c)
        m.load_state_dict(sd)
        sargs = types.SimpleNamespace(**vars(margs), inference_steps=workloads.steps_of(c) if c == 4 else 20)
        t_load = [0.0]

        def load(i):
            t1 = time.perf_counter()
            g = lib.unit(i)
            t_load[0] += time.perf_counter() - t1
            return g
        n_done, n_failed = [0], [0]
        per_batch = workloads.DEFAULT_UNITS[c] * workloads.SAMPLES
        D.screen(2, load, m, sargs, samples_per_complex=2, max_poses=per_batch, seed=1, on_result=lambda i, arr: None)      # warm-up
        torch.cuda.synchronize()
        t_load[0] = 0.0
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        t0 = time.perf_counter()
        marks = []          # (seconds since the call, poses delivered so far) per finished complex

        def sink(i, arr):
            if arr is None:                     # a unit screen() gave up on (reason in its `failed` list): counted, not timed as work
                n_failed[0] += 1
                return
            n_done[0] += arr.shape[0]
            marks.append((time.perf_counter() - t0, n_done[0]))
        D.screen(n_lib, load, m, sargs, costs=[lib.cost(i) for i in range(n_lib)], samples_per_complex=workloads.SAMPLES, max_poses=per_batch,
                 seed=607201, on_result=sink)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # rate once the pipeline is full: from the moment the first device batch has been delivered to the end of the call
        t_fill = next((t for t, n in marks if n >= per_batch), dt)
        steady = (n_done[0] - per_batch) * sargs.inference_steps / (dt - t_fill) if dt > t_fill and n_done[0] > per_batch else None
        e2e_screen = {"what": f"distributed.screen (streaming) over a lazily built configs[{c - 1}] library: {n_lib} units x {workloads.SAMPLES} samples, "
                              f"device batches of {per_batch} poses, on-device start randomisation + noise (seed)", "poses": n_done[0],
                      "steps": sargs.inference_steps, "precision": prec_c, "seconds": dt, "value": n_done[0] * sargs.inference_steps / dt,
                      "unit": "pose-steps/s", "failed_units": n_failed[0], "first_batch_delivered_s": round(t_fill, 3), "steady_value": steady,
                      "steady_note": "poses after the first device batch / time after its delivery: the rate of a long library, where the start-up "
                                     "(units of the first batch built and packed before anything runs) is paid once; a kernel trace of this call "
                                     "shows no GPU idle gap >= 0.3 ms between the first launch and the last write-back (tools/trace_gaps.py)",
                      "host_unit_build_s": round(t_load[0], 3), "max_rss_mib_before": rss0 // 1024,
                      "max_rss_mib_after": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024}

    n_x = int(pb.x_ptr[-1])

    def launch_work(pbx, ecounts, layer, convs, prec, s0=False):
        """`s0`: the launch is a normals conv as s0-only node-side launch + DP_EDGES_LIST launch over its listed pairs (both inside the bracket).
        Work of ONE launch of interaction layer `layer` that carries `convs` (numbering of DpWork.prof_conv): node-form algorithmic flops
        (DESIGN.md 5.1), SURVEY.md 8d's edge-side flops and operator-boundary bytes, and the MFMA flops the node-side kernel issues.
        `prec`: which node-side kernel - "f32" (fp32 MFMA, 16-node groups), "bf16" (bf16 operands, 16-node groups) or "f16x3" (two fp16 limbs
        per operand, 8-node groups)."""
        nl_, np__ = [int(x) for x in pbx.nl], [int(x) for x in pbx.np_]
        nB = len(nl_)
        n_x_, n_ll_ = int(pbx.x_ptr[-1]), ecounts["ll"]
        lig_side = convs[0] < 3
        n_scat = sum(nl_) if lig_side else sum(np__)
        edges = {c: (n_ll_ if c == 0 else n_x_) for c in convs}
        gather = {c: (sum(nl_) if c in (0, 4, 5) else sum(np__)) for c in convs}
        n_edges = sum(edges.values())
        cols, wrows = (NODE_COLS_S0, NODE_W_S0) if s0 else (NODE_COLS, NODE_W)
        node_flops = sum(edges[c] * 2 * 61 * (20 + cols[layer]) + n_scat * 2 * 61 * (21 + wrows[layer]) + gather[c] * 2 * 61 * 20 for c in convs)
        n_listed = int(pbx.struct.n_xs) if s0 else 0
        list_flops = n_listed * (2 * (60 * 60 + 60 * NUMEL_NOT_L0[layer]) + LIST_TP_FLOPS[layer])        # edge-side count of the listed pairs' paths
        node_flops += list_flops
        bf = prec in ("bf16", "f16x3")            # 16x16x32 MFMAs on the 16-bit pipe
        grp = 8 if prec == "f16x3" else 16        # scatter nodes per workgroup

        def pass_units(n_slots):      # 32-slot passes of a node; fp32 kernel: a last pass of <= 16 slots is half a pass (16-bit kernels: GEMM-A / GEMM-B MFMAs are not halved)
            full, rem = divmod(n_slots, 32)
            return full + (0 if rem == 0 else (0.5 if (rem <= 16 and not bf) else 1.0))
        scat_i = nl_ if lig_side else np__
        slots = {c: (nl_ if (c == 0 or not lig_side) else np__) for c in convs}
        passes = sum(((scat_i[i] + grp - 1) // grp) * pass_units(slots[c][i]) for c in convs for i in range(nB))
        per_wave = (NODE_MFMA_PER_WAVE_F16_S0 if s0 else {"bf16": NODE_MFMA_PER_WAVE_BF16, "f16x3": NODE_MFMA_PER_WAVE_F16}.get(prec, NODE_MFMA_PER_WAVE))[layer]
        issued = passes * 4 * per_wave * (16384.0 if bf else 2048.0) + 3 * list_flops      # (+ the list launch's three limb products)
        return dict(edges=n_edges, node_flops=node_flops, listed_pairs=n_listed, edge_flops=ALG_FLOPS_PER_EDGE[layer] * n_edges, alg_bytes=ALG_BYTES_PER_EDGE[layer] * n_edges,
                    issued_flops=issued, workgroups=int(sum((n + grp - 1) // grp for n in scat_i)))

    def step_weighted(pbx, ecounts, table, prec, peak):
        """sum of node-form flops / sum of launch time over every NODE-SIDE launch of a step (from the per-launch table)"""
        if not table:
            return None
        fl = tm = iss = 0.0
        rows = []
        for r_ in table:
            if not r_["node_side"] or not (r_["ms"] > 0):
                continue
            cv = list(range(r_["first_conv"], r_["first_conv"] + r_["convs"]))
            wk = launch_work(pbx, ecounts, r_["layer"], cv, prec, s0=(r_.get("kind") == 2))
            fl += wk["node_flops"]; tm += r_["ms"]; iss += wk["issued_flops"]
            rows.append({"layer": r_["layer"], "convs": cv, "ms": round(r_["ms"], 4), "TFLOPs": wk["node_flops"] / (r_["ms"] * 1e-3) / 1e12})
            if r_.get("kind") == 2:
                rows[-1]["s0_only_plus_listed_pairs"] = wk["listed_pairs"]
        if tm <= 0:
            return None
        return {"achieved": fl / (tm * 1e-3) / 1e12, "frac": fl / (tm * 1e-3) / 1e12 / peak, "issued_frac": iss / (tm * 1e-3) / 1e12 / peak,
                "node_launch_ms_per_step": tm, "launches": rows,
                "note": "sum of node-form algorithmic flops / sum of launch times over every node-side launch of a step; launch times from an untimed "
                        "extra pass that brackets each launch with HIP events (DpWork.prof_events), one step at a time.  When dp_forward runs the two sides "
                        "of a layer on two streams (the default with the f16x3 node-side kernels) the bracketed launches OVERLAP and stretch each other: "
                        "node_launch_ms_per_step then exceeds ms_per_step and this ratio is a lower bound"}

    other = None
    if not a.no_other and a.config == 2 and a.precision is None and a.conv_algo is None and a.poses is None:
        # every other BASELINE config in the same driver-run line (never the headline `value`): one short pass each at its default
        # per-GPU size, library-default conv formulation, the config's own precision; one schedule wrap (config 5: the first 20 steps of
        # its 100-step schedule), 3 warm-up steps
        other = []
        for c in (1, 3, 4, 5):
            try:
                t_b = time.perf_counter()
                w2 = Workload(c, workloads.DEFAULT_UNITS[c])
                prec_c = PREC_OF_CONFIG.get(c, "f32")
                n_steps = min(w2.S_total, 20)
                r2 = w2.timed_run(model_for(prec_c), None, n_steps, 3, prof=(c == 5), table_steps=(5 if c == 5 else 0))
                tot = torch.tensor([w2.B], device=dev, dtype=torch.float64)
                if dist is not None:
                    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
                rec = {"config": f"configs[{c - 1}]", "workload": workloads.describe(c, w2.units), "poses_per_gpu": w2.B,
                       "poses_total": int(tot.item()), "precision": prec_c, "conv_algo": r2["algo"], "edge_side_convs": r2["edge_side_convs"], "steps": n_steps,
                       "ms_per_step": r2["elapsed"] / n_steps * 1e3, "value": float(tot.item()) * n_steps / r2["elapsed"], "unit": "pose-steps/s",
                       "finite": r2["finite"], "kabsch_flagged_poses": r2["bad"], "edge_counts": w2.edge_counts,
                       "build_and_run_s": round(time.perf_counter() - t_b, 2)}
                if c == 5 and r2["kernel_ms"] > 0:
                    # configs[4]'s own roofline: the bf16-operand node-side kernel against the dense bf16 MFMA peak
                    f5, n5, ns5 = w2.prof_launch
                    wk = launch_work(w2.pb, w2.edge_counts, PROF_LAYER, list(range(f5, f5 + n5)), prec_c)
                    km = r2["kernel_ms"]
                    tr5, tr5_src = measured_traffic(5, w2.B, prec_c, r2["algo"]) if world == 1 else (None, None)
                    rec["roofline"] = {"bound": "mfma", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "kernel_ms": km, "traffic": tr5,
                                       "traffic_note": (f"HBM bytes per launch from rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE), {tr5_src}: mostly scratch - the "
                                                        "kernels park 138 / 242 / 214 registers (fused ligand-side launches of layers 1-3; cross pairs 69-100: tools/kernel_meta.py)") if tr5 is not None else None,
                                       "kernel": f"layer-{PROF_LAYER} ligand-side launch ({n5} conv(s)), bf16 operands / fp32 accumulate, node-side={bool(ns5)}",
                                       "achieved": wk["node_flops"] / (km * 1e-3) / 1e12, "frac": wk["node_flops"] / (km * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                       "issued_frac": wk["issued_flops"] / (km * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                       "step_weighted": step_weighted(w2.pb, w2.edge_counts, r2["launch_table"], prec_c, MFMA_BF16_PEAK_TFLOPS),
                                       "note": "this kernel is bound by VALU issue (the tensor-product operand z is built per node on the VALU) and LDS, not by "
                                               "the matrix pipe: SQ counters in profiles/r4_bf16_node_pmc.csv (DESIGN.md 5.4)"}
                other.append(rec)
                del w2
            except Exception as e:      # noqa: BLE001 - an informational pass must never sink the headline line
                other.append({"config": f"configs[{c - 1}]", "error": str(e)[:200]})
    # The profiled launch: the one that carries layer-3 phore->ligand, the longest launch of a step.  Node-side it is a FUSED launch
    # (dp_tp_conv_node_fused): the ligand-ligand conv + phore->ligand + phore->ligand(normals) of the layer back to back into one set of
    # accumulators, so the algorithmic work of the launch is the sum over its convs.  Edge-side it is the single conv.
    first, count, launch_node = prof_launch
    convs = list(range(first, first + count))
    node_kern = main["node_side_convs"] if precision == "f32" else precision        # which node-side kernel ran: f32 | f16x3 | bf16
    wk = launch_work(pb, edge_counts, PROF_LAYER, convs, node_kern)
    launch_edges, alg_bytes, flops = wk["edges"], wk["alg_bytes"], wk["edge_flops"]
    conv_names = {0: "ligand-ligand", 1: "phore->ligand", 2: "phore->ligand(normals)"}
    kernel_name = (f"layer-{PROF_LAYER} ligand-side launch = {' + '.join(conv_names[c] for c in convs)}"
                   + (" fused into one node-side launch" if count > 1 else "") + f", conv_algo={algo_used}, precision={precision}"
                   + (f", node-side kernel {node_kern}" if launch_node else ""))
    on16 = precision != "f32" or (launch_node and node_kern == "f16x3")        # the launch's MFMAs run on the 16-bit matrix pipe
    peak = MFMA_BF16_PEAK_TFLOPS if on16 else MFMA_F32_PEAK_TFLOPS
    if launch_node:
        rf_flops = wk["node_flops"]
        rf_note = (f"node-side formulation, summed over the {count} conv(s) of the launch: 2*61*(20+{NODE_COLS[PROF_LAYER]}) flop per edge (edge part of radial layer 1 + "
                   f"GEMM-A) + 2*61*(21+{NODE_W[PROF_LAYER]}) flop per scatter node (its layer-1 block + GEMM-B) + 2*61*20 per gather node, DESIGN.md 5.1; "
                   "the edge-side count of SURVEY.md 8d is reported beside it")
        issued = wk["issued_flops"]
        rf_issue = {"executed_flops_per_launch": issued, "achieved": issued / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None,
                    "frac": issued / (kern_ms * 1e-3) / 1e12 / peak if kern_ms > 0 else None, "workgroups": wk["workgroups"],
                    "note": "MFMA flops actually issued (padding included; fp32 kernel: a half pass counted as half of a 32-slot pass, the lig-lig "
                            "conv's second layer-1 sweep over bond edges not counted; f16x3 kernel: 3 limb products per GEMM1 / GEMM-A MFMA, 2 W2 limbs "
                            "x (T_hi | T_lo in N) per GEMM-B MFMA) / launch time = matrix-pipe occupancy"}
    else:
        rf_flops, rf_note, rf_issue = flops, "SURVEY.md 8d: 2*(60*60+60*numel)+TP flop per edge-conv (edge-side formulation)", None
    sw_main = step_weighted(pb, edge_counts, main["launch_table"], node_kern, peak)
    sw_single = None
    if single is not None and "error" not in single:
        sw_single = step_weighted(pb, edge_counts, single["launch_table"], node_kern, peak)
        if sw_single is not None:
            sw_single["ms_per_step_single_stream"] = single["elapsed"] / min(a.steps, 20) * 1e3
            sw_single["note"] = ("the same step with DIFFPHORE_TWO_STREAMS=0: nothing overlaps, every launch's time is its own (this arm is slower as a whole: "
                                 "ms_per_step_single_stream)")
    whole_step = None
    if sw_main is not None:
        fl_step = sw_main["achieved"] * 1e12 * sw_main["node_launch_ms_per_step"] * 1e-3          # sum of node-form flops over the node-side launches
        ms_step = elapsed / a.steps * 1e3
        whole_step = {"node_form_flops_per_step": fl_step, "achieved": fl_step / (ms_step * 1e-3) / 1e12, "frac": fl_step / (ms_step * 1e-3) / 1e12 / peak,
                      "unit": "TFLOP/s", "note": "node-form algorithmic flops of every node-side launch of a step (the normals convs counted as what they "
                                                 "evaluate: l = 0 paths on every pair + the other paths on the listed pairs) / ms_per_step / the 16-bit MFMA peak: the "
                                                 "fraction of the pipe the WHOLE step sustains, serial head and tail of the step included"}
    traffic, traffic_src = measured_traffic(a.config, B, precision, algo_used) if world == 1 else (None, None)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else float("nan")

    if rank == 0:
        dtype = {"f32": "f32", "bf16x9": "f32 (bf16x9 split, exact products, fp32 accumulate)",
                 "bf16x6": "f32 (bf16x6 split, ~2^-24 products, fp32 accumulate)", "bf16x3": "bf16x3 split (~2^-17 products), fp32 accumulate",
                 "bf16": "bf16 operands, fp32 accumulate"}[precision]
        if precision == "f32" and (main["edge_side_convs"] != "f32" or main["node_side_convs"] != "f32"):
            nd = ("f16x3 = every GEMM operand as two fp16 limbs (22 significand bits, exact power-of-two scales), limb cross products on "
                  "v_mfma_f32_16x16x32_f16, fp32 accumulate" if main["node_side_convs"] == "f16x3" else "exact fp32 MFMA")
            ed = ("f16x3 = both operands of the radial contraction as two fp16 limbs, 3 cross products on v_mfma_f32_32x32x16_f16, fp32 accumulate"
                  if main["edge_side_convs"] == "f16x3" else (main["edge_side_convs"] + " split" if main["edge_side_convs"] != "f32" else "exact fp32 MFMA"))
            dtype = (f"f32-grade (fp32 model held to the fp32 parity tolerances; node-side convs: {nd}; edge-side convs = phore-phore, torsion bonds "
                     f"[, ligand-ligand above 32 atoms]: {ed}; exact-fp32-MFMA-everywhere arm: value_exact_fp32)")
        rank_rate = [p_ * a.steps / t_ for p_, t_ in zip(rank_poses, main["rank_seconds"])]
        out = {
            "metric": "poses/sec (samples x steps) on the reverse-diffusion hot path",
            "value": total_poses * a.steps / elapsed,
            "unit": "pose-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": workloads.describe(a.config, units) + f", one reverse-diffusion step per bench step ({S_total}-step schedule, "
                                   f"state reset at every wrap), random-init weights",
                       "poses_per_gpu": B, "poses_total": total_poses, f"poses_per_sec_at_{S_total}_steps": total_poses * a.steps / elapsed / S_total,
                       "parallelism": f"independent poses sharded over {world} GPU(s), RCCL weight broadcast only",
                       "kabsch_flagged_poses": bad, "finite": finite, "precision": precision, "conv_algo": algo_used,
                       "edge_side_convs": main["edge_side_convs"], "node_side_convs": main["node_side_convs"],
                       "node_side_convs_note": "what the node-side launches (ligand-ligand + cross convs: 21 per forward) run: 'f16x3' = the fp32-grade two-limb "
                                               "fp16 kernel (dp_conv_node_f16_impl.h; library default DpWork.node_f16 = 1, same parity tolerances as fp32: "
                                               "tests/test_conv_f16_gpu.py 2-4e-7 of max vs the fp64 oracle), 'f32' = v_mfma_f32_16x16x4_f32 (DIFFPHORE_NODE_F16=0)",
                       "edge_side_convs_note": "what the convs that stay on the edge-side kernel (phore-phore, torsion bonds; ligand-ligand above 32 atoms) run: "
                                               "'f16x3' = 2-way fp16 operand split, 3 cross products, fp32 accumulate (fp32-grade: same parity tolerances; the library default, "
                                               "DpWork.edge_bf16x6 = 1 - the 3-way bf16 split 'bf16x6' until round 4), 'f32' = exact fp32 MFMA (alt_runs[0] times that arm)",
                       "library": dict(_build_info()),
                       "ranks": {"pose_steps_per_s": rank_rate, "seconds": main["rank_seconds"], "poses": rank_poses,
                                 "imbalance_max_over_mean": (max(main["rank_seconds"]) / (sum(main["rank_seconds"]) / len(main["rank_seconds"]))),
                                 "weight_broadcast_ms": bcast_ms.get(precision)},
                       "edge_counts": edge_counts, "alt_runs": alt, "e2e": e2e, "e2e_one_complex": e2e_one, "e2e_screen": e2e_screen, "other_configs": other},
            # The dominant kernel generates the per-edge TP weights on the matrix pipe in registers (they never reach HBM), so it is
            # bound by the matrix pipe (fp32 model) or by VALU issue (bf16 model); SURVEY.md 8d asks for both views when the radial MLP is fused into
            # the TP kernel.  alg_flops_per_edge is the EDGE-SIDE count of SURVEY.md 8d (radial GEMM per edge); the node-side formulation computes
            # the same result with fewer flops, so its `frac` under this count can exceed what the pipe could
            # deliver for the edge-side form - `mfma_issue` is what that launch really issues.
            "roofline": {"bound": "mfma", "achieved": rf_flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None,
                         "peak": peak, "unit": "TFLOP/s",
                         "frac": rf_flops / (kern_ms * 1e-3) / 1e12 / peak if kern_ms > 0 else None,
                         "alg_flops_per_launch": rf_flops, "flop_count": rf_note,
                         "step_weighted": step_weighted(pb, edge_counts, main["launch_table"], node_kern, peak),
                         "step_weighted_single_stream": sw_single,
                         "whole_step": whole_step,
                         "launch_note": ("the bracketed launch is the layer-3 ligand-side launch: layer 3 has no pharmacophore side, so it is the one conv launch "
                                         "of a step that runs ALONE on the chip (its events measure the kernel, not an overlap) - not the longest by wall: "
                                         "the layer-2 ligand-side launch is, sharing the chip with the layer's pharmacophore side (step_weighted.launches; "
                                         "step_weighted_single_stream has every launch on its own)"),
                         "edge_side_count": {"achieved": flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None,
                                             "frac": flops / (kern_ms * 1e-3) / 1e12 / peak if kern_ms > 0 else None,
                                             "note": "SURVEY.md 8d's per-edge count (radial GEMM per edge) x edges / launch time: what an edge-side "
                                                     "kernel would have to sustain for this launch time; > 1 is possible for the node-side form"},
                         "mfma_issue": rf_issue,
                         "traffic": traffic,
                         "kernel": "fused TP conv, " + kernel_name,
                         "kernel_ms": kern_ms, "edges_per_launch": launch_edges, "convs_in_launch": [conv_names[c] for c in convs],
                         "alg_flops_per_edge": ALG_FLOPS_PER_EDGE[PROF_LAYER], "alg_bytes_per_edge": ALG_BYTES_PER_EDGE[PROF_LAYER],
                         "hbm_equivalent": {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                                            "note": "algorithmic bytes of the e3nn operator boundary (per-edge weights counted as "
                                                    "HBM input, SURVEY.md 8d) / launch time; the metric's 'fused TP-scatter HBM GB/s'"},
                         "fp32_pipe_equivalent": ({"frac": rf_flops / (kern_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                                   "note": "the same algorithmic flops against the fp32-input MFMA peak (157.3 TFLOP/s), the pipe that bounded this "
                                                           "launch through round 4 (0.44): what moving the association to the 16-bit pipe bought"}
                                                  if (on16 and precision == "f32" and kern_ms > 0) else None),
                         "bound_note": ("the f16x3 kernel is bound by VALU ISSUE (limb extraction: v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 per operand pair, "
                                        "the tensor-product operand z built per node), not by the matrix pipe: SQ_VALU_MFMA_BUSY / (4 SQ_BUSY_CU) and the "
                                        "instruction mix in profiles/r5_f16_node_pmc.csv, timing ablations in profiles/r5_experiments.md") if (on16 and precision == "f32") else None,
                         "traffic_note": (f"HBM bytes per launch from rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction), {traffic_src}, "
                                          f"kernel sources {kernel_source_hash()}") if traffic is not None else
                                         "null: no committed PMC profile matches this library build / workload (profiles/conv_hbm_traffic.json)"},
        }
        if alt and "value" in alt[0]:
            # the exact-fp32-MFMA-everywhere arm as top-level fields (the driver record keeps only the head of config.alt_runs)
            out["value_exact_fp32"] = alt[0]["value"]
            out["ms_per_step_exact_fp32"] = alt[0]["ms_per_step"]
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(graphs, sd, cfg, margs, a.cpu_poses, a.cpu_steps, S_total)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _build_info():
    from diffphore_amd import _lib
    return _lib.build_info()


if __name__ == "__main__":
    main()
