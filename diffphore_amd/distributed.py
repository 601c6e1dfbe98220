"""Multi-GPU plumbing: independent poses sharded over ranks, one weight broadcast, optional result gather.

The reference has no distributed code; poses are independent through all reverse-diffusion steps
(sampling.py:213-255, BatchNorm in eval mode), so there is no data-path collective.  One process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm); "gloo" works for the CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional, List, Sequence, Tuple

import torch


def pose_cost(g) -> int:
    """Work estimate of one pose per step: dense cross edges dominate (4 of 6 conv families)."""
    nl, npn = int(g["ligand"].pos.shape[0]), int(g["phore"].pos.shape[0])
    return nl * npn + 25 * nl + 8 * npn


def shard_indices(costs: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first partition of pose indices into `world` shards (deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def shard_poses(graphs: Sequence, rank: int, world: int) -> Tuple[List[int], List]:
    idx = shard_indices([pose_cost(g) for g in graphs], world)[rank]
    return idx, [graphs[i] for i in idx]


def broadcast_weights(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """The job's only collective: rank `src`'s packed parameter blob to every rank (in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def gather_positions(local_idx: Sequence[int], local_pos: Sequence[torch.Tensor], n_total: int, dst: int = 0):
    """Collect final ligand coordinates on rank `dst` in the original pose order (CPU tensors; tiny: 12 B per atom)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = [None] * n_total
        for i, p in zip(local_idx, local_pos):
            out[i] = p
        return out
    payload = (list(local_idx), [p.cpu() for p in local_pos])
    gathered = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    out = [None] * n_total
    for idx, pos in gathered:
        for i, p in zip(idx, pos):
            out[i] = p
    return out


def fit_complexes(graphs: Sequence, model, args, samples_per_complex: int = 40, device="cuda", batch_size: int = 20,
                  max_poses: int = 8192):
    """The sampler-facing part of `inference.py::fit` (:139-228) on this rank's share of `graphs`:

        data_list = [deepcopy(graph) x N] ; randomize_position(...) ; sampling_phore(...) ; collect ligand_pos

    for every complex, but with ALL complexes x samples of the rank batched into device-resident launches instead of one
    40-pose call per complex.  Returns (indices of the complexes this rank owns, list of [N, n_atoms, 3] arrays in that order).
    Scoring / SDF writing (AncPhore, RDKit) stay with the caller.  `args` needs the reference's fields: inference_steps,
    tr/rot/tor_sigma_min/max, no_torsion, no_random, ode, no_final_step_noise (missing ones default like inference.py:54-96).
    """
    import copy

    import torch.distributed as dist

    from .diffusion import get_t_schedule, t_to_sigma_fn
    from .sampling import sampling_phore
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    idx, mine = shard_poses(graphs, rank, world)
    steps = int(getattr(args, "inference_steps", 20))
    sched = get_t_schedule(steps)
    out = []
    # One fused call: every device batch (whole complexes, <= max_poses poses) is packed once, randomised and sampled on the device,
    # and the host stage of the next batch overlaps the device loop of the current one (sampling_phore's pipeline).
    per = max(1, max_poses // samples_per_complex)
    data_list = [g.sample_copy() if hasattr(g, "sample_copy") else copy.deepcopy(g) for g in mine for _ in range(samples_per_complex)]
    sampling_phore(data_list, model, steps, sched, sched, sched, device, t_to_sigma_fn(args), args,
                   no_random=getattr(args, "no_random", False), ode=getattr(args, "ode", False), batch_size=batch_size,
                   no_final_step_noise=getattr(args, "no_final_step_noise", False), max_poses=per * samples_per_complex,
                   randomize=dict(no_torsion=getattr(args, "no_torsion", False), no_random=False, tr_sigma_max=args.tr_sigma_max))
    for c in range(len(mine)):
        poses = data_list[c * samples_per_complex:(c + 1) * samples_per_complex]
        out.append(torch.stack([p["ligand"].pos for p in poses]).numpy())
    return idx, out


def screen_plan(n_units: int, costs, rank: int, world: int, samples_per_complex: int = 40, max_poses: int = 8192) -> List[List[int]]:
    """Which units rank `rank` of `world` processes, cut into device batches (lists of unit indices, in processing order).  Pure host
    arithmetic on the cost list: every rank computes its own plan without building or exchanging anything.  With `costs` the shard is
    the LPT partition (`shard_indices`) walked heaviest first - similar sizes share a device batch (its launch geometry follows the
    largest pose), and inside a launch the workgroups, dispatched in pose order, end on the cheapest poses instead of on a straggler;
    without, units are dealt round-robin.  A batch holds whole complexes, at most `max_poses` poses (at least one complex)."""
    if costs is not None:
        if len(costs) != n_units:
            raise ValueError("costs needs one entry per unit")
        mine = sorted(shard_indices(costs, world)[rank], key=lambda i: (-costs[i], i))
    else:
        mine = list(range(rank, n_units, world))
    if int(n_units) * samples_per_complex >= 2 ** 31:
        raise ValueError("pose uids are 32-bit: n_units * samples_per_complex must stay below 2^31")
    per = max(1, max_poses // samples_per_complex)
    return [mine[lo:lo + per] for lo in range(0, len(mine), per)]


LIMITS = dict(atoms=256, phore_nodes=256, torsions=128)       # per-pose kernel limits (engine.PackedBatch)


def _unit_problem(g) -> Optional[str]:
    """Why a complex cannot go on the device (None = fine): the checks `PackedBatch` would fail the WHOLE device batch with."""
    try:
        nl, npn = int(g["ligand"].pos.shape[0]), int(g["phore"].pos.shape[0])
        nt = int(torch.as_tensor(g["ligand"].edge_mask).sum())
    except Exception as e:      # noqa: BLE001 - a malformed unit is the caller's data problem, not the run's
        return f"malformed complex: {type(e).__name__}: {e}"
    if nl < 1 or npn < 1:
        return "empty ligand or pharmacophore"
    if nl > LIMITS["atoms"] or npn > LIMITS["phore_nodes"] or nt > LIMITS["torsions"]:
        return f"over the kernel limits ({nl} atoms / {npn} pharmacophore nodes / {nt} torsions; limits {LIMITS})"
    return None


def screen(n_units: int, load_unit, model, args, *, costs: Sequence[int] = None, samples_per_complex: int = 40, device="cuda",
           max_poses: int = 8192, seed: int = 0, on_result=None, failed: Optional[list] = None, retries: int = 2):
    """Streaming form of `fit_complexes` for library-scale screening (BASELINE configs[3] / configs[4]: 10^4-10^5 complexes x 40 samples
    per GPU).  `load_unit(i)` builds complex i on demand (featurisation, parsing, ... stay with the caller); `costs[i]` (optional, cheap:
    e.g. atoms x pharmacophore nodes) drives the LPT shard, otherwise units are dealt round-robin.  This rank

        * takes ITS indices only (no rank ever builds or holds another rank's units),
        * walks them in device batches of <= max_poses poses (whole complexes), building batch k+1 - `load_unit` calls included - on the
          host while the GPU samples batch k (`sampling_phore_stream`),
        * draws start randomisation and step noise on the device from (`seed`, pose uid = unit * samples + sample): the NOISE of a pose does
          not depend on the world size, the batch size or the order of the stream; its trajectory is reproduced to fp32 rounding (the conv
          formulation and workgroup slicing follow the batch a pose lands in: <= 2e-5 A, so near-tie rankings may differ between layouts),
        * hands each finished complex to `on_result(i, poses [samples, n_atoms, 3] float32)` and drops it; with on_result=None the arrays
          are collected and returned (only sensible for small runs),
        * isolates failures per complex, as the reference's loop does with its try / except and SVD retries (src/inference.py:200-219): a
          `load_unit` exception, a complex over the kernel limits or a sample whose Kabsch alignment fails never ends the run.  A
          Kabsch-flagged complex is re-sampled up to `retries` times with a fresh noise key (the reference re-draws and retries up to five
          times; a complex whose status carries bit 2 - non-finite SCORES, which with the two-limb fp16 kernels of an fp32 model can mean a value
          beyond fp16's range - is first run again with the SAME key on the exact-fp32 kernels, see below, so what reaches the fresh-key
          retries is flagged in the reference's arithmetic too; a degenerate Kabsch alignment, bit 0, goes straight to the retries); what still fails
          is reported as `on_result(i, None)` (or a None entry of the returned list) and, when the caller passes a list as `failed`, appended to
          it as (i, reason).

    Returns (indices owned by this rank in processing order - heaviest first when `costs` is given -, list of pose arrays or None).  Peak host memory: two device batches of
    pose objects, whatever the library size."""
    import torch.distributed as dist

    from .diffusion import get_t_schedule, t_to_sigma_fn
    from .sampling import sampling_phore_stream
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    batches = screen_plan(n_units, costs, rank, world, samples_per_complex, max_poses)
    mine = [i for bt in batches for i in bt]
    ramp = os.environ.get("DIFFPHORE_SCREEN_RAMP", "1") != "0"      # 0: A/B switch - the first device batch whole, as the others
    if ramp and len(batches) >= 3 and len(batches[0]) >= 8:
        # geometric ramp: nothing runs until the first device batch is built and packed, so the first full batch is dealt as 1/8, 1/8, 1/4, 1/2 of
        # its units - the GPU starts after an eighth of the host work and each piece is built while the previous (at most equally long) one
        # runs; poses do not depend on the batching (seeded noise per pose uid)
        b0 = batches[0]
        e = max(1, len(b0) // 8)
        batches = [b0[:e], b0[e:2 * e], b0[2 * e:4 * e], b0[4 * e:]] + batches[1:]
    elif ramp and len(batches) >= 3 and len(batches[0]) >= 4:
        q = len(batches[0]) // 4
        batches = [batches[0][:q], batches[0][q:]] + batches[1:]
    steps = int(getattr(args, "inference_steps", 20))
    sched = get_t_schedule(steps)

    bad: dict = {}                                   # unit -> reason (never reached the device, or still flagged after the retries)

    def parts_of(batch_list, loaded):
        for bt in batch_list:
            poses, uids, units = [], [], []
            for i in bt:
                try:
                    g = load_unit(i)
                    why = _unit_problem(g)
                except Exception as e:      # noqa: BLE001 - isolate the unit, keep streaming
                    g, why = None, f"load_unit raised {type(e).__name__}: {e}"
                if why is not None:
                    bad[i] = why
                    continue
                units.append(i)
                for s_ in range(samples_per_complex):
                    poses.append(g.sample_copy() if hasattr(g, "sample_copy") else __import__("copy").deepcopy(g))
                    uids.append(i * samples_per_complex + s_)
            loaded.append(units)
            if poses:
                yield poses, uids

    kept = {} if on_result is None else None         # on_result=None: arrays are collected (small runs) and returned in processing order

    def deliver(i, arr):
        """a settled unit: its poses, or None (reason in `bad`)"""
        if arr is None and failed is not None:
            failed.append((i, bad[i]))
        if on_result is not None:
            on_result(i, arr)
        else:
            kept[i] = arr

    def settle(res_of_batch, final):
        """hand over what finished; return the units that were flagged (Kabsch) - they are failures only when `final`"""
        flagged = []
        for i, arr in res_of_batch:
            if arr is not None:
                deliver(i, arr)
            elif final:
                bad[i] = "Kabsch alignment failed to converge (every attempt)"
                done_bad.add(i)
                deliver(i, None)
            else:
                flagged.append(i)
        return flagged

    def run(batch_list, run_seed, final, exact=False):
        """stream `batch_list`; every finished complex is handed over at once, units that never reach the device are failures; -> flagged units"""
        loaded, flagged = [], []
        stream = sampling_phore_stream(parts_of(batch_list, loaded), model, steps, sched, sched, sched, device, t_to_sigma_fn(args), args, seed=run_seed,
                                       no_random=getattr(args, "no_random", False), ode=getattr(args, "ode", False),
                                       no_final_step_noise=getattr(args, "no_final_step_noise", False), isolate=True, exact=exact, replay=False,
                                       randomize=dict(no_torsion=getattr(args, "no_torsion", False), no_random=False, tr_sigma_max=args.tr_sigma_max))
        k = 0

        def report_bad():                             # units that never reached the device (load failure, over the limits)
            for i in list(bad):
                if i not in done_bad:
                    done_bad.add(i)
                    deliver(i, None)
        for part, status in stream:
            while not loaded[k]:                      # a batch whose units all failed to load produces no device batch
                k += 1
            out = []
            for c, i in enumerate(loaded[k]):
                lo, hi = c * samples_per_complex, (c + 1) * samples_per_complex
                st = [int(x) for x in status[lo:hi]]
                if any(x & 2 for x in st):
                    # status bit 1 (DpWork.status): the ligand graph repeats an ordered atom pair - deterministic, a new noise key cannot help:
                    # fail the unit at once with its own reason instead of re-sampling it `retries` times as a Kabsch failure
                    bad[i] = "ligand graph repeats an ordered atom pair (not representable in the node-side pair tables; status bit 1)"
                    continue
                if any(x & 4 for x in st) and not exact:
                    over_range.add(i)                 # status bit 2: non-finite scores out of (possibly) the two-limb fp16 kernels
                out.append((i, None if any(st) else torch.stack([p["ligand"].pos for p in part[lo:hi]]).numpy()))
            k += 1
            flagged += settle(out, final)
            report_bad()
        report_bad()
        return flagged

    done_bad: set = set()
    # An fp32 model runs its convs on the two-limb fp16 kernels, fp32-grade only inside fp16's exponent range (engine.Engine.limb_kernels): a
    # complex they bring back flagged is first run again with the SAME noise key on the exact-fp32 kernels - poses are independent and their noise
    # is keyed by (seed, pose uid), so that IS the exact-fp32 result of the original run - and only what is flagged in the reference's own
    # arithmetic too goes on to the fresh-key retries (which stay on the exact kernels).  Only the flagged complexes are re-run, not their batches.
    # Which units: the ones whose status carries bit 2 (non-finite SCORES while the pose was fine: DpWork.status) from a run that was not exact already;
    # a degenerate Kabsch alignment (bit 0) fails the same way in any arithmetic and goes straight to the fresh-key retries.
    replay_ok = getattr(getattr(model, "device_model", None), "precision", None) == "f32" and os.environ.get("DIFFPHORE_EXACT_REPLAY", "1") != "0"
    over_range: set = set()
    seed0 = int(seed) & 0xFFFFFFFFFFFFFFFF
    retry = run(batches, seed0, retries <= 0 and not replay_ok)
    per = max(1, max_poses // samples_per_complex)
    cut = lambda units: [units[lo:lo + per] for lo in range(0, len(units), per)]      # noqa: E731
    limb = replay_ok and bool(over_range)
    if retry and limb:
        again = [i for i in retry if i in over_range]
        retry = [i for i in retry if i not in over_range] + run(cut(again), seed0, False, exact=True)
    if retry and retries <= 0 and replay_ok:
        # no fresh-key retries were asked for: what is still flagged after the exact replay is a failure
        for i in retry:
            bad[i] = "Kabsch alignment failed to converge (every attempt)"
            done_bad.add(i)
            deliver(i, None)
        retry = []
    # retries of the Kabsch-flagged complexes with a fresh noise key each (the pose uids stay: only the seed changes)
    for attempt in range(1, retries + 1):
        if not retry:
            break
        retry = run(cut(retry), (int(seed) + 0x9E3779B97F4A7C15 * attempt) & 0xFFFFFFFFFFFFFFFF, attempt == retries, exact=limb)
    collected = None if kept is None else [kept.get(i) for i in mine]
    return mine, collected
