"""Drop-in for the reference sampler entry points (src/utils/sampling.py:16-63, 174-280) on the HIP path.

`sampling_phore` keeps the reference signature.  All N poses are packed ONCE into a device-resident batch; the whole
reverse-diffusion loop then runs on the GPU (score model + conformer update per step, no host synchronisation and no
per-step collate / H2D / D2H as in sampling.py:210-254); results are written back into `data_list` at the end.

Extra, optional keywords:
    noise      pre-drawn normals {'tr': [S,N,3], 'rot': [S,N,3], 'tor': [S,sum T]} (pose-major) so that a CPU oracle and
               this path consume identical noise.  With noise=None the numbers are drawn from the global torch CPU
               generator in exactly the reference's order (per `batch_size` chunk: z_tr, z_rot, z_tor; sampling.py:230-243).
    engine     a prepared `diffphore_amd.engine.Engine` to reuse a packed batch across calls.
    seed       int: draw the noise ON THE DEVICE instead (dp_noise_normals: Philox-4x32-10 keyed by `seed`, counter = (pose uid, step,
               component)); with `randomize=` the start randomisation too (dp_randomize_draws).  Nothing is drawn or uploaded by the
               host, and a pose's trajectory depends only on (seed, its uid), not on batching or sharding.  `pose_uid` (sequence of
               ints, one per entry of data_list; default: the index) carries the global identity of a pose across ranks.  Not the
               reference's random stream: use `noise=` / the global generators for runs that must be compared with a CPU run.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from . import _lib
from .engine import DpStepParams, Engine, PackedBatch, _declare, _stream, d2h, h2d


GRAPH_MAX_POSES = 0        # `auto` never captures: the loop is GPU-bound down to 8 poses, a replay gains <= 0.5 % and the capture of a
                           # run's ~760 launches costs ~2 ms per call (40 poses: 35.4 vs 32.4 ms end to end); DIFFPHORE_GRAPH=1 forces it


def _step_params(t_idx, inference_steps, tr_schedule, rot_schedule, tor_schedule, t_to_sigma, model_args, ode, zero_noise) -> DpStepParams:
    """Scalars of one reverse step (sampling.py:205-244), evaluated in float64 like the reference, stored as float32."""
    last = t_idx == inference_steps - 1
    t_tr, t_rot, t_tor = tr_schedule[t_idx], rot_schedule[t_idx], tor_schedule[t_idx]
    dt_tr = tr_schedule[t_idx] - tr_schedule[t_idx + 1] if not last else tr_schedule[t_idx]
    dt_rot = rot_schedule[t_idx] - rot_schedule[t_idx + 1] if not last else rot_schedule[t_idx]
    dt_tor = tor_schedule[t_idx] - tor_schedule[t_idx + 1] if not last else tor_schedule[t_idx]
    tr_sigma, rot_sigma, tor_sigma = t_to_sigma(t_tr, t_rot, t_tor)
    tr_g = float(tr_sigma) * np.sqrt(2 * np.log(model_args.tr_sigma_max / model_args.tr_sigma_min))
    rot_g = 2 * float(rot_sigma) * np.sqrt(np.log(model_args.rot_sigma_max / model_args.rot_sigma_min))
    tor_g = float(tor_sigma) * np.sqrt(2 * np.log(model_args.tor_sigma_max / model_args.tor_sigma_min))
    sp = DpStepParams()
    half = 0.5 if ode else 1.0
    sp.tr_c1, sp.rot_c1, sp.tor_c1 = half * tr_g ** 2 * dt_tr, half * rot_g ** 2 * dt_rot, half * tor_g ** 2 * dt_tor
    nz = 0.0 if (ode or zero_noise) else 1.0
    sp.tr_c2, sp.rot_c2, sp.tor_c2 = nz * tr_g * np.sqrt(dt_tr), nz * rot_g * np.sqrt(dt_rot), nz * tor_g * np.sqrt(dt_tor)
    sp.no_torsion = 1 if getattr(model_args, "no_torsion", False) else 0
    return sp


LAZY_HEAD = 2      # noisy steps whose normals are drawn before the first launch (see _host_stage_noise_only)


def _draw_rest(Z):
    """Draw the steps `_host_stage_noise_only` left for later into the host tables; returns the first of them (None: nothing was left)."""
    rest = Z.pop("_rest", None) if Z is not None else None
    if rest is None:
        return None
    steps, N, nt, batch_size = rest
    for t in steps:
        dd = _draw_reference_order(N, nt, batch_size)
        for i, k in enumerate(("tr", "rot", "tor")):
            Z[k][t] = dd[i]
    return steps[0]


def _draw_reference_order(N, T_per_pose: np.ndarray, batch_size: int):
    """One step's normals from the global CPU generator in the reference's order (per chunk: tr, rot, tor)."""
    ztr, zrot, ztor = [], [], []
    for lo in range(0, N, batch_size):
        b = min(batch_size, N - lo)
        ztr.append(torch.normal(mean=0, std=1, size=(b, 3)))
        zrot.append(torch.normal(mean=0, std=1, size=(b, 3)))
        ztor.append(torch.normal(mean=0, std=1, size=(int(T_per_pose[lo:lo + b].sum()),)))
    return torch.cat(ztr), torch.cat(zrot), torch.cat(ztor)


def _noisy_steps(inference_steps, ode, no_random, no_final_step_noise):
    return [t for t in range(inference_steps) if not (ode or no_random or (no_final_step_noise and t == inference_steps - 1))]


def _device_noise(seed, noise, pose_uid, lo, n, device):
    """(seed, int32 device tensor of the uids of poses lo .. lo+n) when the run draws its noise on the device, else None."""
    if seed is None or noise is not None:
        return None
    if pose_uid is None:
        uid = torch.arange(lo, lo + n, dtype=torch.int32)
    else:
        uid = torch.as_tensor(list(pose_uid[lo:lo + n]), dtype=torch.int32)
        if uid.numel() != n:
            raise ValueError("pose_uid needs one entry per pose of data_list")
    return int(seed), h2d(uid, device)


def _randomize_on_device(pb, dev_noise, randomize):
    """`randomize_position` with its draws made on the device (dp_randomize_draws) - nothing crosses PCIe."""
    if randomize.get("keep_update", False):
        raise NotImplementedError("keep_update needs the host-drawn start randomisation (seed=None)")
    L = _declare()
    seed, uid = dev_noise
    dev, B, nT = pb.device, pb.B, int(pb.tor_ptr[-1])
    no_torsion, no_random = bool(randomize["no_torsion"]), bool(randomize["no_random"])
    theta = torch.empty(max(nT, 1), dtype=torch.float32, device=dev) if (not no_torsion and nT) else None
    rm = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
    tr = torch.empty(B, 3, dtype=torch.float32, device=dev) if not no_random else None
    _lib.check(L.dp_randomize_draws(seed & 0xFFFFFFFFFFFFFFFF, C.byref(pb.struct), uid.data_ptr(), 0, float(randomize["tr_sigma_max"]),
                                    _lib.ptr(theta), rm.data_ptr(), _lib.ptr(tr), _stream()), "dp_randomize_draws")
    _randomize_launch(pb, theta, rm, tr, no_torsion)


def _host_stage(part, device, inference_steps, tr_schedule, rot_schedule, tor_schedule, ode, no_random, no_final_step_noise, batch_size, noise,
                randomize=None, rz=None, on_dev=False):
    """Everything that needs no device: pack the graphs, draw the start randomisation (if asked and not pre-drawn: `rz`) and then the
    run's normals - the reference's order for one complex: randomize_position(...) first, sampling_phore(...) after - and tabulate the
    times."""
    pb = PackedBatch(part, device, upload=False)
    if randomize is not None and rz is None and not on_dev:
        rz = _randomize_draws(part, pb.nt, randomize["no_torsion"], randomize["no_random"], randomize["tr_sigma_max"],
                              randomize.get("keep_update", False), randomize.get("draws"))
    return _host_stage_noise_only(pb, len(part), inference_steps, tr_schedule, rot_schedule, tor_schedule, ode, no_random,
                                  no_final_step_noise, batch_size, noise, on_dev) + (rz,)


def _enqueue_run(eng, Z, T_all, part, inference_steps, tr_schedule, rot_schedule, tor_schedule, t_to_sigma, model_args, no_random, ode,
                 no_final_step_noise, visualization_list, dev_noise=None):
    """Upload the run's noise / times and enqueue every step; no host synchronisation unless keep_update / visualisation ask for it."""
    pb, dev = eng.pb, eng.pb.device
    no_torsion = bool(getattr(model_args, "no_torsion", False))
    keep_update = bool(getattr(model_args, "keep_update", False))
    eng.ws.t["status"].zero_()     # dp_conformer_update ORs its flag in: a run on a reused engine must not inherit an old failure
    whole = not keep_update and visualization_list is None and os.environ.get("DIFFPHORE_WHOLE_LOOP", "1") != "0"
    g = os.environ.get("DIFFPHORE_GRAPH", "auto")
    use_graph = g == "1" or (g == "auto" and pb.B <= GRAPH_MAX_POSES)
    lazy = Z is not None and "_rest" in Z
    if lazy and not (whole and not use_graph):
        _draw_rest(Z)
        lazy = False
    Zh = Z
    if Z is not None and not lazy:
        Z = {k: h2d(v.to(torch.float32).contiguous(), dev) for k, v in Z.items()}
    T_all = h2d(T_all.contiguous(), dev)
    sps = []
    for t_idx in range(inference_steps):
        last = t_idx == inference_steps - 1
        zero = no_random or (no_final_step_noise and last)
        sps.append(_step_params(t_idx, inference_steps, tr_schedule, rot_schedule, tor_schedule, t_to_sigma, model_args, ode, zero))
    if whole:
        # the whole loop in one library call (dp_sample): times and noise are device tables indexed by the step.  Enqueued eagerly; the
        # hipGraph form (capture once, replay) is for callers that rerun the same engine many times (dp_sample_graph_*)
        seed, uid = dev_noise if dev_noise is not None else (None, None)

        def tables(Zd, lo, hi):
            z_tor = Zd["tor"][lo:hi] if (Zd is not None and not no_torsion and Zd["tor"].numel()) else None
            return (Zd["tr"][lo:hi] if Zd is not None else None, Zd["rot"][lo:hi] if Zd is not None else None, z_tor)
        if lazy:
            # two calls: the steps whose normals exist, then - drawn while the GPU runs those - the others (same schedule, same tables)
            split = Zh["_rest"][0][0]
            head = {k: h2d(Zh[k][:split].contiguous(), dev) for k in ("tr", "rot", "tor")}
            eng.sample(sps[:split], T_all[:split], *tables(head, 0, split), graph=False)
            _draw_rest(Zh)
            tail = {k: h2d(Zh[k][split:].contiguous(), dev) for k in ("tr", "rot", "tor")}
            eng.sample(sps[split:], T_all[split:], *tables(tail, 0, inference_steps - split), graph=False)
            eng._keep = (eng._keep, head, tail)            # both calls' tables stay alive until the run is collected
            return
        eng.sample(sps, T_all, *tables(Z, 0, inference_steps), graph=use_graph, seed=seed, pose_uid=uid)
        return
    for t_idx in range(inference_steps):
        last = t_idx == inference_steps - 1
        zero = no_random or (no_final_step_noise and last)
        sp = sps[t_idx]
        pb.set_time_row(T_all[t_idx])
        z_tr = z_rot = z_tor = None
        if not (ode or zero) and dev_noise is not None:
            seed, uid = dev_noise
            nT = int(pb.tor_ptr[-1])
            z_tr, z_rot = torch.empty(pb.B, 3, device=dev), torch.empty(pb.B, 3, device=dev)
            z_tor = torch.empty(nT, device=dev) if (not no_torsion and nT) else None
            _lib.check(eng.L.dp_noise_normals(seed & 0xFFFFFFFFFFFFFFFF, t_idx, C.byref(pb.struct), uid.data_ptr(), 0, z_tr.data_ptr(),
                                              z_rot.data_ptr(), _lib.ptr(z_tor), _stream()), "dp_noise_normals")
        elif not (ode or zero):
            z_tr, z_rot, z_tor = Z["tr"][t_idx].contiguous(), Z["rot"][t_idx].contiguous(), Z["tor"][t_idx].contiguous()
            z_tor = z_tor if (not no_torsion and z_tor.numel()) else None
        eng.sample_step(sp, z_tr, z_rot, z_tor)
        if keep_update:
            _record_updates(part, eng, sp, z_tr, z_rot, z_tor, no_torsion)
        if visualization_list is not None:
            pos = pb.pos.cpu()
            for idx, vis in enumerate(visualization_list):
                lo, hi = int(pb.lig_ptr[idx]), int(pb.lig_ptr[idx + 1])
                vis.add((pos[lo:hi] + part[idx].original_center).detach().cpu(), part=1, order=t_idx + 2)


def _check_status(eng, synced=False):
    """Per-pose status words of the device loop -> the exceptions the reference's callers expect.  `synced`: the caller has already
    waited for the loop (the download runs on the copy stream and is not ordered behind the launch stream by itself)."""
    if not synced:
        torch.cuda.current_stream(eng.pb.device).synchronize()
    status = d2h(eng.ws.t["status"])
    if bool((status & 2).any()):
        raise ValueError("ligand graph lists the same ordered atom pair twice among its bond (or radius) edges for poses "
                         f"{torch.nonzero(status & 2).flatten().tolist()}: not representable by the node-side ligand-ligand conv "
                         "(set DIFFPHORE_CONV_ALGO=edge)")
    if bool((status & 4).any()):
        # non-finite SCORES (DpWork.status bit 2).  The reference would hand them to its Kabsch SVD, which raises the text its callers string-match
        # (inference.py:211); on the two-limb fp16 kernels the cause may be their exponent range, which the reference does not have: say how to leave it
        poses = torch.nonzero(status & 4).flatten().tolist()
        hint = (" - this engine runs the two-limb fp16 kernels (fp32-grade inside fp16's range only): run the batch on Engine(exact=True) / "
                "DIFFPHORE_NODE_F16=0 DIFFPHORE_EDGE_X6=0 (sampling_phore does that by itself unless engine= / write_back=False / keep_update is used)"
                if eng.limb_kernels() else "")
        raise RuntimeError(f"linalg.svd: the score model returned non-finite scores for poses {poses}, the alignment failed to converge{hint}")
    if bool((status & 1).any()):
        # the reference raises from torch.linalg.svd and the caller string-matches this text (inference.py:211)
        raise RuntimeError(f"linalg.svd: Kabsch alignment failed to converge for poses {torch.nonzero(status & 1).flatten().tolist()}")


def _finish_run(eng, part, done=None, raise_on_status=True):
    """Wait for the batch's loop (`done`: an event recorded behind it; None = the whole launch stream) and write the final state back.
    The downloads run on the copy stream: on the launch stream they would queue behind the NEXT batch's loop.  Returns the per-pose status
    words (0 = ok, bit 0 = Kabsch / SVD failure, bit 1 = unrepresentable ligand multigraph, bit 2 = non-finite scores); with `raise_on_status` a non-zero one raises
    the exception the reference's callers expect for the WHOLE batch, without it the caller sorts the poses out (library screens)."""
    pb = eng.pb
    if done is not None:
        done.synchronize()
    else:
        torch.cuda.current_stream(pb.device).synchronize()
    if raise_on_status:
        _check_status(eng, synced=True)
    status = d2h(eng.ws.t["status"]).tolist()
    pos, norm = d2h(pb.pos), d2h(pb.norm)
    for i, g in enumerate(part):
        lo, hi = int(pb.lig_ptr[i]), int(pb.lig_ptr[i + 1])
        g["ligand"].pos = pos[lo:hi]                   # (disjoint row ranges of the batch's host copy: no per-pose clone)
        g["ligand"].norm = norm[lo:hi]
    return status


def sampling_phore(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args,
                   no_random=False, ode=False, visualization_list=None, confidence_model=None, confidence_data_list=None,
                   confidence_model_args=None, batch_size=20, no_final_step_noise=False, noise=None, engine: Optional[Engine] = None,
                   write_back=True, max_poses: int = 8192, timing: Optional[dict] = None, randomize: Optional[dict] = None,
                   seed: Optional[int] = None, pose_uid=None):
    """`randomize={'no_torsion','no_random','tr_sigma_max'[,'keep_update','draws']}` fuses `randomize_position` into the call: every
    device batch is packed once, randomised on the device and sampled, without the pack / D2H / re-pack round trip of two calls."""
    if confidence_model is not None:
        raise NotImplementedError("confidence models are outside the accelerated path (SURVEY.md 8b)")
    N = len(data_list)
    common = (inference_steps, tr_schedule, rot_schedule, tor_schedule)
    if engine is not None:   # caller-managed engine: refresh the diffusion state from data_list, one device batch
        engine.pb.set_state(torch.cat([g["ligand"].pos for g in data_list]), torch.cat([g["ligand"].norm.reshape(-1, 33) for g in data_list]))
        if randomize is not None:
            raise ValueError("randomize= needs a batch packed by this call (engine=None)")
        dev_noise = _device_noise(seed, noise, pose_uid, 0, N, engine.pb.device)
        _, Z, T_all = _host_stage_noise_only(engine.pb, N, *common, ode, no_random, no_final_step_noise, batch_size, noise, dev_noise is not None)
        _enqueue_run(engine, Z, T_all, data_list, *common, t_to_sigma, model_args, no_random, ode, no_final_step_noise, visualization_list,
                     dev_noise)
        if write_back:
            _finish_run(engine, data_list)
        return data_list, None
    # Device-batch the pose list (`max_poses` per batch).  Poses are independent, so results do not depend on the batching.  With more
    # than one batch and noise=None the global generators are still consumed in the reference's order for the WHOLE list - start
    # randomisation of every pose first, then step-major normals per `batch_size` chunk - by drawing everything before the first batch
    # (S x N x (6 + T) floats on the host); the batches then take their slices.  The host stage of batch k+1 (packing, uploads) runs
    # while the GPU works on batch k: the step loop is enqueued without any host synchronisation.
    rz_all = None
    on_dev = seed is not None and noise is None
    if N > max_poses and noise is None and not on_dev:
        nt_all = np.asarray([int(torch.as_tensor(g["ligand"].edge_mask).sum()) for g in data_list], dtype=np.int64)
        if randomize is not None:
            rz_all = _randomize_draws(data_list, nt_all, randomize["no_torsion"], randomize["no_random"], randomize["tr_sigma_max"],
                                      randomize.get("keep_update", False), randomize.get("draws"))
        steps = _noisy_steps(inference_steps, ode, no_random, no_final_step_noise)
        if steps:
            d = [_draw_reference_order(N, nt_all, batch_size) for _ in steps]
            noise = {k: torch.zeros((inference_steps,) + tuple(d[0][i].shape)) for i, k in enumerate(("tr", "rot", "tor"))}
            for t, dd in zip(steps, d):
                for i, k in enumerate(("tr", "rot", "tor")):
                    noise[k][t] = dd[i]
    parts, subs, rzs, t_off = [], [], [], 0
    for lo in range(0, N, max_poses):
        part = data_list[lo:lo + max_poses]
        need = N > max_poses and (noise is not None or rz_all is not None)
        nt_part = sum(int(torch.as_tensor(g["ligand"].edge_mask).sum()) for g in part) if need else 0
        subs.append(noise if (noise is None or N <= max_poses) else
                    {"tr": noise["tr"][:, lo:lo + max_poses], "rot": noise["rot"][:, lo:lo + max_poses], "tor": noise["tor"][:, t_off:t_off + nt_part]})
        if rz_all is not None:
            th, rm, tr = rz_all
            rzs.append((th[t_off:t_off + nt_part].contiguous() if (th is not None and nt_part) else None, rm[lo:lo + max_poses].contiguous(),
                        tr[lo:lo + max_poses].contiguous() if tr is not None else None))
        else:
            rzs.append(None)
        parts.append(part)
        t_off += nt_part
    def gen():
        for k, part in enumerate(parts):
            lo = k * max_poses
            yield part, subs[k], rzs[k], (list(pose_uid[lo:lo + len(part)]) if pose_uid is not None else range(lo, lo + len(part)))

    for _ in _run_parts(gen(), model, device, common, t_to_sigma, model_args, no_random, ode, no_final_step_noise, batch_size,
                        visualization_list, randomize, seed if on_dev else None, write_back, timing):
        pass
    return data_list, None


def _run_parts(parts_iter, model, device, common, t_to_sigma, model_args, no_random, ode, no_final_step_noise, batch_size,
               visualization_list, randomize, seed, write_back, timing=None, isolate=False, exact=False, replay=True):
    """The device-batch pipeline behind `sampling_phore` and `sampling_phore_stream`.  `parts_iter` yields
    (pose list, noise tables | None, pre-drawn start randomisation | None, pose uids); the NEXT item is pulled - whatever lazy loading
    the iterator does included - and host-staged while the GPU runs the current batch (the step loop is enqueued without any host
    synchronisation), so at most two batches of host objects are alive.  Yields each pose list once its results are written back; with
    `isolate` it yields (pose list, per-pose status words) and a flagged pose does not raise (the caller drops or retries its complex)."""
    import time as _time
    on_dev = seed is not None
    _t0 = _time.time()
    it = iter(parts_iter)

    def pull():
        item = next(it, None)
        if item is None:
            return None
        part, sub, rz, uids = item
        return (part, uids) + _host_stage(part, device, *common, ode, no_random, no_final_step_noise, batch_size, sub, randomize, rz, on_dev)

    n_seen = 0

    def launch(item, first):
        """upload, (randomise,) and enqueue the whole loop of one device batch; nothing here waits for work already queued on the launch
        stream, so the GPU runs the previous batch meanwhile"""
        nonlocal n_seen, _t0
        part, uids, pb, Z, T_all, rz = item
        pb.upload()
        dev_noise = _device_noise(seed, None, uids, 0, len(part), pb.device) if on_dev else None
        if rz is not None:
            _randomize_launch(pb, *rz, randomize["no_torsion"])
        elif randomize is not None and on_dev:
            _randomize_on_device(pb, dev_noise, randomize)
        eng = Engine(model.device_model, pb, exact=exact)
        if timing is not None and first:
            torch.cuda.synchronize()
            timing["pack_and_noise_s"] = _time.time() - _t0
            _t0 = _time.time()
        vis = visualization_list[n_seen:n_seen + len(part)] if visualization_list is not None else None      # this device batch's recorders
        n_seen += len(part)
        _enqueue_run(eng, Z, T_all, part, *common, t_to_sigma, model_args, no_random, ode, no_final_step_noise, vis, dev_noise)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(pb.device))
        return part, eng, done, item

    replayable = replay and visualization_list is None and not bool(getattr(model_args, "keep_update", False)) and \
        not (randomize or {}).get("keep_update", False) and os.environ.get("DIFFPHORE_EXACT_REPLAY", "1") != "0"

    def replay_exact(item, eng):
        """A batch of an fp32 model that ran the two-limb fp16 kernels came back with status bit 2 (the scores of some pose went non-finite while
        the pose was fine; a degenerate Kabsch alignment - bit 0 - is not replayed: it fails the same way in any arithmetic).  Those kernels are fp32-grade only inside fp16's exponent range (Engine.limb_kernels), the reference has no such
        limit: run the batch once more on the exact-fp32 kernels - same start state (the host graphs are untouched until write-back), same
        start randomisation, same noise - before anything is reported.  What is still flagged then is flagged in the reference's own
        arithmetic."""
        part, uids, pb, Z, T_all, rz = item
        pb.set_state(torch.cat([g["ligand"].pos for g in part]), torch.cat([g["ligand"].norm.reshape(-1, 33) for g in part]))
        dev_noise = _device_noise(seed, None, uids, 0, len(part), pb.device) if on_dev else None
        if rz is not None:
            _randomize_launch(pb, *rz, randomize["no_torsion"])
        elif randomize is not None and on_dev:
            _randomize_on_device(pb, dev_noise, randomize)
        eng2 = Engine(model.device_model, pb, conv_algo=eng.conv_algo, exact=True)
        _enqueue_run(eng2, Z, T_all, part, *common, t_to_sigma, model_args, no_random, ode, no_final_step_noise, None, dev_noise)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(pb.device))
        if timing is not None:
            timing["exact_replays"] = timing.get("exact_replays", 0) + 1
        return eng2, done

    item = pull()
    cur = launch(item, True) if item is not None else None
    while cur is not None:
        part, eng, done, cur_item = cur
        item = pull()                                # host stage of the next batch (the caller's lazy loading included) ...
        nxt = launch(item, False) if item is not None else None      # ... and its loop, queued behind the running one
        if write_back and replayable and eng.limb_kernels():
            done.synchronize()
            if bool((d2h(eng.ws.t["status"]) & 4).any()):
                eng, done = replay_exact(cur_item, eng)
        if timing is not None and nxt is None:
            done.synchronize()
            timing["loop_s"] = _time.time() - _t0
            _t0 = _time.time()
        status = None
        if write_back:
            status = _finish_run(eng, part, done, raise_on_status=not isolate)
        if timing is not None and nxt is None:
            timing["write_back_s"] = _time.time() - _t0
        yield (part, status) if isolate else part
        cur = nxt


def sampling_phore_stream(parts, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args, *, seed: int,
                          no_random=False, ode=False, no_final_step_noise=False, randomize: Optional[dict] = None, write_back=True,
                          isolate=False, exact=False, replay=True):
    """`sampling_phore` over a STREAM of device batches for library-scale runs (virtual screening shards 10^5 ligands x 40
    samples per GPU; the reference's loop is `for graph in dataset: [deepcopy(graph) x N] -> randomize -> sample`, inference.py:139-228).
    `parts` is an iterable (typically a generator that builds its poses lazily) of `(pose_list, pose_uids)`; each pose list is one device
    batch.  Noise - and with `randomize=` the start randomisation - is drawn on the device from (`seed`, pose uid), so nothing depends on
    how the stream is cut into batches or ranks.  A generator: yields each pose list after its final positions are written back
    (`g['ligand'].pos`); the caller consumes and drops it, the pipeline keeps at most two batches of host objects alive.  `isolate`: yield
    (pose list, per-pose status words) and never raise for a flagged pose (Kabsch failure): one bad ligand must not end a library run.
    `exact`: every batch on the exact-fp32 kernels (Engine(exact=True)); `replay=False`: no whole-batch exact replay of a flagged batch - for
    callers that re-run the flagged COMPLEXES themselves (distributed.screen)."""
    if seed is None:
        raise ValueError("the streaming sampler draws its noise on the device: seed is required")
    common = (inference_steps, tr_schedule, rot_schedule, tor_schedule)
    yield from _run_parts(((p, None, None, u) for p, u in parts), model, device, common, t_to_sigma, model_args, no_random, ode,
                          no_final_step_noise, 20, None, randomize, int(seed), write_back, isolate=isolate, exact=exact, replay=replay)


def _host_stage_noise_only(pb, N, inference_steps, tr_schedule, rot_schedule, tor_schedule, ode, no_random, no_final_step_noise, batch_size, noise,
                           on_dev=False):
    """`_host_stage` for an already packed batch (`on_dev`: the noise is drawn on the device, nothing to prepare here)."""
    steps = _noisy_steps(inference_steps, ode, no_random, no_final_step_noise)
    Z = None
    if steps and not on_dev:
        if noise is not None:
            Z = {k: noise[k].to(torch.float32) for k in ("tr", "rot", "tor")}
        else:
            # step-major, then per `batch_size` chunk: tr, rot, tor - nothing else consumes the generator inside the reference's loop.
            # Only the first LAZY_HEAD steps are drawn here; `_enqueue_run` draws the others - same generator, same order, nothing in
            # between - AFTER it has queued those first steps, so the ~1000 small torch.normal calls of a 20-step run (3-4 ms for 512
            # poses) run while the GPU is already sampling instead of in front of it.
            head = steps[:LAZY_HEAD] if len(steps) > LAZY_HEAD + 1 else steps
            d = [_draw_reference_order(N, pb.nt, batch_size) for _ in head]
            Z = {k: torch.zeros((inference_steps,) + tuple(d[0][i].shape)) for i, k in enumerate(("tr", "rot", "tor"))}
            for t, dd in zip(head, d):
                for i, k in enumerate(("tr", "rot", "tor")):
                    Z[k][t] = dd[i]
            if len(head) < len(steps):
                Z["_rest"] = (steps[len(head):], N, pb.nt, batch_size)
    T_all = torch.stack([torch.tensor([float(tr_schedule[t]), float(rot_schedule[t]), float(tor_schedule[t])], dtype=torch.float32)
                         for t in range(inference_steps)])
    return pb, Z, T_all


def _record_updates(data_list, eng, sp, z_tr, z_rot, z_tor, no_torsion):
    """model_args.keep_update bookkeeping of modify_conformer (diffusion_utils.py:71-77): per-step perturbations and poses."""
    ws, pb = eng.ws, eng.pb
    tr = sp.tr_c1 * ws.tr + (sp.tr_c2 * z_tr if z_tr is not None else 0)
    rot = sp.rot_c1 * ws.rot + (sp.rot_c2 * z_rot if z_rot is not None else 0)
    tor = None
    if not no_torsion and ws.n_tor:
        tor = sp.tor_c1 * ws.tor[:ws.n_tor] + (sp.tor_c2 * z_tor[:ws.n_tor] if z_tor is not None else 0)
        tor = tor.cpu().numpy()
    tr, rot, pos = tr.cpu().numpy(), rot.cpu().numpy(), pb.pos.cpu().numpy()
    for i, g in enumerate(data_list):
        lo, hi = int(pb.lig_ptr[i]), int(pb.lig_ptr[i + 1])
        t0, t1 = int(pb.tor_ptr[i]), int(pb.tor_ptr[i + 1])
        for name, val in (("rvs_rot_update", rot[i]), ("rvs_tr_update", tr[i:i + 1]),
                          ("rvs_tor_update", tor[t0:t1] if tor is not None else None), ("docked_poses", pos[lo:hi].copy())):
            prev = getattr(g, name) if name in g else []
            setattr(g, name, prev + [val])


def _randomize_draws(data_list, nt, no_torsion, no_random, tr_sigma_max, keep_update=False, draws=None):
    """Host half of `randomize_position`: the reference's global-generator calls in its order (np.random.uniform per graph, then
    per graph scipy `R.random()` and `torch.normal`, sampling.py:32-57).  Returns CPU tensors (theta | None, rot [N,3,3], tr | None)."""
    thetas, rots, trs = [], [], []
    if not no_torsion:
        for i, g in enumerate(data_list):
            T = int(nt[i])
            th = draws[i]["tor"] if draws is not None else np.random.uniform(low=-np.pi, high=np.pi, size=T)
            thetas.append(np.asarray(th, dtype=np.float64).reshape(-1))
            if keep_update:
                g.fw_tor_update = np.asarray(th)
    for i, g in enumerate(data_list):
        rot = draws[i]["rot"] if draws is not None else R.random().as_matrix()
        rots.append(np.asarray(rot, dtype=np.float32))
        if not no_random:
            tr = draws[i]["tr"] if draws is not None else torch.normal(mean=0, std=tr_sigma_max, size=(1, 3))
            trs.append(torch.as_tensor(tr).float().reshape(1, 3))
            if keep_update:
                g.fw_tr_update = trs[-1].numpy()
        if keep_update:
            g.fw_rot_update = rots[-1]
    theta = torch.from_numpy(np.concatenate(thetas).astype(np.float32)) if thetas and sum(t.size for t in thetas) else None
    rm = torch.from_numpy(np.ascontiguousarray(np.stack(rots)))
    tr = torch.cat(trs).contiguous() if trs else None
    return theta, rm, tr


def _randomize_launch(pb, theta, rm, tr, no_torsion):
    """Device half: `dp_randomize_position` on an uploaded batch (state stays on the device)."""
    L = _declare()
    L.dp_randomize_position.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_void_p]
    L.dp_randomize_position.restype = C.c_int
    dev = pb.device
    theta, rm, tr = h2d(theta, dev), h2d(rm, dev), h2d(tr, dev)
    _lib.check(L.dp_randomize_position(C.byref(pb.struct), _lib.ptr(theta), rm.data_ptr(), _lib.ptr(tr), 1 if no_torsion else 0,
                                       _stream()), "dp_randomize_position")


def randomize_position(data_list, no_torsion, no_random, tr_sigma_max, keep_update=False, device="cuda", draws=None):
    """sampling.py:16-63.  Random numbers come from the same global generators, in the same order, as the reference
    (np.random.uniform per graph, then per graph scipy `R.random()` and `torch.normal`); the geometry runs in
    `dp_randomize_position`.  `draws` (list of dicts 'tor','rot','tr') overrides the RNG for reproducible tests."""
    pb = PackedBatch(data_list, device)
    theta, rm, tr = _randomize_draws(data_list, pb.nt, no_torsion, no_random, tr_sigma_max, keep_update, draws)
    _randomize_launch(pb, theta, rm, tr, no_torsion)
    pos, norm = pb.pos.cpu(), pb.norm.cpu()
    for i, g in enumerate(data_list):
        lo, hi = int(pb.lig_ptr[i]), int(pb.lig_ptr[i + 1])
        g["ligand"].pos = pos[lo:hi].clone()
        g["ligand"].norm = norm[lo:hi].clone()


# ---------------------------------------------------------------------------------------------------------------------------
# the single-step API and the best-of-R branching sampler
# ---------------------------------------------------------------------------------------------------------------------------
def _g_params(model_args, tr_sigma, rot_sigma, tor_sigma):
    tr_g = float(tr_sigma) * np.sqrt(2 * np.log(model_args.tr_sigma_max / model_args.tr_sigma_min))
    rot_g = 2 * float(rot_sigma) * np.sqrt(np.log(model_args.rot_sigma_max / model_args.rot_sigma_min))
    tor_g = float(tor_sigma) * np.sqrt(2 * np.log(model_args.tor_sigma_max / model_args.tor_sigma_min))
    return tr_g, rot_g, tor_g


def _perturbations(eng, sp, z_tr, z_rot, z_tor, no_torsion):
    """What the step applied: c1*score + c2*z, as CPU tensors / numpy in the reference's return types."""
    ws = eng.ws
    tr = (sp.tr_c1 * ws.tr + (sp.tr_c2 * z_tr if z_tr is not None else 0)).cpu()
    rot = (sp.rot_c1 * ws.rot + (sp.rot_c2 * z_rot if z_rot is not None else 0)).cpu()
    tor = None
    if not no_torsion:
        tor = (sp.tor_c1 * ws.tor[:ws.n_tor] + (sp.tor_c2 * z_tor[:ws.n_tor] if z_tor is not None else 0)).cpu().numpy()
    return tr, rot, tor


def sample_step(complex_graph_batch, model, model_args, tr_sigma, rot_sigma, tor_sigma, delta_t=0.05, no_random=False, ode=False,
                noise=None):
    """`src/utils/sampling.py:501-559`: ONE reverse step on a collated batch whose `complex_t` is already set (the calibrated
    training sampler's entry point).  Returns `(data_list, tor_perturb, tr_perturb, rot_perturb)` like the reference;
    `noise = (z_tr [b,3], z_rot [b,3], z_tor [sum T])` overrides the global-generator draws (tr, rot, tor order, :530-546)."""
    eng = model.engine_for(complex_graph_batch)
    pb = eng.pb
    pb.set_state(torch.as_tensor(complex_graph_batch["ligand"].pos), torch.as_tensor(complex_graph_batch["ligand"].norm))
    ct = complex_graph_batch.complex_t
    pb.set_time(ct["tr"], ct["rot"], ct["tor"])
    no_torsion = bool(getattr(model_args, "no_torsion", False))
    tr_g, rot_g, tor_g = _g_params(model_args, tr_sigma, rot_sigma, tor_sigma)
    sp = DpStepParams()
    half = 0.5 if ode else 1.0
    sp.tr_c1, sp.rot_c1, sp.tor_c1 = half * tr_g ** 2 * delta_t, half * rot_g ** 2 * delta_t, half * tor_g ** 2 * delta_t
    nz = 0.0 if (ode or no_random) else 1.0
    sp.tr_c2, sp.rot_c2, sp.tor_c2 = nz * tr_g * np.sqrt(delta_t), nz * rot_g * np.sqrt(delta_t), nz * tor_g * np.sqrt(delta_t)
    sp.no_torsion = 1 if no_torsion else 0
    b, nT = pb.B, int(pb.tor_ptr[-1])
    z_tr = z_rot = z_tor = None
    if nz:
        if noise is not None:
            z_tr, z_rot, z_tor = noise
        else:
            z_tr, z_rot = torch.normal(mean=0, std=1, size=(b, 3)), torch.normal(mean=0, std=1, size=(b, 3))
            z_tor = torch.normal(mean=0, std=1, size=(nT,)) if not no_torsion else None
        z_tr, z_rot = z_tr.to(pb.device, torch.float32).contiguous(), z_rot.to(pb.device, torch.float32).contiguous()
        z_tor = z_tor.to(pb.device, torch.float32).contiguous() if (z_tor is not None and not no_torsion and nT) else None
    eng.sample_step(sp, z_tr, z_rot, z_tor)
    _check_status(eng)
    tr_p, rot_p, tor_p = _perturbations(eng, sp, z_tr, z_rot, z_tor, no_torsion)
    data_list = complex_graph_batch.to_data_list()
    pos, norm = pb.pos.cpu(), pb.norm.cpu()
    for i, g in enumerate(data_list):
        lo, hi = int(pb.lig_ptr[i]), int(pb.lig_ptr[i + 1])
        g["ligand"].pos = pos[lo:hi].clone()
        g["ligand"].norm = norm[lo:hi].clone()
    if getattr(model_args, "keep_update", False):
        _record_updates(data_list, eng, sp, z_tr, z_rot, z_tor, no_torsion)
    return data_list, tor_p, tr_p, rot_p


def sampling_phore_with_fitscore(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args,
                                 no_random=False, ode=False, visualization_list=None, confidence_model=None, confidence_data_list=None,
                                 confidence_model_args=None, batch_size=32, no_final_step_noise=False, fitscore_fn=None, noise=None):
    """`src/utils/sampling.py:283-444`: reverse diffusion where every step draws `model_args.random_samples` noise candidates
    per pose and keeps the one with the best pharmacophore fit score.

    The score model runs ONCE per step; the R candidate conformer updates are R launches of `dp_conformer_update` from the same
    saved state (the reference deep-copies the graph R times and runs `modify_conformer` in Python for each).
    `fitscore_fn(dock_pose [b*R, n_heavy, 3] (pose-major, candidate-minor), data_list) -> b*R scores` replaces the AncPhore
    call (`calculate_fitscore`, :447-498, closed binary); it is required when `random_samples > 1`.
    Deliberate deviations: (1) candidate selection is per pose (the reference's `view(random_samples, -1)` indexing, :413,
    is only consistent for batches of one pose); (2) `ode=True` with torsions dies with NameError in the reference (:372-378
    use `tor_z` that the ODE branch never defines) and raises NotImplementedError here; (3) the ODE translation update of the
    reference is `0.5*g*sqrt(dt)*score` (:344, not squared) - reproduced.
    With `random_samples <= 1`, `no_random` or on the noise-free last step this is `sampling_phore`'s update.
    `fitscore_fn="overlap"` selects the built-in on-device surrogate (`diffphore_amd.fitness`, not AncPhore).
    `noise` = {'tr': [S,R,N,3], 'rot': [S,R,N,3], 'tor': [S,R,sum T]} overrides the generator (R = 1 when not branching)."""
    if confidence_model is not None:
        raise NotImplementedError("confidence models are outside the accelerated path (SURVEY.md 8b)")
    N = len(data_list)
    R = int(getattr(model_args, "random_samples", 0) or 0)
    no_torsion = bool(getattr(model_args, "no_torsion", False))
    if ode and not no_torsion:
        raise NotImplementedError("sampling_phore_with_fitscore(ode=True) with torsions is undefined in the reference (NameError at sampling.py:378)")
    if R > 1 and not no_random and fitscore_fn is None:
        raise ValueError("random_samples > 1 needs fitscore_fn (the reference calls the AncPhore binary here)")
    if N > 65536:
        # one device batch by construction (the candidates of a step are scored together); the reference calls this with one complex's
        # samples (tens of poses)
        raise ValueError("sampling_phore_with_fitscore packs all poses into one device batch: split the list (<= 65536 poses per call)")
    eng = Engine(model.device_model, PackedBatch(data_list, device))
    eng.ws.t["status"].zero_()
    pb = eng.pb
    dev = pb.device
    nT = int(pb.tor_ptr[-1])
    heavy = torch.cat([torch.not_equal(torch.as_tensor(g["ligand"].x)[:, 0], 0) for g in data_list]).numpy()
    centers = np.concatenate([np.repeat(torch.as_tensor(g.original_center).reshape(1, 3).numpy() if "original_center" in g
                                        else np.zeros((1, 3)), int(pb.nl[i]), 0) for i, g in enumerate(data_list)])
    for t_idx in range(inference_steps):
        last = no_final_step_noise and t_idx == inference_steps - 1
        branching = (not no_random) and R > 1 and not last
        sp = _step_params(t_idx, inference_steps, tr_schedule, rot_schedule, tor_schedule, t_to_sigma, model_args, ode, no_random or last)
        if ode:   # :344 - the reference's ODE translation coefficient is not squared
            dt_tr = tr_schedule[t_idx] - tr_schedule[t_idx + 1] if t_idx < inference_steps - 1 else tr_schedule[t_idx]
            tr_g, _, _ = _g_params(model_args, *t_to_sigma(tr_schedule[t_idx], rot_schedule[t_idx], tor_schedule[t_idx]))
            sp.tr_c1 = 0.5 * tr_g * np.sqrt(dt_tr)
        pb.set_time(float(tr_schedule[t_idx]), float(rot_schedule[t_idx]), float(tor_schedule[t_idx]))
        eng = eng.forward_in_range()[0]            # (scores beyond the two-limb fp16 kernels' range: this step and the following ones on exact fp32)
        reps = R if branching else 1
        noisy = not (ode or no_random or last)
        cand_pos, cand_norm = [], []
        pos0, norm0 = pb.pos.clone(), pb.norm.clone()
        drawn = None
        if noisy and noise is None:
            # the reference's draw order (:352-381): per `batch_size` chunk  tr_z [R,b,3], rot_z [R,b,3], tor_z [R, sum T of the chunk]
            ztr, zrot, ztor = [], [], []
            for lo in range(0, N, batch_size):
                b_ = min(batch_size, N - lo)
                shp = (reps, b_, 3) if reps > 1 else (b_, 3)
                ztr.append(torch.normal(mean=0, std=1, size=shp).reshape(reps, b_, 3))
                zrot.append(torch.normal(mean=0, std=1, size=shp).reshape(reps, b_, 3))
                tc = int(pb.nt[lo:lo + b_].sum())
                if not no_torsion:
                    ztor.append(torch.normal(mean=0, std=1, size=(reps, tc) if reps > 1 else (tc,)).reshape(reps, tc))
            drawn = (torch.cat(ztr, 1), torch.cat(zrot, 1), torch.cat(ztor, 1) if ztor else torch.zeros(reps, 0))
        for j in range(reps):
            z_tr = z_rot = z_tor = None
            if noisy:
                if noise is not None:
                    z_tr, z_rot, z_tor = noise["tr"][t_idx][j], noise["rot"][t_idx][j], noise["tor"][t_idx][j]
                else:
                    z_tr, z_rot, z_tor = drawn[0][j], drawn[1][j], drawn[2][j]
                z_tr, z_rot = z_tr.to(dev, torch.float32).contiguous(), z_rot.to(dev, torch.float32).contiguous()
                z_tor = z_tor.to(dev, torch.float32).contiguous() if (not no_torsion and nT) else None
            if j:
                pb.pos.copy_(pos0)
                pb.norm.copy_(norm0)
            eng.update(sp, z_tr, z_rot, z_tor)
            if branching:
                cand_pos.append(pb.pos.clone())
                cand_norm.append(pb.norm.clone())
        if branching and isinstance(fitscore_fn, str):
            if fitscore_fn != "overlap":
                raise ValueError(f"unknown built-in fit score {fitscore_fn!r}")
            from .fitness import overlap_scores                            # on-device surrogate: no host round trip in the loop
            P = torch.stack(cand_pos, 0)
            ex_l = float(getattr(model_args, "ex_lambda", 1.0))
            best = torch.stack([overlap_scores(pb, P[j], ex_l) for j in range(R)], 1).argmax(dim=1)
            sel = torch.repeat_interleave(best, torch.as_tensor(pb.nl, dtype=torch.long, device=dev))
            idx = torch.arange(sel.numel(), device=dev)
            pb.pos.copy_(P[sel, idx])
            pb.norm.copy_(torch.stack(cand_norm, 0)[sel, idx])
        elif branching:
            P = torch.stack(cand_pos, 0)                                   # [R, sum N_l, 3]
            dock = []
            Pn = P.cpu().numpy()
            for i in range(N):
                lo, hi = int(pb.lig_ptr[i]), int(pb.lig_ptr[i + 1])
                for j in range(R):
                    dock.append((Pn[j, lo:hi] + centers[lo:hi])[heavy[lo:hi]])
            scores = torch.as_tensor(np.asarray(fitscore_fn(dock, data_list), dtype=np.float64)).view(N, R)
            best = scores.argmax(dim=1)
            sel = torch.repeat_interleave(best, torch.as_tensor(pb.nl, dtype=torch.long)).to(dev)
            idx = torch.arange(sel.numel(), device=dev)
            pb.pos.copy_(P[sel, idx])
            pb.norm.copy_(torch.stack(cand_norm, 0)[sel, idx])
        if visualization_list is not None:
            pos = pb.pos.cpu()
            for idx_, vis in enumerate(visualization_list):
                lo, hi = int(pb.lig_ptr[idx_]), int(pb.lig_ptr[idx_ + 1])
                vis.add((pos[lo:hi] + data_list[idx_].original_center).detach().cpu(), part=1, order=t_idx + 2)
    _check_status(eng)
    pos, norm = pb.pos.cpu(), pb.norm.cpu()
    for i, g in enumerate(data_list):
        lo, hi = int(pb.lig_ptr[i]), int(pb.lig_ptr[i + 1])
        g["ligand"].pos = pos[lo:hi].clone()
        g["ligand"].norm = norm[lo:hi].clone()
    return data_list, None
