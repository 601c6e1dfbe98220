"""World-size-2 gloo test of the multi-GPU plumbing (sharding, weight-blob broadcast, result gather) on CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffphore_amd import distributed as D
from diffphore_amd import synthetic


def test_shard_indices_partition_and_balance():
    costs = [960, 960, 1580, 300, 2560, 2560, 700, 960, 100, 1200]
    for world in (1, 2, 3, 8):
        shards = D.shard_indices(costs, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs)
    assert D.shard_indices(costs, 2) == D.shard_indices(costs, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        graphs = synthetic.synthetic_pairs(55, 7, (10, 20), (0, 4), (8, 20))
        # weights: rank 0 owns the real blob, the others start from garbage
        blob = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.full((1000,), float("nan"))
        D.broadcast_weights(blob, src=0)
        ok_blob = bool(torch.equal(blob, torch.arange(1000, dtype=torch.float32)))
        idx, mine = D.shard_poses(graphs, rank, world)
        # stand-in for the sampler: every rank "finishes" its poses by adding its pose index
        local = [g["ligand"].pos + float(i) for i, g in zip(idx, mine)]
        allpos = D.gather_positions(idx, local, len(graphs), dst=0)
        if rank == 0:
            ok = all(torch.allclose(allpos[i], graphs[i]["ligand"].pos + float(i)) for i in range(len(graphs)))
            q.put((ok_blob, ok, len(idx)))
        else:
            q.put((ok_blob, allpos is None, len(idx)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_broadcast_shard_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(r[0] and r[1] for r in res)
    assert sum(r[2] for r in res) == 7


def test_screen_library_units_are_pure_functions_of_seed_and_index():
    """synthetic.ScreenLibrary: sizes / cost of a unit without building it, identical units from any rank in any order, and the
    replicated side (the pharmacophore of a virtual screen, the ligand of target fishing) shared by every unit."""
    import torch
    from diffphore_amd import distributed as D
    from diffphore_amd import synthetic
    lib = synthetic.ScreenLibrary(4, 50)
    costs = [lib.cost(i) for i in range(len(lib))]
    for i in (0, 7, 49):
        g = lib.unit(i)
        nl, nt, npn = lib.sizes(i)
        assert g["ligand"].pos.shape[0] == nl and g["phore"].pos.shape[0] == npn == 79 and int(g["ligand"].edge_mask.sum()) <= nt
        assert D.pose_cost(g) == costs[i]
        assert torch.equal(lib.unit(i)["ligand"].pos, g["ligand"].pos)
    assert torch.equal(lib.unit(3)["phore"].pos, lib.unit(30)["phore"].pos) and not torch.equal(lib.unit(3)["ligand"].pos[:5], lib.unit(30)["ligand"].pos[:5])
    assert len({lib.sizes(i)[0] for i in range(50)}) > 10                       # 15..45 atoms
    # every unit is owned by exactly one rank, whatever the world size
    for world in (1, 2, 8):
        sh = D.shard_indices(costs, world)
        assert sorted(i for s_ in sh for i in s_) == list(range(50))
    lib5 = synthetic.ScreenLibrary(5, 10)
    a, b = lib5.unit(1), lib5.unit(8)                                           # the same ligand, placed relative to each phore's centre
    assert torch.allclose(a["ligand"].pos + a["original_center"], b["ligand"].pos + b["original_center"], atol=1e-5)
    assert torch.equal(a["ligand"].x, b["ligand"].x) and a["phore"].pos.shape[0] == lib5.sizes(1)[2] != b["phore"].pos.shape[0]


def test_screen_plan_covers_every_unit_once_and_walks_heaviest_first():
    """distributed.screen_plan: pure host arithmetic every rank does on the cost list alone"""
    from diffphore_amd import distributed as D
    from diffphore_amd import synthetic
    lib = synthetic.ScreenLibrary(4, 203)
    costs = [lib.cost(i) for i in range(len(lib))]
    for world in (1, 2, 8):
        seen, loads = [], []
        for rank in range(world):
            batches = D.screen_plan(len(lib), costs, rank, world, samples_per_complex=40, max_poses=640)
            flat = [i for bt in batches for i in bt]
            assert all(1 <= len(bt) <= 16 for bt in batches) and all(len(bt) == 16 for bt in batches[:-1])      # 640 // 40 complexes
            assert [costs[i] for i in flat] == sorted((costs[i] for i in flat), reverse=True)                  # heaviest first
            seen += flat
            loads.append(sum(costs[i] for i in flat))
        assert sorted(seen) == list(range(len(lib)))
        assert max(loads) - min(loads) <= max(costs)                                                          # LPT balance
    # no cost list: round-robin; one complex larger than a device batch still gets its own batch
    assert D.screen_plan(7, None, 1, 3, 40, 8192) == [[1, 4]]
    assert D.screen_plan(3, [1, 2, 3], 0, 1, samples_per_complex=40, max_poses=8) == [[2], [1], [0]]
    with pytest.raises(ValueError):
        D.screen_plan(3, [1, 2], 0, 1)
    with pytest.raises(ValueError):
        D.screen_plan(2 ** 27, None, 0, 1, samples_per_complex=40)


def _plan_worker(rank, world, port, q):
    import torch.distributed as dist
    from diffphore_amd import distributed as D
    from diffphore_amd import synthetic
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = synthetic.ScreenLibrary(5, 37)
        plan = D.screen_plan(len(lib), [lib.cost(i) for i in range(len(lib))], dist.get_rank(), dist.get_world_size(), 40, 400)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(plan, gathered, dst=0)
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_plan_disjoint_shards_without_exchanging_anything():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    plans = q.get(timeout=100)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    a, b = ([i for bt in pl for i in bt] for pl in plans)
    assert not set(a) & set(b) and sorted(a + b) == list(range(37)) and all(len(bt) <= 10 for pl in plans for bt in pl)


def test_unit_problem_names_what_would_sink_a_device_batch():
    """distributed._unit_problem: the per-unit checks that keep one bad complex from failing the whole device batch of a screen"""
    from diffphore_amd import distributed as D
    g = synthetic.synthetic_pairs(5, 1, 12, 2, 9)[0]
    assert D._unit_problem(g) is None
    assert "malformed" in D._unit_problem({"ligand": None})
    lim = dict(D.LIMITS)
    try:
        D.LIMITS["atoms"] = 11
        assert "kernel limits" in D._unit_problem(g)
        D.LIMITS.update(lim)
        D.LIMITS["torsions"] = 1
        assert "kernel limits" in D._unit_problem(g)
    finally:
        D.LIMITS.update(lim)
