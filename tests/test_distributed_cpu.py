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
