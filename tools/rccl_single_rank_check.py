"""RCCL smoke on ONE GPU: the collectives bench.py / distributed.py issue on the nccl (= RCCL) backend - broadcast of a weight-blob-sized tensor,
all_reduce(MAX), all_gather, barrier - in a one-rank group (the GPU boxes of the build pool have one device; the gloo tests cover the multi-rank
logic, this covers "the RCCL library loads and its kernels run on this image").  python tools/rccl_single_rank_check.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
blob = torch.randn(6_000_000, device=dev)
ref = blob.clone()
torch.cuda.synchronize()
t = time.perf_counter()
dist.broadcast(blob, src=0)
torch.cuda.synchronize()
print("broadcast 24 MB ms", round((time.perf_counter() - t) * 1e3, 2), "equal", bool(torch.equal(blob, ref)))
te = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(te, op=dist.ReduceOp.MAX)
out = [torch.zeros(2, device=dev, dtype=torch.float64)]
dist.all_gather(out, torch.tensor([0.25, 512.0], device=dev, dtype=torch.float64))
dist.barrier()
torch.cuda.synchronize()
print("all_reduce", te.item(), "all_gather", out[0].tolist(), "backend", dist.get_backend(), "nccl version", torch.cuda.nccl.version())
dist.destroy_process_group()
print("ok")
