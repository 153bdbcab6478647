#!/usr/bin/env python3
"""Is a gloo all_gather of CUDA tensors ordered after the kernels that produce its input (ranks sharing ONE GPU)?
torchrun --nproc-per-node 8 tools/experiments/gloo_gather_check.py      (experiment driver, GPU box only)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from misonet_amd.pipeline import gather_outputs   # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bad = 0
big = torch.randn(4096, 4096, device=dev)
for it in range(40):
    # a long-running producer chain on the current stream, then the gather WITHOUT a host synchronisation in between
    z = big
    for _ in range(6):
        z = (z @ big) * 1e-3
    local = torch.full((2, 2, 1001, 129), float(rank * 100 + it), device=dev, dtype=torch.float32) + z[0, 0] * 0
    local = torch.complex(local, -local)
    allout = gather_outputs(local, 2 * world)
    torch.cuda.synchronize()
    for r in range(world):
        want = float(r * 100 + it)
        got = allout[2 * r].real
        if not bool((got == want).all()):
            bad += 1
            print(f"[rank {rank}] it {it}: shard of rank {r} wrong: {int((got != want).sum())} of {got.numel()} elements, e.g. {float(got.flatten()[0])} (want {want})", flush=True)
dist.barrier()
print(f"[rank {rank}] gloo all_gather of CUDA tensors: {bad} wrong shards", flush=True)
