#!/usr/bin/env python3
"""Why is a step slower once torch.distributed is initialised?  Times the two-pipeline GOP loop of bench.py
(no gather) after different amounts of process-group setup.  Usage: python tools/dist_overhead.py MODE
MODE: plain | init_lazy | init_eager | init_eager_barrier | init_eager_late_barrier | init_lazy_late_barrier"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.gop import GopCoder, TorchArena, torch_allocator
from new_bloom_filter_repo_amd.synthetic import make_gop, P_KSTAR_2_3

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
if mode != "plain":
    kw = {"device_id": device} if mode.startswith("init_eager") else {}
    dist.init_process_group("nccl", rank=0, world_size=1, **kw)
    if mode == "init_eager_barrier":
        dist.barrier()
W, H, F = 1920, 1080, 30
n, pairs = W * H, F - 1
streams = [torch.cuda.current_stream(device), torch.cuda.Stream(device)]
ctxs = [nat.Context(0, s.cuda_stream) for s in streams]
arenas = [TorchArena(device, GopCoder.record_bytes(n, pairs)) for _ in range(2)]
coders = []
for k in range(2):
    coders.append(GopCoder(ctxs[k], W, H, F, allocator=torch_allocator(device), out_allocator=arenas[k],
                           frames_block=coders[0].frames if k else None))
coders[0].load_frames(np.stack(make_gop(2000, W, H, F, p=P_KSTAR_2_3)))
torch.cuda.synchronize()
for rep in range(4):
    host = 0.0
    if mode.endswith("late_barrier") and rep in (1, 2):
        dist.barrier()                       # like bench.py: a collective right before the timed loop
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(60):
        with torch.cuda.stream(streams[s % 2]):
            h0 = time.perf_counter()
            coders[s % 2].encode()
            host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-20s rep %d: %.1f us/step, host time inside encode() %.1f us/step, threads %d, cpus %d" % (
        mode, rep, dt / 60 * 1e6, host / 60 * 1e6, len(os.listdir("/proc/self/task")), len(os.sched_getaffinity(0))), flush=True)
if mode != "plain":
    dist.destroy_process_group()
