#!/usr/bin/env python3
"""Does it matter WHERE a context's scratch and a coder's buffers land in HBM?  Re-creates the context + coder of one 1080p GOP a dozen times
with allocations of random sizes in between (kept alive, so every round gets other addresses) and prints every kernel's time alone, plus the
device addresses of the coder's buffers.  Usage (GPU box): python tools/placement_probe.py [rounds=12] [bits=8]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.gop import GopCoder, torch_allocator
from new_bloom_filter_repo_amd.synthetic import make_gop

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, F = 1920, 1080, 30
device = torch.device("cuda", 0)
frames = np.stack(make_gop(4242, W, H, F, dtype=np.uint8 if bits == 8 else np.uint16))
rng = np.random.default_rng(7)
ballast = []
keep = nat.Context(0)                               # holds the hash table's reference: the table itself stays where it is
for r in range(rounds):
    stream = torch.cuda.Stream(device)
    ctx = nat.Context(0, stream.cuda_stream)
    coder = GopCoder(ctx, W, H, F, sample_bytes=bits // 8, allocator=torch_allocator(device), planar_luma=True, keep_interleaved=False)
    coder.load_frames(frames)
    if r == 0:
        k = GopCoder(keep, W, H, F, sample_bytes=bits // 8, planar_luma=True, keep_interleaved=False)
        k.load_frames(frames); k.encode(); keep.sync()
    for _ in range(5):
        coder.encode()
    ctx.sync()
    ctx.timing_reset(); ctx.timing(True)
    for _ in range(30):
        coder.encode()
    ctx.sync(); ctx.timing(False)
    t = {k2: v[0] / 30 * 1e3 for k2, v in ctx.timing_read().items() if v[1]}
    print("round %2d: total %.1f us | %s | luma %x masks %x filters %x witness %x" % (r, sum(t.values()), "  ".join("%s %.1f" % kv for kv in t.items()),
          coder.luma.ptr, coder.masks.ptr, coder.filters.ptr, coder.witness.ptr), flush=True)
    coder.close(); ctx.close()
    del coder, ctx, stream
    ballast.append(torch.empty(int(rng.integers(1, 64)) * (1 << 20) + int(rng.integers(0, 4096)) * 256, dtype=torch.uint8, device=device))
    if r % 3 == 2:
        torch.cuda.empty_cache()
