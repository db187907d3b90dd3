#!/usr/bin/env python3
"""What the mask stage costs the overlapped step: the four-pipeline step of bench.py (rbf_encode_runs: mask -> host parameters -> insert ->
reduce -> query -> compaction) next to the SAME pipelines running only the Bloom half on the masks and parameters the full step left in HBM
(rbf_bloom_encode_batch: insert -> reduce -> query -> compaction, no mask kernel, no host round trip).
Usage (GPU box): python tools/step_without_mask.py [bits=8] [gops_per_call=4] [pipelines=4]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.gop import GopCoder, torch_allocator
from new_bloom_filter_repo_amd.synthetic import make_gop

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
GPC = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 4
W, H, F = 1920, 1080, 30
n, FB = W * H, 30 * GPC
dtype = np.uint8 if bits == 8 else np.uint16
device = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device) for _ in range(NP)]
ctxs = [nat.Context(0, s.cuda_stream) for s in streams]
coders = []
for k in range(NP):
    c = GopCoder(ctxs[k], W, H, FB, sample_bytes=bits // 8, allocator=torch_allocator(device), planar_luma=True, keep_interleaved=False, run_starts=[F * g for g in range(1, GPC)])
    c.load_frames(np.concatenate([np.stack(make_gop(7000 + 16 * k + j, W, H, F, dtype=dtype)) for j in range(GPC)]))
    coders.append(c)
L = nat.lib()
coded = GPC * (F - 1)


def full(c):
    c.encode()


def bloom_params(c):
    """The block's parameter rows as rbf_bloom_encode_batch takes them: a pair across a keyframe is a frame that is not coded (m = 0)."""
    if getattr(c, "_bloom_params", None) is None:
        c._bloom_params = (nat.FilterParams * c.pairs)()
        for i in range(c.pairs):
            skipped = int(c.params[i].m) == 0 and int(c.params[i].floor_k) == nat.PAIR_SKIPPED
            c._bloom_params[i].m, c._bloom_params[i].floor_k, c._bloom_params[i].threshold = int(c.params[i].m), 0 if skipped else int(c.params[i].floor_k), int(c.params[i].threshold)
    return c._bloom_params


def bloom_only(c):
    nat.check(L.rbf_bloom_encode_batch(c.ctx.handle, c.masks.ptr, c.mask_stride, n, c.pairs, bloom_params(c), ctypes.byref(c.seeds),
                                       c.filters.ptr, c.filter_stride, c.witness.ptr, c.witness_stride, c.stats.ptr))


def timed(fn, steps):
    for _ in range(3 * NP):
        for c in coders:
            fn(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        fn(coders[s % NP])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for c in coders:
    c.encode()
torch.cuda.synchronize()
steps = 160 if GPC == 1 else 60
t_full = timed(full, steps)
t_bloom = timed(bloom_only, steps)
t_full2 = timed(full, steps)
print("%d-bit, %d GOP(s) per call, %d pipelines: full step %.4f ms (%.0f Mpixel/s; again %.4f) | Bloom half only (masks resident, no host round trip) %.4f ms (%.0f Mpixel/s) | difference %.1f us per step"
      % (bits, GPC, NP, t_full * 1e3, coded * n / t_full / 1e6, t_full2 * 1e3, t_bloom * 1e3, coded * n / t_bloom / 1e6, (t_full - t_bloom) * 1e6))
