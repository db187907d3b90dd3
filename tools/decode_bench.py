#!/usr/bin/env python3
"""A6 on the GPU: decode (filter + witness -> mask) throughput for one 1080p GOP whose coded rows are
already in HBM, checked against the masks the encoder saw.  Usage: python tools/decode_bench.py [W H F]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.gop import GopCoder
from new_bloom_filter_repo_amd.synthetic import make_gop, P_KSTAR_2_3

W, H, F = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 30)
ctx = nat.Context(0)
coder = GopCoder(ctx, W, H, F)
coder.load_frames(np.stack(make_gop(2000, W, H, F, p=P_KSTAR_2_3)))
coder.encode()
ctx.sync()
n, pairs = W * H, F - 1
out = ctx.alloc(coder.mask_stride * pairs)
L = nat.lib()


def decode():
    nat.check(L.rbf_bloom_decode_batch(ctx.handle, coder.filters.ptr, coder.filter_stride, coder.witness.ptr, coder.witness_stride,
                                       n, pairs, coder.params, ctypes.byref(coder.seeds), out.ptr, coder.mask_stride))


for _ in range(3):
    decode()
ctx.sync()
ctx.timing_reset()
ctx.timing(True)
t0 = time.perf_counter()
steps = 30
for _ in range(steps):
    decode()
ctx.sync()
dt = (time.perf_counter() - t0) / steps
ctx.timing(False)
same = np.array_equal(out.download(), coder.masks.numpy(ctx)[:coder.mask_stride * pairs])
print("decode %dx%d x %d inter-frames: %.1f us/GOP = %.1f Gpixel/s (with per-kernel events); masks identical: %s" % (
    W, H, pairs, dt * 1e6, pairs * n / dt / 1e9, same))
print("kernels (us/GOP):", {k: round(v[0] / steps * 1e3, 1) for k, v in ctx.timing_read().items() if v[1]})
