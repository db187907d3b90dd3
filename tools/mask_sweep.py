"""Temporal chunks of the GOP mask kernel (rbf_ctx_force_generic bits 8-12) against its time alone, planar luma, 1080p x 30.
Usage (GPU box): python tools/mask_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.gop import GopCoder
from new_bloom_filter_repo_amd.synthetic import make_gop, P_KSTAR_2_3

ctx = nat.Context(0)
W, H, F = 1920, 1080, 30
coder = GopCoder(ctx, W, H, F, planar_luma=True, keep_interleaved=False)
frames = np.stack(make_gop(2000, W, H, F, p=P_KSTAR_2_3))
coder.load_frames(frames)
for chunks in (0, 1, 2, 3, 4, 5, 6, 8):
    ctx.force_generic(chunks << 8)
    for _ in range(3):
        coder.encode()
    ctx.sync()
    ctx.timing_reset()
    ctx.timing(1 << nat.K_MASK)
    for _ in range(20):
        coder.encode()
    ctx.sync()
    ctx.timing(False)
    t = ctx.timing_read()["mask"]
    print("chunks %s: mask %.1f us" % (chunks or "auto", t[0] / t[1] * 1000))
