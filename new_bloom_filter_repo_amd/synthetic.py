"""Synthetic YUV444 frame sequences for tests and bench (no network, no datasets).

Recipe: `rng = np.random.default_rng(seed)`; keyframe of uniform random
samples; every following frame changes a Bernoulli(p) set of pixels -- a non-zero luma
residual plus fresh chroma -- so that with threshold 0 the luma mask marks exactly the
pixels that changed and reconstruction from (mask, values) is exact.

p = P_KSTAR_2_3 makes the reference's parameter rule pick k* = 2.3
(k = log2((1-p) ln^2 2 / p), improved_video_compressor.py:185).
"""
import math

import numpy as np

P_KSTAR_2_3 = 1.0 / (1.0 + 2.0 ** 2.3 / math.log(2.0) ** 2)   # 0.08888997000980829


def next_frame(rng, frame, p, scratch=None):
    """Return a new frame differing from `frame` on a Bernoulli(p) pixel set.  `scratch`: reusable (H, W) float64 array for
    the Bernoulli draws (same stream as rng.random((h, w)); a fresh 16 MB array per frame costs more in page faults than
    everything else here)."""
    h, w = frame.shape[:2]
    bits = 8 * frame.dtype.itemsize
    if scratch is None:
        scratch = np.empty((h, w), dtype=np.float64)
    change = rng.random(out=scratch) < p
    idx = np.flatnonzero(change)                  # raster order = the order boolean-mask assignment uses (same frames, ~10x faster)
    cnt = int(idx.size)
    out = frame.copy()
    flat = out.reshape(-1, out.shape[2])
    if bits == 8:
        resid = rng.integers(1, 256, cnt, dtype=np.uint16)
        flat[idx, 0] = ((flat[idx, 0].astype(np.uint16) + resid) & 0xFF).astype(np.uint8)
    else:
        # residuals in 1..32767: avoids the int16 blind spot |d| == 32768 (np.abs(int16 -32768) < 0)
        resid = rng.integers(1, 32768, cnt, dtype=np.uint32)
        flat[idx, 0] = ((flat[idx, 0].astype(np.uint32) + resid) & 0xFFFF).astype(np.uint16)
    flat[idx, 1] = rng.integers(0, 1 << bits, cnt, dtype=frame.dtype)
    flat[idx, 2] = rng.integers(0, 1 << bits, cnt, dtype=frame.dtype)
    return out


def make_gop(seed, width, height, nframes, p=P_KSTAR_2_3, dtype=np.uint8):
    """nframes interleaved YUV444 frames of shape (H, W, 3)."""
    rng = np.random.default_rng(seed)
    bits = 8 * np.dtype(dtype).itemsize
    frames = [rng.integers(0, 1 << bits, (height, width, 3), dtype=dtype)]
    scratch = np.empty((height, width), dtype=np.float64)
    for _ in range(nframes - 1):
        frames.append(next_frame(rng, frames[-1], p, scratch))
    return frames


def make_clip_shard(seed, width, height, first, stop, interval=30, p=P_KSTAR_2_3, dtype=np.uint8, threads=None):
    """Frames [first, stop) of a long synthetic clip, as an array (stop-first, H, W, 3).  The clip is a sequence
    of independent GOPs of `interval` frames (GOP g = make_gop(seed * 1000 + g, ...)), so any rank can produce
    its shard -- including a halo frame -- without generating the frames before it, and every rank sees the
    same clip."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    g = first // interval
    while g * interval < stop:
        jobs.append((g, max(first, g * interval), min(stop, (g + 1) * interval)))
        g += 1

    def one(job):                                 # the GOPs are independent streams: one thread each (numpy drops the GIL in the big ops)
        g, lo, hi = job
        return make_gop(seed * 1000 + g, width, height, hi - g * interval, p=p, dtype=dtype)[lo - g * interval:]
    with ThreadPoolExecutor(max(1, min(len(jobs), threads or 16))) as pool:
        parts = list(pool.map(one, jobs))
    return np.stack([f for part in parts for f in part])


def make_mask(seed, n, p):
    """Flat 0/1 uint8 vector with Bernoulli(p) ones (the reference's `binary_input`)."""
    return (np.random.default_rng(seed).random(n) < p).astype(np.uint8)
