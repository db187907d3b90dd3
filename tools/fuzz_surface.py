#!/usr/bin/env python3
"""Surface soak: random small videos through ImprovedVideoCompressor -- the GOP-batched route (random block sizes over 1-3 GPU lanes) and
the frame-by-frame route must write the same container, and both must decode to the original frames bit for bit.
Usage: python tools/fuzz_surface.py SECONDS [SEED]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import new_bloom_filter_repo_amd as pkg
from new_bloom_filter_repo_amd.synthetic import next_frame

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int.from_bytes(os.urandom(4), "little")
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
t_end, cases, inter = time.time() + budget, 0, 0
while time.time() < t_end:
    W, H = int(rng.integers(1, 150)), int(rng.integers(1, 90))
    dtype = [np.uint8, np.uint16][int(rng.integers(0, 2))]
    F = int(rng.integers(1, 9))
    interval = int(rng.integers(1, 6))
    top = np.iinfo(dtype).max
    frames = [rng.integers(0, top + 1, (H, W, 3), dtype=dtype)]
    for _ in range(F - 1):
        kind = rng.random()
        if kind < 0.15:
            frames.append(frames[-1].copy())                                   # static frame
        elif kind < 0.25:
            f = frames[-1].copy(); f[int(rng.integers(0, H)), int(rng.integers(0, W)), 1 + int(rng.integers(0, 2))] ^= 1
            frames.append(f)                                                   # chroma-only change -> keyframe fallback
        else:
            frames.append(next_frame(rng, frames[-1], float(rng.choice([0.002, 0.05, 0.0889, 0.2, 0.4, 0.9]))))
    lanes = int(rng.integers(1, 4))
    block = [None, None, 2, 3, 5, 9][int(rng.integers(0, 6))]                  # blocks that do not line up with the keyframe interval too
    desc = dict(seed=seed, case=cases, W=W, H=H, dtype=np.dtype(dtype).name, F=F, interval=interval, lanes=lanes, block=block)
    a = pkg.ImprovedVideoCompressor(keyframe_interval=interval, verbose=False, gpu_lanes=lanes, block_frames=block)
    b = pkg.ImprovedVideoCompressor(keyframe_interval=interval, verbose=False)
    b.gop_batching = False
    a.compress_video([f.copy() for f in frames], None, input_color_space="YUV")
    b.compress_video([f.copy() for f in frames], None, input_color_space="YUV")
    if a.last_compressed_frames != b.last_compressed_frames:
        print("CONTAINER MISMATCH", desc); sys.exit(1)
    blob = a._container(a.last_compressed_frames)
    for comp in (a, b):
        dec = comp.decompress_video(compressed_frames=comp._parse_container(blob))
        if len(dec) != F or not all(np.array_equal(x, np.asarray(getattr(y, "data", y))) for x, y in zip(frames, dec)):
            print("DECODE MISMATCH", desc, "batched" if comp is a else "frame-by-frame"); sys.exit(1)
    inter += sum(1 for ty, _ in a.last_compressed_frames if ty == 2)
    a.close()
    b.close()
    cases += 1
print("ok: %d videos, %d inter-frames, both routes identical and lossless in %.0f s (seed %d)" % (cases, inter, budget, seed))
