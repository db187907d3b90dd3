import sys, numpy as np
sys.path.insert(0, '/root/repo')
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.gop import GopCoder
from new_bloom_filter_repo_amd.synthetic import make_gop, P_KSTAR_2_3
ctx = nat.Context(0)
W, H, F = 1920, 1080, 30
coder = GopCoder(ctx, W, H, F)
frames = np.stack(make_gop(2000, W, H, F, p=P_KSTAR_2_3))
coder.load_frames(frames)
for chunks in (3, 4, 5, 6, 8):
    ctx.force_generic(chunks << 8)
    for _ in range(3): coder.encode()
    ctx.sync(); ctx.timing_reset(); ctx.timing(1 << nat.K_MASK)
    for _ in range(10): coder.encode()
    ctx.timing(False); t = ctx.timing_read()["mask"]
    print("chunks", chunks, "mask us", round(t[0] / t[1] * 1000, 1))
