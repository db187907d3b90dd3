"""The launch shapes bench.py times, checked against the CPU oracle frame by frame (not HIP vs HIP):

* BASELINE config 2: 1920x1080 YUV444 uint8 x 30-frame GOP through GopCoder.encode() (= rbf_encode_gop, default
  kernels: 29-frame frames-inner query, (29 frames x 8 slices) LDS insert), FOUR contexts in flight on four
  HIP streams sharing the resident frames -- every pipeline's 29 (mask, k, l, filter, witness, counts) against
  oracle.residual_mask / orc_compress, then decoded back on the GPU.
* BASELINE config 4: 3840x2160 x 9 frames (two-kernel insert over LDS tiles, tiled query) the same way, plus the tiled
  insert it replaced and a 2560x1440 GOP (one 136 KB tile per frame).
* BASELINE config 5, single-GPU half: 1920x1080 uint16 x 30 frames, GOP record vs the oracle and the whole
  ImprovedVideoCompressor round trip under verify_lossless / verify_bit_exact
  (verify_true_lossless.py:241-249,338-492 semantics).
"""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd import params as P
from new_bloom_filter_repo_amd.engine import BloomEngine
from new_bloom_filter_repo_amd.gop import GopCoder
from new_bloom_filter_repo_amd.synthetic import make_gop

pytestmark = pytest.mark.gpu

U8P = ctypes.POINTER(ctypes.c_uint8)


def oracle_gop(oracle, frames, seeds=P.SEEDS_VIDEO):
    """Per inter-frame: (mask bits, k, l, bit_array, witness) from the CPU oracle, frames spread over host threads."""
    F = frames.shape[0]
    n = frames.shape[1] * frames.shape[2]
    L = oracle.lib()
    sd = (ctypes.c_uint64 * 3)(*seeds)

    def one(f):
        mask = oracle.residual_mask(np.ascontiguousarray(frames[f][..., 0]), np.ascontiguousarray(frames[f + 1][..., 0]), 0.0).reshape(-1)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        ones = int(mask.sum())
        p = np.uint64(ones) / n
        k, l = oracle.optimal_params(n, p)
        if p >= P.P_STAR or l == 0 or l >= n:
            return mask, ones, 0.0, 0, None, None
        bit_array = np.zeros(l, dtype=np.uint8)
        witness = np.zeros(n, dtype=np.uint8)
        w = L.orc_compress(mask.ctypes.data_as(U8P), n, l, ctypes.c_double(k), sd, bit_array.ctypes.data_as(U8P), witness.ctypes.data_as(U8P))
        return mask, ones, k, l, bit_array, witness[:w].copy()
    with ThreadPoolExecutor(min(F - 1, os.cpu_count() or 1)) as pool:
        return list(pool.map(one, range(F - 1)))


def check_records(res, want, n, tag):
    assert len(res) == len(want)
    for f, (r, (mask, ones, k, l, bit_array, witness)) in enumerate(zip(res, want)):
        assert np.array_equal(np.unpackbits(r["mask"])[:n], mask), (tag, f, "mask")
        assert r["ones"] == ones, (tag, f, "ones")
        assert (r["k"], r["l"]) == (k, l), (tag, f, "geometry", r["k"], r["l"], k, l)
        if l == 0:
            assert r["witness_bits"] == 0, (tag, f)
            continue
        assert np.array_equal(np.unpackbits(r["filter"])[:l], bit_array), (tag, f, "filter")
        assert r["filter_ones"] == int(bit_array.sum()), (tag, f, "filter popcount")
        assert r["witness_bits"] == len(witness), (tag, f, "witness length", r["witness_bits"], len(witness))
        assert np.array_equal(np.unpackbits(r["witness"])[:len(witness)], witness), (tag, f, "witness")


def decode_back(ctx, res, n, tag):
    eng = BloomEngine(ctx)
    coded = [r for r in res if r["l"]]
    dec = eng.decode(n, [(r["l"], r["floor_k"], r["threshold"]) for r in coded], [r["filter"] for r in coded], [r["witness"] for r in coded])
    for i, r in enumerate(coded):
        assert np.array_equal(dec[i][:(n + 7) // 8], r["mask"]), (tag, i, "decode")
    eng.close()


def test_config2_1080p_gop_four_pipelines_vs_oracle(oracle):
    import torch
    W, H, F = 1920, 1080, 30
    n = W * H
    frames = np.stack(make_gop(2000, W, H, F))
    want = oracle_gop(oracle, frames)
    device = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device) for _ in range(4)]
    ctxs = [nat.Context(0, s.cuda_stream) for s in streams]
    coders = []
    for c in ctxs:
        coders.append(GopCoder(c, W, H, F, channels=3, sample_bytes=1, frames_block=coders[0].frames if coders else None))
    coders[0].load_frames(frames)
    ctxs[0].sync()
    for rep in range(3):                                     # the bench's rotation: all four pipelines overlap on the GPU
        for k in range(4):
            with torch.cuda.stream(streams[k]):
                coders[k].encode()
    torch.cuda.synchronize(device)
    for k in range(4):
        res = coders[k].results()
        check_records(res, want, n, "pipeline %d" % k)
        if k == 0:
            decode_back(ctxs[0], res, n, "pipeline 0")
    for c in coders:
        c.close()
    for c in ctxs:
        c.close()


def test_two_phase_gop_interleaved_over_two_contexts_vs_oracle(oracle):
    """rbf_encode_gop_begin / _poll / _finish (SURVEY 8b's masks -> host -> blooms split): ONE host thread drives two contexts with
    DIFFERENT GOPs and issues begin(k + 1) before finish(k), three rounds; both contexts' records equal the CPU oracle's frame by frame.
    Also the contract's errors: a second begin on a busy context, finish / poll without begin, and a bad argument that must leave the
    context usable and the caller's buffers untouched (validation in front of the token and the witness clear)."""
    import time
    import torch
    W, H, F = 1920, 1080, 12
    n = W * H
    gops = [np.stack(make_gop(2100 + k, W, H, F)) for k in range(2)]
    want = [oracle_gop(oracle, g) for g in gops]
    device = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device) for _ in range(2)]
    ctxs = [nat.Context(0, s.cuda_stream) for s in streams]
    coders = [GopCoder(c, W, H, F, channels=3, sample_bytes=1, planar_luma=True, keep_interleaved=False) for c in ctxs]
    for c, g in zip(coders, gops):
        c.load_frames(g)
    L = nat.lib()
    # errors first
    for fn in (lambda: coders[0].encode_finish(), lambda: coders[0].encode_ready()):
        with pytest.raises(nat.RbfError) as e:
            fn()
        assert e.value.code == nat.RBF_EINVAL
    coders[0].witness.buf.upload(np.full(64, 0xAB, dtype=np.uint8))
    c0 = coders[0]
    with pytest.raises(nat.RbfError) as e:             # mask stride 7: refused before anything is enqueued or cleared
        nat.check(L.rbf_encode_gop_begin(c0.ctx.handle, c0.luma.ptr, c0.luma_bytes, F, W, H, W, 1, 1, 0, None, ctypes.byref(c0.seeds),
                                         c0.masks.ptr, 7, c0.ones.ptr, c0.filters.ptr, c0.filter_stride, c0.witness.ptr, c0.witness_stride, c0.stats.ptr))
    assert e.value.code == nat.RBF_EINVAL
    ctxs[0].sync()
    assert (c0.witness.buf.download(64) == 0xAB).all(), "a refused begin cleared the caller's witness rows"
    coders[0].encode_begin()
    with pytest.raises(nat.RbfError) as e:
        coders[0].encode_begin()
    assert e.value.code == nat.RBF_EINVAL
    coders[0].encode_finish()
    ctxs[0].sync()
    check_records(coders[0].results(), want[0], n, "after the refused calls")
    # the interleaving the bench uses
    for rep in range(3):
        coders[0].encode_begin()
        coders[1].encode_begin()
        coders[0].encode_finish()
        coders[1].encode_finish()
    coders[0].encode_begin()
    t0 = time.time()
    while not coders[0].encode_ready():                # poll never blocks; the counts arrive within the mask kernel's time
        assert time.time() - t0 < 30
    coders[1].encode_begin()
    coders[1].encode_finish()
    coders[0].encode_finish()
    torch.cuda.synchronize(device)
    for k in range(2):
        check_records(coders[k].results(), want[k], n, "context %d" % k)
    for c in coders:
        c.close()
    for c in ctxs:
        c.close()


def test_back_to_back_gops_without_sync_vs_oracle(oracle):
    """Two contexts, each rotating over THREE different resident GOPs, four rounds back to back without a sync in between (a GOP's mask
    stage is enqueued right behind the previous GOP's compaction on the same stream and reuses its output rows): every GOP's records, read
    right after its encode, equal the CPU oracle's; pack() right behind the encode sees finished witness rows; the decode direction gives
    the masks back."""
    import torch
    from new_bloom_filter_repo_amd.dist import unpack_device_record
    W, H, F = 1920, 1080, 10
    n = W * H
    gops = [[np.stack(make_gop(2300 + 10 * k + g, W, H, F, p=[0.0889, 0.2, 0.03][g])) for g in range(3)] for k in range(2)]
    want = [[oracle_gop(oracle, g) for g in gk] for gk in gops]
    device = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device) for _ in range(2)]
    ctxs = [nat.Context(0, s.cuda_stream) for s in streams]
    coders = [GopCoder(c, W, H, F, channels=3, sample_bytes=1, planar_luma=True, keep_interleaved=False, resident_gops=3) for c in ctxs]
    for c, gk in zip(coders, gops):
        for g in range(3):
            c.load_frames(gk[g], g)
    for rnd in range(4):                                    # no sync inside: GOP g + 1's mask stage meets GOP g's compaction
        for g in range(3):
            for c in coders:
                c.encode(g)
    for g in (0, 2, 1):                                     # ... and now every GOP, checked right behind its encode
        for k, c in enumerate(coders):
            c.encode(g)
        for k, c in enumerate(coders):
            res = c.results()
            check_records(res, want[k][g], n, "context %d gop %d" % (k, g))
            if k == 0:
                rows = unpack_device_record(c.pack().numpy(c.ctx), n)     # pack right behind the encode, on the same stream
                for r, w in zip(rows, res):
                    assert r["witness_bits"] == w["witness_bits"] and np.array_equal(r["witness"], w["witness"]), (g, "packed witness")
                decode_back(ctxs[0], res, n, "gop %d" % g)
    for c in coders:
        c.close()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("W,H,F,force", [(3840, 2160, 9, 0), (3840, 2160, 5, 128), (2560, 1440, 6, 0), (2560, 1440, 4, 1 << 15)],
                         ids=["2160p_records_hashed_2_tiles", "2160p_tiled_insert_tab", "1440p_records_hashed_1_tile", "1440p_again_no_table_rewrite"])
def test_config4_large_frames_gop_vs_oracle(oracle, W, H, F, force):
    """BASELINE config 4 (3840x2160 x 9) on its default kernels -- k_insert_positions (positions hashed: the pixel-index
    table would be 265 MB) + k_insert_records over 2 LDS tiles, k_query_f64t with 2 stages per frame -- and the tiled
    k_insert_tab it replaced; 2560x1440 takes the same kernels with ONE 136 KB tile per frame."""
    n = W * H
    frames = np.stack(make_gop(4000 + W, W, H, F))
    want = oracle_gop(oracle, frames)
    with nat.Context(0) as ctx:
        ctx.force_generic(force)
        coder = GopCoder(ctx, W, H, F, channels=3, sample_bytes=1)
        coder.load_frames(frames)
        for _ in range(2):
            coder.encode()
            res = coder.results()
            check_records(res, want, n, "%dx%d" % (W, H))
        decode_back(ctx, res, n, "%dx%d" % (W, H))
        coder.close()


def test_config5_1080p_uint16_gop_and_surface_round_trip(oracle):
    from new_bloom_filter_repo_amd import ImprovedVideoCompressor
    from new_bloom_filter_repo_amd.verify import verify_bit_exact
    W, H, F = 1920, 1080, 30
    n = W * H
    frames = np.stack(make_gop(5000, W, H, F, dtype=np.uint16))
    want = oracle_gop(oracle, frames)
    with nat.Context(0) as ctx:
        coder = GopCoder(ctx, W, H, F, channels=3, sample_bytes=2)
        coder.load_frames(frames)
        coder.encode()
        res = coder.results()
        check_records(res, want, n, "uint16")
        decode_back(ctx, res, n, "uint16")
        coder.close()
        # the product surface on the same clip: keyframe every 30 -> 1 keyframe + 29 Bloom inter-frames
        comp = ImprovedVideoCompressor(noise_tolerance=0, min_diff_threshold=0, max_diff_threshold=10, keyframe_interval=30, ctx=ctx)
        originals = [f.copy() for f in frames]
        out = comp.compress_video([f for f in frames], None, input_color_space="YUV")
        assert out["frame_count"] == F and out["keyframes"] == 1
        decoded = comp.decompress_video(compressed_frames=comp.last_compressed_frames)
        v = comp.verify_lossless(originals, decoded)
        assert v["lossless"] and v["exact_frame_matches"] == F, v
        b = verify_bit_exact(originals, decoded, color_space="YUV")
        assert b["success"] and b["exact_matches"] == F and b["different_frames"] == 0, b
        comp.close()


@pytest.mark.parametrize("dtype,size,thr", [(np.uint8, (640, 360), 0.0), (np.uint8, (1920, 1080), 0.0), (np.uint16, (640, 360), 0.0), (np.uint8, (322, 181), 3.0)],
                         ids=["u8_360p", "u8_1080p", "u16_360p", "u8_ragged_thr3"])
def test_planar_luma_layout_equals_interleaved_and_oracle(oracle, dtype, size, thr):
    """The planar-Y resident layout (bench.py's default, rbf_extract_luma_batch / GopCoder(planar_luma=True)): Y planes extracted
    on the device (interleaved frames kept: the changed-value gather still works) and on the host (luma only crosses PCIe, several
    resident GOP slots) give the records of the interleaved layout and of the CPU oracle; the gathered values are the same."""
    W, H = size
    n, F = W * H, 6
    gops = [np.stack(make_gop(500 + g, W, H, F, p=0.07, dtype=dtype)) for g in range(2)]
    sb = np.dtype(dtype).itemsize
    ctx = nat.Context(0)
    inter = GopCoder(ctx, W, H, F, sample_bytes=sb, threshold=thr)
    dev = GopCoder(ctx, W, H, F, sample_bytes=sb, threshold=thr, planar_luma=True, keep_interleaved=True)
    host = GopCoder(ctx, W, H, F, sample_bytes=sb, threshold=thr, planar_luma=True, keep_interleaved=False, resident_gops=2)
    for g, gop in enumerate(gops):
        host.load_frames(gop, g)
    for g, gop in enumerate(gops):
        inter.load_frames(gop)
        dev.load_frames(gop)
        inter.encode(); dev.encode(); host.encode(g)
        a, b, c = inter.results(), dev.results(), host.results()
        for f in range(F - 1):
            for key in ("ones", "k", "l", "witness_bits", "filter_ones"):
                assert a[f][key] == b[f][key] == c[f][key], (g, f, key)
            for key in ("mask", "filter", "witness"):
                assert np.array_equal(a[f][key], b[f][key]) and np.array_equal(a[f][key], c[f][key]), (g, f, key)
        if thr == 0.0:
            check_records(c, oracle_gop(oracle, gop), n, "planar gop %d" % g)
        va, vb = inter.gather_values(), dev.gather_values()
        assert all(np.array_equal(x, y) for x, y in zip(va, vb))
    with pytest.raises(ValueError):
        host.gather_values()
    for c in (inter, dev, host):
        c.close()
    ctx.close()


@pytest.mark.parametrize("W,H,p", [(2560, 1440, 0.01), (3840, 2160, 0.004)], ids=["1440p_sparse", "2160p_sparse"])
def test_sparse_large_frames_single_tile_insert_hashes_instead_of_building_a_table(oracle, W, H, p):
    """Nearly static large frames: the filter fits ONE LDS tile, so the single-tile k_insert_tab runs -- but the pixel-index
    table of such a geometry (118 / 265 MB) is past the 96 MB cutoff, so that kernel hashes its set positions itself
    (k_insert_tab<0, true>) and no table is ever allocated.  Records against the oracle, and decoded back."""
    n, F = W * H, 3
    frames = np.stack(make_gop(4100 + W, W, H, F, p=p))
    want = oracle_gop(oracle, frames)
    assert all(l > 0 for _, _, _, l, _, _ in want)
    with nat.Context(0) as ctx:
        with GopCoder(ctx, W, H, F, channels=3, sample_bytes=1) as coder:
            coder.load_frames(frames)
            coder.encode()
            res = coder.results()
            check_records(res, want, n, "%dx%d sparse" % (W, H))
            decode_back(ctx, res, n, "%dx%d sparse" % (W, H))


def test_fused_mask_tail_equals_separate_finish_kernel(oracle):
    """rbf_encode_gop hands the ones counts to the host (and clears the witness rows / stats) inside the GOP mask kernel when
    that kernel covers the whole frame (640x360 = 225 segments of 1024 pixels); RBF_OPT_SEPARATE_FINISH keeps the k_finish_ones
    launch, and 322x181 (no whole segments at the end) takes that path by itself.  Same records, repeatedly (the ticket and the
    accumulator must come back to zero), against the oracle."""
    for W, H in ((640, 360), (322, 181)):
        n, F = W * H, 7
        frames = np.stack(make_gop(4300 + W, W, H, F, p=0.06))
        want = oracle_gop(oracle, frames)
        with nat.Context(0) as ctx:
            with GopCoder(ctx, W, H, F) as coder:
                coder.load_frames(frames)
                for rep in range(3):
                    ctx.option(nat.OPT_SEPARATE_FINISH, rep == 1)
                    coder.encode()
                    check_records(coder.results(), want, n, "%dx%d rep %d" % (W, H, rep))


def test_every_floor_k_branch_of_the_query_kernel_in_one_1080p_batch(oracle):
    """One 1080p GOP whose inter-frames have densities 0.3 ... 0.003, i.e. floor(k*) = 0, 0, 1, 2, 3, 3, 4, 5, 6, 7 in ONE launch of
    k_query_u64w (class loops 1 ... 5 in rows -- 5 single-buffered -- and the plain pass for 0, 6 and 7; ten different activation
    thresholds for the rank search; the host orders the frames by class, the records name their output rows), one with
    floor(k*) <= 3 in k_query_u64 (the 111-register kernel), and one with an ODD number of coded frames of one class (the packed
    pass counts end on a single frame).  Every frame against the oracle."""
    from new_bloom_filter_repo_amd.synthetic import next_frame
    W, H = 1920, 1080
    n = W * H
    for dens, ks in (([0.3, 0.2, 0.12, 0.0889, 0.05, 0.03, 0.02, 0.012, 0.006, 0.003], [0, 1, 2, 3, 4, 5, 6, 7]),
                     ([0.2, 0.12, 0.0889, 0.05, 0.03], [0, 1, 2, 3]),
                     ([0.0889, 0.0889, 0.0889], [2])):
        rng = np.random.default_rng(20260927 + len(dens))
        frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)]
        for p in dens:
            frames.append(next_frame(rng, frames[-1], p))
        frames = np.stack(frames)
        want = oracle_gop(oracle, frames)
        assert sorted({int(k) for (_, _, k, l, _, _) in want if l}) == ks
        ctx = nat.Context(0)
        coder = GopCoder(ctx, W, H, len(frames), planar_luma=True)
        coder.load_frames(frames)
        coder.encode()
        res = coder.results()
        check_records(res, want, n, "floor_k %s" % ks)
        decode_back(ctx, res, n, "floor_k %s" % ks)
        coder.close()
        ctx.close()


def test_tiled_query_kernels_by_floor_k_at_2160p(oracle):
    """3840x2160 (two LDS tiles per frame), k_query_s64t: batches with floor(k*) = 0, 0, 1, 2 / 2, 3 / 4, 5, 6 -- every instantiation
    of its frame body that keeps the probe positions in registers (0 ... 4) and the one that walks them again per tile (5 and up;
    round 3 sent every batch with floor(k*) > 2 to k_query_r64t).  Every frame against the oracle."""
    from new_bloom_filter_repo_amd.synthetic import next_frame
    W, H = 3840, 2160
    n = W * H
    for seed, dens, ks in ((1, [0.3, 0.2, 0.12, 0.0889], [0, 1, 2]), (2, [0.0889, 0.05], [2, 3]), (3, [0.02, 0.012, 0.006], [4, 5, 6])):
        rng = np.random.default_rng(seed)
        frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)]
        for p in dens:
            frames.append(next_frame(rng, frames[-1], p))
        frames = np.stack(frames)
        want = oracle_gop(oracle, frames)
        assert sorted({int(k) for (_, _, k, l, _, _) in want if l}) == ks
        ctx = nat.Context(0)
        coder = GopCoder(ctx, W, H, len(frames), planar_luma=True, keep_interleaved=False)
        coder.load_frames(frames)
        coder.encode()
        check_records(coder.results(), want, n, "2160p floor_k %s" % ks)
        coder.close()
        ctx.close()
