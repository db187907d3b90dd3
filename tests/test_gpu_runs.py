"""Multi-run blocks (rbf_encode_runs: several keyframe-delimited GOPs in ONE mask / insert / reduce / query / compact launch sequence)
against the CPU oracle, frame by frame.  Reference semantics: every inter-frame is coded on its own (improved_video_compressor.py:
198-266); frame t is a keyframe of the caller's stream iff the caller says so (t % keyframe_interval == 0), and the pair
in front of a keyframe is not coded at all.
"""
import ctypes

import numpy as np
import pytest

from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd import params as P
from new_bloom_filter_repo_amd.dist import unpack_device_record
from new_bloom_filter_repo_amd.gop import GopCoder
from new_bloom_filter_repo_amd.synthetic import make_gop, next_frame
from test_gpu_bench_shape import check_records, decode_back, oracle_gop

pytestmark = pytest.mark.gpu


def make_runs(seed, W, H, runs, dtype=np.uint8):
    """runs: list of lists of densities; run r = a fresh keyframe followed by one frame per density.  Returns (frames array,
    run_starts) -- run_starts are the frame indices of the keyframes after the first."""
    rng = np.random.default_rng(seed)
    bits = 8 * np.dtype(dtype).itemsize
    frames, starts = [], []
    for r, dens in enumerate(runs):
        if r:
            starts.append(len(frames))
        frames.append(rng.integers(0, 1 << bits, (H, W, 3), dtype=dtype))
        for p in dens:
            frames.append(next_frame(rng, frames[-1], p))
    return np.stack(frames), starts


def check_block(oracle, ctx, frames, starts, planar, tag, pack=True):
    F, H, W = frames.shape[:3]
    n = W * H
    sb = frames.dtype.itemsize
    coder = GopCoder(ctx, W, H, F, sample_bytes=sb, planar_luma=planar, keep_interleaved=True, run_starts=starts)
    coder.load_frames(frames)
    coder.encode()
    res = coder.results()
    want = oracle_gop(oracle, frames)
    skipped = {t - 1 for t in starts}
    coded_res, coded_want = [], []
    for f in range(F - 1):
        if f in skipped:
            r = res[f]
            assert r.get("skipped") and not r["mask"].any() and r["ones"] == 0 and r["witness_bits"] == 0, (tag, f)
            assert int(coder.params[f].floor_k) == nat.PAIR_SKIPPED and int(coder.params[f].m) == 0 and coder.k[f] == 0.0
        else:
            assert not res[f].get("skipped"), (tag, f)
            coded_res.append(res[f])
            coded_want.append(want[f])
    check_records(coded_res, coded_want, n, tag)
    decode_back(ctx, coded_res, n, tag)
    if pack:                                       # the device-packed record: a skipped pair is a header row without payload
        block = coder.pack()
        ctx.sync()
        rows = unpack_device_record(block.numpy(ctx), n)
        assert len(rows) == F - 1
        for f, (g, r) in enumerate(zip(rows, res)):
            if f in skipped:
                assert g.get("skipped") and g["witness_bits"] == 0 and "filter" not in g and "mask" not in g, (tag, f)
                continue
            assert g["l"] == r["l"] and g["witness_bits"] == r["witness_bits"] and np.array_equal(g["witness"], r["witness"]), (tag, f)
            if r["l"]:
                assert np.array_equal(g["filter"], r["filter"]) and g["k"] == r["k"], (tag, f)
            else:
                assert np.array_equal(g["mask"], r["mask"]), (tag, f)
    coder.close()
    return res


# run 0: k* = 2.3; run 1: a keyframe directly followed by the next keyframe (a run of length 1: nothing to code); run 2: floor(k*) = 5,
# 0, a frame the reference passes through (p >= P_STAR), one it passes through because it is nearly static (p <= 1e-4), floor(k*) = 3;
# run 3: floor(k*) = 1
RUNS = [[0.0889] * 4, [], [0.01, 0.2, 0.4, 0.00004, 0.05], [0.15, 0.15]]


@pytest.mark.parametrize("force", [0, 8, 1], ids=["default", "barrett", "generic"])
@pytest.mark.parametrize("planar", [True, False], ids=["planar", "interleaved"])
def test_four_run_block_vs_oracle_frame_by_frame(oracle, force, planar):
    W, H = 640, 360                                # whole 1024-pixel segments: the fast mask kernel with the chunk table and the fused tail
    frames, starts = make_runs(51, W, H, RUNS)
    assert starts == [5, 6, 12]
    with nat.Context(0) as ctx:
        ctx.force_generic(force)
        res = check_block(oracle, ctx, frames, starts, planar, ("runs", force, planar))
        fks = sorted({r["floor_k"] for r in res if r["l"]})
        assert fks == [0, 1, 2, 3, 5], fks           # mixed floor(k*) inside one launch sequence
        assert sum(1 for r in res if not r.get("skipped") and r["l"] == 0) == 2        # the two passthrough frames


def test_runs_ragged_frame_generic_mask_kernel(oracle):
    """A frame size that is not whole 1024-pixel segments: the generic mask kernel codes the tail, and a skipped pair reaches it as the
    threshold INT32_MAX."""
    frames, starts = make_runs(52, 322, 181, [[0.09, 0.09], [0.09], [], [0.2, 0.05]])
    with nat.Context(0) as ctx:
        check_block(oracle, ctx, frames, starts, True, "ragged")
        check_block(oracle, ctx, frames, starts, False, "ragged-interleaved")


def test_runs_uint16(oracle):
    frames, starts = make_runs(53, 640, 360, [[0.0889, 0.0889], [0.0889, 0.05]], dtype=np.uint16)
    with nat.Context(0) as ctx:
        check_block(oracle, ctx, frames, starts, True, "u16")


def test_runs_more_runs_than_the_chunk_table_holds(oracle):
    """600 tiny frames, a keyframe every second frame: 300 runs do not fit the mask kernel's 256-entry chunk table (uniform chunks, the
    skipped pairs through their thresholds), and 599 pairs are more than one MAX_BATCH chunk of the Bloom kernels."""
    W, H = 64, 32
    runs = [[0.09]] * 300
    frames, starts = make_runs(54, W, H, runs)
    with nat.Context(0) as ctx:
        res = check_block(oracle, ctx, frames, starts, True, "many-runs")
        assert sum(1 for r in res if r.get("skipped")) == 299


def test_block_equals_its_runs_coded_one_by_one(oracle):
    """HIP against HIP, cheap and exhaustive over rows: the block's rows are the rows of every run coded as its own GOP."""
    W, H = 640, 360
    frames, starts = make_runs(55, W, H, [[0.0889] * 3, [0.05] * 2, [0.12] * 3])
    bounds = [0] + starts + [len(frames)]
    with nat.Context(0) as ctx:
        coder = GopCoder(ctx, W, H, len(frames), planar_luma=True, keep_interleaved=False, run_starts=starts)
        coder.load_frames(frames)
        coder.encode()
        block = coder.results()
        coder.close()
        for a, b in zip(bounds[:-1], bounds[1:]):
            c = GopCoder(ctx, W, H, b - a, planar_luma=True, keep_interleaved=False)
            c.load_frames(frames[a:b])
            c.encode()
            for j, r in enumerate(c.results()):
                g = block[a + j]
                for key in ("ones", "k", "l", "floor_k", "threshold", "witness_bits", "filter_ones"):
                    assert g[key] == r[key], (a, j, key)
                for key in ("mask", "filter", "witness"):
                    assert np.array_equal(g[key], r[key]), (a, j, key)
            c.close()


def test_1080p_four_gops_in_one_call_vs_oracle(oracle):
    """The shape bench.py's `batched_gops` leg times: 4 x 30 frames of 1920x1080, planar luma, one rbf_encode_runs."""
    W, H, F, G = 1920, 1080, 30, 4
    frames = np.concatenate([np.stack(make_gop(7100 + g, W, H, F)) for g in range(G)])
    starts = [F * g for g in range(1, G)]
    with nat.Context(0) as ctx:
        res = check_block(oracle, ctx, frames, starts, True, "1080p x 4 GOPs", pack=False)
        assert sum(1 for r in res if r.get("skipped")) == G - 1 and len(res) == G * F - 1


def test_begin_rejects_a_filter_stride_the_planner_could_outgrow():
    """Both begins check the filter stride against rbf_filter_stride_min BEFORE they touch the stream or the caller's buffers (the filters
    are only planned in the second half)."""
    W, H, F = 640, 360, 3
    n = W * H
    L = nat.lib()
    need = int(L.rbf_filter_stride_min(n))
    assert need * 8 >= int(0.31606 * n) and need % 8 == 0
    with nat.Context(0) as ctx:
        c = GopCoder(ctx, W, H, F, planar_luma=True, keep_interleaved=False)
        c.load_frames(np.stack(make_gop(1, W, H, F)))
        ctx.sync()
        before = c.masks.numpy(ctx).copy()
        sd = ctypes.byref(c.seeds)
        rc = L.rbf_encode_gop_begin(ctx.handle, c.luma.ptr, c.luma_bytes, F, W, H, W, 1, 1, 0, None, sd, c.masks.ptr, c.mask_stride, c.ones.ptr,
                                    c.filters.ptr, need - 8, c.witness.ptr, c.witness_stride, c.stats.ptr)
        assert rc == nat.RBF_EINVAL and b"rbf_filter_stride_min" in L.rbf_last_error()
        ctx.sync()
        assert np.array_equal(c.masks.numpy(ctx), before)
        c.encode()                                 # the context is still usable
        ctx.sync()
        c.close()


@pytest.mark.parametrize("W,H", [(640, 360), (322, 181)], ids=["whole_segments", "ragged"])
def test_runs_with_per_pair_thresholds(oracle, W, H):
    """Per-pair thresholds (the adaptive rule's thr_floors, improved_video_compressor.py:804-805) together with run starts: the fast mask
    kernel then takes the chunk table AND the threshold table, the generic kernel the thresholds with INT32_MAX for the skipped pairs."""
    frames, starts = make_runs(57, W, H, [[0.0889, 0.0889, 0.2], [0.05, 0.0889], [0.0889]])
    F, n = len(frames), W * H
    thr = [0, 3, 0, 9, 1, 0, 5, 2][:F - 1]
    with nat.Context(0) as ctx:
        coder = GopCoder(ctx, W, H, F, planar_luma=True, keep_interleaved=False, run_starts=starts)
        coder.thr_tab = (ctypes.c_int32 * (F - 1))(*thr)
        coder.load_frames(frames)
        coder.encode()
        res = coder.results()
        skipped = {t - 1 for t in starts}
        L = oracle.lib()
        sd = (ctypes.c_uint64 * 3)(*P.SEEDS_VIDEO)
        U8P = ctypes.POINTER(ctypes.c_uint8)
        for f in range(F - 1):
            r = res[f]
            if f in skipped:
                assert r.get("skipped") and not r["mask"].any() and r["ones"] == 0
                continue
            mask = np.ascontiguousarray(oracle.residual_mask(np.ascontiguousarray(frames[f][..., 0]), np.ascontiguousarray(frames[f + 1][..., 0]), float(thr[f])).reshape(-1), dtype=np.uint8)
            assert np.array_equal(np.unpackbits(r["mask"])[:n], mask), (f, "mask")
            k, l = oracle.optimal_params(n, np.uint64(int(mask.sum())) / n)
            if l == 0 or l >= n:
                assert r["l"] == 0
                continue
            bit_array, witness = np.zeros(l, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
            w = L.orc_compress(mask.ctypes.data_as(U8P), n, l, ctypes.c_double(k), sd, bit_array.ctypes.data_as(U8P), witness.ctypes.data_as(U8P))
            assert (r["k"], r["l"]) == (k, l) and np.array_equal(np.unpackbits(r["filter"])[:l], bit_array), (f, "filter")
            assert r["witness_bits"] == w and np.array_equal(np.unpackbits(r["witness"])[:w], witness[:w]), (f, "witness")
        coder.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16], ids=["u8", "u16"])
@pytest.mark.parametrize("pairs", [1, 2, 3, 127, 128, 129, 299], ids=lambda p: "pairs%d" % p)
def test_mask_kernel_one_long_chunk_counts_vs_oracle(oracle, dtype, pairs):
    """The GOP mask kernel keeps a wave's per-pair counts in ONE register (two pairs per lane, two pairs per DPP tree) and files them
    into LDS every 128 pairs: chunks of 1, 2, 3 pairs (the unrolled loop's tails), of exactly 128, of 129 and of 299 pairs (two
    flushes + a tail) in ONE temporal chunk (force bit 8: mask_chunks = 1), masks and counts against the oracle's A1."""
    W, H = 64, 32                                  # two whole 1024-pixel segments: the fast kernel, two waves
    n = W * H
    rng = np.random.default_rng(900 + pairs)
    frames = [rng.integers(0, 1 << (8 * np.dtype(dtype).itemsize), (H, W, 3), dtype=dtype)]
    for i in range(pairs):
        frames.append(next_frame(rng, frames[-1], [0.0889, 0.3, 0.0, 1.0][i % 4] if i % 7 else 0.02))
    frames = np.stack(frames)
    with nat.Context(0) as ctx:
        ctx.force_generic(1 << 8)
        coder = GopCoder(ctx, W, H, pairs + 1, sample_bytes=np.dtype(dtype).itemsize, planar_luma=True, keep_interleaved=False)
        coder.load_frames(frames)
        coder.encode()
        res = coder.results()
        for f in range(pairs):
            want = oracle.residual_mask(np.ascontiguousarray(frames[f][..., 0]), np.ascontiguousarray(frames[f + 1][..., 0]), 0.0).reshape(-1)
            assert np.array_equal(np.unpackbits(res[f]["mask"])[:n], want), (pairs, f)
            assert res[f]["ones"] == int(want.sum()), (pairs, f, res[f]["ones"], int(want.sum()))
        coder.close()


@pytest.mark.parametrize("force", [0, 1, 8], ids=["default", "generic", "barrett"])
@pytest.mark.parametrize("W,H", [(640, 360), (322, 181)], ids=["whole_segments", "ragged"])
def test_outputs_do_not_depend_on_what_the_buffers_held_before(oracle, force, W, H):
    """Nobody clears witness rows any more (k_chunk_offsets zeroes the dwords the compaction's workgroups share, the compaction writes the
    rest) and the fast insert path never touches the filter row of a pair that is not coded: so every output buffer is filled with 0xFF
    before rbf_encode_runs, and afterwards every coded pair's filter and witness must equal the oracle's INCLUDING the zero padding up to
    the 64-bit word that rbf_pack_records and results() rely on; masks, counts and stats likewise.  Then the same through
    rbf_bloom_encode_batch (BloomEngine.encode)."""
    frames, starts = make_runs(58, W, H, [[0.0889, 0.2, 0.0889], [0.05], [0.4, 0.0889]])      # one passthrough frame (p >= P_STAR) among them
    F, n = len(frames), W * H
    L = nat.lib()
    with nat.Context(0) as ctx:
        ctx.force_generic(force)
        coder = GopCoder(ctx, W, H, F, planar_luma=True, keep_interleaved=False, run_starts=starts)
        coder.load_frames(frames)
        for blk in (coder.masks, coder.filters, coder.witness, coder.stats, coder.ones):
            nat.check(L.rbf_memset(ctx.handle, blk.ptr, 0xFF, blk.nbytes))
        ctx.sync()
        coder.encode()
        ctx.sync()
        want = [dict(zip(("mask", "ones", "k", "l", "filter", "witness"), row)) for row in oracle_gop(oracle, frames)]
        skipped = {t - 1 for t in starts}
        pairs = F - 1
        masks = coder.masks.numpy(ctx)[:pairs * coder.mask_stride].reshape(pairs, coder.mask_stride)
        filt = coder.filters.numpy(ctx)[:pairs * coder.filter_stride].reshape(pairs, coder.filter_stride)
        wit = coder.witness.numpy(ctx)[:pairs * coder.witness_stride].reshape(pairs, coder.witness_stride)
        stats = coder.stats.numpy(ctx)[:pairs * 8 * nat.STATS_PER_FRAME].view(np.uint64).reshape(pairs, nat.STATS_PER_FRAME)
        ones = coder.ones.numpy(ctx)[:8 * pairs].view(np.uint64)
        coded_masks = []
        for f in range(pairs):
            if f in skipped:
                assert not masks[f].any() and ones[f] == 0 and not stats[f].any(), f
                continue
            w = want[f]
            mask_row = np.zeros(coder.mask_stride, np.uint8)
            mask_row[:(n + 7) // 8] = np.packbits(w["mask"])
            assert np.array_equal(masks[f], mask_row), (f, "mask row incl. padding")
            assert int(ones[f]) == int(w["mask"].sum())
            l = w["l"]
            assert int(coder.params[f].m) == l
            if not l:
                assert stats[f, 0] == 0, f
                continue
            fb = np.zeros((l + 63) // 64 * 8, np.uint8)
            fb[:(l + 7) // 8] = np.packbits(w["filter"])
            assert np.array_equal(filt[f, :len(fb)], fb), (f, "filter incl. padding to the 64-bit word")
            wb = len(w["witness"])
            assert int(stats[f, 0]) == wb and int(stats[f, 1]) == int(w["filter"].sum())
            wr = np.zeros((wb + 63) // 64 * 8, np.uint8)
            wr[:(wb + 7) // 8] = np.packbits(np.asarray(w["witness"], dtype=np.uint8))
            assert np.array_equal(wit[f, :len(wr)], wr), (f, "witness incl. padding to the 64-bit word")
            coded_masks.append((f, w))
        # the packed record copies whole 64-bit words: it must not carry any of the poison
        block = coder.pack()
        ctx.sync()
        rows = unpack_device_record(block.numpy(ctx), n)
        for f, w in coded_masks:
            assert rows[f]["witness_bits"] == len(w["witness"]) and np.array_equal(np.unpackbits(rows[f]["witness"])[:len(w["witness"])], np.asarray(w["witness"], dtype=np.uint8))
        coder.close()
        # ---- rbf_bloom_encode_batch on poisoned filter / witness / stats buffers
        from new_bloom_filter_repo_amd.engine import BloomEngine
        eng = BloomEngine(ctx)
        sel = [w for _, w in coded_masks]
        eng.upload_masks(np.stack([np.packbits(w["mask"]) for w in sel]), n)
        plist = [P.filter_params(w["k"], w["l"]) for w in sel]
        fstride = max(nat.packed_stride(p[0]) for p in plist)
        for name, nbytes in (("filters", len(sel) * fstride), ("witness", len(sel) * nat.packed_stride(n)), ("stats", len(sel) * nat.STATS_PER_FRAME * 8)):
            b = eng._buf(name, nbytes)
            nat.check(L.rbf_memset(ctx.handle, b.ptr, 0xFF, b.nbytes))
        ctx.sync()
        got = eng.encode(n, plist)
        for g, w in zip(got, sel):
            assert np.array_equal(np.unpackbits(g["filter"])[:w["l"]], w["filter"]) and not (np.unpackbits(g["filter"])[w["l"]:]).any()
            wb = len(w["witness"])
            assert g["witness_bits"] == wb and np.array_equal(np.unpackbits(g["witness"])[:wb], np.asarray(w["witness"], dtype=np.uint8))
            assert not np.unpackbits(g["witness"])[wb:].any()
        eng.close()
