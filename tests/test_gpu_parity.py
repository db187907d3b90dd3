"""GPU parity: the HIP path (through the C ABI) vs the committed reference fixtures and the CPU oracle.
Bit-exact everywhere (integer / bit work).  Run with `-m gpu` on an MI355X."""
import hashlib

import numpy as np
import pytest

from conftest import load_json, load_npz
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd import params as P
from new_bloom_filter_repo_amd.engine import BloomEngine
from new_bloom_filter_repo_amd.synthetic import make_gop, make_mask, P_KSTAR_2_3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["lds_double_buffer", "lds_single_buffer", "lds_barrett_only", "hash_in_insert", "lds_tiled_1KiB", "lds_tiled_8KiB",
                                        "lds_tiled_8KiB_insert_tab", "generic"])
def eng(request):
    """Every LIVE kernel family must be bit-exact (round 4 pruned the superseded FP64 query kernels: k_query_f64 / f64t / p4 / r64 / r64t /
    s64 / s64w left the library in round 4; git keeps them): the LDS-resident fast path -- k_query_u64 (FP64 reductions, frame records
    and LDS-DMA staging; default whenever every filter of the batch has 2^15 <= m < 2^23 and fits LDS twice) with k_insert_tab
    gathering from the pixel-index hash table; "lds_single_buffer" / "lds_barrett_only": the integer Barrett kernels (k_query_lds,
    k_insert_lds), so both forms of h mod m are pinned to the same fixtures; "hash_in_insert": the insert kernel hashes its set
    positions itself; the LDS-tiled path that 4K-class filters take (forced here onto small filters with 1 KiB / 8 KiB tiles so that
    every test crosses tile boundaries): k_query_s64t for every floor(k*) (probe positions kept in registers across the tiles for
    floor(k*) <= 4, walked again per tile otherwise), inserts through k_insert_positions + k_insert_records inside rbf_encode_gop
    ("lds_tiled_8KiB_insert_tab" keeps the tiled k_insert_tab there too); and the generic global-memory path."""
    ctx = nat.Context(0)
    ctx.force_generic({"lds_double_buffer": 0, "lds_single_buffer": 2, "lds_barrett_only": 8, "hash_in_insert": 32, "lds_tiled_1KiB": 4 << 16, "lds_tiled_8KiB": 32 << 16,
                       "lds_tiled_8KiB_insert_tab": (32 << 16) | 128, "generic": 1}[request.param])
    e = BloomEngine(ctx)
    yield e
    e.close()
    ctx.close()


def unpack(a, nbits):
    return np.unpackbits(np.asarray(a, dtype=np.uint8))[:nbits]


def encode_one(eng, mask_bits, seeds, guard=True):
    """Host flow of BloomFilterCompressor.compress for one vector; returns None on passthrough."""
    n = len(mask_bits)
    p = np.uint64(mask_bits.sum()) / n
    if p >= P.P_STAR:
        return None
    k, l = P.optimal_params(n, p)
    if l == 0 or (guard and l >= n):
        return None
    pl = [P.filter_params(k, l)]
    eng.upload_masks(np.packbits(mask_bits)[None, :], n)
    r = eng.encode(n, pl, seeds)[0]
    r["k"], r["l"], r["plist"] = k, l, pl
    return r


def test_g3_fixtures_bit_exact(eng):
    meta = load_json("g3_320x180.json")["cases"]
    z = load_npz("g3_320x180.npz")
    checked = 0
    for rec in meta:
        n = rec["W"] * rec["H"]
        mask = unpack(z[rec["case"] + "_mask"], n)
        for vname, prefix, guard in (("video", "video", True), ("bloom_compress", "bc", False)):
            v = rec["variants"][vname]
            r = encode_one(eng, mask, tuple(v["seeds"]), guard)
            if v["passthrough"]:
                assert r is None
                continue
            assert float(r["k"]).hex() == v["k_hex"] and r["l"] == v["l"]
            assert r["witness_bits"] == v["wlen"] and r["filter_ones"] == v["bits_set"]
            assert np.array_equal(r["filter"], z["%s_%s_filter" % (rec["case"], prefix)])
            assert np.array_equal(r["witness"], z["%s_%s_witness" % (rec["case"], prefix)])
            dec = eng.decode(n, r["plist"], [r["filter"]], [r["witness"]], tuple(v["seeds"]))
            assert np.array_equal(unpack(dec[0], n), mask)
            checked += 1
        if "string" in rec["variants"]:       # config 1: rational_bloom_filter.py seeds (0, 1, ceil k)
            v = rec["variants"]["string"]
            k = float.fromhex(v["k_hex"])
            assert list(P.string_filter_seeds(k)) == v["seeds"]
            pl = [P.filter_params(k, v["l"])]
            eng.upload_masks(np.packbits(mask)[None, :], n)
            r = eng.encode(n, pl, tuple(v["seeds"]))[0]
            assert np.array_equal(r["filter"], z[rec["case"] + "_str_filter"])
            assert r["filter_ones"] == v["bits_set"] and r["witness_bits"] == v["passed"]
            checked += 1
    assert checked >= 12


def test_residual_mask_vs_oracle_and_fixture(eng, oracle):
    z = load_npz("g3_320x180.npz")
    for rec in load_json("g3_320x180.json")["cases"][:5]:
        frames = np.stack(make_gop(rec["seed"], rec["W"], rec["H"], 2, p=rec["density_req"]))
        masks, ones = eng.residual_masks(frames, 0.0)
        n = rec["W"] * rec["H"]
        assert np.array_equal(masks[0][:n // 8], z[rec["case"] + "_mask"])
        assert int(ones[0]) == rec["ones"]
    # thresholds, planar 2-D frames, ragged width (n % 64 != 0)
    rng = np.random.default_rng(77)
    a = rng.integers(0, 256, (3, 37, 53), dtype=np.uint8)
    for thr in (0.0, 0.5, 3.0, 17.9, 255.0, -1.0):
        masks, ones = eng.residual_masks(a, thr)
        for f in range(2):
            want = oracle.residual_mask(a[f], a[f + 1], thr).reshape(-1)
            assert np.array_equal(unpack(masks[f], 37 * 53), want), thr
            assert int(ones[f]) == int(want.sum())


def test_residual_mask_uint16_wrap(eng, oracle):
    z = load_npz("g5_masks.npz")
    for r in load_json("g5_masks.json")["rows"]:
        name, thr = r["name"], r["thr"]
        prev, curr = z[name + "_prev"], z[name + "_curr"]
        masks, ones = eng.residual_masks(np.stack([prev, curr]), thr)
        want = z["%s_mask_%s" % (name, str(thr).replace(".", "_"))]
        assert np.array_equal(unpack(masks[0], want.size), want.reshape(-1)), (name, thr)
        assert int(ones[0]) == r["ones"]


@pytest.mark.parametrize("dtype,channels", [(np.uint8, 1), (np.uint8, 3), (np.uint16, 1), (np.uint16, 3)])
def test_residual_mask_gop_layouts(eng, oracle, dtype, channels):
    """GOP-streaming mask kernel (16-byte aligned flat frames) + generic tail, every sample layout,
    5 frames, thresholds incl. the int16 wrap region for 16-bit samples."""
    W, H, F = 112, 75, 5                      # n = 8400 = 8 segments of 1024 + a 208-pixel tail
    rng = np.random.default_rng(1234)
    hi = 256 if dtype == np.uint8 else 65536
    shape = (F, H, W) if channels == 1 else (F, H, W, channels)
    frames = rng.integers(0, hi, shape, dtype=dtype)
    frames[1:][rng.random(frames[1:].shape) < 0.7] = 0      # plenty of equal / extreme pairs
    frames[2] = frames[1]                                   # an all-zero mask
    if dtype == np.uint16:                                  # |d| == 32768: numpy's abs(int16) stays negative -> bit 0, even at threshold 0
        lum = frames if channels == 1 else frames[..., 0]
        lum[3, 0, :6] = [32768, 0, 40000, 7232, 65535, 32767]
        lum[4, 0, :6] = [0, 32768, 7232, 40000, 32767, 65535]
    for thr in (0.0, 2.5, 100.0, 32767.0, -3.0):
        masks, ones = eng.residual_masks(frames, thr)
        for f in range(F - 1):
            a = frames[f] if channels == 1 else frames[f][:, :, 0]
            b = frames[f + 1] if channels == 1 else frames[f + 1][:, :, 0]
            want = oracle.residual_mask(a, b, thr).reshape(-1)
            assert np.array_equal(unpack(masks[f], W * H), want), (thr, f)
            assert int(ones[f]) == int(want.sum())


def test_batch_equals_oracle(eng, oracle):
    """A ragged batch: 5 frames of 211x97 (n % 1024 != 0, n % 64 != 0), different densities."""
    W, H = 211, 97
    n = W * H
    masks = [make_mask(100 + f, n, p) for f, p in enumerate((0.004, 0.05, P_KSTAR_2_3, 0.2, 0.3))]
    plist, ks = [], []
    for m in masks:
        k, l = P.optimal_params(n, np.uint64(m.sum()) / n)
        ks.append(k)
        plist.append(P.filter_params(k, l))
    eng.upload_masks(np.stack([np.packbits(m) for m in masks]), n)
    res = eng.encode(n, plist)
    for f, m in enumerate(masks):
        bm, wit, p, _, _ = oracle.compress(m)
        assert np.array_equal(unpack(res[f]["filter"], plist[f][0]), bm)
        assert res[f]["witness_bits"] == len(wit)
        assert np.array_equal(unpack(res[f]["witness"], len(wit)), np.array(wit, dtype=np.uint8))
        assert res[f]["filter_ones"] == int(bm.sum())
    dec = eng.decode(n, plist, [r["filter"] for r in res], [r["witness"] for r in res])
    for f, m in enumerate(masks):
        assert np.array_equal(unpack(dec[f], n), m)


@pytest.mark.parametrize("idx", [0, 1])
def test_g4_fullsize_digests(eng, idx):
    """1920x1080 and 3840x2160 at k* = 2.3: SHA-256 of filter / witness made by the reference."""
    r = load_json("g4_fullsize.json")["rows"][idx]
    n = r["W"] * r["H"]
    x = make_mask(r["seed"], n, r["p_req"])
    assert hashlib.sha256(np.packbits(x).tobytes()).hexdigest() == r["mask_sha256"]
    res = encode_one(eng, x, P.SEEDS_VIDEO)
    assert float(res["k"]).hex() == r["k_hex"] and res["l"] == r["l"]
    assert res["witness_bits"] == r["wlen"] and res["filter_ones"] == r["bits_set"]
    assert hashlib.sha256(res["filter"].tobytes()).hexdigest() == r["filter_sha256"]
    assert hashlib.sha256(res["witness"].tobytes()).hexdigest() == r["witness_sha256"]
    dec = eng.decode(n, res["plist"], [res["filter"]], [res["witness"]])
    assert np.array_equal(unpack(dec[0], n), x)          # encode -> decode round trip at full size


def test_edge_geometries(eng, oracle):
    """Tiny vectors, m == 1, floor_k == 0 (k < 1), floor_k large, all-pass and no-pass filters."""
    for n, p, seed in ((1, 1.0, 1), (63, 0.1, 2), (64, 0.1, 3), (65, 0.2, 4), (1023, 0.05, 5), (1025, 0.3, 6), (4097, 0.001, 7)):
        m = make_mask(seed, n, p)
        for k, l in ((0.1, 1), (0.7, 5), (1.0, 3), (2.5, 7), (12.25, 100), (3.2062923944339987, max(1, n // 3))):
            pl = [P.filter_params(k, l)]
            eng.upload_masks(np.packbits(m)[None, :], n)
            r = eng.encode(n, pl)[0]
            f = oracle.RationalFilter(l, k)
            for i in np.flatnonzero(m):
                f.add_index(int(i))
            assert np.array_equal(unpack(r["filter"], l), f.bit_array), (n, k, l)
            wit = [int(m[i]) for i in range(n) if f.check_index(i)]
            assert r["witness_bits"] == len(wit)
            assert np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8))
            if len(wit):
                dec = eng.decode(n, pl, [r["filter"]], [r["witness"]])
                assert np.array_equal(unpack(dec[0], n), m)


def test_adversarial_filter_lengths(eng, oracle):
    """Reductions mod m for awkward m (tiny, powers of two +-1, primes, > 2^20 so that the tiled kernels
    are used even by default) and k* values with extreme fractions."""
    n = 3000
    mask = make_mask(321, n, 0.12)
    ones = np.flatnonzero(mask)
    for m, k in ((2, 1.5), (3, 2.0), (4, 0.999999), (5, 3.7), (7, 1.0000001), (8, 2.5), (255, 4.25), (256, 2.3), (257, 2.3),
                 (65535, 2.9), (65536, 2.1), (65537, 5.5), (1048575, 2.3), (1048576, 2.3), (1048577, 2.3), (1398101, 3.3),
                 (4194303, 2.3)):
        pl = [P.filter_params(k, m)]
        eng.upload_masks(np.packbits(mask)[None, :], n)
        r = eng.encode(n, pl)[0]
        f = oracle.RationalFilter(m, k)
        for i in ones:
            f.add_index(int(i))
        assert np.array_equal(unpack(r["filter"], m), f.bit_array), (m, k)
        assert r["filter_ones"] == int(f.bit_array.sum())
        wit = [int(mask[i]) for i in range(n) if f.check_index(i)]
        assert r["witness_bits"] == len(wit), (m, k)
        assert np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8)), (m, k)
        dec = eng.decode(n, pl, [r["filter"]], [r["witness"]])
        assert np.array_equal(unpack(dec[0], n), mask), (m, k)


def test_filter_lengths_up_to_the_abi_limit(oracle):
    """m beyond 2^30 (where the small-m reduction no longer applies) up to the ABI limit 2^32 - 1:
    set bits, witness and round trip against positions computed with the oracle's primitives (a
    byte-per-bit oracle filter of 4 Gbit would not fit the host)."""
    import math
    ctx = nat.Context(0)
    eng = BloomEngine(ctx)
    n = 2000
    mask = make_mask(99, n, 0.15)
    s1, s2, sa = P.SEEDS_VIDEO
    for m, k in (((1 << 30) + 7, 2.3), ((1 << 31) + 11, 1.5), ((1 << 32) - 1, 3.25)):
        fk, pa = math.floor(k), k - math.floor(k)

        def positions(i):
            h1, h2 = oracle.hash_index(i, s1), oracle.hash_index(i, s2)
            cnt = fk + (1 if oracle.normalize(oracle.hash_index(i, sa)) < pa else 0)
            return [oracle.position(h1, h2, j, m) for j in range(cnt)]
        want = set()
        for i in np.flatnonzero(mask):
            want.update(positions(int(i)))
        pl = [P.filter_params(k, m)]
        eng.upload_masks(np.packbits(mask)[None, :], n)
        r = eng.encode(n, pl)[0]
        filt = r["filter"]
        assert r["filter_ones"] == len(want)
        idx = np.fromiter(want, dtype=np.int64)
        assert np.all((filt[idx >> 3] >> (7 - (idx & 7))) & 1)
        nz = np.flatnonzero(filt)
        assert int(np.unpackbits(filt[nz]).sum()) == len(want)             # and nothing else is set
        wit = [int(mask[i]) for i in range(n) if all(p in want for p in positions(i))]
        assert r["witness_bits"] == len(wit)
        assert np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8))
        dec = eng.decode(n, pl, [r["filter"]], [r["witness"]])
        assert np.array_equal(unpack(dec[0], n), mask)
    eng.close()
    ctx.close()


def test_three_gigapixel_vector_round_trip(oracle):
    """n > 2^31 (10-digit keys, 64-bit bit offsets everywhere): encode -> decode reproduces the mask, the
    counters agree with the data, and sampled keys from the top of the range sit where the oracle's
    hash puts them.  Size-independent properties only: the CPU oracle would need minutes for 3 Gpixel."""
    import math
    n = 3 * (1 << 30) + 5
    rng = np.random.default_rng(2024)
    nb = (n + 7) // 8
    packed = rng.integers(0, 256, nb, dtype=np.uint8)
    packed &= rng.integers(0, 256, nb, dtype=np.uint8)
    packed &= rng.integers(0, 256, nb, dtype=np.uint8)                      # density 1/8
    packed[-1] &= 0xFF << (8 * nb - n) & 0xFF                               # pad bits are zero
    ones = int(np.bitwise_count(packed).sum(dtype=np.int64))
    k, l = P.optimal_params(n, np.uint64(ones) / n)
    assert (1 << 29) < l < (1 << 32) and 1 < k < 2
    pl = [P.filter_params(k, l)]
    ctx = nat.Context(0)
    eng = BloomEngine(ctx)
    eng.upload_masks(packed[None, :], n)
    r = eng.encode(n, pl)[0]
    filt = r["filter"]
    assert r["filter_ones"] == int(np.bitwise_count(filt).sum(dtype=np.int64))
    assert ones <= r["witness_bits"] <= n
    # sampled set pixels from the last bytes of the vector: 10-digit decimal keys
    s1, s2, sa = P.SEEDS_VIDEO
    fk, pa = math.floor(k), k - math.floor(k)
    tail = np.flatnonzero(np.unpackbits(packed[-4096:])) + (nb - 4096) * 8
    assert len(tail) > 1000 and tail.min() > 3_000_000_000
    for i in tail[::17]:
        i = int(i)
        h1, h2 = oracle.hash_index(i, s1), oracle.hash_index(i, s2)
        cnt = fk + (1 if oracle.normalize(oracle.hash_index(i, sa)) < pa else 0)
        for j in range(cnt):
            pos = oracle.position(h1, h2, j, l)
            assert (filt[pos >> 3] >> (7 - (pos & 7))) & 1, (i, j)
    dec = eng.decode(n, pl, [filt], [r["witness"]])
    assert np.array_equal(dec[0][:nb], packed)
    eng.close()
    ctx.close()


def test_random_geometry_fuzz(oracle):
    """Seeded fuzz over frame sizes, dtypes, channel layouts, GOP lengths, densities and seed variants: the one-call
    GOP encoder against the oracle (mask, geometry, filter, witness), and decode back to the mask."""
    from new_bloom_filter_repo_amd.gop import GopCoder
    from new_bloom_filter_repo_amd.synthetic import next_frame
    rng = np.random.default_rng(20250926)
    ctx = nat.Context(0)
    eng = BloomEngine(ctx)
    for case in range(120):
        W, H = int(rng.integers(1, 260)), int(rng.integers(1, 120))
        C = int(rng.choice([1, 3]))
        dtype = [np.uint8, np.uint16][int(rng.integers(0, 2))]
        F = int(rng.integers(2, 6))
        seeds = [P.SEEDS_VIDEO, P.SEEDS_BLOOM_COMPRESS, (7, 11, 13)][case % 3]
        top = np.iinfo(dtype).max
        frames = [rng.integers(0, top + 1, (H, W, 3), dtype=dtype)]
        for _ in range(F - 1):
            p = float(rng.choice([0.0, 0.0005, 0.01, 0.05, 0.0889, 0.15, 0.25, 0.31, 0.33, 0.6, 1.0]))
            frames.append(next_frame(rng, frames[-1], p) if p else frames[-1].copy())
        frames = np.stack(frames)
        if C == 1:
            frames = np.ascontiguousarray(frames[..., 0])
        n = W * H
        coder = GopCoder(ctx, W, H, F, channels=C, sample_bytes=np.dtype(dtype).itemsize, seeds=seeds)
        coder.load_frames(frames)
        coder.encode()
        for f, r in enumerate(coder.results()):
            y0, y1 = (frames[f], frames[f + 1]) if C == 1 else (frames[f][..., 0], frames[f + 1][..., 0])
            want = oracle.residual_mask(np.ascontiguousarray(y0), np.ascontiguousarray(y1), 0.0).reshape(-1)
            assert np.array_equal(unpack(r["mask"], n), want), (case, f)
            bm, wit, p, _, _ = oracle.compress(want, seeds=seeds)
            if len(wit) == 0:
                assert r["l"] == 0 and r["witness_bits"] == 0, (case, f)
                continue
            k, l = oracle.optimal_params(n, p)
            assert (r["k"], r["l"]) == (k, l), (case, f)
            assert np.array_equal(unpack(r["filter"], l), bm), (case, f, W, H, l)
            assert r["witness_bits"] == len(wit) and np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8)), (case, f)
            dec = eng.decode(n, [P.filter_params(k, l)], [r["filter"]], [r["witness"]], seeds=seeds)
            assert np.array_equal(unpack(dec[0], n), want), (case, f)
        coder.close()
    eng.close()
    ctx.close()


@pytest.mark.parametrize("force", [32 << 16, (32 << 16) | (1 << 14), (32 << 16) | 128, 4 << 16, 0],
                         ids=["records_8KiB_tiles", "records_hashed_8KiB_tiles", "insert_tab_8KiB_tiles", "records_1KiB_tiles", "default"])
def test_gop_insert_paths_with_tiles(oracle, force):
    """rbf_encode_gop on filters of several LDS tiles (forced by a tile cap on a 640x360 GOP, m ~ 60-70 kbit, inside the
    FP64 reduction's range): the two-kernel insert (k_insert_positions + k_insert_records, taken because the call knows the
    masks' set-bit counts; with the positions gathered from the pixel-index hash table or hashed in the kernel, as frames of
    more than ~6 Mpixel do) and the tiled k_insert_tab must all reproduce the oracle's filter and witness bytes, also for a
    frame without changes (m = 0, no records) and for a dense one that is not Bloom-coded."""
    from new_bloom_filter_repo_amd.gop import GopCoder
    from new_bloom_filter_repo_amd.synthetic import next_frame
    rng = np.random.default_rng(4242)
    W, H, n = 640, 360, 640 * 360
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)]
    for p in (0.0889, 0.0, 0.05, 0.6, 0.2, 0.0889):
        frames.append(next_frame(rng, frames[-1], p) if p else frames[-1].copy())
    frames = np.stack(frames)
    ctx = nat.Context(0)
    ctx.force_generic(force)
    eng = BloomEngine(ctx)
    coder = GopCoder(ctx, W, H, len(frames))
    coder.load_frames(frames)
    for _ in range(2):                                        # second pass: hash table written by the previous query kernel
        coder.encode()
        coded = 0
        for f, r in enumerate(coder.results()):
            want = oracle.residual_mask(np.ascontiguousarray(frames[f][..., 0]), np.ascontiguousarray(frames[f + 1][..., 0]), 0.0).reshape(-1)
            assert np.array_equal(unpack(r["mask"], n), want), f
            bm, wit, p, _, _ = oracle.compress(want)
            if len(wit) == 0:
                assert r["l"] == 0 and r["witness_bits"] == 0, f
                continue
            coded += 1
            k, l = oracle.optimal_params(n, p)
            assert (r["k"], r["l"]) == (k, l), f
            assert np.array_equal(unpack(r["filter"], l), bm), f
            assert r["witness_bits"] == len(wit) and np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8)), f
            dec = eng.decode(n, [P.filter_params(k, l)], [r["filter"]], [r["witness"]])
            assert np.array_equal(unpack(dec[0], n), want), f
        assert coded == 4
    coder.close()
    eng.close()
    ctx.close()


def test_shared_hash_table_across_contexts(oracle):
    """Contexts of one process share the pixel-index hash table of a (device, frame size, seeds) triple: the first one builds
    it, the others wait for it, it outlives the context that built it, and other seeds or sizes get their own.  Every
    context must still reproduce the oracle (a stale or half-built table would corrupt the filters)."""
    from new_bloom_filter_repo_amd.gop import GopCoder
    from new_bloom_filter_repo_amd.synthetic import next_frame
    rng = np.random.default_rng(99)
    W, H, n = 640, 360, 640 * 360
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)]
    for p in (0.0889, 0.05, 0.2):
        frames.append(next_frame(rng, frames[-1], p))
    frames = np.stack(frames)
    small = np.ascontiguousarray(frames[:, :300, :500])

    def check(coder, fr, seeds):
        hh, ww = fr.shape[1], fr.shape[2]
        coder.encode()
        for f, r in enumerate(coder.results()):
            want = oracle.residual_mask(np.ascontiguousarray(fr[f][..., 0]), np.ascontiguousarray(fr[f + 1][..., 0]), 0.0).reshape(-1)
            bm, wit, p, _, _ = oracle.compress(want, seeds=seeds)
            k, l = oracle.optimal_params(hh * ww, p)
            assert (r["k"], r["l"]) == (k, l), f
            assert np.array_equal(unpack(r["filter"], l), bm), f
            assert r["witness_bits"] == len(wit) and np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8)), f

    ctxs = [nat.Context(0) for _ in range(3)]
    coders = [GopCoder(c, W, H, len(frames)) for c in ctxs]
    for c in coders:
        c.load_frames(frames)
    for c in coders:                                           # builder first, then two that take the same table
        check(c, frames, P.SEEDS_VIDEO)
    coders[0].close(); ctxs[0].close()                         # the builder goes away; the table must not
    check(coders[1], frames, P.SEEDS_VIDEO)
    other = GopCoder(ctxs[1], W, H, len(frames), seeds=P.SEEDS_BLOOM_COMPRESS)      # same context, other seeds: another table
    other.load_frames(frames)
    check(other, frames, P.SEEDS_BLOOM_COMPRESS)
    check(coders[1], frames, P.SEEDS_VIDEO)                    # ... and back
    sm = GopCoder(ctxs[2], 500, 300, len(small))               # other frame size on a context that held the 640x360 table
    sm.load_frames(small)
    check(sm, small, P.SEEDS_VIDEO)
    check(coders[2], frames, P.SEEDS_VIDEO)
    for c in (other, sm, coders[1], coders[2]):
        c.close()
    ctxs[1].close(); ctxs[2].close()


def test_shared_hash_table_from_concurrent_threads(oracle):
    """Four host threads, each with its own context and stream, start coding the same geometry at the same moment (ctypes
    releases the GIL inside the library): exactly one of them builds the shared pixel-index hash table, the others must
    wait for it on the device -- every thread's filters and witnesses are checked against the oracle, several rounds, with
    the contexts torn down in between so that the table is also freed and rebuilt."""
    import threading
    from new_bloom_filter_repo_amd.gop import GopCoder
    from new_bloom_filter_repo_amd.synthetic import next_frame
    rng = np.random.default_rng(321)
    W, H, n = 704, 396, 704 * 396
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)]
    for p in (0.0889, 0.12, 0.05):
        frames.append(next_frame(rng, frames[-1], p))
    frames = np.stack(frames)
    want = []
    for f in range(len(frames) - 1):
        mask = oracle.residual_mask(np.ascontiguousarray(frames[f][..., 0]), np.ascontiguousarray(frames[f + 1][..., 0]), 0.0).reshape(-1)
        bm, wit, p, _, _ = oracle.compress(mask)
        want.append((oracle.optimal_params(n, p), bm, np.array(wit, dtype=np.uint8)))
    errors = []

    def worker(barrier, tid):
        try:
            ctx = nat.Context(0)
            coder = GopCoder(ctx, W, H, len(frames))
            coder.load_frames(frames)
            barrier.wait()
            for _ in range(3):
                coder.encode()
                for f, r in enumerate(coder.results()):
                    (k, l), bm, wit = want[f]
                    assert (r["k"], r["l"]) == (k, l), (tid, f)
                    assert np.array_equal(unpack(r["filter"], l), bm), (tid, f)
                    assert r["witness_bits"] == len(wit) and np.array_equal(unpack(r["witness"], len(wit)), wit), (tid, f)
            coder.close()
            ctx.close()
        except BaseException as e:                             # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    for _ in range(3):                                          # table built, shared, freed; three times
        barrier = threading.Barrier(4)
        threads = [threading.Thread(target=worker, args=(barrier, t)) for t in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors


@pytest.mark.parametrize("force", [0, 8, 16 << 16], ids=["default", "barrett_only", "tiles_4KiB"])
def test_long_gop_129_inter_frames_vs_oracle(oracle, force):
    """A GOP of 130 frames = 129 inter-frames: the library splits it into batches of MAX_BATCH = 128 + 1.  The first batch fills
    every per-batch table to the brim -- 128 thresholds for the activation ranks (binary search from step 64, ranks up to 128),
    128 geometries in the kernel arguments, 2 insert slices per frame -- with repeated densities (equal thresholds) and unchanged
    frames (m = 0) in it.  960x544 luma frames keep every coded filter inside the FP64 kernels' range (m >= 2^15, asserted),
    from k* = 0.17 (floor 0) to k* = 5.6 (floor 5: the runtime-k pass)."""
    from new_bloom_filter_repo_amd.gop import GopCoder
    from new_bloom_filter_repo_amd.synthetic import next_frame
    rng = np.random.default_rng(1300)
    W, H, F = 960, 544, 130
    n = W * H
    ps = [0.0889, 0.0889, 0.12, 0.0, 0.05, 0.3, 0.2, 0.01, 0.0889, 0.15]
    frames = [rng.integers(0, 256, (H, W), dtype=np.uint8)]
    for t in range(F - 1):
        p = ps[t % len(ps)]
        frames.append(next_frame(rng, frames[-1][..., None].repeat(3, axis=2), p)[..., 0].copy() if p else frames[-1].copy())
    frames = np.stack(frames)
    ctx = nat.Context(0)
    ctx.force_generic(force)
    eng = BloomEngine(ctx)
    coder = GopCoder(ctx, W, H, F, channels=1)
    coder.load_frames(frames)
    coder.encode()
    coded = 0
    for f, r in enumerate(coder.results()):
        want = oracle.residual_mask(frames[f], frames[f + 1], 0.0).reshape(-1)
        assert np.array_equal(unpack(r["mask"], n), want), f
        bm, wit, p, _, _ = oracle.compress(want)
        if len(wit) == 0:
            assert r["l"] == 0 and r["witness_bits"] == 0, f
            continue
        coded += 1
        k, l = oracle.optimal_params(n, p)
        assert (r["k"], r["l"]) == (k, l), f
        assert np.array_equal(unpack(r["filter"], l), bm), f
        assert r["witness_bits"] == len(wit) and np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8)), f
        if f % 16 == 0:
            dec = eng.decode(n, [P.filter_params(k, l)], [r["filter"]], [r["witness"]])
            assert np.array_equal(unpack(dec[0], n), want), f
    assert coded >= 100
    assert min(r["l"] for r in coder.results() if r["l"]) >= 1 << 15       # else the batch would have taken the Barrett kernels
    coder.close()
    eng.close()
    ctx.close()


@pytest.mark.parametrize("force", [0, 32 << 16, 8, 1 << 14 | 32 << 16], ids=["default", "tiles_8KiB", "barrett_only_no_split", "tiles_hashed_records"])
def test_mixed_batch_is_split_over_the_kernel_families(oracle, force):
    """A GOP whose filters straddle the FP64 kernels' range (2^15 <= m < 2^23): busy frames next to nearly static ones (a few
    hundred changed pixels: m of a few thousand bits) and unchanged ones.  rbf_encode_gop codes such a batch in two passes --
    Barrett kernels for the small filters, FP64 kernels for the rest, one compaction -- and every frame must still match the
    oracle, whichever frame comes first, also when a knob rules the split out."""
    from new_bloom_filter_repo_amd.gop import GopCoder
    from new_bloom_filter_repo_amd.synthetic import next_frame
    rng = np.random.default_rng(5150)
    W, H, n = 640, 360, 640 * 360
    for order in ((0.0889, 0.002, 0.05, 0.0, 0.0005, 0.12, 0.001), (0.0008, 0.0889, 0.0, 0.2, 0.003)):
        frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)]
        for p in order:
            frames.append(next_frame(rng, frames[-1], p) if p else frames[-1].copy())
        frames = np.stack(frames)
        ctx = nat.Context(0)
        ctx.force_generic(force)
        eng = BloomEngine(ctx)
        coder = GopCoder(ctx, W, H, len(frames))
        coder.load_frames(frames)
        for _ in range(2):
            coder.encode()
            small = big = 0
            for f, r in enumerate(coder.results()):
                want = oracle.residual_mask(np.ascontiguousarray(frames[f][..., 0]), np.ascontiguousarray(frames[f + 1][..., 0]), 0.0).reshape(-1)
                assert np.array_equal(unpack(r["mask"], n), want), f
                bm, wit, p, _, _ = oracle.compress(want)
                if len(wit) == 0:
                    assert r["l"] == 0 and r["witness_bits"] == 0, f
                    continue
                k, l = oracle.optimal_params(n, p)
                small += l < (1 << 15)
                big += l >= (1 << 15)
                assert (r["k"], r["l"]) == (k, l), f
                assert np.array_equal(unpack(r["filter"], l), bm), f
                assert r["witness_bits"] == len(wit) and np.array_equal(unpack(r["witness"], len(wit)), np.array(wit, dtype=np.uint8)), f
                dec = eng.decode(n, [P.filter_params(k, l)], [r["filter"]], [r["witness"]])
                assert np.array_equal(unpack(dec[0], n), want), f
            assert small >= 2 and big >= 2
            # ... and ONE decode call over all coded frames (the decoder splits its query the same way)
            res = [r for r in coder.results() if r["l"]]
            masks = [np.unpackbits(r["mask"])[:n] for r in res]
            dec = eng.decode(n, [P.filter_params(r["k"], r["l"]) for r in res], [r["filter"] for r in res], [r["witness"] for r in res])
            for j, want in enumerate(masks):
                assert np.array_equal(unpack(dec[j], n), want), j
        coder.close()
        eng.close()
        ctx.close()


def test_key_length_boundary_at_ten_million(eng, oracle):
    """Indices around 10^7 (7- and 8-character keys in the same wave) through every kernel family: the LDS kernels'
    fixed-length and shared-prefix hash paths must hand over to the generic one exactly at the boundary."""
    import ctypes
    n = 10_000_640                                            # the last 640 positions have 8-character keys
    mask = np.zeros(n, dtype=np.uint8)
    rng = np.random.default_rng(77)
    mask[rng.integers(0, n, 18000)] = 1
    mask[9_999_990:10_000_012] = [1, 0] * 11                   # straddle the boundary
    k, l = 2.3, 400_003                                       # a filter that fits LDS (50 KB), not the optimal one for this n
    pl = [P.filter_params(k, l)]
    eng.upload_masks(np.packbits(mask)[None, :], n)
    r = eng.encode(n, pl)[0]
    bit_array = np.zeros(l, dtype=np.uint8)
    witness = np.zeros(n, dtype=np.uint8)
    seeds = (ctypes.c_uint64 * 3)(*P.SEEDS_VIDEO)
    w = oracle.lib().orc_compress(mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n, l, k, seeds,
                                  bit_array.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), witness.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    assert np.array_equal(unpack(r["filter"], l), bit_array)
    assert r["witness_bits"] == w and np.array_equal(unpack(r["witness"], w), witness[:w])
    dec = eng.decode(n, pl, [r["filter"]], [r["witness"]])
    assert np.array_equal(unpack(dec[0], n), mask)


def test_key_length_boundary_at_hundred_million(oracle):
    """8- and 9-character keys (indices around 10^8) through the default LDS kernels with a filter that fits LDS."""
    import ctypes
    n = 100_000_640
    mask = np.zeros(n, dtype=np.uint8)
    rng = np.random.default_rng(78)
    mask[rng.integers(0, n, 30000)] = 1
    mask[99_999_985:100_000_015] = [1, 1, 0] * 10
    k, l = 1.7, 611_158
    pl = [P.filter_params(k, l)]
    ctx = nat.Context(0)
    e = BloomEngine(ctx)
    e.upload_masks(np.packbits(mask)[None, :], n)
    r = e.encode(n, pl)[0]
    bit_array = np.zeros(l, dtype=np.uint8)
    witness = np.zeros(n, dtype=np.uint8)
    seeds = (ctypes.c_uint64 * 3)(*P.SEEDS_VIDEO)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    w = oracle.lib().orc_compress(mask.ctypes.data_as(u8p), n, l, k, seeds, bit_array.ctypes.data_as(u8p), witness.ctypes.data_as(u8p))
    assert np.array_equal(unpack(r["filter"], l), bit_array)
    assert r["witness_bits"] == w and np.array_equal(unpack(r["witness"], w), witness[:w])
    e.close()
    ctx.close()
