"""CPU-only: host-side logic of the product package (no GPU compute calls):
parameter math vs the reference fixtures, the C ABI's host helpers vs the Python ones, the C-ABI
surface (every symbol include/rbf.h declares is exported), keyframe codec / container / verifiers."""
import ctypes
import math
import os
import re
import struct

import numpy as np
import pytest

from conftest import REPO, load_json, load_npz
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd import params as P
from new_bloom_filter_repo_amd.synthetic import make_gop, P_KSTAR_2_3


def test_abi_header_matches_binding_and_library():
    hdr = open(os.path.join(REPO, "include", "rbf.h")).read()
    declared = set(re.findall(r"\b(rbf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(nat.exported_symbols()), declared ^ set(nat.exported_symbols())
    lib = nat.lib()                       # loads librbf_hip.so (needs no GPU)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rbf_version() == 4
    assert isinstance(lib.rbf_last_error(), bytes)
    # same number of parameters in the header and in the ctypes prototypes
    clean = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name, args in re.findall(r"\b(rbf_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", clean):
        count = 0 if args.strip() == "void" else args.count(",") + 1
        assert count == len(nat._PROTOS[name][1]), (name, count, len(nat._PROTOS[name][1]))


def test_no_cpu_fallback_message(monkeypatch):
    monkeypatch.setattr(nat, "LIB_PATH", "/nonexistent/librbf_hip.so")
    monkeypatch.setattr(nat, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "new_bloom_filter_repo_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.lower(), f


def test_params_match_reference_fixture():
    for r in load_json("g2_params.json")["rows"]:
        p = np.uint64(r["ones"]) / r["n"]
        k, l = P.optimal_params(r["n"], p)
        assert float(k).hex() == r["k_hex"] and int(l) == r["l"]


def test_c_host_math_matches_python():
    lib = nat.lib()
    for r in load_json("g2_params.json")["rows"]:
        k, l = ctypes.c_double(), ctypes.c_uint64()
        assert lib.rbf_optimal_params(r["n"], r["ones"], ctypes.byref(k), ctypes.byref(l)) == 0
        assert float(k.value).hex() == r["k_hex"] and l.value == r["l"], r
        if r["l"]:
            fk, t = ctypes.c_uint32(), ctypes.c_uint64()
            assert lib.rbf_activation_threshold(k.value, ctypes.byref(fk), ctypes.byref(t)) == 0
            assert (fk.value, t.value) == P.activation_threshold(k.value)
    # plan_batch incl. passthrough decisions
    n = 57600
    ones = [0, 3, 62, 1696, 5261, 17894, 18692, 18693, 23048, 57600]
    arr = (ctypes.c_uint64 * len(ones))(*ones)
    par = (nat.FilterParams * len(ones))()
    ks = (ctypes.c_double * len(ones))()
    assert lib.rbf_plan_batch(n, arr, len(ones), 1, par, ks) == 0
    for i, c in enumerate(ones):
        p = np.uint64(c) / n
        k, l = (0, 0) if p >= P.P_STAR else P.optimal_params(n, p)
        if l == 0 or l >= n:
            assert par[i].m == 0
        else:
            assert (par[i].m, par[i].floor_k, par[i].threshold) == P.filter_params(k, l) and ks[i] == k


def test_activation_threshold_equals_float_predicate(oracle):
    g = load_json("g2b_activation.json")
    for row in g["activation"]:
        k = float.fromhex(row["k_hex"])
        fk, t = P.activation_threshold(k)
        assert fk == math.floor(k)
        for i, want in zip(row["indices"], row["activated"]):
            assert int(oracle.hash_index(i, row["seed"]) < t) == want
    for k in (0.1, 0.5, 1.0, 2.0, 2.3, 2.9999999999999996, 12.999999999999998, 5e-17 + 3):
        fk, t = P.activation_threshold(k)
        frac = k - math.floor(k)
        for h in (t - 1, t, t + 1, 0, (1 << 64) - 1):
            if 0 <= h < (1 << 64):
                assert (h / (2 ** 64 - 1) < frac) == (h < t), (k, h)
        fk2, t2 = ctypes.c_uint32(), ctypes.c_uint64()
        assert nat.lib().rbf_activation_threshold(k, ctypes.byref(fk2), ctypes.byref(t2)) == 0
        assert (fk2.value, t2.value) == (fk, t)


def test_argument_errors_without_gpu():
    lib = nat.lib()
    k, l = ctypes.c_double(), ctypes.c_uint64()
    assert lib.rbf_optimal_params(0, 0, ctypes.byref(k), ctypes.byref(l)) == nat.RBF_EINVAL
    assert b"n > 0" in lib.rbf_last_error()
    assert lib.rbf_optimal_params(10, 11, ctypes.byref(k), ctypes.byref(l)) == nat.RBF_EINVAL
    fk, t = ctypes.c_uint32(), ctypes.c_uint64()
    assert lib.rbf_activation_threshold(float("nan"), ctypes.byref(fk), ctypes.byref(t)) == nat.RBF_ERANGE
    assert lib.rbf_ctx_sync(None) == nat.RBF_EINVAL


def test_keyframe_codec_matches_reference_bytes():
    from new_bloom_filter_repo_amd.frame_codec import FixedVideoCompressor
    z, meta = load_npz("g10_keyframes.npz"), load_json("g10_keyframes.json")
    fx = FixedVideoCompressor(verbose=False)
    frames = make_gop(meta["seed"], meta["W"], meta["H"], 3, p=0.2)
    assert fx.compress_frame(frames[0]) == z["plain"].tobytes()
    assert fx.compress_frame(fx.add_yuv_info_to_frame(frames[1])) == z["wrapped"].tobytes()
    assert fx.compress_frame(frames[2][:, :, 0].copy()) == z["gray"].tobytes()
    u16 = make_gop(10001, 24, 16, 1, dtype=np.uint16)[0]
    assert fx.compress_frame(u16) == z["u16"].tobytes()
    for blob, want in ((z["plain"], frames[0]), (z["wrapped"], frames[1]), (z["gray"], frames[2][:, :, 0]), (z["u16"], u16)):
        got = fx.decompress_frame(blob.tobytes())
        assert got.dtype == want.dtype and np.array_equal(got, want)


def test_all_keyframe_container_is_reference_compatible():
    from new_bloom_filter_repo_amd.video_compressor import ImprovedVideoCompressor
    z, meta = load_npz("g10_keyframes.npz"), load_json("g10_keyframes.json")
    frames = make_gop(meta["seed"], meta["W"], meta["H"], 3, p=0.2)
    comp = ImprovedVideoCompressor(keyframe_interval=1, verbose=False)      # every frame a keyframe: no GPU involved
    res = comp.compress_video([f.copy() for f in frames], None, input_color_space="YUV")
    assert sorted(res) == meta["result_keys"]
    for key, val in meta["stable"].items():
        assert res[key] == val, key
    assert comp._container(comp.last_compressed_frames) == z["container"].tobytes()
    assert comp._container_size(comp.last_compressed_frames) == len(z["container"].tobytes())
    mixed = [(1, b"abc"), (2, b"defgh"), (2, b"")]
    assert comp._container_size(mixed) == len(comp._container(mixed))
    # and the reference-written container decodes
    recs = comp._parse_container(z["container"].tobytes())
    dec = comp.decompress_video(compressed_frames=recs)
    assert all(np.array_equal(a, b) for a, b in zip(frames, dec))
    with pytest.raises(ValueError):
        comp.compress_video([], None)
    with pytest.raises(ValueError):
        comp._parse_container(b"XXXX\0\0\0\0")
    with pytest.raises(ValueError):
        comp.decompress_video()


def test_verifiers_match_reference_dicts():
    from new_bloom_filter_repo_amd.frame_codec import YUVFrame
    from new_bloom_filter_repo_amd.verify import verify_bit_exact, verify_lossless
    g = load_json("g8_verify.json")
    frames = make_gop(g["seed"], g["W"], g["H"], g["nframes"], p=g["p"])
    same = [f.copy() for f in frames]
    off = [f.copy() for f in frames]
    off[1][3, 4, 1] ^= 1
    off[2][0, 0, 0] = (int(off[2][0, 0, 0]) + 7) % 256
    off[2][5, 6, 2] = (int(off[2][5, 6, 2]) + 100) % 256
    wrap = lambda fs: [YUVFrame(f) for f in fs]
    for name, (a, b) in {"identical": (frames, same), "pixels_off": (frames, off), "count_mismatch": (frames, same[:2])}.items():
        for aa, bb in ((a, b), (wrap(a), wrap(b))):          # plain ndarrays and wrappers give the same dicts
            got = verify_lossless(aa, bb)
            for key, val in g["results"][name]["verify_lossless"].items():
                if val == "inf":
                    assert got[key] == float("inf")
                else:
                    assert got[key] == val, (name, key)
            gb = verify_bit_exact(aa, bb)
            for key, val in g["results"][name]["verify_bit_exact"].items():
                if key == "diff_details":
                    assert [d.get("differences_found") for d in gb[key]] == [d.get("differences_found") for d in val]
                else:
                    assert gb[key] == val, (name, key)


def test_synthetic_density_gives_kstar_2_3():
    k, _ = P.optimal_params(10 ** 9, P_KSTAR_2_3)
    assert abs(k - 2.3) < 1e-12


def test_bloom_compress_containers_parse_reference_blobs():
    """CPU part of the bloom_compress.py front-ends: header parsing / binarisation helpers."""
    from new_bloom_filter_repo_amd.bloom_compress import BloomFilterCompressor as BC
    z, meta = load_npz("g11_bloom_compress.npz"), load_json("g11_bloom_compress.json")
    bm, wit, p, n, k, shape = BC._unpack_compressed_data(BC, z["img_blob"].tobytes())
    assert shape == z["img"].shape and n == meta["img"]["n"] == shape[0] * shape[1]
    assert BC._pack_compressed_data(BC, bm, wit, p, n, k, shape) == z["img_blob"].tobytes()
    assert int(BC._binarize_image(z["img"], 127).sum()) == meta["img"]["ones"]
    bm, wit, p, n, k, tl, bd = BC._unpack_text_data(BC, z["sparse_text_blob"].tobytes())
    assert tl == len(meta["sparse_text"]["text"]) and bd == 8 and n == 8 * tl
    assert BC._pack_text_data(BC, bm, wit, p, n, k, tl, bd) == z["sparse_text_blob"].tobytes()
    txt = meta["text"]["text"]
    assert BC._debinarize_text(BC._binarize_text(txt, 8), 8) == txt


def test_header_is_plain_c():
    """include/rbf.h is the drop-in boundary: it must compile as C99 and as C++ on its own."""
    import subprocess
    hdr = os.path.join(REPO, "include", "rbf.h")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                ["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_c_consumer_links_and_agrees(tmp_path):
    """A plain C program built against include/rbf.h and linked with librbf_hip.so (what a cgo / JNI / FFI binding
    does) gets the same filter geometry as params.py and the fixtures."""
    import subprocess
    pkg = os.path.join(REPO, "new_bloom_filter_repo_amd")
    exe = str(tmp_path / "abi_smoke")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "c", "abi_smoke.c"),
                        "-L", pkg, "-lrbf_hip", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cases = [(2073600, 184322), (57600, 5120), (8294400, 737288), (2073600, 100), (2073600, 700000), (1000, 0), (57600, 17000)]
    out = subprocess.run([exe] + [str(v) for c in cases for v in c], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "version 4" and lines[-1].startswith("error ") and len(lines[-1]) > 8
    assert lines[-2] == "record_max %d" % (8 * (4 + 8 * 29) + 29 * 2 * ((2073600 + 63) // 64 * 8))
    for (n, ones), line in zip(cases, lines[1:]):
        f = line.split()
        k, l = P.optimal_params(n, np.uint64(ones) / n)
        assert (int(f[1]), int(f[2])) == (n, ones) and float.fromhex(f[3]) == k and int(f[4]) == l, line
        guard = l == 0 or l >= n or np.uint64(ones) / n >= P.P_STAR
        if k > 0:
            m, fk, thr = P.filter_params(k, l)
            assert (int(f[5]), int(f[6])) == (fk, thr), line
            assert [int(x) for x in f[8:11]] == ([0, 0, 0] if guard else [m, fk, thr]), line
        else:
            assert [int(x) for x in f[8:11]] == [0, 0, 0], line


def test_wire_record_layout_against_reference_blob():
    """frame_codec.parse_record / build_record (no GPU): the reference's own _compress_frame_differences blob
    (fixture G7) parses into the expected fields and rebuilds to the same bytes; the float64-k variant differs
    only in the k field."""
    import zlib
    from new_bloom_filter_repo_amd.frame_codec import build_record, parse_record
    z = load_npz("g7_g9_frame_codec.npz")
    blob = z["blob"].tobytes()
    d = parse_record("reference", blob)
    mask = z["mask"].reshape(-1)
    n = mask.size
    assert d["n"] == n and abs(d["p"] - mask.sum() / n) < 1e-7 and d["value_count"] == len(z["values"])
    assert np.array_equal(np.frombuffer(zlib.decompress(d["values_z"]), dtype=np.uint8), z["values"])
    k, l = P.optimal_params(n, np.uint64(mask.sum()) / n)
    assert d["bitmap_bits"] == l and np.float32(k) == np.float32(d["k"]) and len(d["bitmap"]) == (l + 7) // 8
    assert len(d["witness"]) == (d["witness_bits"] + 7) // 8
    again = build_record("reference", d["p"], d["n"], d["k"], d["bitmap_bits"], d["bitmap"], d["witness_bits"], d["witness"],
                         d["value_count"], d["values_z"])
    assert again == blob
    wide = build_record("f64", d["p"], d["n"], k, d["bitmap_bits"], d["bitmap"], d["witness_bits"], d["witness"], d["value_count"], d["values_z"])
    assert len(wide) == len(blob) + 4 and wide[:8] == blob[:8] and wide[16:] == blob[12:]
    assert parse_record("f64", wide)["k"] == k
    with pytest.raises(ValueError):
        parse_record("reference", blob[:-5])


def test_fp64_reduction_is_exact(tmp_path):
    """mod_m_f64 (the query / insert kernels' h mod m through one v_fma_f64 + one v_mad_u32_u24) restated step by step
    in C against 64-bit integer arithmetic: random h, h next to multiples of m over the whole quotient range, the
    extremes of the 64-bit range, the whole span of eligible m."""
    import subprocess
    exe = str(tmp_path / "mod_f64_check")
    r = subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-Werror", os.path.join(REPO, "tests", "c", "mod_f64_check.c"), "-lm", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe, "3000"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout[-2000:]


def test_activation_rank_is_equivalent_to_the_threshold_compare():
    """k_query_r64 keeps, per pixel, rank = #{coded thresholds t_j <= h_act} (one byte) instead of the 64-bit activation hash,
    and tests rank <= c_f with c_f = #{t_j < T_f} (rbf_kernels_r64.h; the host side is rank_table() in rbf_api.hip).  Restated
    here in Python integers: for any multiset of thresholds -- repeated values, 0, 2^64 - 1 -- and any hash,
    (h_act < T_f) == (rank <= c_f) for every frame f."""
    import bisect
    import random
    rnd = random.Random(2026)
    special = [0, 1, 2 ** 63, 2 ** 64 - 2, 2 ** 64 - 1]
    for _ in range(300):
        nfr = rnd.randint(1, 128)
        ts = [rnd.choice(special) if rnd.random() < 0.15 else rnd.getrandbits(64) for _ in range(nfr)]
        if rnd.random() < 0.5:                                   # repeated thresholds
            ts = [ts[rnd.randrange(max(1, nfr // 3))] for _ in range(nfr)]
        srt = sorted(ts)
        c = [bisect.bisect_left(srt, t) for t in ts]             # std::lower_bound in rank_table()
        hashes = [rnd.getrandbits(64) for _ in range(40)] + special + [t for t in ts[:8]] + [max(0, t - 1) for t in ts[:8]]
        for h in hashes:
            rank = bisect.bisect_right(srt, h)                   # the kernel's branch-free upper_bound
            assert rank <= 128
            for f in range(nfr):
                assert (h < ts[f]) == (rank <= c[f]), (h, ts[f], rank, c[f])


def test_graft_entry_build_check_follows_the_header():
    """__graft_entry__.build() compares the library's ABI version with include/rbf.h (round 4: a literal 2 survived the bump to 3)."""
    src = open(os.path.join(REPO, "__graft_entry__.py"), encoding="utf-8").read()
    assert "RBF_ABI_VERSION" in src and "rbf_version() ==" not in src


def test_yuv_frame_planes_are_lazy_copies_and_blocks_are_zero_copy_views():
    """Host-side helpers of the plugin surface (no GPU): YUVFrame.yuv_info behaves like the reference's dict of plane copies
    (fixed_video_compressor.py:292-296) but copies a plane on first use; _as_block hands contiguous frames to the upload without a copy."""
    from new_bloom_filter_repo_amd.frame_codec import FixedVideoCompressor, YUVFrame
    from new_bloom_filter_repo_amd.video_compressor import _as_block
    clip = np.random.default_rng(5).integers(0, 256, (4, 6, 8, 3), dtype=np.uint8)
    f = YUVFrame(clip[1])
    info = f.yuv_info
    assert hasattr(f, "yuv_info") and info.get("format", "x") == "YUV444" and "u_plane" in info and info.get("nope") is None
    assert not any(k in dict.keys(info) for k in ("y_plane", "u_plane", "v_plane"))            # nothing copied yet
    u = info["u_plane"]
    assert u.flags.c_contiguous and np.array_equal(u, clip[1][:, :, 1]) and not np.shares_memory(u, clip) and info["u_plane"] is u
    with pytest.raises(KeyError):
        info["w_plane"]
    fx = FixedVideoCompressor()
    back = fx.decompress_frame(fx.compress_frame(f))                                        # the keyframe record carries the three planes
    assert np.array_equal(back.data, clip[1]) and np.array_equal(back.yuv_info["v_plane"], clip[1][:, :, 2])
    blk = _as_block([clip[i] for i in range(4)])
    assert np.shares_memory(blk, clip) and np.array_equal(blk, clip)
    blk = _as_block([clip[i] for i in (0, 2)])
    assert not np.shares_memory(blk, clip) and np.array_equal(blk, clip[[0, 2]])
    c16 = clip.astype(np.uint16) * 257
    blk = _as_block([c16[i] for i in range(1, 4)])
    assert blk.dtype == np.uint16 and np.shares_memory(blk, c16) and np.array_equal(blk, c16[1:])


def test_filter_stride_min_covers_every_plan():
    """rbf_encode_gop_begin / rbf_encode_runs_begin refuse a filter stride below rbf_filter_stride_min(n) BEFORE the filters are planned:
    the bound has to cover whatever rbf_plan_batch (improved_video_compressor.py:161-196, :211-225) can produce for n pixels."""
    import ctypes
    L = nat.lib()
    for n in (1, 63, 64, 1000, 57600, 230400, 2073600, 8294400, 33177600, 2 ** 32 - 1):
        smin = int(L.rbf_filter_stride_min(n))
        assert smin % 8 == 0 and smin >= 8
        cnt = 4001 if n > 4001 else n + 1
        ones = (ctypes.c_uint64 * cnt)(*sorted({min(n, int(round(i * n / (cnt - 1)))) for i in range(cnt)} | {min(n, int(0.13183 * n)), min(n, int(0.13183 * n) + 1)})[:cnt])
        cnt = len(ones)
        params = (nat.FilterParams * cnt)()
        nat.check(L.rbf_plan_batch(n, ones, cnt, 1, params, None))
        worst = max(int(p.m) for p in params)
        assert (worst + 63) // 64 * 8 <= smin, (n, worst, smin)
        if n >= 57600:
            assert smin * 8 <= 0.32 * n + 128            # ... without asking for more than GopCoder.strides gives
