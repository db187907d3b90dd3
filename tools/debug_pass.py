import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from new_bloom_filter_repo_amd import _native as nat, params as P
from new_bloom_filter_repo_amd.engine import BloomEngine
from new_bloom_filter_repo_amd.synthetic import make_mask
from oracle import oracle as orc
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = nat.Context(0); ctx.force_generic(mode)
eng = BloomEngine(ctx)
z = np.load('/root/repo/tests/golden/g3_320x180.npz')
n = 57600
for case in ("c0", "c1", "c2"):
    m = np.unpackbits(z[case + "_mask"])[:n]
    k, l = P.optimal_params(n, np.uint64(m.sum()) / n)
    pl = [P.filter_params(k, l)]
    eng.upload_masks(np.packbits(m)[None, :], n)
    r = eng.encode(n, pl)[0]
    f = orc.RationalFilter(l, k)
    for i in np.flatnonzero(m): f.add_index(int(i))
    feq = np.array_equal(np.unpackbits(r["filter"])[:l], f.bit_array)
    want = np.array([f.check_index(i) for i in range(n)], dtype=np.uint8)
    ones = np.packbits(np.ones(n, dtype=np.uint8))
    dec = eng.decode(n, pl, [r["filter"]], [ones])
    got = np.unpackbits(dec[0])[:n]
    bad = np.flatnonzero(got != want)
    print(case, "k", k, "l", l, "words", (l + 31) // 32, "filter eq", feq, "witness_bits", r["witness_bits"], "want", int(want.sum()),
          "decode-pass", int(got.sum()), "mismatch", len(bad))
    # which filter words do the failing pixels probe?
    words = []
    fk = int(np.floor(k)); T = pl[0][2]
    for i in bad[:200]:
        h1 = orc.hash_index(int(i), P.SEEDS_VIDEO[0]); h2 = orc.hash_index(int(i), P.SEEDS_VIDEO[1]); ha = orc.hash_index(int(i), 999)
        pos = [(h1 + j * h2) % l for j in range(fk + (1 if ha < T else 0))]
        words.append([p // 32 for p in pos])
    if len(bad):
        flat = np.array([w for ws in words for w in ws])
        print("   probe-word histogram of failing pixels (top):", np.bincount(flat).argsort()[::-1][:8], np.sort(np.bincount(flat))[::-1][:8])
        print("   bad lanes", np.flatnonzero(np.bincount(bad % 64, minlength=64)), "bad it", np.bincount((bad // 64) % 8, minlength=8))
