import sys, numpy as np
sys.path.insert(0, '.')
from new_bloom_filter_repo_amd import _native as nat, params as P
from new_bloom_filter_repo_amd.engine import BloomEngine
from new_bloom_filter_repo_amd.synthetic import make_mask
n = 1920*1080
x = make_mask(7, n, 0.0889)
def run(flag):
    ctx = nat.Context(0); ctx.force_generic(flag); e = BloomEngine(ctx)
    p = np.uint64(x.sum())/n; k,l = P.optimal_params(n,p); pl=[P.filter_params(k,l)]
    e.upload_masks(np.packbits(x)[None,:], n); r = e.encode(n, pl)[0]
    return r
a = run(0)
for tile_kib in (32, 64, 8):
    b = run((tile_kib*4) << 16)
    print("tile", tile_kib, "filter equal", np.array_equal(a["filter"], b["filter"]), "wbits", a["witness_bits"], b["witness_bits"], "witness equal", np.array_equal(a["witness"], b["witness"]))
    if a["witness_bits"] != b["witness_bits"]:
        pass
