"""Host-side float64 parameter math of the Bloom compressor.

This stays on the host, in CPython's `math`, on purpose: (k, l) come from libm log/log2/pow in
float64 and the filter geometry must match the reference to the last ulp
(improved_video_compressor.py:161-196).  The device only ever sees integers: the filter length
m = l, floor(k*) and the activation threshold T.
"""
import math

P_STAR = 0.32453                               # improved_video_compressor.py:150
SEEDS_VIDEO = (0x12345678, 0x87654321, 999)    # improved_video_compressor.py:62-63,94
SEEDS_BLOOM_COMPRESS = (0, 1, 999)             # bloom_compress.py:163-164,195
_D = (1 << 64) - 1


def string_filter_seeds(k_star):
    """Seeds of rational_bloom_filter.RationalBloomFilter: (0, 1, ceil(k*)) (:100-101,134)."""
    return (0, 1, math.ceil(k_star))


def optimal_params(n, p):
    """(k, l) for a length-n vector of density p; (0, 0) when Bloom coding does not apply."""
    if p <= 0.0001 or p >= P_STAR:
        return 0, 0
    ln2 = math.log(2)
    k = math.log2((1 - p) * (ln2 ** 2) / p)
    if math.isnan(k) or k <= 0:
        return 0, 0
    l = int(p * n * k * (1 / ln2))
    return max(0.1, k), max(1, l)


def activation_threshold(k_star):
    """(floor_k, T): the extra hash fires iff XXH64(key, act_seed) < T.

    The reference tests `h / (2**64 - 1) < k* - floor(k*)` in float64, where int/int is the
    correctly rounded quotient (improved_video_compressor.py:94-97).  That predicate is monotone
    in h, so it equals `h < T` with T the smallest h whose quotient reaches the fraction."""
    floor_k = math.floor(k_star)
    frac = k_star - floor_k
    if not frac > 0.0:
        return floor_k, 0
    lo, hi = 0, _D                      # invariant: hi / D >= frac (1.0 >= frac), answer in [lo, hi]
    while lo < hi:
        mid = (lo + hi) // 2
        if mid / _D < frac:
            lo = mid + 1
        else:
            hi = mid
    return floor_k, lo


def filter_params(k_star, l):
    """(m, floor_k, T) triple the C ABI takes for one filter."""
    floor_k, t = activation_threshold(k_star)
    return int(l), int(floor_k), int(t)
