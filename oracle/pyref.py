"""TEST INFRASTRUCTURE ONLY -- literal pure-Python restatement (small cases).

Follows the reference line by line with Python big ints and a pure-Python XXH64,
so that it has no dependency on the `xxhash` wheel nor on the C oracle.  Used to
cross-check oracle/rbf_oracle.c and to run where only a handful of indices are
needed.  Never imported by the product package.

Reference anchors (/root/reference):
  improved_video_compressor.py:39-138   RationalBloomFilter
  improved_video_compressor.py:161-196  _calculate_optimal_params
  improved_video_compressor.py:198-307  compress / decompress
  rational_bloom_filter.py:9-182        StandardBloomFilter / RationalBloomFilter (string keys)
XXH64: public xxHash specification (PyPI xxhash>=2.0.0, requirements.txt:10).
"""
import math

M64 = (1 << 64) - 1
P1 = 0x9E3779B185EBCA87
P2 = 0xC2B2AE3D27D4EB4F
P3 = 0x165667B19E3779F9
P4 = 0x85EBCA77C2B2AE63
P5 = 0x27D4EB2F165667C5

SEEDS_VIDEO = (0x12345678, 0x87654321, 999)   # improved_video_compressor.py:62-63,94
SEEDS_BLOOM_COMPRESS = (0, 1, 999)            # bloom_compress.py:163-164,195


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _round(acc, inp):
    acc = (acc + inp * P2) & M64
    return (_rotl(acc, 31) * P1) & M64


def _merge(h, v):
    h ^= _round(0, v)
    return (h * P1 + P4) & M64


def xxh64(data: bytes, seed: int = 0) -> int:
    n = len(data)
    p = 0
    if n >= 32:
        v1 = (seed + P1 + P2) & M64
        v2 = (seed + P2) & M64
        v3 = seed & M64
        v4 = (seed - P1) & M64
        while p + 32 <= n:
            v1 = _round(v1, int.from_bytes(data[p:p + 8], "little")); p += 8
            v2 = _round(v2, int.from_bytes(data[p:p + 8], "little")); p += 8
            v3 = _round(v3, int.from_bytes(data[p:p + 8], "little")); p += 8
            v4 = _round(v4, int.from_bytes(data[p:p + 8], "little")); p += 8
        h = (_rotl(v1, 1) + _rotl(v2, 7) + _rotl(v3, 12) + _rotl(v4, 18)) & M64
        h = _merge(h, v1); h = _merge(h, v2); h = _merge(h, v3); h = _merge(h, v4)
    else:
        h = (seed + P5) & M64
    h = (h + n) & M64
    while p + 8 <= n:
        h ^= _round(0, int.from_bytes(data[p:p + 8], "little"))
        h = (_rotl(h, 27) * P1 + P4) & M64
        p += 8
    if p + 4 <= n:
        h ^= (int.from_bytes(data[p:p + 4], "little") * P1) & M64
        h = (_rotl(h, 23) * P2 + P3) & M64
        p += 4
    while p < n:
        h ^= (data[p] * P5) & M64
        h = (_rotl(h, 11) * P1) & M64
        p += 1
    h ^= h >> 33; h = (h * P2) & M64
    h ^= h >> 29; h = (h * P3) & M64
    h ^= h >> 32
    return h


def hash_str(item, seed):
    return xxh64(str(item).encode("utf-8"), seed)


class RationalFilter:
    """improved_video_compressor.py:39-138 with the seeds as parameters."""

    def __init__(self, size, k_star, seeds=SEEDS_VIDEO):
        self.size = size
        self.k_star = k_star
        self.floor_k = math.floor(k_star)
        self.p_activation = k_star - self.floor_k
        self.bit_array = [0] * size
        self.h1_seed, self.h2_seed, self.act_seed = seeds

    def position(self, item, i):
        h1 = hash_str(item, self.h1_seed)
        h2 = hash_str(item, self.h2_seed)
        return (h1 + i * h2) % self.size          # unbounded ints: no 2^64 wrap

    def activated(self, item):
        return hash_str(item, self.act_seed) / (2 ** 64 - 1) < self.p_activation

    def add(self, item):
        for i in range(self.floor_k):
            self.bit_array[self.position(item, i)] = 1
        if self.activated(item):
            self.bit_array[self.position(item, self.floor_k)] = 1

    def check(self, item):
        for i in range(self.floor_k):
            if self.bit_array[self.position(item, i)] == 0:
                return False
        if self.activated(item):
            if self.bit_array[self.position(item, self.floor_k)] == 0:
                return False
        return True


class StandardFilter:
    """rational_bloom_filter.py:9-41."""

    def __init__(self, m, k):
        self.size = m
        self.hash_count = int(k)
        self.bit_array = [0] * m

    def add(self, item):
        for i in range(self.hash_count):
            self.bit_array[hash_str(item, i) % self.size] = 1

    def contains(self, item):
        return all(self.bit_array[hash_str(item, i) % self.size] for i in range(self.hash_count))


P_STAR = 0.32453


def optimal_params(n, p):
    """improved_video_compressor.py:161-196 (identical text in bloom_compress.py:30-64)."""
    if p <= 0.0001:
        return 0, 0
    if p >= P_STAR:
        return 0, 0
    q = 1 - p
    L = math.log(2)
    k = math.log2(q * (L ** 2) / p)
    if math.isnan(k) or k <= 0:
        return 0, 0
    gamma = 1 / L
    l = int(p * n * k * gamma)
    return max(0.1, k), max(1, l)


def compress(bits, seeds=SEEDS_VIDEO, guard_l_ge_n=True):
    """improved_video_compressor.py:198-266.  bits: sequence of 0/1.
    Returns (bit_array | passthrough input, witness, p, n, ratio)."""
    n = len(bits)
    ones = sum(int(b) for b in bits)
    p = ones / n
    if p >= P_STAR:
        return list(bits), [], p, n, 1.0
    k, l = optimal_params(n, p)
    if l == 0 or (guard_l_ge_n and l >= n):
        return list(bits), [], p, n, 1.0
    f = RationalFilter(l, k, seeds)
    for i in range(n):
        if bits[i] == 1:
            f.add(i)
    witness = [int(bits[i]) for i in range(n) if f.check(i)]
    return f.bit_array, witness, p, n, (l + len(witness)) / n


def decompress(bitmap, witness, n, k, seeds=SEEDS_VIDEO):
    """improved_video_compressor.py:268-307."""
    if len(witness) == 0:
        return list(bitmap)
    f = RationalFilter(len(bitmap), k, seeds)
    f.bit_array = list(bitmap)
    out = [0] * n
    w = 0
    for i in range(n):
        if f.check(i):
            out[i] = witness[w]
            w += 1
    return out
