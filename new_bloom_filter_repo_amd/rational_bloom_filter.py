"""RationalBloomFilter -- the reference's filter class (improved_video_compressor.py:39-138) on the GPU.

Same constructor, attributes and methods; the bit array lives in device memory as a packed
(numpy.packbits order) vector and every add / check runs in the HIP kernels behind
rbf_filter_insert_indices / rbf_filter_query_indices.  `bit_array` materialises the reference's
one-byte-per-bit np.uint8 view on demand (and can be assigned, as BloomFilterCompressor.decompress
does at :290).  add_indices / check_indices are the vectorised forms one should use from Python.

Seed variants: pass seeds=(h1, h2, act); the defaults are the video codec's.
"""
import math

import numpy as np

from . import _native as nat
from . import params as P
from .engine import DeviceFilter


class RationalBloomFilter:
    def __init__(self, size, k_star, seeds=None, ctx=None):
        if size < 1:
            raise ValueError("size must be >= 1")
        self.size = int(size)
        self.k_star = k_star
        self.floor_k = math.floor(k_star)
        self.p_activation = k_star - self.floor_k
        seeds = tuple(seeds) if seeds is not None else P.SEEDS_VIDEO
        self.h1_seed, self.h2_seed, self.act_seed = seeds
        _fk, self._threshold = P.activation_threshold(k_star)
        self._dev = DeviceFilter(ctx or nat.default_context(), self.size)

    def close(self):
        """Return the filter's device memory (also happens when the object is dropped)."""
        self._dev.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def _seeds(self):
        return (self.h1_seed, self.h2_seed, self.act_seed)

    # ---- the reference's per-index methods
    def add_index(self, index):
        self.add_indices([index])

    def check_index(self, index):
        return bool(self.check_indices([index])[0])

    # ---- vectorised
    def add_indices(self, indices):
        self._dev.insert(_as_indices(indices), self.floor_k, self._threshold, self._seeds)

    def check_indices(self, indices):
        return self._dev.query(_as_indices(indices), self.floor_k, self._threshold, self._seeds)

    @property
    def bit_array(self):
        return self._dev.bits()

    @bit_array.setter
    def bit_array(self, value):
        self._dev.set_bits(value)


def _as_indices(indices):
    a = np.asarray(indices)
    if a.size and (a.min() < 0 or a.max() > 0xFFFFFFFF):
        raise ValueError("indices must fit in uint32")
    return a.astype(np.uint32)


class StringRationalBloomFilter:
    """rational_bloom_filter.RationalBloomFilter (rational_bloom_filter.py:74-214): string keys,
    seeds (0, 1) and activation seed ceil(k*); add / contains (+ list forms add_many / contains_many)."""

    def __init__(self, m, k_star, ctx=None):
        self.size = int(m)
        self.k_star = k_star
        self.floor_k = math.floor(k_star)
        self.ceil_k = math.ceil(k_star)
        self.p_activation = k_star - self.floor_k
        self.h1_seed, self.h2_seed = 0, 1
        _fk, self._threshold = P.activation_threshold(k_star)
        self._dev = DeviceFilter(ctx or nat.default_context(), self.size)

    def close(self):
        """Return the filter's device memory (also happens when the object is dropped)."""
        self._dev.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def _seeds(self):
        return (self.h1_seed, self.h2_seed, self.ceil_k)

    def add(self, item):
        self.add_many([item])

    def contains(self, item):
        return bool(self.contains_many([item])[0])

    def add_many(self, items):
        self._dev.insert_keys(list(items), self.floor_k, self._threshold, self._seeds)

    def contains_many(self, items):
        return self._dev.query_keys(list(items), self.floor_k, self._threshold, self._seeds)

    @property
    def bit_array(self):
        return self._dev.bits().tolist()

    @staticmethod
    def get_optimal_size(n, p):
        return int(math.ceil(-(n * math.log(p)) / (math.log(2) ** 2)))

    @staticmethod
    def get_optimal_hash_count(m, n):
        return max(0.1, (m / n) * math.log(2))


class StandardBloomFilter:
    """rational_bloom_filter.StandardBloomFilter (rational_bloom_filter.py:9-71): k independent
    hashes XXH64(str(item), seed=j) % m."""

    def __init__(self, m, k, ctx=None):
        self.size = int(m)
        self.hash_count = int(k)
        self._dev = DeviceFilter(ctx or nat.default_context(), self.size)

    def close(self):
        """Return the filter's device memory (also happens when the object is dropped)."""
        self._dev.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add(self, item):
        self.add_many([item])

    def contains(self, item):
        return bool(self.contains_many([item])[0])

    def add_many(self, items):
        if self.hash_count > 0:
            self._dev.insert_keys(list(items), 0, 0, (0, 0, 0), standard_k=self.hash_count)

    def contains_many(self, items):
        items = list(items)
        if self.hash_count <= 0:
            return np.ones(len(items), dtype=bool)          # no hash functions: `all([])` is True
        return self._dev.query_keys(items, 0, 0, (0, 0, 0), standard_k=self.hash_count)

    @property
    def bit_array(self):
        return self._dev.bits().tolist()

    @staticmethod
    def get_optimal_size(n, p):
        return int(math.ceil(-(n * math.log(p)) / (math.log(2) ** 2)))

    @staticmethod
    def get_optimal_hash_count(m, n):
        return max(1, int(round((m / n) * math.log(2))))
