"""BloomFilterCompressor -- the reference's Bloom+witness coder (improved_video_compressor.py:140-307)
with the insert / query / decode loops on the GPU.

compress() / decompress() keep the reference's signatures and return types (byte-per-bit np.uint8
bitmap, witness as a list of ints); compress_batch() / decompress_batch() are the forms that keep a
whole GOP on the device and exchange packed bit vectors.
"""
import numpy as np

from . import _native as nat
from . import params as P
from .engine import BloomEngine


class BloomFilterCompressor:
    P_STAR = P.P_STAR                                      # improved_video_compressor.py:150

    def __init__(self, verbose=False, seeds=None, guard_l_ge_n=True, ctx=None):
        """seeds: (h1, h2, act) hash seeds, default the video codec's; guard_l_ge_n=False gives the
        older bloom_compress.py behaviour (no `l >= n` passthrough, bloom_compress.py:264)."""
        self.verbose = verbose
        self.seeds = tuple(seeds) if seeds is not None else P.SEEDS_VIDEO
        self.guard_l_ge_n = guard_l_ge_n
        self._ctx = ctx
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = BloomEngine(self._ctx or nat.default_context())
        return self._engine

    def _calculate_optimal_params(self, n, p):
        return P.optimal_params(n, p)

    # ------------------------------------------------------------------ reference surface
    def compress(self, binary_input):
        binary_input = np.asarray(binary_input)
        n = len(binary_input)
        if n == 0:
            raise ValueError("empty input")
        ones_count = np.sum(binary_input)
        p = ones_count / n
        if p >= self.P_STAR:
            if self.verbose:
                print(f"Density {p:.4f} is >= threshold {self.P_STAR}, compression not effective")
            return binary_input, [], p, n, 1.0
        k, l = self._calculate_optimal_params(n, p)
        if l == 0 or (self.guard_l_ge_n and l >= n):
            return binary_input, [], p, n, 1.0
        if self.verbose:
            print(f"Input length: {n}, Density: {p:.4f}")
            print(f"Optimal parameters: k={k:.4f}, l={l}")
        eng = self.engine
        eng.upload_masks(np.packbits(binary_input.astype(np.uint8))[None, :], n)
        r = eng.encode(n, [P.filter_params(k, l)], self.seeds)[0]
        bit_array = np.unpackbits(r["filter"])[:l]
        witness = np.unpackbits(r["witness"])[:r["witness_bits"]].tolist()
        ratio = (l + len(witness)) / n
        if self.verbose:
            print(f"Bloom filter size: {l} bits")
            print(f"Witness size: {len(witness)} bits")
            print(f"Compression ratio: {ratio:.4f}")
            print(f"Bloom filter test pass rate: {len(witness) / n:.4f}")
        return bit_array, witness, p, n, ratio

    def decompress(self, bloom_bitmap, witness, n, k):
        if len(witness) == 0:
            return bloom_bitmap                              # passthrough: the bitmap IS the data (:282-284)
        bloom_bitmap = np.asarray(bloom_bitmap, dtype=np.uint8)
        l = len(bloom_bitmap)
        out = self.engine.decode(n, [P.filter_params(k, l)], [np.packbits(bloom_bitmap)],
                                 [np.packbits(np.asarray(witness, dtype=np.uint8))], self.seeds)
        return np.unpackbits(out[0])[:n]

    # ------------------------------------------------------------------ batch (packed) forms
    def plan(self, n, ones):
        """Per-frame (k, l) or None for a passthrough frame, exactly as compress() decides."""
        out = []
        for c in ones:
            p = np.uint64(c) / n
            k, l = (0, 0) if p >= self.P_STAR else self._calculate_optimal_params(n, p)
            out.append(None if (l == 0 or (self.guard_l_ge_n and l >= n)) else (k, l))
        return out

    def compress_batch(self, masks_packed, n):
        """masks_packed: uint8 [F, ceil(n/8)].  Returns a list of dicts with packed filter/witness
        (or {"passthrough": True}) and the (k, l) used."""
        masks_packed = np.atleast_2d(np.asarray(masks_packed, dtype=np.uint8))
        ones = [int(np.unpackbits(m)[:n].sum()) for m in masks_packed]
        plan = self.plan(n, ones)
        plist = [P.filter_params(*kl) if kl else (0, 0, 0) for kl in plan]
        eng = self.engine
        eng.upload_masks(masks_packed, n)
        res = eng.encode(n, plist, self.seeds)
        out = []
        for kl, r, c in zip(plan, res, ones):
            if kl is None:
                out.append({"passthrough": True, "ones": c})
            else:
                r.update({"passthrough": False, "k": kl[0], "l": kl[1], "ones": c})
                out.append(r)
        return out

    def decompress_batch(self, records, n):
        """Inverse of compress_batch for the non-passthrough records; returns packed masks [F, ceil(n/8)]."""
        plist = [P.filter_params(r["k"], r["l"]) for r in records]
        return self.engine.decode(n, plist, [r["filter"] for r in records], [r["witness"] for r in records], self.seeds)
