"""Image / text front-ends of the older reference module (bloom_compress.py:348-618)
on the GPU coder: seeds (0, 1, 999), no `l >= n` passthrough guard (bloom_compress.py:264), and the
network-byte-order ('!') container of that module, byte for byte.

Note the reference stores k as float32 ('!f', :428): decoding with the rounded k can flip an
activation decision.  For parity the bytes are kept; `decompress_*` accepts `k`
overrides for callers that kept the exact value.
"""
import struct

import numpy as np

from . import params as P
from .bloom_compressor import BloomFilterCompressor as _GpuCompressor


class BloomFilterCompressor(_GpuCompressor):
    """bloom_compress.BloomFilterCompressor with its image / text helpers."""

    def __init__(self, ctx=None):
        super().__init__(verbose=False, seeds=P.SEEDS_BLOOM_COMPRESS, guard_l_ge_n=False, ctx=ctx)

    # ---- binarisation (:66-142)
    @staticmethod
    def _binarize_image(image, threshold=127):
        image = np.asarray(image)
        if image.ndim > 2 and image.shape[2] > 1:
            image = np.mean(image, axis=2).astype(np.uint8)       # plain channel average
        return (image > threshold).astype(np.uint8).flatten()

    @staticmethod
    def _binarize_text(text, bit_depth=8):
        raw = text.encode("ascii", errors="replace") if bit_depth == 8 else text.encode("utf-8")
        return np.unpackbits(np.frombuffer(raw, dtype=np.uint8))

    @staticmethod
    def _debinarize_text(binary_array, bit_depth=8):
        binary_array = np.asarray(binary_array, dtype=np.uint8)
        if len(binary_array) % 8:
            binary_array = np.pad(binary_array, (0, 8 - len(binary_array) % 8), "constant")
        raw = np.packbits(binary_array).tobytes()
        return raw.decode("ascii", errors="replace") if bit_depth == 8 else raw.decode("utf-8", errors="replace")

    # ---- containers
    @staticmethod
    def _pack_streams(bloom_bitmap, witness):
        return (struct.pack("!I", len(bloom_bitmap)) + struct.pack("!I", len(witness)) +
                np.packbits(np.asarray(bloom_bitmap, dtype=np.uint8)).tobytes() +
                np.packbits(np.array(witness, dtype=np.uint8)).tobytes())

    @staticmethod
    def _unpack_streams(data, off):
        l, wl = struct.unpack_from("!II", data, off)
        off += 8
        nb, nw = (l + 7) // 8, (wl + 7) // 8
        bitmap = np.unpackbits(np.frombuffer(data, dtype=np.uint8, count=nb, offset=off))[:l]
        witness = np.unpackbits(np.frombuffer(data, dtype=np.uint8, count=nw, offset=off + nb))[:wl].tolist()
        return bitmap, witness

    def _pack_compressed_data(self, bloom_bitmap, witness, p, n, k, original_shape):
        head = struct.pack("!f", p) + struct.pack("!I", n) + struct.pack("!f", k) + struct.pack("!B", len(original_shape))
        head += b"".join(struct.pack("!I", d) for d in original_shape)
        return head + self._pack_streams(bloom_bitmap, witness)

    def _unpack_compressed_data(self, data):
        p, n, k, nd = struct.unpack_from("!fIfB", data, 0)
        shape = struct.unpack_from("!%dI" % nd, data, 13)
        bitmap, witness = self._unpack_streams(data, 13 + 4 * nd)
        return bitmap, witness, p, n, k, tuple(shape)

    def _pack_text_data(self, bloom_bitmap, witness, p, n, k, text_length, bit_depth):
        head = (struct.pack("!f", p) + struct.pack("!I", n) + struct.pack("!f", k) +
                struct.pack("!I", text_length) + struct.pack("!B", bit_depth))
        return head + self._pack_streams(bloom_bitmap, witness)

    def _unpack_text_data(self, data):
        p, n, k, text_length, bit_depth = struct.unpack_from("!fIfIB", data, 0)
        bitmap, witness = self._unpack_streams(data, 17)
        return bitmap, witness, p, n, k, text_length, bit_depth

    # ---- image (arrays instead of paths are accepted too; PIL only when a path is given)
    def compress_image(self, image, threshold=127, output_path=None):
        if isinstance(image, str):
            from PIL import Image
            image = np.array(Image.open(image))
        img = np.asarray(image)
        binary = self._binarize_image(img, threshold)
        bitmap, witness, p, n, ratio = self.compress(binary)
        k, _ = self._calculate_optimal_params(n, p)
        blob = self._pack_compressed_data(bitmap, witness, p, n, k, img.shape)
        if output_path:
            with open(output_path, "wb") as f:
                f.write(blob)
        return blob, ratio

    def decompress_image(self, compressed_data, output_path=None, k=None):
        bitmap, witness, p, n, k32, shape = self._unpack_compressed_data(compressed_data)
        binary = self.decompress(bitmap, witness, n, k32 if k is None else k)
        out = np.asarray(binary).reshape(shape[:2]) * 255
        if output_path:
            from PIL import Image
            Image.fromarray(out.astype(np.uint8)).save(output_path)
        return out

    # ---- text
    def compress_text(self, text, bit_depth=8, output_path=None):
        binary = self._binarize_text(text, bit_depth)
        bitmap, witness, p, n, ratio = self.compress(binary)
        k, _ = self._calculate_optimal_params(n, p)
        blob = self._pack_text_data(bitmap, witness, p, n, k, len(text), bit_depth)
        if output_path:
            with open(output_path, "wb") as f:
                f.write(blob)
        return blob, ratio

    def decompress_text(self, compressed_data, output_path=None, k=None):
        bitmap, witness, p, n, k32, text_length, bit_depth = self._unpack_text_data(compressed_data)
        binary = self.decompress(bitmap, witness, n, k32 if k is None else k)
        text = self._debinarize_text(binary, bit_depth)[:text_length]
        if output_path:
            with open(output_path, "w", encoding="utf-8") as f:
                f.write(text)
        return text
