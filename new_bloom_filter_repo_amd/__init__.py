"""MI355X-native rational Bloom-filter residual coder.

Host surface with the reference's class names; every insert / query / decode runs in hand-written
HIP kernels (librbf_hip.so, include/rbf.h), loaded on first use -- there is no CPU fallback.
"""
__version__ = "0.1.0"

_LAZY = {
    "RationalBloomFilter": "rational_bloom_filter",
    "StringRationalBloomFilter": "rational_bloom_filter",
    "StandardBloomFilter": "rational_bloom_filter",
    "BloomFilterCompressor": "bloom_compressor",
    "VideoFrameCompressor": "frame_codec",
    "FixedVideoCompressor": "frame_codec",
    "YUVFrame": "frame_codec",
    "ImprovedVideoCompressor": "video_compressor",
    "verify_lossless": "verify",
    "verify_bit_exact": "verify",
    "GopCoder": "gop",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError(name)
