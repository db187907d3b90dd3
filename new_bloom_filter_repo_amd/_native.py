"""ctypes binding of librbf_hip.so (include/rbf.h).  No CPU fallback: if the HIP library is
missing or a call fails, the caller gets an exception."""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RBF_LIB_PATH") or os.path.join(_HERE, "librbf_hip.so")      # RBF_LIB_PATH: A/B runs of differently built libraries

RBF_OK = 0
RBF_EINVAL = -22
RBF_ENOMEM = -12
RBF_EIO = -5
RBF_ERANGE = -34

K_MASK, K_INSERT, K_QUERY, K_STITCH, K_EXPAND, K_GATHER, K_SCATTER, K_INDEX, K_REDUCE, K_SCAN, K_NOISE, K_PACK, K_HASHTAB = range(13)
KERNEL_NAMES = ["mask", "insert", "query", "stitch", "expand", "gather", "scatter", "index", "reduce", "scan", "noise", "pack", "hashtab"]
STATS_PER_FRAME = 4
PAIR_SKIPPED = 0xFFFFFFFF                 # params[p].floor_k of a pair across a keyframe (rbf_encode_runs)
OPT_SEPARATE_FINISH, OPT_INSERT_SLICES = 2, 4    # rbf_ctx_option keys (include/rbf.h); keys 1, 3, 5 and 99 of earlier ABIs are gone with what they selected


class FilterParams(ctypes.Structure):
    _fields_ = [("m", ctypes.c_uint32), ("floor_k", ctypes.c_uint32), ("threshold", ctypes.c_uint64)]


class Seeds(ctypes.Structure):
    _fields_ = [("h1", ctypes.c_uint64), ("h2", ctypes.c_uint64), ("act", ctypes.c_uint64)]


class RbfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librbf_hip error %d: %s" % (code, msg))
        self.code = code


_vp, _u64, _u32, _i32, _int = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int
_i32p = ctypes.POINTER(ctypes.c_int32)
_PROTOS = {
    "rbf_version": (_int, []),
    "rbf_last_error": (ctypes.c_char_p, []),
    "rbf_device_count": (_int, [ctypes.POINTER(_int)]),
    "rbf_ctx_create": (_int, [_int, _vp, ctypes.POINTER(_vp)]),
    "rbf_ctx_destroy": (_int, [_vp]),
    "rbf_ctx_sync": (_int, [_vp]),
    "rbf_malloc": (_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "rbf_free": (_int, [_vp, _vp]),
    "rbf_memset": (_int, [_vp, _vp, _int, ctypes.c_size_t]),
    "rbf_memcpy_h2d": (_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "rbf_memcpy_d2h": (_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "rbf_memcpy_d2d": (_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "rbf_timing_enable": (_int, [_vp, _int]),
    "rbf_timing_reset": (_int, [_vp]),
    "rbf_ctx_force_generic": (_int, [_vp, _int]),
    "rbf_ctx_option": (_int, [_vp, _int, ctypes.c_int64]),
    "rbf_timing_read": (_int, [_vp, _int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_u64)]),
    "rbf_optimal_params": (_int, [_u64, _u64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_u64)]),
    "rbf_activation_threshold": (_int, [ctypes.c_double, ctypes.POINTER(_u32), ctypes.POINTER(_u64)]),
    "rbf_plan_batch": (_int, [_u64, ctypes.POINTER(_u64), _u32, _int, ctypes.POINTER(FilterParams), ctypes.POINTER(ctypes.c_double)]),
    "rbf_encode_gop": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _i32, _i32p, ctypes.POINTER(Seeds),
                              _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp, ctypes.POINTER(FilterParams), ctypes.POINTER(ctypes.c_double)]),
    "rbf_encode_gop_begin": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _i32, _i32p, ctypes.POINTER(Seeds),
                                    _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp]),
    "rbf_encode_gop_poll": (_int, [_vp, ctypes.POINTER(_int)]),
    "rbf_filter_stride_min": (_u64, [_u64]),
    "rbf_encode_runs_begin": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _i32, _i32p, _vp, ctypes.POINTER(Seeds),
                                     _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp]),
    "rbf_encode_runs": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _i32, _i32p, _vp, ctypes.POINTER(Seeds),
                               _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp, ctypes.POINTER(FilterParams), ctypes.POINTER(ctypes.c_double)]),
    "rbf_encode_gop_finish": (_int, [_vp, ctypes.POINTER(FilterParams), ctypes.POINTER(ctypes.c_double)]),
    "rbf_residual_mask_batch": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _i32, _i32p, _vp, _u64, _vp]),
    "rbf_record_max_bytes": (_u64, [_u32, _u64]),
    "rbf_pack_records": (_int, [_vp, _u32, _u64, ctypes.POINTER(FilterParams), ctypes.POINTER(ctypes.c_double),
                                _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _u64]),
    "rbf_gather_values_batch": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _u32, _vp, _u64, _vp, _u64, _vp, _vp]),
    "rbf_bgr_to_gray_batch": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _vp]),
    "rbf_extract_luma_batch": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _vp]),
    "rbf_noise_moments_batch": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u64, _u32, _u32, _vp, _vp]),
    "rbf_bloom_encode_batch": (_int, [_vp, _vp, _u64, _u64, _u32, ctypes.POINTER(FilterParams), ctypes.POINTER(Seeds),
                                      _vp, _u64, _vp, _u64, _vp]),
    "rbf_bloom_decode_batch": (_int, [_vp, _vp, _u64, _vp, _u64, _u64, _u32, ctypes.POINTER(FilterParams),
                                      ctypes.POINTER(Seeds), _vp, _u64]),
    "rbf_filter_insert_indices": (_int, [_vp, _vp, ctypes.POINTER(FilterParams), ctypes.POINTER(Seeds), _vp, _u64]),
    "rbf_filter_query_indices": (_int, [_vp, _vp, ctypes.POINTER(FilterParams), ctypes.POINTER(Seeds), _vp, _u64, _vp]),
    "rbf_filter_insert_keys": (_int, [_vp, _vp, ctypes.POINTER(FilterParams), ctypes.POINTER(Seeds), _u32, _vp, _vp, _u64]),
    "rbf_filter_query_keys": (_int, [_vp, _vp, ctypes.POINTER(FilterParams), ctypes.POINTER(Seeds), _u32, _vp, _vp, _u64, _vp]),
    "rbf_gather_values": (_int, [_vp, _vp, _u32, _u32, _u64, _u32, _u32, _u32, _vp, _vp, _vp]),
    "rbf_scatter_values": (_int, [_vp, _vp, _u32, _u32, _u64, _u32, _u32, _u32, _vp, _vp]),
}

_lib = None


def exported_symbols():
    """Names include/rbf.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(_PROTOS)


def lib():
    """Load librbf_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != RBF_OK:
        msg = lib().rbf_last_error()
        raise RbfError(code, msg.decode("utf-8", "replace") if msg else "")
    return code


def device_count():
    c = _int(0)
    try:
        check(lib().rbf_device_count(ctypes.byref(c)))
    except RbfError:
        return 0
    return c.value


class DeviceBuffer:
    """A block of device memory.  Freed by .free(), when the last reference to it goes away, or with its
    Context -- whichever comes first (the context only keeps a weak reference, so dropping a filter /
    coder / compressor object returns its HBM)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = _vp()
        check(lib().rbf_malloc(ctx.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value
        ctx._buffers.add(self)

    def free(self):
        if self.ptr:
            ptr, self.ptr = self.ptr, None
            self.ctx._buffers.discard(self)
            if self.ctx.handle:                       # a closed context has already released its blocks
                check(lib().rbf_free(self.ctx.handle, ptr))

    close = free

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def zero(self):
        check(lib().rbf_memset(self.ctx.handle, self.ptr, 0, self.nbytes))
        return self

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes, (offset, arr.nbytes, self.nbytes)
        check(lib().rbf_memcpy_h2d(self.ctx.handle, self.ptr + offset, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, nbytes=None, offset=0, dtype=np.uint8):
        nbytes = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(nbytes, dtype=np.uint8)
        if nbytes:
            check(lib().rbf_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + offset, nbytes))
        return out.view(dtype)


class Context:
    """One HIP stream + scratch.  stream: an existing hipStream_t handle (int) or None."""

    def __init__(self, device=0, stream=None):
        h = _vp()
        check(lib().rbf_ctx_create(int(device), _vp(stream) if stream else None, ctypes.byref(h)))
        self.handle = h.value
        self.device = int(device)
        self._buffers = weakref.WeakSet()     # live blocks, so close() can release what is still referenced

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def sync(self):
        check(lib().rbf_ctx_sync(self.handle))

    def close(self):
        if self.handle:
            for b in list(self._buffers):
                b.free()
            handle, self.handle = self.handle, None
            check(lib().rbf_ctx_destroy(handle))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def option(self, option, value):
        """rbf_ctx_option: further tuning knobs (OPT_SEPARATE_FINISH: the GOP mask kernel's tail as its own launch)."""
        check(lib().rbf_ctx_option(self.handle, int(option), int(value)))

    def force_generic(self, on):
        """Testing knob: never use the LDS-resident fast-path kernels."""
        check(lib().rbf_ctx_force_generic(self.handle, int(on)))

    # ---- timing
    def timing(self, on):
        """on: False/0 off, True/1 every kernel, or an int bit mask of kernel ids (1 << K_QUERY ...)."""
        check(lib().rbf_timing_enable(self.handle, int(on)))

    def timing_reset(self):
        check(lib().rbf_timing_reset(self.handle))

    def timing_read(self):
        out = {}
        for kid, name in enumerate(KERNEL_NAMES):
            ms, cnt = ctypes.c_double(0), _u64(0)
            check(lib().rbf_timing_read(self.handle, kid, ctypes.byref(ms), ctypes.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out


_default_ctx = None


def default_context():
    """Process-wide context on device LOCAL_RANK (or 0)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx


def packed_stride(nbits):
    """Bytes of a packed bit vector buffer: whole 64-bit words."""
    return ((int(nbits) + 63) // 64) * 8


def params_array(plist):
    arr = (FilterParams * len(plist))()
    for i, (m, floor_k, thr) in enumerate(plist):
        arr[i].m, arr[i].floor_k, arr[i].threshold = int(m), int(floor_k), int(thr)
    return arr
