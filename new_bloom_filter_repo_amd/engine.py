"""Batch-level host driver of the HIP path: numpy in, numpy out, everything in between on the GPU.

`BloomEngine` owns the device buffers of one batch geometry (n pixels, F frames) and exposes
the three C-ABI stages -- residual masks (A1), insert+query/witness (A4+A5), decode (A6) -- plus
the host step between them: the float64 parameter math (params.py) that must stay on the host.
"""
import ctypes

import numpy as np

from . import _native as nat
from . import params as P


class BloomEngine:
    def __init__(self, ctx=None):
        self.ctx = ctx or nat.default_context()
        self._bufs = {}

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, nbytes):
        b = self._bufs.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self.ctx.alloc(max(int(nbytes), 8))
            self._bufs[name] = b
        return b

    def close(self):
        for b in self._bufs.values():
            b.free()
        self._bufs = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ A1
    def _upload_frames(self, frames, min_frames):
        frames = np.ascontiguousarray(frames)
        if frames.ndim not in (3, 4):
            raise ValueError("frames must be (F, H, W) or (F, H, W, C)")
        if frames.dtype not in (np.uint8, np.uint16):
            raise ValueError("8- or 16-bit unsigned samples expected")
        if frames.shape[0] < min_frames:
            raise ValueError("need at least %d frame%s" % (min_frames, "s" if min_frames > 1 else ""))
        F, H, W = frames.shape[:3]
        C = frames.shape[3] if frames.ndim == 4 else 1
        fb = self._buf("frames", frames.nbytes).upload(frames)
        return fb, F, H, W, C, frames.dtype.itemsize

    def residual_masks(self, frames, threshold, luma_only=True, adaptive=None):
        """frames: array (F, H, W[, C]) uint8/uint16 (channel 0 = luma).  threshold: one number for
        every pair, a sequence of F-1 numbers, or None with adaptive=(noise_tolerance, min_thr,
        max_thr) for the reference's per-frame noise-adaptive threshold (:746-766).  Returns
        (masks_packed uint8 [F-1, stride], ones uint64 [F-1]); masks stay on the device as well
        for the following encode."""
        fb, F, H, W, C, sb = self._upload_frames(frames, 2)
        n = H * W
        stride = nat.packed_stride(n)
        mb = self._buf("masks", (F - 1) * stride)
        ob = self._buf("ones", (F - 1) * 8)
        if threshold is None:
            if adaptive is None:
                raise ValueError("threshold=None needs adaptive=(noise_tolerance, min_thr, max_thr)")
            threshold = self._adaptive_floors(fb, F, H, W, C, sb, *adaptive)
        thr, thr_tab = 0, None
        if np.ndim(threshold) == 0:
            thr = threshold_floor(threshold)
        else:
            if len(threshold) != F - 1:
                raise ValueError("need one threshold per frame pair")
            thr_tab = (ctypes.c_int32 * (F - 1))(*[threshold_floor(t) for t in threshold])
        self.thresholds = [thr] * (F - 1) if thr_tab is None else list(thr_tab)
        nat.check(nat.lib().rbf_residual_mask_batch(
            self.ctx.handle, fb.ptr, H * W * C * sb, F, W, H, W * C * sb, C * sb, sb, thr, thr_tab,
            mb.ptr, stride, ob.ptr))
        masks = mb.download((F - 1) * stride).reshape(F - 1, stride)
        ones = ob.download((F - 1) * 8, dtype=np.uint64).copy()
        self.n, self.mask_stride = n, stride
        return masks, ones

    def bgr_to_gray(self, frames):
        """cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) for every (H, W, >=3) frame of the stack (:794-795):
        OpenCV 4.x's integer path, (B*3735 + G*19235 + R*9798 + 2^14) >> 15.  Returns (F, H, W)."""
        fb, F, H, W, C, sb = self._upload_frames(frames, 1)
        if C < 3:
            raise ValueError("BGR frames need at least 3 channels")
        gb = self._buf("gray", F * H * W * sb)
        nat.check(nat.lib().rbf_bgr_to_gray_batch(self.ctx.handle, fb.ptr, H * W * C * sb, F, W, H, W * C * sb, C * sb, sb, gb.ptr))
        return gb.download(F * H * W * sb, dtype=np.uint8 if sb == 1 else np.uint16).reshape(F, H, W).copy()

    # ------------------------------------------------------------------ A1, adaptive threshold
    def _noise(self, fb, first, count, H, W, C, sb, want_planes):
        """Launch the 5x5-median residual kernel on frames [first, first+count) of the uploaded block."""
        mo = self._buf("moments", count * 16)
        nz = self._buf("noise", count * H * W * 4) if want_planes else None
        fs = H * W * C * sb
        nat.check(nat.lib().rbf_noise_moments_batch(
            self.ctx.handle, fb.ptr + first * fs, fs, count, W, H, W * C * sb, C * sb, sb,
            mo.ptr, nz.ptr if nz else None))
        moments = mo.download(count * 16, dtype=np.int64).reshape(count, 2).copy()
        planes = nz.download(count * H * W * 4, dtype=np.float32).reshape(count, H, W) if nz else None
        return moments, planes

    def noise_moments(self, frames):
        """Exact (sum, sum of squares) of luma - medianBlur5(luma) per frame: int64 [F, 2]."""
        fb, F, H, W, C, sb = self._upload_frames(frames, 1)
        return self._noise(fb, 0, F, H, W, C, sb, False)[0]

    def noise_levels(self, frames):
        """VideoFrameCompressor._estimate_noise_level for every frame (:727-744): np.float32 [F].
        The median residual comes from the GPU; the float32 np.std is taken here, with numpy, on
        the same (H, W) float32 array the reference would hold, so the result has the same bits."""
        fb, F, H, W, C, sb = self._upload_frames(frames, 1)
        planes = self._noise(fb, 0, F, H, W, C, sb, True)[1]
        return np.array([np.std(planes[f]) for f in range(F)], dtype=np.float32)

    def _adaptive_floors(self, fb, F, H, W, C, sb, noise_tolerance, min_thr, max_thr):
        """floor() of the reference's adaptive threshold of frames 1..F-1 (the `curr` frame of each
        pair, :805).  Fast path: the exact integer moments give the standard deviation to ~1e-16;
        numpy's float32 pairwise np.std differs from that by < 1e-5 relative, so unless an integer
        lies inside that band around the clamped threshold the floor is already decided.  Otherwise
        (rare) the frame's noise plane is downloaded and the reference's float32 math is replayed."""
        n = H * W
        moments = self._noise(fb, 1, F - 1, H, W, C, sb, False)[0]
        floors = []
        for i in range(F - 1):
            s1, s2 = int(moments[i, 0]), int(moments[i, 1])
            lo, hi = adaptive_threshold_band(n, s1, s2, noise_tolerance, min_thr, max_thr)
            if lo != hi:
                plane = self._noise(fb, 1 + i, 1, H, W, C, sb, True)[1][0]
                lo = threshold_floor(adaptive_threshold(np.std(plane), noise_tolerance, min_thr, max_thr))
                self.adaptive_exact_fallbacks = getattr(self, "adaptive_exact_fallbacks", 0) + 1
            floors.append(lo)
        return floors

    def adaptive_thresholds(self, frames, noise_tolerance=10.0, min_thr=3.0, max_thr=30.0):
        """Integer thresholds (floors) the reference's adaptive rule gives frames 1..F-1."""
        fb, F, H, W, C, sb = self._upload_frames(frames, 2)
        return self._adaptive_floors(fb, F, H, W, C, sb, noise_tolerance, min_thr, max_thr)

    # ------------------------------------------------------------------ A4 + A5
    def upload_masks(self, masks_packed, n):
        """masks_packed: uint8 [F, >=ceil(n/8)] numpy.packbits rows (pad bits zero)."""
        masks_packed = np.atleast_2d(np.asarray(masks_packed, dtype=np.uint8))
        F = masks_packed.shape[0]
        stride = nat.packed_stride(n)
        rows = np.zeros((F, stride), dtype=np.uint8)
        nb = (n + 7) // 8
        rows[:, :nb] = masks_packed[:, :nb]
        self._buf("masks", F * stride).upload(rows)
        self.n, self.mask_stride = n, stride
        return F

    def encode(self, n, plist, seeds=P.SEEDS_VIDEO, download=True):
        """Insert + query for the F masks currently in the device mask buffer.
        plist: F tuples (m, floor_k, T).  Returns list of dicts (filter, witness packed uint8,
        witness_bits, filter_ones) when download, else None (results stay on the device)."""
        F = len(plist)
        stride = nat.packed_stride(n)
        fstride = max(nat.packed_stride(p[0]) for p in plist)
        wstride = nat.packed_stride(n)
        mb = self._bufs["masks"]
        fb = self._buf("filters", F * fstride)
        wb = self._buf("witness", F * wstride)
        sb = self._buf("stats", F * nat.STATS_PER_FRAME * 8)
        arr = nat.params_array(plist)
        sd = nat.Seeds(*[int(s) for s in seeds])
        nat.check(nat.lib().rbf_bloom_encode_batch(
            self.ctx.handle, mb.ptr, stride, n, F, arr, ctypes.byref(sd),
            fb.ptr, fstride, wb.ptr, wstride, sb.ptr))
        self.filter_stride, self.witness_stride, self.nframes = fstride, wstride, F
        if not download:
            return None
        stats = sb.download(F * nat.STATS_PER_FRAME * 8, dtype=np.uint64).reshape(F, nat.STATS_PER_FRAME)
        filt = fb.download(F * fstride).reshape(F, fstride)
        wit = wb.download(F * wstride).reshape(F, wstride)
        out = []
        for f in range(F):
            wbits = int(stats[f, 0])
            out.append({"filter": filt[f, :(plist[f][0] + 7) // 8].copy(),
                        "witness": wit[f, :(wbits + 7) // 8].copy(),
                        "witness_bits": wbits, "filter_ones": int(stats[f, 1])})
        return out

    # ------------------------------------------------------------------ A6
    def decode(self, n, plist, filters_packed, witnesses_packed, seeds=P.SEEDS_VIDEO):
        """filters_packed / witnesses_packed: lists of packed uint8 arrays.  Returns masks
        packed uint8 [F, ceil(n/8)]."""
        F = len(plist)
        stride = nat.packed_stride(n)
        fstride = max(nat.packed_stride(p[0]) for p in plist)
        wstride = max([nat.packed_stride(len(w) * 8) for w in witnesses_packed] + [8])
        frows = np.zeros((F, fstride), dtype=np.uint8)
        wrows = np.zeros((F, wstride), dtype=np.uint8)
        for f in range(F):
            fp = np.asarray(filters_packed[f], dtype=np.uint8)
            wp = np.asarray(witnesses_packed[f], dtype=np.uint8)
            frows[f, :len(fp)] = fp
            wrows[f, :len(wp)] = wp
        fb = self._buf("filters", F * fstride).upload(frows)
        wb = self._buf("witness", F * wstride).upload(wrows)
        mb = self._buf("masks", F * stride)
        arr = nat.params_array(plist)
        sd = nat.Seeds(*[int(s) for s in seeds])
        nat.check(nat.lib().rbf_bloom_decode_batch(
            self.ctx.handle, fb.ptr, fstride, wb.ptr, wstride, n, F, arr, ctypes.byref(sd), mb.ptr, stride))
        out = mb.download(F * stride).reshape(F, stride)
        return out[:, :(n + 7) // 8].copy()


def threshold_floor(threshold):
    """int32 t with (d > threshold) == (d > t) for every integer d (float compare, :808)."""
    if threshold != threshold:           # NaN: comparison is always False
        return 2 ** 31 - 1
    import math
    t = math.floor(threshold)
    return int(max(-2 ** 31, min(2 ** 31 - 1, t)))


def adaptive_threshold(noise_level, noise_tolerance, min_thr, max_thr):
    """VideoFrameCompressor._adaptive_diff_threshold's clamp (:756-760); with a np.float32 noise
    level the product stays float32, as it does in the reference."""
    return max(min_thr, min(max_thr, noise_level * noise_tolerance))


ADAPTIVE_GUARD = 1e-4      # >> the float32 pairwise-summation error of np.std (measured < 2e-6)


def adaptive_threshold_band(n, s1, s2, noise_tolerance, min_thr, max_thr):
    """(floor_lo, floor_hi) of the adaptive threshold from the exact moments of the noise plane:
    equal when numpy's float32 rounding cannot change the integer threshold."""
    import math
    var_num = n * s2 - s1 * s1                       # n^2 * variance, exact
    std = math.sqrt(var_num) / n if var_num > 0 else 0.0
    lo = adaptive_threshold(std * (1.0 - ADAPTIVE_GUARD), noise_tolerance, min_thr, max_thr)
    hi = adaptive_threshold(std * (1.0 + ADAPTIVE_GUARD), noise_tolerance, min_thr, max_thr)
    return threshold_floor(min(lo, hi)), threshold_floor(max(lo, hi))


class DeviceFilter:
    """One packed Bloom filter of m bits living in device memory (RationalBloomFilter's storage)."""

    def __init__(self, ctx, m):
        self.ctx, self.m = ctx, int(m)
        self.nbytes = nat.packed_stride(self.m)
        self.buf = ctx.alloc(self.nbytes).zero()

    def close(self):
        self.buf.free()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _params(self, floor_k, threshold):
        return nat.FilterParams(self.m, int(floor_k), int(threshold))

    def insert(self, indices, floor_k, threshold, seeds):
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        if idx.size == 0:
            return
        ib = self.ctx.alloc(idx.nbytes).upload(idx)
        p, sd = self._params(floor_k, threshold), nat.Seeds(*[int(s) for s in seeds])
        nat.check(nat.lib().rbf_filter_insert_indices(self.ctx.handle, self.buf.ptr, ctypes.byref(p), ctypes.byref(sd), ib.ptr, idx.size))
        ib.free()

    def query(self, indices, floor_k, threshold, seeds):
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        if idx.size == 0:
            return np.zeros(0, dtype=bool)
        ib = self.ctx.alloc(idx.nbytes).upload(idx)
        ob = self.ctx.alloc(idx.size)
        p, sd = self._params(floor_k, threshold), nat.Seeds(*[int(s) for s in seeds])
        nat.check(nat.lib().rbf_filter_query_indices(self.ctx.handle, self.buf.ptr, ctypes.byref(p), ctypes.byref(sd), ib.ptr, idx.size, ob.ptr))
        out = ob.download(idx.size).astype(bool)
        ib.free(); ob.free()
        return out

    def _keys(self, items):
        keys = [str(x).encode("utf-8") for x in items]
        offs = np.zeros(len(keys) + 1, dtype=np.uint32)
        offs[1:] = np.cumsum([len(k) for k in keys])
        blob = np.frombuffer(b"".join(keys) + b"\0" * 8, dtype=np.uint8)
        return self.ctx.alloc(blob.nbytes).upload(blob), self.ctx.alloc(offs.nbytes).upload(offs), len(keys)

    def insert_keys(self, items, floor_k, threshold, seeds, standard_k=0):
        """String keys str(item) (rational_bloom_filter.py add); standard_k > 0 = StandardBloomFilter."""
        if len(items) == 0:
            return
        kb, ob, cnt = self._keys(items)
        p, sd = self._params(floor_k, threshold), nat.Seeds(*[int(s) for s in seeds])
        nat.check(nat.lib().rbf_filter_insert_keys(self.ctx.handle, self.buf.ptr, ctypes.byref(p), ctypes.byref(sd), int(standard_k), kb.ptr, ob.ptr, cnt))
        kb.free(); ob.free()

    def query_keys(self, items, floor_k, threshold, seeds, standard_k=0):
        if len(items) == 0:
            return np.zeros(0, dtype=bool)
        kb, ob, cnt = self._keys(items)
        out = self.ctx.alloc(cnt)
        p, sd = self._params(floor_k, threshold), nat.Seeds(*[int(s) for s in seeds])
        nat.check(nat.lib().rbf_filter_query_keys(self.ctx.handle, self.buf.ptr, ctypes.byref(p), ctypes.byref(sd), int(standard_k), kb.ptr, ob.ptr, cnt, out.ptr))
        res = out.download(cnt).astype(bool)
        kb.free(); ob.free(); out.free()
        return res

    def bits(self):
        """np.uint8[m], one byte per bit (the reference's `bit_array`)."""
        return np.unpackbits(self.buf.download(self.nbytes))[:self.m]

    def set_bits(self, bit_array):
        bit_array = np.asarray(bit_array, dtype=np.uint8).reshape(-1)
        if bit_array.size != self.m:
            raise ValueError("bit_array must have %d entries" % self.m)
        row = np.zeros(self.nbytes, dtype=np.uint8)
        pk = np.packbits(bit_array)
        row[:pk.size] = pk
        self.buf.upload(row)

    def free(self):
        self.buf.free()


def gather_values(ctx, frame, mask_packed):
    """Changed-pixel values of `frame` (H, W[, C]) at the mask's '1' pixels, raster order (A2)."""
    frame = np.ascontiguousarray(frame)
    H, W = frame.shape[:2]
    C = frame.shape[2] if frame.ndim == 3 else 1
    sb = frame.dtype.itemsize
    n = H * W
    fb = ctx.alloc(frame.nbytes).upload(frame)
    row = np.zeros(nat.packed_stride(n), dtype=np.uint8)
    row[:(n + 7) // 8] = np.asarray(mask_packed, dtype=np.uint8)[:(n + 7) // 8]
    mb = ctx.alloc(row.nbytes).upload(row)
    vb = ctx.alloc(max(frame.nbytes, 8))
    cb = ctx.alloc(8)
    nat.check(nat.lib().rbf_gather_values(ctx.handle, fb.ptr, W, H, W * C * sb, C * sb, sb, C, mb.ptr, vb.ptr, cb.ptr))
    cnt = int(cb.download(8, dtype=np.uint64)[0])
    vals = vb.download(cnt * C * sb).view(frame.dtype).copy()
    for b in (fb, mb, vb, cb):
        b.free()
    return vals


def apply_chain(ctx, base, masks_packed, values_list, chunk_frames=64, chunk_bytes=256 << 20):
    """A8 for a run of inter-frames: frame t = frame t-1 with `values_list[t]` written at mask t's '1' pixels
    (improved_video_compressor.py:849-909).  The run is rebuilt ON THE DEVICE in chunks of at most `chunk_frames` frames and `chunk_bytes`
    bytes (1080p YUV444: 41 frames; an 8K 16-bit frame: one at a time -- device block and host block stay bounded whatever the frame
    size): one upload of the chunk's masks, one of its values, then per frame a device-to-device copy of its predecessor and one
    scatter, no host round trip in between; the chunk's frames come back in ONE download.
    Returns the list of frames.  They are VIEWS of the downloaded chunk blocks: keeping one of them alive keeps its whole chunk alive."""
    base = np.ascontiguousarray(base)
    H, W = base.shape[:2]
    C = base.shape[2] if base.ndim == 3 else 1
    sb = base.dtype.itemsize
    n = H * W
    stride = nat.packed_stride(n)
    fbytes = base.nbytes
    total = len(masks_packed)
    out = []
    if total == 0:
        return out
    L = nat.lib()
    per = max(1, min(int(chunk_frames), total, int(chunk_bytes) // max(1, fbytes)))
    fb = ctx.alloc((per + 1) * fbytes)                            # slot 0: the predecessor of the chunk's first frame
    mb = ctx.alloc(per * stride)
    vcap = 8
    for c0 in range(0, total, per):
        vcap = max(vcap, sum(np.asarray(v).nbytes for v in values_list[c0:c0 + per]))
    vb = ctx.alloc(vcap)
    prev = base
    for c0 in range(0, total, per):
        cnt = min(per, total - c0)
        rows = np.zeros((cnt, stride), dtype=np.uint8)
        vals, offs = [], []
        off = 0
        for j in range(cnt):
            rows[j, :(n + 7) // 8] = np.asarray(masks_packed[c0 + j], dtype=np.uint8)[:(n + 7) // 8]
            v = np.ascontiguousarray(values_list[c0 + j], dtype=base.dtype).reshape(-1)
            offs.append(off)
            off += v.nbytes
            vals.append(v)
        fb.upload(prev, 0)
        mb.upload(rows)
        if off:
            vb.upload(np.concatenate(vals))
        for j in range(cnt):
            dst = fb.ptr + (j + 1) * fbytes
            nat.check(L.rbf_memcpy_d2d(ctx.handle, dst, fb.ptr + j * fbytes, fbytes))
            nat.check(L.rbf_scatter_values(ctx.handle, dst, W, H, W * C * sb, C * sb, sb, C, mb.ptr + j * stride, vb.ptr + offs[j]))
        block = fb.download(cnt * fbytes, offset=fbytes).view(base.dtype).reshape((cnt,) + base.shape)
        out += [block[j] for j in range(cnt)]
        prev = block[cnt - 1]
    for b in (fb, mb, vb):
        b.free()
    return out


def scatter_values(ctx, frame, mask_packed, values):
    """Copy of `frame` with `values` written at the mask's '1' pixels, raster order (A8)."""
    frame = np.ascontiguousarray(frame)
    H, W = frame.shape[:2]
    C = frame.shape[2] if frame.ndim == 3 else 1
    sb = frame.dtype.itemsize
    n = H * W
    values = np.ascontiguousarray(values, dtype=frame.dtype).reshape(-1)
    fb = ctx.alloc(frame.nbytes).upload(frame)
    row = np.zeros(nat.packed_stride(n), dtype=np.uint8)
    row[:(n + 7) // 8] = np.asarray(mask_packed, dtype=np.uint8)[:(n + 7) // 8]
    mb = ctx.alloc(row.nbytes).upload(row)
    vb = ctx.alloc(max(values.nbytes, 8))
    if values.nbytes:
        vb.upload(values)
    nat.check(nat.lib().rbf_scatter_values(ctx.handle, fb.ptr, W, H, W * C * sb, C * sb, sb, C, mb.ptr, vb.ptr))
    out = fb.download(frame.nbytes).view(frame.dtype).reshape(frame.shape).copy()
    for b in (fb, mb, vb):
        b.free()
    return out
