"""GOP-level driver: frames resident in HBM -> residual masks -> (host: filter geometry) ->
Bloom insert + query/witness, one C-ABI call per block of frames: one GOP (rbf_encode_gop) or several GOPs whose keyframes
are marked (rbf_encode_runs: ONE launch sequence for all of them).

Device memory comes from an allocator callback so the same code runs on library-owned buffers
(default) or on torch tensors (bench.py / dist.py pass `torch_allocator`, which lets RCCL move the
results without a copy)."""
import ctypes

import numpy as np

from . import _native as nat
from . import params as P


class _OwnedBlock:
    def __init__(self, buf):
        self.buf, self.ptr, self.nbytes = buf, buf.ptr, buf.nbytes

    def numpy(self, ctx, nbytes=None):
        return self.buf.download(nbytes)


class _TorchBlock:
    def __init__(self, t):
        self.tensor, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()

    def numpy(self, ctx, nbytes=None):
        a = self.tensor.cpu().numpy().view(np.uint8).reshape(-1)
        return a if nbytes is None else a[:nbytes]


def owned_allocator(ctx):
    return lambda nbytes: _OwnedBlock(ctx.alloc(nbytes))


def torch_allocator(device):
    import torch
    return lambda nbytes: _TorchBlock(torch.zeros((int(nbytes) + 7) // 8, dtype=torch.int64, device=device))


class TorchArena:
    """Carves consecutive 256-byte aligned blocks out of ONE torch tensor, so that a GOP's output
    record (filters | witnesses | stats) is a single contiguous tensor a collective can move."""

    def __init__(self, device, nbytes):
        import torch
        self.tensor = torch.zeros((int(nbytes) + 7) // 8, dtype=torch.int64, device=device)
        self.used = 0

    def __call__(self, nbytes):
        nbytes = (int(nbytes) + 255) // 256 * 256
        assert self.used + nbytes <= self.tensor.numel() * 8, "arena exhausted"
        view = self.tensor[self.used // 8:(self.used + nbytes) // 8]
        self.used += nbytes
        return _TorchBlock(view)


class GopCoder:
    """Encodes the nframes-1 inter-frame residual masks of one GOP of (H, W, C) frames."""

    def __init__(self, ctx, width, height, nframes, channels=3, sample_bytes=1, seeds=P.SEEDS_VIDEO,
                 allocator=None, threshold=0.0, out_allocator=None, frames_block=None, adaptive=None,
                 planar_luma=False, keep_interleaved=True, resident_gops=1, luma_block=None, run_starts=None):
        """allocator: device memory source (default: library-owned); out_allocator: separate source for
        the output record (filters, witnesses, stats); frames_block: share another coder's frame buffer.
        threshold=None with adaptive=(noise_tolerance, min_thr, max_thr): per-frame noise-adaptive
        thresholds (improved_video_compressor.py:746-766) -- lossy, like the reference's default.
        planar_luma: keep the GOP's Y planes as one dense block (the reference's YUVFrame carries the same plane,
        fixed_video_compressor.py:292-296) and run the mask stage on it: a third of the interleaved bytes.  The
        interleaved frames are then needed by gather_values() only (keep_interleaved=False: luma alone is resident,
        3x more GOPs per byte of HBM).  resident_gops: room for that many GOPs of frames; encode(gop=g) codes the g-th.
        luma_block: share an existing block of Y planes (like frames_block).
        run_starts: frame indices (within the block of `nframes` frames) that are KEYFRAMES of the caller's stream: each starts a new run,
        and the pair in front of it is not coded (rbf_encode_runs: several GOPs in ONE launch sequence; results() marks those pairs
        `skipped`).  The reference codes frame by frame (improved_video_compressor.py:198-266); batching whole GOPs is this package's.
        """
        from .engine import threshold_floor
        self.ctx, self.W, self.H, self.F, self.C, self.sb = ctx, width, height, nframes, channels, sample_bytes
        self.n = width * height
        self.pairs = nframes - 1
        self.seeds = nat.Seeds(*[int(s) for s in seeds])
        if threshold is None and adaptive is None:
            raise ValueError("threshold=None needs adaptive=(noise_tolerance, min_thr, max_thr)")
        self.thr = 0 if threshold is None else threshold_floor(threshold)
        self.adaptive = adaptive if threshold is None else None
        self.thr_tab = None
        self.set_run_starts(run_starts)
        base_alloc = allocator or owned_allocator(ctx)
        self._blocks = []

        def alloc(nbytes):                        # remember what this coder allocated, for close()
            b = base_alloc(nbytes)
            self._blocks.append(b)
            return b
        self._alloc = alloc
        self.frame_bytes = self.n * channels * sample_bytes
        self.mask_stride, self.filter_stride, self.witness_stride = self.strides(self.n)
        oalloc = self._out_alloc = out_allocator or alloc
        self.planar_luma, self.keep_interleaved, self.resident_gops = bool(planar_luma), bool(keep_interleaved), int(resident_gops)
        if self.planar_luma and self.adaptive is not None:
            raise ValueError("noise-adaptive thresholds read the interleaved frames: planar_luma needs an explicit threshold")
        self.luma_bytes = self.n * sample_bytes
        self.luma = (luma_block if luma_block is not None else alloc(self.luma_bytes * nframes * self.resident_gops)) if self.planar_luma else None
        if frames_block is not None:
            self.frames = frames_block
        elif self.planar_luma and not self.keep_interleaved:
            self.frames = None
        else:
            self.frames = alloc(self.frame_bytes * nframes * self.resident_gops)
        self.masks, self.ones = alloc(self.mask_stride * self.pairs), alloc(8 * self.pairs)
        self.filters, self.witness = oalloc(self.filter_stride * self.pairs), oalloc(self.witness_stride * self.pairs)
        self.stats = oalloc(8 * nat.STATS_PER_FRAME * self.pairs)
        self.params, self.k = (nat.FilterParams * self.pairs)(), (ctypes.c_double * self.pairs)()
        self.record = None
        if self.adaptive is not None:
            self.moments = alloc(16 * self.pairs)
            self.noise_plane = None                       # allocated on the first exact fallback

    def set_run_starts(self, run_starts):
        """Name the keyframes inside the block for the following encode() calls (see the constructor): a coder is sized by its frame
        count, not by where its runs start, so one coder serves every block of a stream whatever the keyframe interval."""
        self.run_starts = None
        self.skipped = [False] * self.pairs
        if run_starts:
            self.run_starts = (ctypes.c_uint8 * self.F)()
            for t in run_starts:
                if not 0 <= int(t) < self.F:
                    raise ValueError("run start %r outside the block of %d frames" % (t, self.F))
                if int(t) > 0:
                    self.run_starts[int(t)] = 1
                    self.skipped[int(t) - 1] = True

    @staticmethod
    def strides(n):
        """(mask, filter, witness) row strides in bytes for frames of n pixels."""
        fstride = (nat.packed_stride(int(n * 0.32) + 64) + 15) // 16 * 16     # l <= 0.317 n for every density
        return nat.packed_stride(n), fstride, nat.packed_stride(n)

    @staticmethod
    def record_bytes(n, pairs):
        """Bytes a TorchArena needs for the output records (filters | witnesses | stats) of `pairs` frames."""
        _, fs, ws = GopCoder.strides(n)
        r = lambda x: (x + 255) // 256 * 256
        return r(fs * pairs) + r(ws * pairs) + r(8 * nat.STATS_PER_FRAME * pairs)

    def close(self):
        """Free the library-owned blocks this coder allocated (torch-backed blocks die with their tensors)."""
        for b in self._blocks:
            if isinstance(b, _OwnedBlock):
                b.buf.free()
        self._blocks = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def load_frames(self, frames, gop=0):
        """Upload one GOP of interleaved (F, H, W, C) frames into slot `gop`.  With planar_luma the Y planes are extracted on
        the device (rbf_extract_luma_batch) when the interleaved frames are kept, else on the host (only luma crosses PCIe)."""
        frames = np.ascontiguousarray(frames)
        assert frames.nbytes == self.frame_bytes * self.F, (frames.shape, frames.dtype)
        assert 0 <= gop < self.resident_gops
        if self.frames is not None:
            dst = self.frames.ptr + gop * self.frame_bytes * self.F
            nat.check(nat.lib().rbf_memcpy_h2d(self.ctx.handle, dst, frames.ctypes.data, frames.nbytes))
            if self.planar_luma:
                nat.check(nat.lib().rbf_extract_luma_batch(self.ctx.handle, dst, self.frame_bytes, self.F, self.W, self.H, self.W * self.C * self.sb,
                                                           self.C * self.sb, self.sb, self.luma.ptr + gop * self.luma_bytes * self.F))
        else:
            self.load_luma(frames.reshape(self.F, self.H, self.W, self.C)[..., 0], gop)

    def load_luma(self, planes, gop=0):
        """Upload one GOP of dense (F, H, W) Y planes (planar_luma coders)."""
        assert self.planar_luma and 0 <= gop < self.resident_gops
        planes = np.ascontiguousarray(planes)
        assert planes.nbytes == self.luma_bytes * self.F, (planes.shape, planes.dtype)
        nat.check(nat.lib().rbf_memcpy_h2d(self.ctx.handle, self.luma.ptr + gop * self.luma_bytes * self.F, planes.ctypes.data, planes.nbytes))

    def _noise(self, first, count, moments_ptr, planes_ptr):
        nat.check(nat.lib().rbf_noise_moments_batch(
            self.ctx.handle, self.frames.ptr + first * self.frame_bytes, self.frame_bytes, count, self.W, self.H,
            self.W * self.C * self.sb, self.C * self.sb, self.sb, moments_ptr, planes_ptr))

    def adaptive_floors(self):
        """Integer thresholds of the reference's adaptive rule for frames 1..F-1 (engine.py has the
        reasoning: exact integer moments decide the floor; the float32 replay is the rare fallback)."""
        from .engine import adaptive_threshold, adaptive_threshold_band, threshold_floor
        self._noise(1, self.pairs, self.moments.ptr, None)
        self.ctx.sync()
        moments = self.moments.numpy(self.ctx)[:16 * self.pairs].view(np.int64).reshape(self.pairs, 2)
        floors = []
        for i in range(self.pairs):
            lo, hi = adaptive_threshold_band(self.n, int(moments[i, 0]), int(moments[i, 1]), *self.adaptive)
            if lo != hi:
                if self.noise_plane is None:
                    self.noise_plane = self._alloc(4 * self.n)
                    self.moments1 = self._alloc(16)
                self._noise(1 + i, 1, self.moments1.ptr, self.noise_plane.ptr)
                self.ctx.sync()
                plane = self.noise_plane.numpy(self.ctx)[:4 * self.n].view(np.float32).reshape(self.H, self.W)
                lo = threshold_floor(adaptive_threshold(np.std(plane), *self.adaptive))
            floors.append(lo)
        return floors

    def encode(self, gop=0):
        """Enqueue one full pass over resident GOP `gop`; returns after the Bloom kernels are enqueued."""
        self.encode_begin(gop)
        self.encode_finish()

    def encode_begin(self, gop=0):
        """First half (rbf_encode_gop_begin): the mask stage of resident GOP `gop` is enqueued; returns without waiting.
        A caller that drives several coders (one context each) begins the next coder's GOP before it finishes this one."""
        if self.adaptive is not None:
            self.thresholds = self.adaptive_floors()
            self.thr_tab = (ctypes.c_int32 * self.pairs)(*self.thresholds)
        self.gop = gop
        if self.planar_luma:                          # dense Y planes: pixel stride = one sample
            src, fstride, pitch, pstride = self.luma.ptr + gop * self.luma_bytes * self.F, self.luma_bytes, self.W * self.sb, self.sb
        else:
            src, fstride, pitch, pstride = self.frames.ptr + gop * self.frame_bytes * self.F, self.frame_bytes, self.W * self.C * self.sb, self.C * self.sb
        nat.check(nat.lib().rbf_encode_runs_begin(
            self.ctx.handle, src, fstride, self.F, self.W, self.H,
            pitch, pstride, self.sb, self.thr, self.thr_tab, self.run_starts, ctypes.byref(self.seeds),
            self.masks.ptr, self.mask_stride, self.ones.ptr,
            self.filters.ptr, self.filter_stride, self.witness.ptr, self.witness_stride, self.stats.ptr))

    def encode_ready(self):
        """True once the mask stage's counts have reached the host (encode_finish will not wait)."""
        ready = ctypes.c_int(0)
        nat.check(nat.lib().rbf_encode_gop_poll(self.ctx.handle, ctypes.byref(ready)))
        return bool(ready.value)

    def encode_finish(self):
        """Second half (rbf_encode_gop_finish): waits for the counts, plans the filters (float64, host), enqueues the Bloom kernels."""
        nat.check(nat.lib().rbf_encode_gop_finish(self.ctx.handle, self.params, self.k))

    def pack(self, block=None):
        """Compact this GOP's output rows into one exact-size record on the device (rbf_pack_records);
        returns the block holding it.  `block`: where to put it (default: a block of the worst-case size)."""
        if block is None:
            if self.record is None:
                self.record = self._out_alloc(int(nat.lib().rbf_record_max_bytes(self.pairs, self.n)))
            block = self.record
        nat.check(nat.lib().rbf_pack_records(
            self.ctx.handle, self.pairs, self.n, self.params, self.k, self.masks.ptr, self.mask_stride,
            self.filters.ptr, self.filter_stride, self.witness.ptr, self.witness_stride, self.stats.ptr,
            block.ptr, block.nbytes // 8 * 8))
        return block

    def gather_values(self, check_uncovered=False):
        """A2 for the whole GOP (rbf_gather_values_batch): list of per-pair arrays of the changed pixels'
        samples (all channels, raster order, frame dtype) taken from frame f+1; with check_uncovered also
        the per-pair count of pixels that changed in some channel although their mask bit is 0."""
        self.ctx.sync()
        ones = self.ones.numpy(self.ctx)[:8 * self.pairs].view(np.uint64)
        total = int(ones.sum())
        if getattr(self, "_values", None) is None or self._values.nbytes < max(8, total * self.C * self.sb):
            self._values = self._alloc(max(8, total * self.C * self.sb))
        if getattr(self, "_voff", None) is None:
            self._voff = self._alloc(8 * (self.pairs + 1))
            self._uncov = self._alloc(8 * self.pairs)
        if self.frames is None:
            raise ValueError("gather_values needs the interleaved frames (keep_interleaved=True)")
        nat.check(nat.lib().rbf_gather_values_batch(
            self.ctx.handle, self.frames.ptr + getattr(self, "gop", 0) * self.frame_bytes * self.F, self.frame_bytes, self.F, self.W, self.H, self.W * self.C * self.sb, self.C * self.sb,
            self.sb, self.C, self.masks.ptr, self.mask_stride, self._values.ptr, total, self._voff.ptr,
            self._uncov.ptr if check_uncovered else None))
        self.ctx.sync()
        off = self._voff.numpy(self.ctx)[:8 * (self.pairs + 1)].view(np.uint64)
        assert int(off[-1]) == total, "mask changed since the ones counts were taken"
        dt = np.uint8 if self.sb == 1 else np.uint16
        flat = self._values.numpy(self.ctx)[:total * self.C * self.sb].view(dt)
        vals = [flat[int(off[f]) * self.C:int(off[f + 1]) * self.C].copy() for f in range(self.pairs)]
        if not check_uncovered:
            return vals
        return vals, self._uncov.numpy(self.ctx)[:8 * self.pairs].view(np.uint64).copy()

    def results_packed(self):
        """What results() returns, through ONE exact-size download: the rows are compacted into a record on the device (rbf_pack_records:
        header + the used bytes of every filter and witness, ~150 KB per 1080p frame instead of ~600 KB of padded rows), the record's
        32-byte header says how many bytes to fetch.  Rows carry `filter` / `witness` (and `mask` only for frames the reference passes
        through uncoded, l == 0) as views of the downloaded record; `ones` comes from the mask stage's counts."""
        from .dist import unpack_device_record, record_used_bytes
        block = self.pack()
        self.ctx.sync()
        used = record_used_bytes(block.numpy(self.ctx, 32))
        raw = block.numpy(self.ctx, (used + 7) // 8 * 8)
        ones = self.ones.numpy(self.ctx)[:8 * self.pairs].view(np.uint64)
        rows = unpack_device_record(raw, self.n, copy=False)
        assert len(rows) == self.pairs
        for f, r in enumerate(rows):
            r["ones"] = int(ones[f])
            if r.get("skipped"):
                assert self.skipped[f] and r["ones"] == 0
        return rows

    def results(self):
        """Download: list of per-frame dicts (mask/filter/witness packed uint8, counts, k, l)."""
        self.ctx.sync()
        def rows(block, stride):                       # blocks may be padded (arena alignment)
            return block.numpy(self.ctx)[:self.pairs * stride].reshape(self.pairs, stride)
        masks = rows(self.masks, self.mask_stride)
        filt = rows(self.filters, self.filter_stride)
        wit = rows(self.witness, self.witness_stride)
        stats = rows(self.stats, 8 * nat.STATS_PER_FRAME).view(np.uint64)
        ones = self.ones.numpy(self.ctx)[:8 * self.pairs].view(np.uint64)
        out = []
        for f in range(self.pairs):
            if self.skipped[f]:                        # the pair across a keyframe: not coded (its mask row is zeros, ones 0)
                assert int(self.params[f].m) == 0 and int(self.params[f].floor_k) == nat.PAIR_SKIPPED and int(ones[f]) == 0
                out.append({"skipped": True, "ones": 0, "k": 0.0, "l": 0, "witness_bits": int(stats[f, 0]), "mask": masks[f, :(self.n + 7) // 8].copy()})
                continue
            m = int(self.params[f].m)
            wb = int(stats[f, 0])
            out.append({"mask": masks[f, :(self.n + 7) // 8].copy(), "ones": int(ones[f]), "k": float(self.k[f]), "l": m,
                        "floor_k": int(self.params[f].floor_k), "threshold": int(self.params[f].threshold),
                        "filter": filt[f, :(m + 7) // 8].copy(), "witness": wit[f, :(wb + 7) // 8].copy(),
                        "witness_bits": wb, "filter_ones": int(stats[f, 1])})
        return out
