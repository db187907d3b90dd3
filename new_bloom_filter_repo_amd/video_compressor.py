"""ImprovedVideoCompressor -- the product surface (improved_video_compressor.py:309-669), with the
inter-frame route the reference never wired in: frame t is a keyframe iff t % keyframe_interval == 0
(zlib, FixedVideoCompressor); every other frame is coded as a luma residual mask against frame t-1
through the GPU Bloom path (VideoFrameCompressor).

Losslessness is kept unconditional, as the reference promises ("True Lossless"): an inter-frame is
only emitted when applying its record to frame t-1 reproduces frame t bit for bit (the luma mask at
threshold 0 must cover every changed pixel); otherwise that frame falls back to a keyframe.

Container: all-keyframe streams are written exactly as the reference does -- 'BFVC' | <I frames |
(<I len | record)* (:398-406) -- so either implementation reads them.  Streams with inter-frames use
magic 'BFV2' and prefix every record with a type byte (1 = keyframe, 2 = inter-frame), following the
type-byte precedent of VideoFrameCompressor.compress_frame (:1053).
"""
import os
import struct
import threading
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import params as P
from .frame_codec import FixedVideoCompressor, VideoFrameCompressor, YUVFrame, build_record, frame_data, parse_record

KEY, INTER = 1, 2
_POPCOUNT8 = np.unpackbits(np.arange(256, dtype=np.uint8)[:, None], axis=1).sum(axis=1).astype(np.uint8)   # numpy 1.x has no bitwise_count


def _popcount(packed):
    """Set bits of a packed uint8 vector."""
    if hasattr(np, "bitwise_count"):             # numpy >= 2: whole 64-bit words
        body = packed[:packed.size // 8 * 8]
        words = body.view(np.uint64) if body.ctypes.data % 8 == 0 else np.frombuffer(body.tobytes(), dtype=np.uint64)
        return int(np.bitwise_count(words).sum(dtype=np.int64)) + int(np.bitwise_count(packed[body.size:]).sum(dtype=np.int64))
    return int(_POPCOUNT8[packed].sum(dtype=np.int64))


def _as_block(data):
    """The frames of a block as ONE contiguous (F, H, W[, C]) array for the upload: a zero-copy view when the caller's frames already lie
    back to back in memory (slices of one decoded clip), a stacked copy otherwise."""
    a = data[0]
    if all(d.flags.c_contiguous for d in data):
        base, nb = a.ctypes.data, a.nbytes
        if nb and all(d.ctypes.data == base + i * nb for i, d in enumerate(data)):
            import ctypes
            raw = np.ctypeslib.as_array((ctypes.c_uint8 * (nb * len(data))).from_address(base))      # (the frames in `data` keep the memory alive)
            return raw.view(a.dtype).reshape((len(data),) + a.shape)
    return np.stack(data)


class _Lane:
    """One GPU lane of the plugin surface: a library context (= one HIP stream) with the GOP coders and the decode engine that live on
    it.  Consecutive blocks of a video alternate over the lanes, each block driven by its own host thread (the C ABI's copies are
    synchronous and ctypes drops the GIL), so block b+1 uploads while block b encodes and block b-1 downloads."""

    def __init__(self, ctx, owned):
        self.ctx, self.owned = ctx, owned
        self.coders = {}
        self.engine = None

    def coder(self, W, H, F, C, sb):
        from .gop import GopCoder
        key = (W, H, F, C, sb)
        c = self.coders.get(key)
        if c is None:
            if len(self.coders) >= 2:            # a stream has at most two block sizes (full blocks and its tail)
                for old in self.coders.values():
                    old.close()
                self.coders = {}
            c = self.coders[key] = GopCoder(self.ctx, W, H, F, channels=C, sample_bytes=sb)
        return c

    def decode_engine(self):
        from .engine import BloomEngine
        if self.engine is None:
            self.engine = BloomEngine(self.ctx)
        return self.engine

    def release(self):
        for c in self.coders.values():
            c.close()
        self.coders = {}
        if self.engine is not None:
            self.engine.close()
            self.engine = None

    def close(self):
        self.release()
        if self.owned:
            self.ctx.close()


def _union_seconds(intervals):
    """Total length of the union of (t0, t1) intervals."""
    total, end = 0.0, None
    for t0, t1 in sorted(intervals):
        if end is None or t0 > end:
            total += t1 - t0
            end = t1
        elif t1 > end:
            total += t1 - end
            end = t1
    return total


class ImprovedVideoCompressor:
    def __init__(self, noise_tolerance=10.0, keyframe_interval=30, min_diff_threshold=3.0,
                 max_diff_threshold=30.0, bloom_threshold_modifier=1.0, batch_size=30,
                 num_threads=None, use_direct_yuv=False, verbose=False, ctx=None, inter_frames=None,
                 gop_batching=True, block_frames=None, gpu_lanes=2):
        """Reference signature (improved_video_compressor.py:318-327) plus five keyword-only extras:
        ctx (library context), gop_batching (False: one set of C-ABI calls per inter-frame instead of one
        per block; both write the same bytes), block_frames (consecutive frames handed to the GPU in ONE
        rbf_encode_runs launch sequence -- several GOPs, cut at the keyframes; default 2 GOPs, at most 128
        frames), gpu_lanes (contexts = HIP streams the blocks alternate over, each block on its own host
        thread: upload, encode and download of neighbouring blocks overlap; 1 = one block at a time) and
        inter_frames -- None (default): YUV input is coded with
        Bloom inter-frames ('BFV2' container, which the reference's decompress_video rejects), anything
        else as keyframes; False: always the reference's all-keyframe 'BFVC' container, readable by the
        reference; True: inter-frames for every colour space (lossless fallback to keyframes per frame)."""
        self.inter_frames = inter_frames
        self.noise_tolerance = noise_tolerance
        self.keyframe_interval = max(1, int(keyframe_interval))
        self.min_diff_threshold = min_diff_threshold
        self.max_diff_threshold = max_diff_threshold
        self.bloom_threshold_modifier = bloom_threshold_modifier
        self.batch_size = batch_size
        self.num_threads = max(1, num_threads or min(64, os.cpu_count() or 1))     # zlib of keyframes / changed values
        self.gop_batching = bool(gop_batching)
        # default block: two keyframe intervals (a multiple of the interval: every block of a stream has the same shape), at most 128 frames.
        # Small blocks start the host's zlib of the changed values early; the GPU's share of a block is a fraction of a millisecond either way.
        self.block_frames = max(2, int(block_frames)) if block_frames else min(128, max(2, 2 * self.keyframe_interval if 2 * self.keyframe_interval <= 128 else self.keyframe_interval))
        self.gpu_lanes = max(1, int(gpu_lanes))
        self.use_direct_yuv = use_direct_yuv
        self.verbose = verbose
        self.compressor = FixedVideoCompressor(verbose=verbose)
        self._ctx = ctx
        self._inter = None
        self.last_compressed_frames = None       # [(type, record bytes)] of the last compress_video call
        self.last_timing = None                  # seconds per stage of the last encode_range / decompress_video (bench.py's e2e_surface leg)
        self.profile_stages = False              # True: synchronise between the stages of a block so that last_timing can tell them apart (bench.py)
        self._lanes = []
        self._tm_lock = threading.Lock()

    def close(self):
        """Return the device memory this compressor holds (the lanes' GOP coders and contexts, the inter-frame codec's
        scratch).  Also happens when the object is dropped; the compressor stays usable afterwards."""
        for lane in self._lanes:
            lane.close()
        self._lanes = []
        if self._inter is not None:
            self._inter.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def inter(self):
        if self._inter is None:
            self._inter = VideoFrameCompressor(keyframe_interval=self.keyframe_interval, use_direct_yuv=True,
                                               verbose=False, ctx=self._ctx)
        return self._inter

    def _get_lanes(self, count):
        """`count` lanes: lane 0 is the caller's context (or the process default), the others are contexts of their own on the same device."""
        from . import _native as nat
        if not self._lanes:
            self._lanes.append(_Lane(self._ctx or nat.default_context(), False))
        while len(self._lanes) < count:
            self._lanes.append(_Lane(nat.Context(self._lanes[0].ctx.device), True))
        return self._lanes[:count]

    def _release_lanes(self):
        for lane in self._lanes:
            lane.release()

    def _tm_add(self, **kw):
        tm = self.last_timing
        if tm is None:
            return
        with self._tm_lock:
            for name, dt in kw.items():
                tm[name] = tm.get(name, 0.0) + dt

    # ------------------------------------------------------------------ encode
    def _encode_inter(self, prev, curr):
        """Record bytes for frame `curr` against `prev`, or None when a keyframe is needed."""
        a, b = frame_data(prev), frame_data(curr)
        if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (np.uint8, np.uint16):
            return None
        if a.ndim == 3 and (a.shape[2] < 3 or a.shape[2] > 4):
            return None
        mask, values, _ = self.inter._calculate_frame_diff(a, b, threshold=0.0)
        changed = (a != b)
        if changed.ndim == 3:
            changed = changed.any(axis=2)
        if np.any(changed & (mask == 0)):        # chroma moved where luma did not: not representable
            return None
        record, _ = self.inter._compress_frame_differences(mask, values)
        return struct.pack("<B", b.dtype.itemsize) + record

    def _encode_block(self, seg, pool, run_starts=(), lane=None, busy=None):
        """Inter-frame records of one block of consecutive frames in one pass over the GPU: seg[0] is only read (a keyframe, or a
        shard's halo frame), every other frame is coded against its predecessor -- except the frames named in `run_starts` (indices
        into seg), which are keyframes of the stream: they start a new run and the pair in front of them is not coded.  One upload,
        ONE rbf_encode_runs launch sequence for all the runs, ONE exact-size download of the packed record (rbf_pack_records), one
        batched gather of the changed values (with the count of changes the luma mask cannot carry); zlib runs in `pool`.
        lane: the context and coders to use (default: lane 0); busy: list that receives the (start, end) time of this block's GPU work.
        Returns a list of futures / None per pair (None = needs a keyframe, or is one), or None when the block cannot be batched
        (mixed shapes or dtypes)."""
        t0 = time.perf_counter()
        data = [frame_data(f) for f in seg]
        a = data[0]
        if a.dtype not in (np.uint8, np.uint16) or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] < 3):
            return None
        if any(d.shape != a.shape or d.dtype != a.dtype for d in data[1:]):
            return None
        H, W = a.shape[:2]
        C = a.shape[2] if a.ndim == 3 else 1
        if C > 4:                                # rbf_gather_values_batch carries at most 4 samples per pixel
            return [None] * (len(seg) - 1)
        if lane is None:
            lane = self._get_lanes(1)[0]
        ctx = lane.ctx
        coder = lane.coder(W, H, len(seg), C, a.dtype.itemsize)
        coder.set_run_starts(list(run_starts))
        block = _as_block(data)
        t1 = time.perf_counter()
        coder.load_frames(block)                 # (synchronous: the frames are pageable host memory)
        t2 = time.perf_counter()
        coder.encode()
        if self.profile_stages:
            ctx.sync()
        t3 = time.perf_counter()
        res = coder.results_packed()
        t4 = time.perf_counter()
        values, uncovered = coder.gather_values(check_uncovered=True)
        t5 = time.perf_counter()
        self._tm_add(stack=t1 - t0, upload=t2 - t1, gpu_encode=t3 - t2, download_rows=t4 - t3, value_gather=t5 - t4)
        if busy is not None:
            busy.append((t1, t5))
        n = H * W
        out = []
        for f, r in enumerate(res):
            if r.get("skipped") or int(uncovered[f]):    # the pair in front of a keyframe; or chroma moved where luma did not: not representable
                out.append(None)
                continue
            p = np.uint64(r["ones"]) / n
            if r["l"]:
                # the filter and the witness were built with the k rbf_plan_batch computed in C: the record carries THAT
                # value (the decoder derives floor_k and T from it), so a last-ulp difference to the Python twin is harmless
                k = r["k"]
                parts = (r["l"], r["filter"], r["witness_bits"], r["witness"])
            else:                                # the reference passes the mask itself through (:215-225)
                k, _l = P.optimal_params(n, p)
                parts = (n, r["mask"], 0, b"")
            vals = values[f]

            def job(p=p, k=k, parts=parts, vals=vals):
                vz = zlib.compress(vals, level=9)
                return struct.pack("<B", vals.dtype.itemsize) + build_record("f64", p, n, k, *parts, len(vals), vz)      # (VideoFrameCompressor's default wire format)
            out.append(pool.submit(job))
        return out

    def encode_range(self, frames, first_index, start, stop, inter_frames=True, release=True):
        """[(type, record)] for the frames with global indices [start, stop); frames[i] is global frame
        first_index + i (a shard passes its halo frame too, dist.halo_start).  Frame t is a keyframe iff
        t % keyframe_interval == 0; the inter-frames are coded in blocks of up to `block_frames` consecutive frames --
        several GOPs per block, ONE launch sequence on the GPU per block, cut at the keyframes.  The blocks alternate over
        `gpu_lanes` contexts, each block on its own host thread; the host's zlib-9 (keyframes: four jobs each; changed values:
        one job per frame) runs on `num_threads` threads under all of it.  release: return the lanes' device memory as soon as the last
        block has left the GPU (False: keep the coders for the next call of the same geometry)."""
        records = {}
        I = self.keyframe_interval
        self.last_timing = tm = {}
        t_all = time.perf_counter()
        busy = []
        with ThreadPoolExecutor(self.num_threads) as pool:
            pending = {}

            def key(t):
                if t not in pending:
                    pending[t] = (KEY, self.compressor.compress_frame_jobs(frames[t - first_index], pool.submit))
            # the keyframes the rule fixes in advance go to the host threads FIRST: their zlib-9 (the longest single jobs, ~0.2 s for a 1080p
            # frame, plus its three planes) then runs under the GPU's blocks instead of behind the last one
            for t in range(start, stop):
                if not inter_frames or t % I == 0 or t - 1 < first_index:
                    key(t)
            blocks = []                                                      # (first frame read, end, run starts)
            t = start
            while t < stop:
                if not inter_frames or t % I == 0 or t - 1 < first_index:
                    t += 1
                    continue
                end = min(stop, t - 1 + self.block_frames)                   # the block reads frames t-1 .. end-1
                blocks.append((t - 1, end, [u - (t - 1) for u in range(t, end) if u % I == 0]))      # keyframes inside the block: new runs
                t = end
            results = [None] * len(blocks)
            if self.gop_batching and blocks:
                import queue
                lanes = self._get_lanes(min(self.gpu_lanes, len(blocks)))
                free = queue.Queue()
                for lane in lanes:
                    free.put(lane)

                def run_block(b):
                    lo, end, starts = blocks[b]
                    lane = free.get()
                    try:
                        return self._encode_block(frames[lo - first_index:end - first_index], pool, starts, lane, busy)
                    finally:
                        free.put(lane)
                if len(lanes) == 1:
                    results = [run_block(b) for b in range(len(blocks))]
                else:
                    with ThreadPoolExecutor(len(lanes)) as gpu_pool:
                        results = list(gpu_pool.map(run_block, range(len(blocks))))
            tm["gpu_phase"] = time.perf_counter() - t_all                    # until the last block's values were on the host
            if release:                                                      # the lanes' blocks of frames, masks, filters and witnesses go back while the
                t_rel = time.perf_counter()                                  # host threads still owe their zlib: not kept between videos
                self._release_lanes()
                tm["release"] = time.perf_counter() - t_rel
            for (lo, end, starts), inter in zip(blocks, results):
                seg = frames[lo - first_index:end - first_index]             # predecessor + the frames lo+1..end-1
                for j in range(1, len(seg)):
                    u = lo + j
                    if u % I == 0:
                        key(u)
                        continue
                    fut = inter[j - 1] if inter is not None else None
                    if inter is None:                                        # not batchable (or gop_batching=False): frame by frame
                        rec = self._encode_inter(seg[j - 1], seg[j])
                        if rec is not None:
                            records[u] = (INTER, rec)
                            continue
                    if fut is not None:
                        pending[u] = (INTER, fut)
                    else:
                        key(u)
            t_wait = time.perf_counter()
            for u, (ty, fut) in pending.items():
                records[u] = (ty, fut() if ty == KEY else fut.result())
            tm["zlib_wait"] = time.perf_counter() - t_wait                   # what the host threads' zlib-9 still owed after the last block left the GPU
        tm["total"] = time.perf_counter() - t_all
        tm["gpu_busy"] = _union_seconds(busy)                                # wall time during which at least one lane had a copy or a kernel in flight
        tm["gpu_busy_frac"] = tm["gpu_busy"] / tm["total"] if tm["total"] > 0 else 0.0
        tm["blocks"], tm["lanes"] = len(blocks), min(self.gpu_lanes, max(1, len(blocks)))
        return [records[u] for u in range(start, stop)]

    def compress_video(self, frames, output_path=None, input_color_space="BGR"):
        """improved_video_compressor.py:358-450, same result dict.  Divergence: with inter-frames enabled
        (see the constructor's `inter_frames`; the default for YUV input) the container is 'BFV2', which only
        this package reads; `inter_frames=False` (or keyframe_interval=1) writes the reference's 'BFVC'."""
        if not frames:
            raise ValueError("No frames provided for compression")
        start = time.time()
        yuv = input_color_space.upper() == "YUV"
        if yuv:
            self.use_direct_yuv = True
            for i in range(len(frames)):
                if not hasattr(frames[i], "yuv_info"):
                    frames[i] = self.compressor.add_yuv_info_to_frame(frames[i])
        original_size = sum(f.nbytes for f in frames)
        use_inter = yuv if self.inter_frames is None else bool(self.inter_frames)
        try:
            records = self.encode_range(frames, 0, 0, len(frames), inter_frames=use_inter)
        finally:
            self._release_lanes()                # (an exception in front of encode_range's own release)
        self.last_compressed_frames = records
        keyframes = sum(1 for ty, _ in records if ty == KEY)
        if output_path:
            blob = self._container(records)
            os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
            with open(output_path, "wb") as f:
                f.write(blob)
            compressed_size = len(blob)
        else:                                    # the container's size without joining ~1 MB per frame into one bytes object nobody asked for
            compressed_size = self._container_size(records)
        ratio = compressed_size / original_size
        elapsed = time.time() - start
        results = {"frame_count": len(frames), "original_size": original_size, "compressed_size": compressed_size,
                   "compression_ratio": ratio, "space_savings": 1.0 - ratio, "compression_time": elapsed,
                   "frames_per_second": len(frames) / elapsed if elapsed > 0 else float("inf"),
                   "keyframes": keyframes, "keyframe_ratio": keyframes / len(frames),
                   "output_path": output_path, "color_space": input_color_space, "overall_ratio": ratio}
        if self.verbose:
            print("\\nCompression Results:")
            print(f"Original Size: {original_size / (1024 * 1024):.2f} MB")
            print(f"Compressed Size: {compressed_size / (1024 * 1024):.2f} MB")
            print(f"Compression Ratio: {ratio:.4f}")
            print(f"Keyframes: {keyframes} ({results['keyframe_ratio'] * 100:.1f}%)")
        return results

    @staticmethod
    def _container(records):
        all_key = all(ty == KEY for ty, _ in records)
        out = [b"BFVC" if all_key else b"BFV2", struct.pack("<I", len(records))]
        for ty, rec in records:
            body = rec if all_key else struct.pack("<B", ty) + rec
            out += [struct.pack("<I", len(body)), body]
        return b"".join(out)

    @staticmethod
    def _container_size(records):
        """len(_container(records)) without building it."""
        extra = 0 if all(ty == KEY for ty, _ in records) else 1
        return 8 + sum(4 + extra + len(rec) for _, rec in records)

    # ------------------------------------------------------------------ decode
    @staticmethod
    def _parse_container(blob):
        magic = blob[:4]
        if magic not in (b"BFVC", b"BFV2"):
            raise ValueError(f"Invalid file format: {magic}")
        (count,) = struct.unpack_from("<I", blob, 4)
        off, records = 8, []
        for _ in range(count):
            (size,) = struct.unpack_from("<I", blob, off)
            body = blob[off + 4: off + 4 + size]
            off += 4 + size
            records.append((KEY, body) if magic == b"BFVC" else (body[0], body[1:]))
        return records

    def decompress_video(self, input_path=None, output_path=None, compressed_frames=None, metadata=None):
        start = time.time()
        t_all = time.perf_counter()
        records = None
        if input_path and os.path.exists(input_path):
            with open(input_path, "rb") as f:
                records = self._parse_container(f.read())
        elif compressed_frames:
            records = [r if isinstance(r, tuple) else (KEY, r) for r in compressed_frames]
        if not records:
            raise ValueError("No compressed frames provided")
        for ty, _ in records:
            if ty not in (KEY, INTER):
                raise ValueError(f"unknown record type {ty}")
        if records[0][0] == INTER:
            raise ValueError("inter-frame without a preceding keyframe")
        self.last_timing = tm = {}
        busy = []
        # the keyframes are independent of everything else: inflate them on the host threads while the inter-frame runs go through the GPU
        key_pool = ThreadPoolExecutor(self.num_threads)
        keys = {j: key_pool.submit(self.compressor.decompress_frame, rec) for j, (ty, rec) in enumerate(records) if ty == KEY}
        runs = []                                                            # (index of the keyframe in front, first record, end)
        i = 0
        while i < len(records):
            if records[i][0] == KEY:
                i += 1
                continue
            j = i
            while j < len(records) and records[j][0] == INTER:
                j += 1
            runs.append((i - 1, i, j))
            i = j
        decoded = {}
        try:
            if self.gop_batching and runs:
                # every run hangs off its own keyframe, so the runs are independent: they alternate over the lanes, each on its own host
                # thread -- the inflate and upload of run r+1 under the device-side rebuild and the download of run r
                import queue
                lanes = self._get_lanes(min(self.gpu_lanes, len(runs)))
                free = queue.Queue()
                for lane in lanes:
                    free.put(lane)

                def run_job(r):
                    k, lo, hi = runs[r]
                    base = keys[k].result()
                    lane = free.get()
                    try:
                        return self._decode_run(base, [rec for _, rec in records[lo:hi]], lane, key_pool, busy)
                    finally:
                        free.put(lane)
                if len(lanes) == 1:
                    outs = [run_job(r) for r in range(len(runs))]
                else:
                    with ThreadPoolExecutor(len(lanes)) as gpu_pool:
                        outs = list(gpu_pool.map(run_job, range(len(runs))))
                for (k, lo, hi), out in zip(runs, outs):
                    decoded[lo] = out
            frames = []
            i = 0
            while i < len(records):
                ty, rec = records[i]
                if ty == KEY:
                    frames.append(keys[i].result())
                    i += 1
                    continue
                j = i
                while j < len(records) and records[j][0] == INTER:
                    j += 1
                if i in decoded:
                    frames += decoded[i]
                else:
                    for _, r in records[i:j]:
                        base = frames[-1]
                        dtype = np.uint8 if r[0] == 1 else np.uint16
                        mask, values = self.inter._decompress_frame_differences(r[1:], base.shape, dtype=dtype)
                        frames.append(self.inter._apply_frame_diff(base, mask, values))
                i = j
        finally:
            key_pool.shutdown(wait=True)
            self._release_lanes()
        tm["total"] = time.perf_counter() - t_all
        tm["gpu_busy"] = _union_seconds(busy)
        tm["gpu_busy_frac"] = tm["gpu_busy"] / tm["total"] if tm["total"] > 0 else 0.0
        tm["runs"], tm["lanes"] = len(runs), min(self.gpu_lanes, max(1, len(runs)))
        if output_path:
            self.save_frames_as_video(frames, output_path)
        if self.verbose:
            print(f"Decompressed {len(frames)} frames in {time.time() - start:.2f} seconds")
        return frames

    def _decode_run(self, base, recs, lane=None, pool=None, busy=None):
        """A run of inter-frame records after `base`: the masks of all Bloom-coded frames are decoded in
        ONE rbf_bloom_decode_batch, the changed values are inflated in threads, and the frames are
        rebuilt in sequence on the device (engine.apply_chain).  lane: the context to use (default: lane 0);
        pool: executor for the inflates (default: a temporary one)."""
        from .engine import apply_chain
        if lane is None:
            lane = self._get_lanes(1)[0]
        t0 = time.perf_counter()
        base_arr = frame_data(base)
        n = base_arr.shape[0] * base_arr.shape[1]
        parsed = []
        for r in recs:
            d = parse_record("f64", r[1:])
            if d["n"] != n:
                raise ValueError("inter-frame record of %d pixels after a frame of %d" % (d["n"], n))
            d["dtype"] = np.uint8 if r[0] == 1 else np.uint16
            parsed.append(d)
        inflate = lambda d: np.frombuffer(zlib.decompress(d["values_z"]), dtype=d["dtype"])[:d["value_count"]]
        own_pool = None
        if pool is None:
            pool = own_pool = ThreadPoolExecutor(self.num_threads)
        val_jobs = [pool.submit(inflate, d) for d in parsed]                 # (run under the mask decode below)
        t1 = time.perf_counter()
        coded = [d for d in parsed if d["witness_bits"] > 0]
        if coded:
            plist = [P.filter_params(d["k"], d["bitmap_bits"]) for d in coded]
            masks = lane.decode_engine().decode(n, plist, [d["bitmap"] for d in coded], [d["witness"] for d in coded])
            for d, m in zip(coded, masks):
                d["mask"] = m
        t2 = time.perf_counter()
        vals = [j.result() for j in val_jobs]
        if own_pool is not None:
            own_pool.shutdown()
        t3 = time.perf_counter()
        masks = [d["mask"] if "mask" in d else d["bitmap"][:(n + 7) // 8] for d in parsed]
        ch = base_arr.shape[2] if base_arr.ndim == 3 else 1
        for i, (m, v) in enumerate(zip(masks, vals)):
            ones = _popcount(np.asarray(m, dtype=np.uint8)[:(n + 7) // 8])
            if len(v) != ones * ch:              # same rule as _apply_frame_diff (:886-903)
                if ch == 1:
                    raise ValueError("changed_values does not match the mask")
                masks[i], vals[i] = np.zeros((n + 7) // 8, np.uint8), v[:0]      # color: frame left untouched
        t4 = time.perf_counter()
        out = apply_chain(lane.ctx, base_arr, masks, vals)
        t5 = time.perf_counter()
        self._tm_add(parse=t1 - t0, mask_decode=t2 - t1, inflate_wait=t3 - t2, check=t4 - t3, apply_chain=t5 - t4)
        if busy is not None:
            busy += [(t1, t2), (t4, t5)]
        return [YUVFrame(f) for f in out] if isinstance(base, YUVFrame) else out

    def verify_lossless(self, original_frames, decompressed_frames):
        return self.compressor.verify_lossless(original_frames, decompressed_frames)

    # ------------------------------------------------------------------ video file I/O (OpenCV, out of scope)
    def save_frames_as_video(self, frames, output_path, fps=30):
        if not frames:
            raise ValueError("No frames provided")
        raise RuntimeError("writing video files needs OpenCV (cv2.VideoWriter, improved_video_compressor.py:525-581), "
                           "which is outside this package's scope; frames are returned as arrays")

    def extract_frames_from_video(self, video_path, max_frames=0, target_fps=None, scale_factor=1.0,
                                  output_color_space="BGR"):
        if not os.path.exists(video_path):
            raise ValueError(f"Video file not found: {video_path}")
        raise RuntimeError("reading video files needs OpenCV (cv2.VideoCapture, improved_video_compressor.py:583-669), "
                           "which is outside this package's scope; pass frames as arrays")
