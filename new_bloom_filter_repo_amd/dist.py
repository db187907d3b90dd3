"""Multi-GPU sharding of the residual coder: one process per GPU, torch.distributed (RCCL = backend
"nccl" on ROCm; "gloo" on CPU for tests).

Frames are independent units on the encoder side (inter-frame t needs only ORIGINAL frames t-1 and
t), so a video shards by contiguous frame ranges with one halo frame and there is no data-path
collective: the only exchange is the gather of the variable-length per-frame records to rank 0,
which writes the container.  RCCL has no gatherv, so lengths are all-gathered first and the payloads then
travel as grouped point-to-point messages of exactly their used size (xGMI gives every peer its own link into
rank 0); rank 0's own records never enter a collective.
"""
import os
import struct
import time

import numpy as np

# Every host-side wait of the gathers is BOUNDED: a peer that died (or a link that does not come up) must surface as an exception on the
# survivors -- with the phase and the ranks that did not answer -- not as a process that sits in a collective until somebody kills it.
# RCCL enqueues its collectives on its own stream and returns; what blocks the host is the read of the result (`.cpu()`), so every
# exchange is issued async_op=True and polled (`is_completed`) against this deadline before anything is read.  gloo blocks inside the
# call and has its own process-group timeout; polling the async work bounds it the same way.
DEFAULT_TIMEOUT_S = float(os.environ.get("RBF_DIST_TIMEOUT_S", "120"))


class CollectiveTimeout(RuntimeError):
    """A collective or point-to-point exchange did not complete within the bound (see DEFAULT_TIMEOUT_S / RBF_DIST_TIMEOUT_S)."""


def wait_work(works, what, timeout_s=None, group=None):
    """Host-side bounded wait for async collectives / grouped point-to-point works; re-raises what the backend recorded.
    RCCL ("nccl"): the work is polled (`is_completed` queries its end event; the first millisecond is a plain spin, so a barrier inside a
    timed region costs what dist.barrier() costs).  gloo: `wait(timeout)` -- gloo's send / receive works only complete inside wait()."""
    import datetime
    import torch.distributed as dist
    bound = DEFAULT_TIMEOUT_S if timeout_s is None else timeout_s
    t0 = time.monotonic()
    deadline = t0 + bound
    polled = dist.get_backend(group) == "nccl"
    pause = 50e-6
    for w in ([works] if not isinstance(works, (list, tuple)) else works):
        if w is None:
            continue
        if not polled:
            left = max(0.05, deadline - time.monotonic())
            try:
                if w.wait(datetime.timedelta(seconds=left)) is False:
                    raise CollectiveTimeout("%s did not complete within %.0f s: a peer is gone or never arrived" % (what, bound))
            except RuntimeError as e:
                if "imeout" in str(e) or "imed out" in str(e):
                    raise CollectiveTimeout("%s did not complete within %.0f s: a peer is gone or never arrived (%s)" % (what, bound, str(e)[:200])) from e
                raise
            continue
        while not w.is_completed():
            now = time.monotonic()
            if now > deadline:
                raise CollectiveTimeout("%s did not complete within %.0f s: a peer is gone or never arrived" % (what, bound))
            if now - t0 > 1e-3:
                time.sleep(pause)
                pause = min(pause * 2, 2e-3)
        w.wait()                                   # completed: a stream-side wait under RCCL, and the place where a backend error surfaces


def shard_range(nframes, world, rank):
    """Contiguous [start, stop) of frame indices coded by `rank` (sizes differ by at most one)."""
    base, extra = divmod(nframes, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def halo_start(start, keyframe_interval):
    """First frame a rank must READ: its first coded frame, or the one before when that frame is an
    inter-frame (it is diffed against the original frame start-1)."""
    return start if start == 0 or start % keyframe_interval == 0 else start - 1


def pack_records(records):
    """[(frame_index, type, bytes)] -> one uint8 array: count | (index, type, len, payload)*."""
    parts = [struct.pack("<I", len(records))]
    for t, ty, rec in records:
        parts += [struct.pack("<IBI", t, ty, len(rec)), bytes(rec)]
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def unpack_records(buf):
    buf = bytes(buf)
    (count,) = struct.unpack_from("<I", buf, 0)
    off, out = 4, []
    for _ in range(count):
        t, ty, ln = struct.unpack_from("<IBI", buf, off)
        off += 9
        out.append((t, ty, buf[off:off + ln]))
        off += ln
    return out


RECORD_MAGIC = 0x3130434552464252          # "RBFREC01", csrc/rbf_kernels_pack.h


def record_used_bytes(buf):
    """Used bytes of a device-packed record (its header is enough: pass at least the first 32 bytes)."""
    head = np.frombuffer(bytes(buf[:32]), dtype="<u8")
    if int(head[0]) != RECORD_MAGIC:
        raise ValueError("not a packed record")
    return int(head[2])


def unpack_device_record(buf, n, copy=True):
    """Parse the block rbf_pack_records wrote (GopCoder.pack): list of per-frame dicts with the
    packed filter / witness bytes (numpy.packbits order), or the packed mask for frames the
    reference does not Bloom-code (l == 0).  buf: bytes-like, numpy array or (device) tensor.
    copy=False: the rows' arrays are views of `buf` (a numpy uint8 array the caller keeps alive)."""
    if hasattr(buf, "cpu"):
        buf = buf.cpu().numpy()
    raw = np.frombuffer(bytes(buf), dtype=np.uint8) if copy or not isinstance(buf, np.ndarray) else buf.view(np.uint8).reshape(-1)
    take = (lambda a: a.copy()) if copy else (lambda a: a)
    head = raw[:32].view("<u8")
    if int(head[0]) != RECORD_MAGIC:
        raise ValueError("not a packed record")
    nframes, used, overflow = int(head[1]), int(head[2]), int(head[3])
    if overflow or used > raw.size:
        raise ValueError("packed record is truncated: %d bytes used, %d present%s" % (used, raw.size, " (overflow flagged)" if overflow else ""))
    rows = raw[32:32 + 64 * nframes].view("<u8").reshape(nframes, 8)
    out = []
    for m, floor_k, thr, kbits, wbits, fones, foff, woff in rows.tolist():
        k = struct.unpack("<d", struct.pack("<Q", kbits))[0]
        rec = {"l": m, "floor_k": floor_k, "threshold": thr, "k": k, "witness_bits": wbits, "filter_ones": fones}
        if m == 0 and floor_k == 0xFFFFFFFF:       # a pair across a keyframe of a multi-run block (rbf_encode_runs): header row only
            rec["skipped"] = True
            out.append(rec)
            continue
        if m:
            rec["filter"] = take(raw[foff:foff + (m + 7) // 8])
        else:
            rec["mask"] = take(raw[foff:foff + (n + 7) // 8])
        rec["witness"] = take(raw[woff:woff + (wbits + 7) // 8])
        out.append(rec)
    return out


class OutboxGather:
    """Batched asynchronous EXACT-SIZE gather of packed records to rank 0 (bench.py's N > 1 path).

    Every step packs one record into the next slot of an outbox of `steps_per_gather` worst-case-size slots; a full
    outbox travels in one exchange on a dedicated communication stream: the used sizes are read from the slots'
    headers (one small device-to-host copy on that stream), all-gathered, and every rank but `dst` sends the used
    bytes of its slots as ONE point-to-point message while `dst` posts one receive per peer (batch_isend_irecv =
    ncclGroupStart/End on RCCL; over xGMI every peer has its own link into `dst`).  Nothing is padded to a common
    size, so a record cannot overflow a slot, and the records of `dst` itself never enter a collective.  Two
    outboxes alternate so that packing never waits for a transfer; the exchange itself (two small host reads: the
    used sizes, then everybody's sizes) runs on a helper thread, so the launching thread is never blocked by it; a
    damaged record raises on EVERY rank, after the size collective.  On CPU (gloo, tests) the same bookkeeping runs
    without streams (and, by default, without the thread).

        slot = og.begin(k)      # on pipeline k's stream: where this step's record goes (int64 tensor)
        ... enqueue the writes into `slot` on pipeline k's stream ...
        og.end(k)               # record written; sends the outbox when this was its last slot
        og.flush()              # send a partly filled outbox, wait for everything in flight
        og.received(ob, rank)   # on dst: the records last received from `rank` in outbox `ob` (list of uint8 tensors)
    """

    def __init__(self, slot_words, steps_per_gather, device, streams=None, group=None, dst=0, threaded=None, timeout_s=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self.group, self.dst = torch, dist, group, dst
        self.timeout_s = timeout_s                 # bound of every host-side wait (None: DEFAULT_TIMEOUT_S)
        self.G, self.slot_words = max(1, int(steps_per_gather)), int(slot_words)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cuda = device.type == "cuda"
        self.device = device
        self.out = [torch.zeros(self.G, self.slot_words, dtype=torch.int64, device=device) for _ in range(2)]
        self.inbox = [[torch.empty(self.G * self.slot_words * 8, dtype=torch.uint8, device=device) if r != dst else None
                       for r in range(self.world)] if self.rank == dst else None for _ in range(2)]
        self.sizes = [None, None]       # per outbox on dst: [rank][slot] used bytes of the last exchange
        self.filled = [0, 0]
        self.pend = [None, None]        # per outbox: (works, keep-alive tensors, event)
        self.s = 0                      # steps begun so far (position in the slot sequence)
        self.sent = 0
        self.bytes_sent = 0
        self.streams = list(streams) if streams else []
        self.comm = torch.cuda.Stream(device) if self.cuda else None
        # The exchange reads the used sizes on the host (device -> host copy, then the all-gathered sizes): two blocking reads.
        # They run on a HELPER THREAD, so the thread that launches the pipelines' kernels never waits for a transfer it does not
        # need (round 2 did both reads inline: a pipeline drain every `steps_per_gather` steps).  The helper executes the
        # exchanges strictly in submission order -- every rank submits the same sequence, so the collectives match up.
        self.threaded = bool(self.cuda if threaded is None else threaded)
        self.error = None
        self._posted = [None, None]     # per outbox: threading.Event set once its exchange has been posted (or failed)
        self._jobs = None
        if self.threaded:
            import queue
            import threading
            self._jobs = queue.Queue()
            self._worker = threading.Thread(target=self._run, name="rbf-outbox-gather", daemon=True)
            self._worker.start()

    def _where(self):
        return self.s % self.G, (self.s // self.G) % 2

    def _peer(self, r):
        """Global rank of group rank r (P2POp wants global ranks)."""
        return r if self.group is None else self._dist.get_global_rank(self.group, r)

    def _run(self):
        if self.cuda:
            self._torch.cuda.set_device(self.device)          # the current device is per thread
        while True:
            job = self._jobs.get()
            if job is None:
                return
            ob, count, ready, posted = job
            try:
                # After an error EVERY rank has raised in the same exchange (the flag travels in the size collective), so every rank
                # skips the same later jobs: nobody is left alone in a collective.
                if self.error is None:
                    self._post(ob, count, ready)
            except BaseException as e:                          # surfaced by the next begin() / flush() of the owning thread
                self.error = e
            finally:
                posted.set()

    def _settle(self, ob):
        """Wait (host) until outbox ob's last exchange has been posted; re-raise what the helper thread caught."""
        ev = self._posted[ob]
        if ev is not None and not ev.wait(2 * (DEFAULT_TIMEOUT_S if self.timeout_s is None else self.timeout_s) + 5):
            raise CollectiveTimeout("OutboxGather: the helper thread did not post the exchange of outbox %d" % ob)
        if self.error is not None:
            raise self.error

    def close(self):
        if self._jobs is not None:
            self._jobs.put(None)
            self._worker.join()
            self._jobs = None

    def _wait(self, ob):
        self._settle(ob)
        if self.pend[ob] is not None:
            works, _keep, ev = self.pend[ob]
            wait_work(works, "OutboxGather: payload exchange of outbox %d" % ob, self.timeout_s, self.group)
            if ev is not None:
                self._torch.cuda.current_stream(self.device).wait_event(ev)
            self.pend[ob] = None

    def begin(self, k=0):
        j, ob = self._where()
        # the first write of every writer stream into this outbox waits for the outbox's previous transfer
        if j < max(1, len(self.streams)):
            if self.cuda and self.streams:
                with self._torch.cuda.stream(self.streams[k]):
                    self._wait_stream(ob, self.streams[k])
            else:
                self._wait(ob)
        return self.out[ob][j]

    def _wait_stream(self, ob, stream):
        self._settle(ob)
        if self.pend[ob] is not None:
            works, _keep, ev = self.pend[ob]
            for w in works:
                w.wait()                                   # NCCL work: makes the current stream wait, does not block the host
            if ev is not None:
                stream.wait_event(ev)

    def end(self, k=0):
        j, ob = self._where()
        self.s += 1
        self.filled[ob] = j + 1
        if j == self.G - 1:
            self._send(ob)

    def _exchange(self, ob, count):
        torch, dist = self._torch, self._dist
        # Everything that can fail on THIS rank alone -- the header read, the checks, the payload's allocation -- happens in front of
        # the size collective, and its failure travels IN the collective (flag 2; 1 = damaged record): the other ranks are already
        # in, or on their way into, that all_gather and would wait for ever for a rank that raised before it.
        used, flag, local, payload = [0] * count, 0, None, None
        try:
            heads = self.out[ob][:count, :4].cpu().numpy().view(np.uint64)          # blocks this (helper) thread only
            self._hook("heads")
            for j, h in enumerate(heads):
                if int(h[0]) != RECORD_MAGIC or int(h[3]) != 0 or int(h[2]) > self.slot_words * 8:
                    flag = 1
                else:
                    used[j] = (int(h[2]) + 7) // 8 * 8
            if flag:
                used = [0] * count
            elif self.rank != self.dst and sum(used):
                payload = torch.cat([self.out[ob][j, :u // 8].view(torch.uint8) for j, u in enumerate(used)])
        except BaseException as e:
            used, flag, local, payload = [0] * count, 2, e, None
        mine = torch.zeros(self.G + 1, dtype=torch.int64, device=self.device)
        mine[:count] = torch.tensor(used, dtype=torch.int64, device=self.device)
        mine[self.G] = flag
        sizes = [torch.zeros(self.G + 1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        wait_work(dist.all_gather(sizes, mine, group=self.group, async_op=True), "OutboxGather: all_gather of the used sizes (exchange %d)" % self.sent, self.timeout_s, self.group)
        sizes = [[int(x) for x in t.cpu().tolist()] for t in sizes]
        if local is not None:
            raise local                                         # after the collective: the peers have seen the flag
        bad = [r for r in range(self.world) if sizes[r][self.G]]
        if bad:
            raise ValueError("a packed record in the outbox of rank(s) %s is damaged, or the rank failed before the exchange" % bad)      # on EVERY rank
        sizes = [t[:self.G] for t in sizes]
        ops, keep = [], []
        if self.rank == self.dst:
            self.sizes[ob] = sizes
            for r in range(self.world):
                if r != self.dst and sum(sizes[r]):
                    ops.append(dist.P2POp(dist.irecv, self.inbox[ob][r][:sum(sizes[r])], self._peer(r), self.group))
        elif payload is not None:
            keep.append(payload)
            ops.append(dist.P2POp(dist.isend, payload, self._peer(self.dst), self.group))
            self.bytes_sent += int(payload.numel())
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, keep

    def _hook(self, where):
        """Test seam (tests/test_dist_cpu.py makes one rank's helper thread fail here, in front of the collective)."""

    def _post(self, ob, count, ready):
        """Exchange `count` slots of outbox ob: on the comm stream behind `ready` (the writers' events).  Helper thread (or inline)."""
        if self.cuda:
            with self._torch.cuda.stream(self.comm):
                for ev in ready:
                    self.comm.wait_event(ev)
                works, keep = self._exchange(ob, count)
                ev = self._torch.cuda.Event()
                ev.record(self.comm)
                self.pend[ob] = (works, keep, ev)
        else:
            works, keep = self._exchange(ob, count)
            self.pend[ob] = (works, keep, None)

    def _send(self, ob):
        count = self.filled[ob]
        # the writers' events are re-recorded by later steps: hand the exchange its own snapshot
        ready = []
        if self.cuda:
            for k, _ in enumerate(self.streams):
                ev = self._torch.cuda.Event()
                ev.record(self.streams[k])
                ready.append(ev)
        if self.threaded:
            import threading
            posted = threading.Event()
            self.pend[ob] = None
            self._posted[ob] = posted
            self._jobs.put((ob, count, ready, posted))
        else:
            self._posted[ob] = None
            self._post(ob, count, ready)
        self.sent += 1

    def flush(self):
        j, ob = self._where()
        if j:                           # a partly filled outbox: only its filled slots travel
            self._send(ob)
            self.s += self.G - j
        for ob in range(2):
            self._wait(ob)
        if self.cuda:
            done = self._torch.cuda.Event()
            done.record(self.comm)
            deadline = time.monotonic() + (DEFAULT_TIMEOUT_S if self.timeout_s is None else self.timeout_s)
            while not done.query():
                if time.monotonic() > deadline:
                    raise CollectiveTimeout("OutboxGather.flush: the communication stream did not drain")
                time.sleep(1e-4)
        if self.error is not None:
            raise self.error

    def received(self, ob, rank):
        """On dst: the records of the last exchange of outbox `ob` that came from `rank`, as a list of uint8 tensors of
        exactly the used size (dst's own records are views of its outbox: they never travelled)."""
        sizes = self.sizes[ob][rank]
        if rank == self.dst:
            return [self.out[ob][j, :u // 8].view(self._torch.uint8) for j, u in enumerate(sizes) if u]
        out, off = [], 0
        for u in sizes:
            if u:
                out.append(self.inbox[ob][rank][off:off + u])
                off += u
        return out


def gather_records(records, dst=0, group=None, device=None, timeout_s=None):
    """Gather every rank's [(frame_index, type, bytes)] to `dst`; returns the merged list sorted by
    frame index on dst, None elsewhere.  Works on any backend (tensors live on `device`).  Exact sizes, like
    gather_device_records: the lengths are all-gathered, then every rank but `dst` sends its bytes as one point-to-point
    message and `dst` posts one receive per peer; dst's own records stay where they are, nothing is padded."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    packed = pack_records(records)
    payload = torch.from_numpy(packed).to(device)
    length = torch.tensor([payload.numel()], dtype=torch.int64, device=device)
    lengths = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    wait_work(dist.all_gather(lengths, length, group=group, async_op=True), "gather_records: all_gather of the record lengths", timeout_s, group)
    lengths = [int(x.item()) for x in lengths]
    ops, inbox = [], {}
    if rank == dst:
        for r in range(world):
            if r != dst:
                inbox[r] = torch.empty(lengths[r], dtype=torch.uint8, device=device)
                ops.append(dist.P2POp(dist.irecv, inbox[r], peer(r), group))
    else:
        ops.append(dist.P2POp(dist.isend, payload, peer(dst), group))
    if ops:
        wait_work(dist.batch_isend_irecv(ops), "gather_records: payload exchange with rank %d" % dst, timeout_s, group)
    if rank != dst:
        return None
    merged = []
    for r in range(world):
        merged += unpack_records(packed.tobytes() if r == dst else inbox[r].cpu().numpy().tobytes())
    merged.sort(key=lambda x: x[0])
    return merged


def gather_device_records(records, device, dst=0, group=None, max_records=None, timeout_s=None):
    """Exact-size gather of device-packed records (GopCoder.pack) to `dst`, RCCL has no gatherv:
    every rank reads the used size of its records from their headers (one small D2H), the sizes are
    all-gathered, then every rank but `dst` sends the used bytes of its records as ONE message and `dst`
    posts one receive per peer -- grouped point-to-point (batch_isend_irecv = ncclGroupStart/End on RCCL),
    so over xGMI each peer uses its own link into `dst`.  The records of `dst` itself never enter a
    collective.  Nothing is padded to a common slot size, so a record cannot overflow.

    records: list of int64 device tensors (one packed record each; any slack after the used bytes is ignored).
    Returns on dst a list (rank-major, record order kept) of uint8 device tensors, one exact-size record each;
    None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    if dist.get_backend(group) == "gloo":          # gloo moves host memory: sizes and payloads are staged through the CPU (what arrives on
        device = torch.device("cpu")               # dst from a peer is a host tensor then; dst's own records stay where they are)
    damaged = 0
    if records:
        heads = torch.stack([r[:4] for r in records]).cpu().numpy().view(np.uint64)
        damaged = int(any(int(h[0]) != RECORD_MAGIC or int(h[3]) != 0 for h in heads))
        used = [0 if damaged else (int(h[2]) + 7) // 8 * 8 for h in heads]
    else:
        used = []
    if max_records is None:                                        # callers that know the largest record count of any rank skip this round
        maxrec = torch.tensor([len(used)], dtype=torch.int64, device=device)
        wait_work(dist.all_reduce(maxrec, op=dist.ReduceOp.MAX, group=group, async_op=True), "gather_device_records: all_reduce of the record counts", timeout_s, group)
        max_records = int(maxrec.item())
    if len(used) > max_records:
        raise ValueError("%d records, but max_records = %d" % (len(used), max_records))
    mine = torch.zeros(max_records + 1, dtype=torch.int64, device=device)
    if used:
        mine[:len(used)] = torch.tensor(used, dtype=torch.int64, device=device)
    mine[max_records] = damaged                                   # sizes + error flag in ONE collective: nobody is left hanging
    sizes = [torch.zeros(max_records + 1, dtype=torch.int64, device=device) for _ in range(world)]
    wait_work(dist.all_gather(sizes, mine, group=group, async_op=True), "gather_device_records: all_gather of the record sizes", timeout_s, group)
    sizes = [[int(x) for x in s.cpu().tolist()] for s in sizes]
    bad = [r for r in range(world) if sizes[r][max_records]]
    if bad:
        raise ValueError("a record to gather is damaged or truncated on rank(s) %s" % bad)
    sizes = [[x for x in s[:max_records] if x] for s in sizes]
    own = [r[:u // 8].view(torch.uint8) for r, u in zip(records, used)]
    ops, inbox = [], {}
    if rank == dst:
        for r in range(world):
            if r != dst and sum(sizes[r]):
                inbox[r] = torch.empty(sum(sizes[r]), dtype=torch.uint8, device=device)
                ops.append(dist.P2POp(dist.irecv, inbox[r], peer(r), group))
    elif used:
        payload = torch.cat(own).to(device)
        ops.append(dist.P2POp(dist.isend, payload, peer(dst), group))
    if ops:                                                        # (bounded: what arrives is read by the caller right away)
        wait_work(dist.batch_isend_irecv(ops), "gather_device_records: payload exchange with rank %d (peers with data: %s)"
                  % (dst, [r for r in range(world) if r != dst and sum(sizes[r])]), timeout_s, group)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        if r == dst:
            out += own
            continue
        off = 0
        for u in sizes[r]:
            out.append(inbox[r][off:off + u])
            off += u
    return out


def encode_video_sharded(frames, first_index, nframes_total, keyframe_interval=30, ctx=None, dst=0, group=None):
    """Code this rank's shard and gather the container records on rank `dst`.

    frames: the frames this rank READS, i.e. global indices [halo_start(start), stop) where
    (start, stop) = shard_range(nframes_total, world, rank); first_index = halo_start(start).
    Returns the container bytes on dst (ImprovedVideoCompressor._container), None elsewhere."""
    import torch.distributed as dist
    from .video_compressor import ImprovedVideoCompressor
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    start, stop = shard_range(nframes_total, world, rank)
    comp = ImprovedVideoCompressor(keyframe_interval=keyframe_interval, ctx=ctx)
    try:
        coded = comp.encode_range(frames, first_index, start, stop)      # blocks of two keyframe intervals over two GPU lanes; the lanes' memory is released inside
    finally:
        comp.close()
    records = [(start + i, ty, rec) for i, (ty, rec) in enumerate(coded)]
    merged = gather_records(records, dst=dst, group=group)
    if merged is None:
        return None
    assert [t for t, _, _ in merged] == list(range(nframes_total)), "missing frames in the gather"
    return ImprovedVideoCompressor._container([(ty, rec) for _, ty, rec in merged])
