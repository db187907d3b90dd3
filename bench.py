#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the rational-Bloom insert+query path on MI355X.

A step = one pass of the hot path over one GOP resident in HBM: residual masks of the 29
inter-frames of a 1920x1080 YUV444 30-frame GOP -> ones counts to the host -> filter geometry
(float64, host) -> Bloom insert -> query + witness compaction (BASELINE.json configs[1], k* = 2.3).
With N > 1 every rank encodes its own GOP (independent frames shard; weak scaling) and the
step compacts its per-frame (filter, witness, stats) rows into one exact-size record on the device and
gathers it to rank 0 over RCCL inside the step (asynchronously, overlapping the next step's kernels).

Prints ONE JSON line on rank 0.  `roofline` is the dominant kernel (query) priced at its
ALGORITHMIC bytes (packed mask in + filter in + witness out) against the 8 TB/s HBM peak, from HIP
event timings taken inside the timed region; `cpu_baseline` is the CPU oracle (scalar C port of the
reference algorithm) timed on one host core on the same masks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=30, help="frames per GOP (frames-1 inter-frames are coded)")
    ap.add_argument("--bits", type=int, default=8, choices=(8, 16))
    ap.add_argument("--density", type=float, default=0.0, help="fraction of changed pixels per inter-frame (0 = 0.08889, i.e. k*=2.3; SURVEY 8d density sweep)")
    ap.add_argument("--streams", type=int, default=4, help="GOP pipelines in flight per GPU (each its own HIP stream)")
    ap.add_argument("--lds-tile-kib", type=int, default=0, help="cap the LDS filter tile (KiB) -> tiled kernels; 0 = auto (BASELINE config 4 sweep)")
    ap.add_argument("--gather-every", type=int, default=16, help="N>1: steps whose records travel in one RCCL gather")
    ap.add_argument("--generic-kernels", action="store_true", help="diagnostic: global-memory insert / query kernels instead of the LDS ones")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather to rank 0")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (and gather) even with one rank (smoke-tests the N>1 path)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of CPU-oracle work (0 = auto, about 10-30 s)")
    ap.add_argument("--verify", action="store_true", help="check the first frames against the CPU oracle after timing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        # Finish RCCL's own setup (its streams, channels, proxy threads) with one collective BEFORE the
        # GOP pipelines' streams exist.  Measured (tools/dist_overhead.py): streams created between an
        # eager communicator init and its first collective end up serialised with each other
        # (260 instead of 222 us/step); created after it, or before a lazy init, they overlap.
        dist.barrier()
        torch.cuda.synchronize(device)
    ncoders = max(1, args.streams)
    streams = [torch.cuda.Stream(device) for _ in range(ncoders)]      # none of them is the null stream

    from new_bloom_filter_repo_amd import _native as nat
    from new_bloom_filter_repo_amd.gop import GopCoder, TorchArena, torch_allocator
    from new_bloom_filter_repo_amd.synthetic import make_gop, P_KSTAR_2_3

    W, H, F = args.width, args.height, args.frames
    n, pairs = W * H, F - 1
    dtype = np.uint8 if args.bits == 8 else np.uint16
    use_gather = use_dist and not args.no_gather
    # `--streams` GOP pipelines (default 2): consecutive steps alternate between them, each with its own
    # HIP stream, library context (scratch) and output record, sharing the resident frames.  The
    # HBM-bound mask kernel and the latency-bound compaction / reduce kernels of one step then overlap
    # the integer-issue-bound insert / query kernels of its neighbour, and (N > 1) the async RCCL gather
    # of step s overlaps the kernels of step s+1.  Every step still does all of its work inside the
    # timed region.
    ctxs = [nat.Context(local_rank, s.cuda_stream) for s in streams]
    if args.lds_tile_kib or args.generic_kernels:
        for c in ctxs:
            c.force_generic(((args.lds_tile_kib * 1024 // 256) << 16) | (1 if args.generic_kernels else 0))      # tile unit: 64 dwords
    ctx = ctxs[0]
    arenas = [TorchArena(device, GopCoder.record_bytes(n, pairs)) for _ in range(ncoders)]
    coders = []
    for k in range(ncoders):
        coders.append(GopCoder(ctxs[k], W, H, F, channels=3, sample_bytes=args.bits // 8, allocator=torch_allocator(device),
                               out_allocator=arenas[k], frames_block=coders[0].frames if k else None))
    coder = coders[0]
    density = args.density or P_KSTAR_2_3
    frames = np.stack(make_gop(1000 * 2 + rank, W, H, F, p=density, dtype=dtype))
    coder.load_frames(frames)
    torch.cuda.synchronize(device)

    gather = use_gather
    # N > 1: every step compacts its output rows into an exact-size record on the device
    # (rbf_pack_records) inside an outbox of `--gather-every` slots; a full outbox goes to rank 0 in ONE
    # asynchronous RCCL gather (fewer, larger collectives; two outboxes alternate so packing never waits
    # for a transfer).  RCCL's gather needs one size on all ranks, so the slot size is agreed once, in setup
    # (max of the ranks' record sizes + 2 %); a record that outgrew its slot would be
    # flagged in its header (checked on rank 0 after the timed region).
    G = max(1, args.gather_every)
    record_max = (int(nat.lib().rbf_record_max_bytes(pairs, n)) + 255) // 256 * 256

    class Slot:                                   # what GopCoder.pack needs of a block
        def __init__(self, tensor):
            self.ptr, self.nbytes = tensor.data_ptr(), tensor.numel() * 8

    probe = [Slot(torch.zeros(record_max // 8, dtype=torch.int64, device=device)) for _ in range(ncoders)] if gather else None
    og = None                                     # dist.OutboxGather once the slot size is agreed
    slots = {}
    state = {"s": 0}

    def step():
        k = state["s"] % ncoders
        state["s"] += 1
        with torch.cuda.stream(streams[k]):
            if not gather:
                coders[k].encode()
                return
            t = og.begin(k)                       # this step's slot (waits stream-side for the outbox's previous transfer)
            coders[k].encode()
            coders[k].pack(slots.setdefault(t.data_ptr(), Slot(t)))
            og.end(k)                             # full outbox -> one asynchronous gather from the comm stream

    def drain():
        if og is not None:
            og.flush()

    # setup (not warm-up steps): every pipeline sizes its scratch once, and for N > 1 the ranks agree on the slot
    for k in range(ncoders):
        coders[k].encode()
        if gather:
            coders[k].pack(probe[k])
    torch.cuda.synchronize(device)
    if gather:
        from new_bloom_filter_repo_amd.dist import OutboxGather
        heads = []
        for k in range(ncoders):
            buf = np.zeros(4, dtype=np.uint64)
            nat.check(nat.lib().rbf_memcpy_d2h(ctxs[k].handle, buf.ctypes.data, probe[k].ptr, 32))
            heads.append(int(buf[2]))
        agreed = torch.tensor([max(heads)], dtype=torch.int64, device=device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
        slot_bytes = min(record_max, (int(agreed.item()) * 102 // 100 + 4096 + 255) // 256 * 256)
        og = OutboxGather(slot_bytes // 8, G, device, streams=streams)
        probe = None
        for _ in range(2 * G):                    # untimed: the first RCCL transfer of both outboxes (connection setup)
            step()
        drain()
        state["s"] = 0
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize(device)
    if not args.no_kernel_timing:
        # HIP events around the DOMINANT kernel only (query): two events per step on the launching
        # stream; bracketing every kernel would add ~40 us of event overhead to a ~360 us step
        for c in ctxs:
            c.timing_reset()
            c.timing(1 << nat.K_QUERY)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    # every pipeline coded the same GOP, concurrently with the others: their output records must be identical
    for k in range(1, ncoders):
        if not torch.equal(arenas[0].tensor, arenas[k].tensor):
            raise SystemExit("pipeline %d produced a different record than pipeline 0" % k)
    ktimes = None
    if not args.no_kernel_timing:
        ktimes = {}
        for c in ctxs:
            c.timing(False)
            for name, (ms, cnt) in c.timing_read().items():
                a = ktimes.get(name, (0.0, 0))
                ktimes[name] = (a[0] + ms, a[1] + cnt)
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    breakdown = None
    if not args.no_kernel_timing:
        # per-kernel breakdown from a few extra, untimed steps with every kernel bracketed
        torch.cuda.synchronize(device)
        ctx.timing_reset()
        ctx.timing(True)
        for _ in range(5):
            coder.encode()
        ctx.timing(False)
        breakdown = {k: round(v[0] / 5, 4) for k, v in ctx.timing_read().items() if v[1]}    # one pipeline alone

    res = coder.results()
    pixels_per_step = pairs * n * world
    value = pixels_per_step * args.steps / elapsed / 1e6

    out = {
        "metric": "Mpixels/s Bloom insert+query, 1080p residuals",
        "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%dx%d YUV444 %d-bit synthetic %d-frame GOP (%d inter-frames/step/GPU), %s, threshold 0"
                               % (W, H, args.bits, F, pairs, "p=%g" % args.density if args.density else "k*=2.3"),
                   "pixels_per_step": pixels_per_step, "gather_to_rank0": bool(gather), "gop_pipelines_per_gpu": ncoders, "pipelines_agree": True,
                   "gather_bytes_per_rank_per_step": og.slot_words * 8 if gather else 0, "steps_per_gather": G if gather else 0,
                   "lds_tile_kib": args.lds_tile_kib or "auto", "generic_kernels": bool(args.generic_kernels),
                   "stages": "residual mask -> host params -> insert -> query+witness"},
    }
    if rank == 0 and gather:
        # what arrived on rank 0 is complete: every slot of every rank has the right magic and frame
        # count, no overflow flag, a size that fits the slot; rank 0's own record matches its rows
        from new_bloom_filter_repo_amd.dist import RECORD_MAGIC, unpack_device_record
        sw = og.slot_words
        for ob in range(2):
            for r in range(world):
                heads = og.received(ob, r)[:, :4].cpu().numpy().view(np.uint64)
                for j in range(G):
                    h = heads[j]
                    if int(h[0]) != RECORD_MAGIC or int(h[1]) != pairs or int(h[3]) != 0 or int(h[2]) > sw * 8:
                        raise SystemExit("gathered record of rank %d (outbox %d slot %d) is damaged: %s" % (r, ob, j, h.tolist()))
        mine = og.received(0, 0)[0].cpu().numpy().view(np.uint8)
        for got, want in zip(unpack_device_record(mine, n), res):
            assert got["witness_bits"] == want["witness_bits"] and np.array_equal(got["witness"], want["witness"]), "gathered record differs"
    if rank == 0:
        l_sum = sum(r["l"] for r in res)
        w_sum = sum(r["witness_bits"] for r in res)
        # ALGORITHMIC bytes of the query launch: packed mask in + filter in + witness out
        alg_bytes = pairs * n / 8 + l_sum / 8 + w_sum / 8
        if ktimes and ktimes["query"][1]:
            q_ms = ktimes["query"][0] / ktimes["query"][1]
            achieved = alg_bytes / (q_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_query_lds", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None if args.density else measured_traffic(W, H, F, args.bits),
                               "avg_launch_ms": round(q_ms, 4),
                               "avg_launch_ms_alone": (breakdown or {}).get("query"),   # same kernel, one pipeline, nothing co-running
                               "algorithmic_bytes_per_launch": int(alg_bytes),
                               "bytes_per_pixel": round(alg_bytes / (pairs * n), 4)}
            # the whole fused path priced as SURVEY 8d does: 2 luma reads + packed mask, filter and witness per pixel
            b_px = 2.0 * (args.bits // 8) + alg_bytes / (pairs * n)
            out["roofline"]["end_to_end"] = {"bytes_per_pixel": round(b_px, 3), "achieved": round(value / world * 1e6 * b_px / 1e9, 1),
                                             "unit": "GB/s per GPU", "frac": round(value / world * 1e6 * b_px / 1e9 / HBM_PEAK_GBPS, 4)}
            out["roofline"]["issue"] = None if args.density else issue_roofline(W, H, F, args.bits, (breakdown or {}).get("query"))
            out["kernels_ms_per_step"] = breakdown
        else:
            out["roofline"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res, n, args.cpu_frames)
        if args.verify:
            verify(res, n)
            out["verified_vs_oracle"] = True
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def issue_roofline(W, H, F, bits, alone_ms):
    """Companion to the HBM roofline ('expect the kernel to sit on the integer-ALU ceiling first; report
    both'): the dominant kernel's measured VALU instruction count (committed rocprofv3 PMC pass) priced at
    the measured issue rate of wave64 integer instructions.  Only valid for the workload it was measured on."""
    path = os.path.join(REPO, "profiles", "r01_query_traffic.json")
    if (W, H, F, bits) != (1920, 1080, 30, 8) or not os.path.exists(path) or not alone_ms:
        return None
    with open(path) as f:
        insts = json.load(f).get("sq_insts_valu_per_launch")
    if not insts:
        return None
    bound_ms = insts * 4.0 / 1024 / 2.34e9 * 1e3          # 4 cycles per wave instruction, 1024 SIMDs, 2.34 GHz
    return {"bound": "valu-int", "wave_instructions_per_launch": int(insts), "issue_bound_ms": round(bound_ms, 4),
            "launch_ms_alone": alone_ms, "frac": round(bound_ms / alone_ms, 3)}


def measured_traffic(W, H, F, bits):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_query_traffic.json; FETCH_SIZE / WRITE_SIZE collected in separate --pmc runs and
    corrected as MI355X_MICROARCH.md prescribes).  Only valid for the workload it was measured on."""
    path = os.path.join(REPO, "profiles", "r01_query_traffic.json")
    if (W, H, F, bits) != (1920, 1080, 30, 8) or not os.path.exists(path):
        return None
    with open(path) as f:
        return int(json.load(f)["hbm_bytes_per_launch"])


def cpu_baseline(res, n, nframes):
    """CPU oracle (scalar C port of the reference loops, oracle/rbf_oracle.c) on the step's own masks:
    residual mask is not included (numpy-trivial); insert + query/witness, one core, >= ~10 s of work."""
    import ctypes
    from oracle import oracle as orc
    L = orc.lib()
    seeds = (ctypes.c_uint64 * 3)(*orc.SEEDS_VIDEO)
    frames = [r for r in res[:nframes or len(res)] if r["l"]]
    masks = [np.unpackbits(r["mask"])[:n] for r in frames]
    t_total, px, passes = 0.0, 0, 0
    while t_total < 10.0 and passes < 6:
        for r, mask in zip(frames, masks):
            bit_array = np.zeros(r["l"], dtype=np.uint8)
            witness = np.zeros(n, dtype=np.uint8)
            t0 = time.perf_counter()
            L.orc_compress(mask.ctypes.data, n, r["l"], ctypes.c_double(r["k"]), seeds, bit_array.ctypes.data, witness.ctypes.data)
            t_total += time.perf_counter() - t0
            px += n
        passes += 1
    # the same port with the frames spread over the host's cores (frames are independent; ctypes drops the GIL)
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(len(frames), os.cpu_count() or 1))

    def one(i):
        r, mask = frames[i], masks[i]
        bit_array = np.zeros(r["l"], dtype=np.uint8)
        witness = np.zeros(n, dtype=np.uint8)
        L.orc_compress(mask.ctypes.data, n, r["l"], ctypes.c_double(r["k"]), seeds, bit_array.ctypes.data, witness.ctypes.data)
    with ThreadPoolExecutor(threads) as pool:
        t0 = time.perf_counter()
        list(pool.map(one, range(len(frames))))
        t_all = time.perf_counter() - t0
    return {"value": round(px / t_total / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d passes over the step's %d masks (%d pixels each), insert+query/witness in the scalar C oracle, %.1f s"
                      % (passes, len(frames), n, t_total),
            "all_cores": {"value": round(len(frames) * n / t_all / 1e6, 1), "unit": "Mpixel/s", "cores": threads, "host_cpus": os.cpu_count(),
                          "sample": "one pass, one frame per thread, %.1f s" % t_all},
            "reference_python_mpixels_per_s": 0.38}


def verify(res, n):
    from oracle import oracle as orc
    for r in res[:2]:
        mask = np.unpackbits(r["mask"])[:n]
        bm, wit, p, _, _ = orc.compress(mask)
        assert np.array_equal(np.unpackbits(r["filter"])[:r["l"]], bm)
        assert r["witness_bits"] == len(wit)
        assert np.array_equal(np.unpackbits(r["witness"])[:len(wit)], np.array(wit, dtype=np.uint8))


if __name__ == "__main__":
    main()
