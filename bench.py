#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the rational-Bloom insert+query path on MI355X.

A step = one pass of the hot path over one GOP resident in HBM: residual masks of the 29 inter-frames of a 1920x1080 YUV444 30-frame
GOP -> ones counts to the host -> filter geometry (float64, host) -> Bloom insert -> query + witness compaction (BASELINE.json
configs[1], k* = 2.3).  `--streams` GOP pipelines are in flight per GPU, EACH WITH ITS OWN GOPs (distinct synthetic frames, ~750 MB of
resident input: the mask kernel reads HBM, not the Infinity Cache).  `--gops-per-call G` makes a step ONE rbf_encode_runs call over G GOPs
(one launch sequence per G x 29 inter-frames); the default stays 1 and the default run adds a `batched_gops` leg with 4.

Timing protocol: --warmup steps, then untimed settling (windows of --steps steps until two agree within 2 % and 30 ms have gone by),
then NINE regions of EXACTLY --steps steps, each between barrier + synchronize, max over ranks; the MEDIAN region is the headline
(`steps` = --steps, `regions_ms` lists all nine).  One long region follows as `steady_state`.

With N > 1 every rank encodes its own GOPs (weak scaling), compacts each step's rows into one exact-size record on the device and
gathers it to rank 0 over RCCL inside the step.  `python bench.py --gpus N` without a launcher environment starts the N ranks ITSELF (one
process per GPU, RCCL over 127.0.0.1); under `python -m torch.distributed.run --nproc-per-node N` it joins the launcher's ranks.
`--backend gloo --one-device` lets the N ranks share GPU 0 (records staged through host memory): the sharded path with the real kernels
on a 1-GPU box.  After the headline the same process group runs BASELINE configs[2] and [4] as written -- ONE clip of 300 frames (keyframe
every 30; 8-bit, then 16-bit) sharded by frame over the ranks, every rank's frames in equal multi-GOP blocks (strong scaling, run_clip())
-- as `clip300` / `clip300_uint16`; `--clip-frames F` runs only that mode.  A single-GPU default run also carries the legs
`decode_1080p`, `interleaved_yuv444`, `batched_gops`, `config4_2160p` (+ `_gop9`) and `e2e_surface` (the plugin surface end to end).

Prints ONE JSON line on rank 0.  After the timed regions every GOP of every pipeline is compared with the CPU oracle
(`verified_vs_oracle`); so is every leg.  `roofline` prices the dominant kernel (query) at its ALGORITHMIC bytes (packed mask in + filter
in + witness out) against the 8 TB/s HBM peak, from its UNCONTENDED launch time (HIP events on the launching stream, 50 launches, one
pipeline alone); `traffic` and `issue` are constants replayed from profiles/ and tagged as such.  `cpu_baseline` is the CPU oracle
(scalar C port of the reference algorithm) timed on the host.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COMM_DEVICE = None              # where small control tensors of the collectives live: the GPU under RCCL, host memory under gloo (init_dist)
MIN_REGION_S = 0.060            # shortest timed region that is reported as the headline
ALONE_LAUNCHES = 50             # launches behind every "alone" per-kernel figure


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=30, help="frames per GOP (frames-1 inter-frames are coded)")
    ap.add_argument("--bits", type=int, default=8, choices=(8, 16))
    ap.add_argument("--density", type=float, default=0.0, help="fraction of changed pixels per inter-frame (0 = 0.08889, i.e. k*=2.3; SURVEY 8d density sweep)")
    ap.add_argument("--streams", type=int, default=4, help="GOP pipelines in flight per GPU (each its own HIP stream, context and GOP)")
    ap.add_argument("--interleaved", action="store_true", help="keep the GOPs as interleaved YUV444 frames (round 1/2 layout: the mask kernel reads 3x its algorithmic bytes) instead of dense Y planes")
    ap.add_argument("--gops-per-pipeline", type=int, default=0, help="resident GOPs a pipeline rotates over (0 = auto: 3 with planar Y, 1 interleaved -- ~750 MB of resident input either way)")
    ap.add_argument("--shared-gop", action="store_true", help="diagnostic: all pipelines read the SAME resident GOP (the round-1 setup; inputs then fit the Infinity Cache)")
    ap.add_argument("--lds-tile-kib", type=int, default=0, help="cap the LDS filter tile (KiB) -> tiled kernels; 0 = auto (BASELINE config 4 sweep)")
    ap.add_argument("--gather-every", type=int, default=16, help="N>1: steps whose records travel in one RCCL gather")
    ap.add_argument("--rebuild-hash-table", action="store_true", help="diagnostic: run k_hash_table in every step instead of taking the table the previous step's query kernel wrote")
    ap.add_argument("--force-bits", type=int, default=0, help="diagnostic: extra rbf_ctx_force_generic bits (see include/rbf.h)")
    ap.add_argument("--generic-kernels", action="store_true", help="diagnostic: global-memory insert / query kernels instead of the LDS ones")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather to rank 0")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (and gather) even with one rank (smoke-tests the N>1 path)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="no per-kernel HIP events (neither in the timed region nor after it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of every pipeline's frames after the timed region")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of CPU-oracle work (0 = auto, about 10-30 s)")
    ap.add_argument("--exact-steps", action="store_true", help="time exactly --steps steps even if that is shorter than %.0f ms" % (MIN_REGION_S * 1e3))
    ap.add_argument("--clip-frames", type=int, default=0, help="strong-scaling mode: one clip of this many frames sharded by frame over the ranks (BASELINE config 3: 300)")
    ap.add_argument("--keyframe-interval", type=int, default=30, help="clip mode: frame t is a keyframe iff t %% interval == 0")
    ap.add_argument("--spawn", action="store_true", help="start the rank processes through bench.py's own launcher even for --gpus 1 (smoke-tests the N>1 launch path)")
    ap.add_argument("--no-clips", action="store_true", help="skip the clip300 / clip300_uint16 legs (BASELINE configs 3 and 5) behind the weak-scaling headline")
    ap.add_argument("--clip-leg-frames", type=int, default=300, help="frames of the clip legs of the default run")
    ap.add_argument("--clip-steps", type=int, default=40, help="timed passes over the clip in the clip legs")
    ap.add_argument("--begin-ahead", type=int, default=0, help="GOPs whose mask stage (rbf_encode_gop_begin) is enqueued before the oldest one is finished (rbf_encode_gop_finish); 0 = the one-call form (default: the fastest feed measured, profiles/r04_feed_sweep.txt), -1 = pipelines - 1")
    ap.add_argument("--host-threads", type=int, default=0, help="1 = one host thread per pipeline (each calls rbf_encode_gop for its own context; ctypes drops the GIL) instead of one thread issuing begin / finish in turn")
    ap.add_argument("--no-legs", action="store_true", help="skip the interleaved / config4_2160p / decode_1080p / batched_gops legs behind the headline")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend of the N > 1 path: nccl (= RCCL over xGMI, the product path) or gloo "
                    "(records staged through host memory; with --one-device it lets N rank processes share ONE GPU, which is how the sharded path is exercised with the real kernels on a 1-GPU box)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses GPU 0 (its own contexts and streams): N processes on one device, for --backend gloo")
    ap.add_argument("--gops-per-call", type=int, default=1, help="GOPs of --frames frames that ONE call (rbf_encode_runs: one mask / insert / reduce / query / compact launch sequence) codes; "
                    "a step is then one such call.  1 = the contract line (one GOP per call); the default run adds a `batched_gops` leg with 4")
    ap.add_argument("--insert-slices", type=int, default=0, help="tuning: RBF_OPT_INSERT_SLICES (0 = auto)")
    ap.add_argument("--clip-pass-slots", type=int, default=0, help="clip mode: sets of coders (own contexts, streams and records) consecutive passes rotate over, so that the mask stage, compaction and packing of "
                    "pass p+1 run under the Bloom kernels of pass p when a rank's share is fewer blocks than it has pipelines; 0 = auto (pipelines // blocks per pass), 1 = one set (round 5)")
    ap.add_argument("--clip-groups", type=int, default=0, help="clip mode: force the number of equal blocks a rank cuts its frames into per pass (0 = auto, see clip_blocks)")
    ap.add_argument("--proxy", default="", help="clip mode, single process: 'N,r' = run what rank r of an N-way split would run (gather stubbed), 'N' = every rank of the split in turn")
    ap.add_argument("--no-shard-proxy", action="store_true", help="skip the shard_proxy leg (one rank's share of configs 3 / 5 at N = 2, 4, 8, timed on this GPU)")
    ap.add_argument("--ballast", type=int, default=0, help="diagnostic: allocate a dummy device block of random size (this seed) in front of every pipeline's first step, so that the contexts' scratch lands elsewhere in HBM (profiles/r06_placement.txt)")
    ap.add_argument("--clip-block-gops", type=int, default=0, help="clip mode: keyframe intervals a rank hands to the GPU in ONE rbf_encode_runs launch sequence; 0 = auto (one block per pipeline and pass, see clip_blocks), 1 = one call per run of inter-frames (round 4)")
    return ap.parse_args()


METRIC = "Mpixels/s Bloom insert+query, 1080p residuals"
PHASE = {"name": "start", "t0": time.time()}


def set_phase(name):
    """Where this rank is (init_dist, setup, warmup, settle, timed, ...): what a failure line names, and -- under bench.py's own launcher --
    a one-line file per rank the launcher reads when a rank dies without a word."""
    PHASE["name"] = name
    d = os.environ.get("RBF_BENCH_LOG_DIR")
    if d:
        try:
            with open(os.path.join(d, "rank%s.phase" % os.environ.get("RANK", "0")), "w") as f:
                f.write(name)
        except OSError:
            pass


def failure_line(error, rank, phase, world, **extra):
    """The ONE JSON line of a run that failed: the contract's keys with value null, plus `error`, `rank` (the rank that failed, or that
    noticed) and `phase`."""
    out = {"metric": METRIC, "value": None, "unit": "Mpixel/s", "n_gpus": world, "higher_is_better": True, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "error": str(error)[:2000], "rank": rank, "phase": phase}
    out.update(extra)
    return out


def print_line(out):
    """Rank 0's one line (result or failure); leaves a marker for bench.py's own launcher."""
    print(json.dumps(out), flush=True)
    d = os.environ.get("RBF_BENCH_LOG_DIR")
    if d:
        try:
            open(os.path.join(d, "rank0.line_printed"), "w").close()
        except OSError:
            pass


def fail(exc, code=1):
    """Any exception on any rank ends here: traceback to stderr, rank 0 prints the failure line (a rank != 0 cannot: stdout carries ONE line,
    rank 0's -- it exits non-zero, the launcher or rank 0's next bounded collective reports it), then the process leaves WITHOUT running
    the distributed teardown (destroy_process_group / RCCL's atexit can wait for peers that are gone)."""
    import traceback
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    sys.stderr.write("[bench rank %d] FAILED in phase %r after %.1f s: %s\n" % (rank, PHASE["name"], time.time() - PHASE["t0"], exc))
    traceback.print_exception(type(exc), exc, exc.__traceback__, file=sys.stderr)
    sys.stderr.flush()
    if rank == 0:
        print_line(failure_line("%s: %s" % (type(exc).__name__, exc), rank, PHASE["name"], world))
    sys.stdout.flush()
    os._exit(code if isinstance(code, int) and code else 1)


def install_term_handler():
    """A launcher that lost one rank sends the others SIGTERM (torch.distributed.run does, bench.py's own does): rank 0 still says where it was."""
    import signal

    def on_term(signum, frame):
        fail(RuntimeError("terminated by the launcher (signal %d): another rank failed or the run was cancelled" % signum), 128 + signum)
    try:
        signal.signal(signal.SIGTERM, on_term)
    except ValueError:                            # not the main thread
        pass


def launch_ranks(nranks, cmd, stdout0=None, log_dir=None):
    """Start `cmd` once per rank (one process per GPU: RANK = LOCAL_RANK = r, WORLD_SIZE = nranks, rendezvous on a free port of
    127.0.0.1) and wait.  Rank 0 inherits stdout (or gets `stdout0`), so its ONE JSON line is the launcher's output; every rank's stderr
    (RCCL banners, tracebacks) goes to <log_dir>/rank<r>.stderr, NCCL_DEBUG defaults to WARN.  A rank that dies takes the others down
    (SIGTERM, exact PIDs); then the launcher prints the tail of every rank's stderr, and -- if rank 0 did not get to print a line -- the
    failure line itself: which rank went first, and the phase it was in.  Returns that rank's exit code."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:                    # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    log_dir = log_dir or os.environ.get("RBF_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="rbf_bench_ranks_")
    os.makedirs(log_dir, exist_ok=True)
    for name in os.listdir(log_dir):
        if name.startswith("rank"):
            os.remove(os.path.join(log_dir, name))
    procs, logs = [], []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RBF_BENCH_LAUNCHER="self-spawned", RBF_BENCH_LOG_DIR=log_dir)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")
        logs.append(open(os.path.join(log_dir, "rank%d.stderr" % r), "wb"))
        procs.append(subprocess.Popen(cmd, env=env, stdout=stdout0 if r == 0 else logs[r], stderr=logs[r]))
    rc, first_failed = 0, None
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc, first_failed = code, procs.index(p)
                for q in alive:                    # one rank failed: the others would wait in a collective until their bound runs out
                    q.terminate()
        time.sleep(0.05)
    for f in logs:
        f.close()
    if rc != 0:
        for r in range(nranks):
            try:
                with open(os.path.join(log_dir, "rank%d.stderr" % r), "rb") as f:
                    tail = f.read()[-3000:].decode("utf-8", "replace")
            except OSError:
                tail = ""
            sys.stderr.write("---- rank %d stderr (tail; whole file: %s) ----\n%s\n" % (r, os.path.join(log_dir, "rank%d.stderr" % r), tail))
        if not os.path.exists(os.path.join(log_dir, "rank0.line_printed")):
            try:
                with open(os.path.join(log_dir, "rank%d.phase" % first_failed)) as f:
                    phase = f.read().strip()
            except OSError:
                phase = "start"
            line = json.dumps(failure_line("rank %d exited with code %d (its stderr: %s)" % (first_failed, rc, os.path.join(log_dir, "rank%d.stderr" % first_failed)),
                                           first_failed, phase, nranks, launcher="bench.py"))
            dest = stdout0 or sys.stdout
            dest.write((line + "\n").encode() if "b" in getattr(dest, "mode", "") else line + "\n")
            dest.flush()
    else:
        sys.stderr.write("[bench launcher] %d ranks finished; per-rank stderr in %s\n" % (nranks, log_dir))
    return rc


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher environment: this script again, once per GPU."""
    sys.exit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))


def all_reduce_b(dist, t, op=None, what="all_reduce"):
    """dist.all_reduce whose result the host may read afterwards: issued async, polled against dist.DEFAULT_TIMEOUT_S (a dead peer raises
    CollectiveTimeout here instead of parking this rank in `.item()` until the watchdog kills it)."""
    from new_bloom_filter_repo_amd.dist import wait_work
    wait_work(dist.all_reduce(t, async_op=True) if op is None else dist.all_reduce(t, op=op, async_op=True), "%s (phase %s)" % (what, PHASE["name"]))


def bounded_barrier(dist, device, what="barrier"):
    """dist.barrier() that cannot outlive a dead peer: an async all_reduce of one word, polled against dist.DEFAULT_TIMEOUT_S."""
    import torch
    from new_bloom_filter_repo_amd.dist import wait_work
    t = torch.zeros(1, dtype=torch.int32, device=COMM_DEVICE)
    wait_work(dist.all_reduce(t, async_op=True), "%s (phase %s)" % (what, PHASE["name"]))
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def init_dist(args):
    import torch
    import torch.distributed as dist
    set_phase("init_dist")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    global COMM_DEVICE
    if args.one_device:
        if args.backend != "gloo":
            raise SystemExit("--one-device needs --backend gloo (RCCL wants one GPU per rank)")
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: only %d GPU(s) visible" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    COMM_DEVICE = device
    use_dist = world > 1 or args.force_dist
    if use_dist and args.backend == "gloo":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        COMM_DEVICE = torch.device("cpu")         # control values (timings, counts) and record payloads travel through host memory
        ones = torch.ones(1, dtype=torch.int64)
        all_reduce_b(dist, ones)
        if int(ones.item()) != world:
            raise SystemExit("gloo saw %d ranks, expected %d" % (int(ones.item()), world))
    elif use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime
        # (the watchdog's bound on a collective that never completes; the gathers' own host-side waits are bounded much tighter: dist.DEFAULT_TIMEOUT_S)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=600))
        # Finish RCCL's own setup (its streams, channels, proxy threads) with one collective BEFORE the
        # GOP pipelines' streams exist.  Measured in round 2 (the one-off tool is in the git history): streams created between an
        # eager communicator init and its first collective end up serialised with each other
        # (260 instead of 222 us/step); created after it, or before a lazy init, they overlap.
        # ... and that collective counts the ranks RCCL really connected (reported as `rccl_ranks`)
        ones = torch.ones(1, dtype=torch.int64, device=device)
        all_reduce_b(dist, ones)
        torch.cuda.synchronize(device)
        if int(ones.item()) != world:
            raise SystemExit("RCCL saw %d ranks, expected %d" % (int(ones.item()), world))
    return world, rank, local_rank, device, use_dist


def main():
    args = parse_args()
    if (args.gpus > 1 or args.spawn) and "RANK" not in os.environ:
        return spawn_ranks(args)
    install_term_handler()
    try:
        return clip_main(args) if args.clip_frames else bench_main(args)
    except SystemExit as e:
        if e.code in (0, None):
            raise
        fail(e, e.code if isinstance(e.code, int) else 1)
    except BaseException as e:                     # (KeyboardInterrupt included: the line still says where)
        fail(e)


def bench_main(args):
    import torch
    import torch.distributed as dist
    world, rank, local_rank, device, use_dist = init_dist(args)
    set_phase("setup")
    ncoders = max(1, args.streams)
    streams = [torch.cuda.Stream(device) for _ in range(ncoders)]      # none of them is the null stream

    from new_bloom_filter_repo_amd import _native as nat
    from new_bloom_filter_repo_amd.gop import GopCoder, TorchArena, torch_allocator
    from new_bloom_filter_repo_amd.synthetic import make_gop, P_KSTAR_2_3

    W, H, F = args.width, args.height, args.frames
    GPC = max(1, args.gops_per_call)              # GOPs per call: a block of GPC * F frames whose frames F, 2F, ... are keyframes (run starts)
    FB = F * GPC                                  # frames of a block
    n, pairs = W * H, FB - 1                      # rows per call; GPC - 1 of them are pairs across a keyframe, which are not coded
    coded_pairs = GPC * (F - 1)
    run_starts = [F * g for g in range(1, GPC)]
    dtype = np.uint8 if args.bits == 8 else np.uint16
    use_gather = use_dist and not args.no_gather and args.backend == "nccl"      # (the outbox gather posts device tensors: RCCL only; under gloo the weak mode runs without it)
    # `--streams` GOP pipelines: consecutive steps rotate over them, each with its own HIP stream, library
    # context (scratch), resident GOP and output record.  The HBM-bound mask kernel and the latency-bound
    # compaction / reduce kernels of one step then overlap the integer-issue-bound insert / query kernels of its
    # neighbours, and (N > 1) the async RCCL gather of step s overlaps the kernels of step s+1.  Every step
    # still does all of its work inside the timed region.
    ctxs = [nat.Context(local_rank, s.cuda_stream) for s in streams]
    if args.lds_tile_kib or args.generic_kernels or args.rebuild_hash_table or args.force_bits:
        for c in ctxs:
            c.force_generic(((args.lds_tile_kib * 1024 // 256) << 16) | (1 if args.generic_kernels else 0) | (16 if args.rebuild_hash_table else 0) | args.force_bits)      # tile unit: 64 dwords
    if args.insert_slices:
        for c in ctxs:
            c.option(nat.OPT_INSERT_SLICES, args.insert_slices)
    ctx = ctxs[0]
    arenas = [TorchArena(device, GopCoder.record_bytes(n, pairs)) for _ in range(ncoders)]
    planar = not args.interleaved
    if args.shared_gop and planar:
        raise SystemExit("--shared-gop is a diagnostic of the interleaved layout: add --interleaved")
    G_res = args.gops_per_pipeline or (3 if planar else 1)      # resident GOPs per pipeline: the inputs stay ~750 MB, well past the Infinity Cache
    coders = []
    for k in range(ncoders):
        coders.append(GopCoder(ctxs[k], W, H, FB, channels=3, sample_bytes=args.bits // 8, allocator=torch_allocator(device),
                               out_allocator=arenas[k], frames_block=coders[0].frames if (k and args.shared_gop) else None,
                               planar_luma=planar, keep_interleaved=not planar, resident_gops=G_res, run_starts=run_starts))
    coder = coders[0]
    density = args.density or P_KSTAR_2_3
    host_gops = []                                # [pipeline][resident gop] -> (F, H, W, 3) host frames
    for k in range(ncoders):
        if k and args.shared_gop:
            host_gops.append(host_gops[0])
            continue
        host_gops.append([np.concatenate([np.stack(make_gop(1000 * 2 + 64 * rank + k + 16 * g + 4096 * j, W, H, F, p=density, dtype=dtype)) for j in range(GPC)])
                          for g in range(G_res)])
        for g in range(G_res):
            coders[k].load_frames(host_gops[k][g], g)
    torch.cuda.synchronize(device)
    per_gop_bytes = host_gops[0][0].nbytes // (3 if planar else 1)
    resident_mb = per_gop_bytes * G_res * (1 if args.shared_gop else ncoders) / 1e6

    gather = use_gather
    # N > 1: every step compacts its output rows into an exact-size record on the device
    # (rbf_pack_records) inside an outbox of `--gather-every` slots; a full outbox goes to rank 0 in ONE
    # asynchronous RCCL gather (fewer, larger collectives; two outboxes alternate so packing never waits
    # for a transfer).  RCCL's gather needs one size on all ranks, so the slot size is agreed once, in setup
    # (max of the ranks' record sizes + 2 %); a record that outgrew its slot is flagged in its header and
    # re-sent through the exact-size path after the timed region (checked on rank 0).
    G = max(1, args.gather_every)
    record_max = (int(nat.lib().rbf_record_max_bytes(pairs, n)) + 255) // 256 * 256

    class Slot:                                   # what GopCoder.pack needs of a block
        def __init__(self, tensor):
            self.ptr, self.nbytes = tensor.data_ptr(), tensor.numel() * 8

    probe = [Slot(torch.zeros(record_max // 8, dtype=torch.int64, device=device)) for _ in range(ncoders)] if gather else None
    og = None                                     # dist.OutboxGather once the slot size is agreed
    slots = {}
    state = {"s": 0}

    # One host thread feeds all pipelines: the mask stage of step s (rbf_encode_gop_begin) is enqueued `ahead` steps before the thread
    # waits for its counts and enqueues its Bloom kernels (rbf_encode_gop_finish), so the thread never stands still while a mask
    # kernel runs (round 3: it span inside every rbf_encode_gop for about half of the step).  ahead = 0 is the one-call form.
    ahead = (ncoders - 1) if args.begin_ahead < 0 else min(args.begin_ahead, ncoders - 1)
    import collections
    pending = collections.deque()
    host_s = {"t": 0.0}                           # time the feeding thread(s) spent inside the library calls of the steps

    def finish_one():
        k = pending.popleft()
        if not gather:
            coders[k].encode_finish()
            return
        with torch.cuda.stream(streams[k]):
            t = og.begin(k)                       # this step's slot (waits stream-side for the outbox's previous transfer)
            coders[k].encode_finish()
            coders[k].pack(slots.setdefault(t.data_ptr(), Slot(t)))
            og.end(k)                             # full outbox -> one asynchronous gather from the comm stream

    def step():
        k = state["s"] % ncoders
        g = (state["s"] // ncoders) % G_res       # a pipeline rotates over its resident GOPs
        state["s"] += 1
        t0 = time.perf_counter()
        coders[k].encode_begin(g)
        pending.append(k)
        if len(pending) > ahead:
            finish_one()
        host_s["t"] += time.perf_counter() - t0

    def run_steps(nsteps):
        if not (args.host_threads and not gather and ncoders > 1):
            for _ in range(nsteps):
                step()
            return
        # one thread per pipeline, each with its own context and stream: step i is pipeline i % ncoders' (i // ncoders)-th GOP
        import threading
        first = state["s"]
        state["s"] += nsteps
        spent = [0.0] * ncoders

        def feed(k):
            t0 = time.perf_counter()
            for i in range(first, first + nsteps):
                if i % ncoders == k:
                    coders[k].encode((i // ncoders) % G_res)
            spent[k] = time.perf_counter() - t0
        threads = [threading.Thread(target=feed, args=(k,)) for k in range(ncoders)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        host_s["t"] += max(spent)

    def drain():
        t0 = time.perf_counter()
        while pending:
            finish_one()
        host_s["t"] += time.perf_counter() - t0
        if og is not None:
            og.flush()

    def barrier():
        if use_dist:
            bounded_barrier(dist, device)
        torch.cuda.synchronize(device)

    # setup (not warm-up steps): every pipeline sizes its scratch once, and for N > 1 the ranks agree on the slot
    ballast = []
    for k in range(ncoders):
        if args.ballast:
            brng = np.random.default_rng(args.ballast * 100 + k)
            ballast.append(ctxs[k].alloc(int(brng.integers(1, 96)) * (1 << 19) + int(brng.integers(0, 2048)) * 256))
        coders[k].encode()
        if gather:
            coders[k].pack(probe[k])
    torch.cuda.synchronize(device)
    if gather:
        from new_bloom_filter_repo_amd.dist import OutboxGather
        # worst-case slots (a record can never overflow one); only the used bytes of a slot travel
        og = OutboxGather(record_max // 8, G, device, streams=streams)
        probe = None
        run_steps(2 * G)                          # untimed: the first RCCL transfer of both outboxes (connection setup)
        drain()
        state["s"] = 0
    set_phase("warmup")
    run_steps(args.warmup)
    drain()
    torch.cuda.synchronize(device)

    def timed(nsteps, with_events):
        if with_events:
            # HIP events around the DOMINANT kernel only (query): two events per step on the launching
            # stream; bracketing every kernel would add ~40 us of event overhead to a ~200 us step
            for c in ctxs:
                c.timing_reset()
                c.timing(1 << nat.K_QUERY)
        barrier()
        host_s["t"] = 0.0
        gc.disable()                              # (no generation-2 collection inside a 2 ms region)
        t0 = time.perf_counter()
        run_steps(nsteps)
        drain()
        torch.cuda.synchronize(device)
        barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
        host_s["region"] = host_s["t"] / nsteps
        kt = {}
        if with_events:
            for c in ctxs:
                c.timing(False)
                for name, (ms, cnt) in c.timing_read().items():
                    a = kt.get(name, (0.0, 0))
                    kt[name] = (a[0] + ms, a[1] + cnt)
        if use_dist:
            te = torch.tensor([elapsed], dtype=torch.float64, device=COMM_DEVICE)
            all_reduce_b(dist, te, dist.ReduceOp.MAX)
            elapsed = float(te.item())
        return elapsed, kt

    # Settling (untimed, after the --warmup steps the caller asked for): the first regions after start-up run 8-10 % slower than the
    # steady state (clocks, caches, the pipelines' queues), so windows of --steps steps are run until two consecutive ones agree within
    # 2 % -- or 30 ms have gone by.  What is then timed is what a warm service does.
    settle = {"windows": 0, "ms": 0.0}
    set_phase("settle")
    if not args.exact_steps:
        prev, spent = None, 0.0
        while settle["windows"] < 200:
            e, _ = timed(args.steps, False)
            settle["windows"] += 1
            spent += e
            agree = prev is not None and abs(e - prev) <= 0.02 * prev
            done = spent >= 0.100 or (agree and spent >= 0.030)     # two windows within 2 % of each other and >= 30 ms behind us, or 100 ms at most
            if use_dist:                          # every rank must leave the loop in the same round
                flag = torch.tensor([1 if done else 0], dtype=torch.int64, device=COMM_DEVICE)
                all_reduce_b(dist, flag, dist.ReduceOp.MIN)
                done = bool(flag.item())
            prev = e
            if done:
                break
        settle["ms"] = round(spent * 1e3, 2)
    # The contract's region: EXACTLY --steps steps between barrier + synchronize, max over ranks.  A region of 20 steps lasts ~2.5 ms, so it
    # is timed `REGIONS` times and the MEDIAN region is the headline (`steps` = --steps); all of them are reported.  One long region
    # (>= MIN_REGION_S) follows as `steady_state`: what the pipelines sustain without the fill and drain of a short region.
    REGIONS = 1 if args.exact_steps else 9
    set_phase("timed")
    regions = [timed(args.steps, False) for _ in range(REGIONS)]      # (no HIP events inside the headline's regions)
    order = sorted(range(REGIONS), key=lambda i: regions[i][0])
    elapsed = regions[order[REGIONS // 2]][0]
    host_region = host_s.get("region", 0.0)
    steps_timed = args.steps
    # one more region with two HIP events per step around the dominant kernel: its latency while the other pipelines run beside it
    ktimes = {} if args.no_kernel_timing else timed(max(args.steps, 4 * ncoders), True)[1]
    steady = None
    if elapsed < MIN_REGION_S and not args.exact_steps:
        nlong = int(MIN_REGION_S * 1.25 / (elapsed / args.steps)) + 1
        if use_dist:                               # every rank must run the same number of steps
            tl = torch.tensor([nlong], dtype=torch.int64, device=COMM_DEVICE)
            all_reduce_b(dist, tl, dist.ReduceOp.MAX)
            nlong = int(tl.item())
        e_long, _ = timed(nlong, False)
        steady = {"steps": nlong, "ms_per_step": round(e_long / nlong * 1e3, 4), "region_ms": round(e_long * 1e3, 2),
                  "value": round(coded_pairs * n * world * nlong / e_long / 1e6, 2), "unit": "Mpixel/s",
                  "note": "one long region: no fill / drain of the pipelines inside it; the headline is the median %d-step region" % args.steps}

    # per-kernel figures with the chip to itself: one pipeline, every kernel bracketed, ALONE_LAUNCHES steps
    set_phase("kernel_timing")
    breakdown = None
    if not args.no_kernel_timing:
        torch.cuda.synchronize(device)
        for i in range(3):
            coder.encode(i % G_res)
        ctx.sync()
        ctx.timing_reset()
        ctx.timing(True)
        for i in range(ALONE_LAUNCHES):
            coder.encode(i % G_res)
        ctx.sync()
        ctx.timing(False)
        breakdown = {k: round(v[0] / ALONE_LAUNCHES, 4) for k, v in ctx.timing_read().items() if v[1]}    # ms per step (a kernel launched in groups counts whole)

    # every resident GOP of every pipeline once more, for the oracle comparison: [(host frames, rows)]
    set_phase("verify")
    checked = []
    for g in range(G_res):
        for k in range(1 if args.shared_gop else ncoders):
            coders[k].encode(g)
        for k in range(1 if args.shared_gop else ncoders):
            checked.append((host_gops[k][g], coders[k].results()))
    res_all = [rows for _, rows in checked]
    res = res_all[0]
    pixels_per_step = coded_pairs * n * world
    value = pixels_per_step * steps_timed / elapsed / 1e6

    out = {
        "metric": METRIC,
        "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": steps_timed, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps_timed * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "requested_steps": args.steps, "steps_timed": steps_timed, "timed_region_ms": round(elapsed * 1e3, 2),
        "regions_ms": [round(r[0] * 1e3, 3) for r in regions], "headline_region": "median of %d regions of exactly %d steps" % (REGIONS, args.steps),
        "settle": settle, "steady_state": steady,
        "host_ms_per_step": round(host_region * 1e3, 4),
        "host_feed": ("one thread per pipeline, each calling rbf_encode_gop on its own context" if (args.host_threads and not gather and ncoders > 1) else
                      "one thread: rbf_encode_gop_begin of step s+%d is enqueued before rbf_encode_gop_finish of step s" % ahead if ahead else
                      "one thread, one blocking rbf_encode_gop per step"),
        "config": {"workload": "%dx%d YUV444 %d-bit synthetic %d-frame GOP%s (%d inter-frames/step/GPU), %s, threshold 0, %s"
                               % (W, H, args.bits, F, "" if GPC == 1 else " x %d GOPs batched into ONE launch sequence per step (rbf_encode_runs)" % GPC, coded_pairs,
                                  "p=%g" % args.density if args.density else "k*=2.3",
                                  "planar Y resident (the mask stage reads luma only)" if planar else "interleaved YUV444 resident"),
                   "gops_per_call": GPC,
                   "layout": "planar Y" if planar else "interleaved", "resident_gops_per_pipeline": G_res,
                   "inputs": "resident in HBM before the timed region: %s; the timed step starts at the mask kernel"
                             % ("Y planes extracted from the YUV444 frames on the host and uploaded once (the `interleaved_yuv444` leg keeps whole frames resident instead)" if planar else "whole interleaved YUV444 frames, uploaded once"),
                   "pixels_per_step": pixels_per_step, "gather_to_rank0": bool(gather), "gop_pipelines_per_gpu": ncoders,
                   "distinct_gop_per_pipeline": not args.shared_gop, "resident_input_mb_per_gpu": round(resident_mb, 1),
                   "gather": "exact-size: all_gather of the used sizes, then one grouped point-to-point message per peer; rank 0's own records are not sent" if gather else None,
                   "steps_per_gather": G if gather else 0,
                   "lds_tile_kib": args.lds_tile_kib or "auto", "generic_kernels": bool(args.generic_kernels),
                   "hash_table": "k_hash_table runs in every step (--rebuild-hash-table)" if args.rebuild_hash_table else
                                 "one pixel-index hash table per (device, frame size, seeds), shared by the pipelines' contexts; built once, before the timed region" if args.streams > 1 else
                                 "built once; rewritten in every step by the query kernel (sole holder: keeps the table in the Infinity Cache for the next insert)",
                   "stages": "residual mask -> host params -> insert -> query+witness",
                   "launcher": os.environ.get("RBF_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "single process"),
                   "rccl_ranks": world if (use_dist and args.backend == "nccl") else 0, "backend": args.backend if use_dist else None,
                   "ranks_share_one_device": bool(args.one_device),
                   "multi_gpu_note": "N>1 on one node was never measured by the builder (1-GPU boxes only); what exists: every rank's share of the N = 2 / 4 / 8 split timed on ONE GPU (the `shard_proxy` leg: a prediction, gather stubbed), world-2 / world-8 runs of the real kernels on ONE device over gloo (tests/test_gpu_dist_shared.py), gloo world-2/3 CPU tests (incl. the self-spawn launcher, a rank killed mid-gather, the failure line) and RCCL world-1 tests"},
    }
    if use_dist:
        # who ran where, and what crossed the links: every rank's device, and the bytes the non-root ranks sent into rank 0 per step
        # (exact-size records: over xGMI on a node, one link per peer)
        props = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "device": "cuda:%d" % local_rank, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
                "bytes_sent_to_rank0": int(og.bytes_sent) if og is not None else 0, "steps_gathered": int(og.s) if og is not None else 0}
        infos = [None] * world
        dist.all_gather_object(infos, mine)
        out["config"]["ranks"] = infos
        sent, steps_g = sum(i["bytes_sent_to_rank0"] for i in infos), max(i["steps_gathered"] for i in infos)
        out["config"]["xgmi_bytes_per_step_into_rank0"] = round(sent / steps_g, 1) if steps_g else 0
        out["config"]["collective_timeout_s"] = __import__("new_bloom_filter_repo_amd.dist", fromlist=["x"]).DEFAULT_TIMEOUT_S
    if rank == 0 and gather:
        out["config"]["gathered_records_parsed_on_rank0"] = check_gathered(og, world, G, pairs, n, res_all)
    if rank == 0:
        l_sum = sum(r["l"] for r in res)
        w_sum = sum(r["witness_bits"] for r in res)
        # ALGORITHMIC bytes of the A5 stage: packed mask in + filter in + witness out
        alg_bytes = coded_pairs * n / 8 + l_sum / 8 + w_sum / 8
        q_alone = (breakdown or {}).get("query")
        if q_alone:
            achieved = alg_bytes / (q_alone * 1e-3) / 1e9
            default_shape = (W, H, F, args.bits, GPC) == (1920, 1080, 30, 8, 1)
            qname = "k_query_u64" if default_shape else "query kernel of this geometry (DESIGN.md 4)"
            rf = {"bound": "hbm", "kernel": qname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                  "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                  "avg_launch_ms": q_alone, "launches_averaged": ALONE_LAUNCHES,
                  "timing": "HIP events on the launching stream, one pipeline alone (nothing co-running), after the timed region",
                  "algorithmic_bytes_per_launch": int(alg_bytes), "bytes_per_pixel": round(alg_bytes / (coded_pairs * n), 4)}
            rf.update(measured_traffic(W, H, F, args.bits, bool(args.density), GPC))
            if ktimes and ktimes.get("query", (0, 0))[1]:
                rf["latency_under_overlap_ms"] = round(ktimes["query"][0] / ktimes["query"][1], 4)   # event pair inside the timed region: includes queueing behind the other pipelines
            c_alone = (breakdown or {}).get("stitch")
            if c_alone:                            # A5 = query + witness compaction: the stage that reads the mask and writes the witness
                st = q_alone + c_alone
                rf["stage_a5"] = {"kernels": [qname, "k_compact_witness"], "ms": round(st, 4),
                                  "achieved": round(alg_bytes / (st * 1e-3) / 1e9, 2), "frac": round(alg_bytes / (st * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
            # the whole fused path priced as SURVEY 8d does: 2 luma reads + packed mask, filter and witness per pixel
            b_px = 2.0 * (args.bits // 8) + alg_bytes / (coded_pairs * n)
            rf["end_to_end"] = {"bytes_per_pixel": round(b_px, 3), "achieved": round(value / world * 1e6 * b_px / 1e9, 1),
                                "unit": "GB/s per GPU", "frac": round(value / world * 1e6 * b_px / 1e9 / HBM_PEAK_GBPS, 4)}
            rf["issue"] = None if args.density else issue_roofline(W, H, F, args.bits, GPC, breakdown, elapsed / steps_timed * 1e3, (steady or {}).get("ms_per_step"))
            out["roofline"] = rf
            out["kernels_ms_per_step_alone"] = breakdown
            out["kernels_alone_note"] = "HIP events around every kernel of %d steps of ONE pipeline after the timed region, nothing co-running (rocprofv3 of the same shape: profiles/r06_rocprofv3_summary.txt)" % ALONE_LAUNCHES
        else:
            out["roofline"] = None
        if world == 1:
            out["pcie_inclusive_mpixels_per_s"] = pcie_inclusive(torch, nat, coder, host_gops[0][0], coded_pairs * n, elapsed / steps_timed)
            if G_res >= 2 and planar:
                out["pcie_overlapped_mpixels_per_s"] = pcie_overlapped(torch, device, coder, streams[0], host_gops[0], coded_pairs * n)
        if not args.no_verify:
            out["verified_vs_oracle"] = verify_all([h for h, _ in checked], res_all, n, len(checked))
            out["verified_vs_oracle"]["pipelines"] = 1 if args.shared_gop else ncoders
            out["verified_vs_oracle"]["gops_per_pipeline"] = G_res
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res, n, args.cpu_frames)
    # ---- further legs of the same line (rank 0 of a single-GPU run only): the interleaved layout next to the planar headline, BASELINE
    #      configs[3] (3840x2160) and the decode direction (A6) of the headline's own records
    if world == 1 and rank == 0 and not args.no_legs and (W, H, F, args.bits, GPC) == (1920, 1080, 30, 8, 1) and not args.density:
        out["decode_1080p"] = decode_leg(torch, nat, coders, host_gops, n, pairs, G_res)
        if planar:
            out["interleaved_yuv444"] = pipelines_leg(torch, nat, device, local_rank, W, H, F, 8, False, 1, ncoders, density, ahead, 160, host_gops=[[h[0]] for h in host_gops], verify=not args.no_verify)
        for c in coders:
            c.close()
        coders = [None]
        torch.cuda.empty_cache()
        # the headline's GOPs, four to a call: ONE mask / insert / reduce / query / compact launch sequence per 116 inter-frames (rbf_encode_runs)
        out["batched_gops"] = pipelines_leg(torch, nat, device, local_rank, W, H, F, 8, True, 1, ncoders, density, ahead, 48, verify=not args.no_verify, seed=5000, gops_per_call=4)
        # BASELINE configs[3]: 3840x2160, 30-frame GOPs (the query kernel's hashing prologue amortised over 29 frames) and the 9-frame GOP of rounds 1-4 beside it
        out["config4_2160p"] = pipelines_leg(torch, nat, device, local_rank, 3840, 2160, 30, 8, True, 1, ncoders, density, ahead, 24, verify=not args.no_verify, seed=4100)
        out["config4_2160p_gop9"] = pipelines_leg(torch, nat, device, local_rank, 3840, 2160, 9, 8, True, 1, ncoders, density, ahead, 60, verify=not args.no_verify, seed=4000)
        out["e2e_surface"] = e2e_surface_leg(nat, local_rank, W, H, density)
    # ---- BASELINE configs[2] and [4] in the same process group: one 300-frame clip sharded by frame (strong scaling), 8- and 16-bit
    if og is not None:
        og.close()
    set_phase("clips")
    if not args.no_clips:
        for c in coders:
            if c is not None:
                c.close()
        del coders, coder, arenas, ctxs, ctx, og, slots, probe
        torch.cuda.empty_cache()
        env = (world, rank, local_rank, device, use_dist)
        proxy = world == 1 and not use_dist and not args.no_shard_proxy and (W, H, args.bits) == (1920, 1080, 8) and not args.density
        cache = {} if proxy else None             # (N > 1: every rank makes its own shard)
        pool = cache if cache is not None else {}  # the pipelines and the arena are shared by the two clip legs at any N
        c8 = run_clip(args, env, args.clip_leg_frames, args.keyframe_interval, 8, args.clip_steps, 2, clip_cache=cache, pool=pool)
        sp = dict(SHARD_PROXY_WHAT)
        if proxy:                                 # 8-bit proxies while the 8-bit clip is cached, then the 16-bit clip
            sp["clip300"] = shard_proxy_leg(args, env, args.clip_leg_frames, args.keyframe_interval, args.clip_steps, cache, 8, c8)
        c16 = run_clip(args, env, args.clip_leg_frames, args.keyframe_interval, 16, args.clip_steps, 2, clip_cache=cache, pool=pool)
        if proxy:
            sp["clip300_uint16"] = shard_proxy_leg(args, env, args.clip_leg_frames, args.keyframe_interval, args.clip_steps, cache, 16, c16)
        for k2 in list(pool):
            if k2[0] == "pipelines":
                for c in pool.pop(k2)[1]:
                    c.close()
        if rank == 0:
            out["clip300"], out["clip300_uint16"] = c8, c16
            if proxy:
                out["shard_proxy"] = sp
    set_phase("shutdown")
    if use_dist:
        bounded_barrier(dist, device, "final barrier")
        dist.destroy_process_group()
    if rank == 0:
        print_line(out)                           # the last thing on stdout (RCCL prints its own lines while it is alive)


def e2e_surface_leg(nat, local_rank, W, H, density, T=300, I=30, block_frames=None, gpu_lanes=2, profile_stages=False):
    """What a caller of the plugin surface gets (improved_video_compressor.py:358-504): ImprovedVideoCompressor.compress_video +
    decompress_video on host-resident YUV444 frames -- one 30-frame GOP and the 300-frame clip of BASELINE configs[2] -- with the time
    of every stage and verify_bit_exact (verify_true_lossless.py:338-492 semantics) of the decoded frames.  The blocks of a clip alternate
    over `gpu_lanes` contexts, each block on its own host thread, so the stage times of different blocks OVERLAP (their sums can exceed the
    wall time); `gpu_busy_frac` is the share of the call's wall time during which at least one lane had a copy or a kernel in flight.  The
    reference's only published time for this surface is 12.45 s for its own clip (results.md:140; other hardware, other content): not
    comparable, reported for orientation only."""
    from new_bloom_filter_repo_amd.synthetic import make_clip_shard
    from new_bloom_filter_repo_amd.video_compressor import ImprovedVideoCompressor
    from new_bloom_filter_repo_amd.verify import verify_bit_exact
    n = W * H
    clip = make_clip_shard(3100, W, H, 0, T, I, p=density)
    out = {}
    with nat.Context(local_rank) as ctx:
        for name, count in (("gop30", I), ("clip300", T)):
            frames = [clip[t] for t in range(count)]
            comp = ImprovedVideoCompressor(keyframe_interval=I, ctx=ctx, block_frames=block_frames, gpu_lanes=gpu_lanes)
            comp.profile_stages = profile_stages
            t0 = time.perf_counter()
            res = comp.compress_video(list(frames), input_color_space="YUV")
            t1 = time.perf_counter()
            tm = dict(comp.last_timing or {})
            dec = comp.decompress_video(compressed_frames=comp.last_compressed_frames)
            t2 = time.perf_counter()
            td = dict(comp.last_timing or {})
            v = verify_bit_exact(frames, dec, color_space="YUV")
            if not v["success"]:
                raise SystemExit("e2e_surface: %s does not round-trip bit-exactly: %s" % (name, v["different_frame_indices"][:5]))
            inter = count - res["keyframes"]
            rnd = lambda d: {k: (round(x, 3) if isinstance(x, float) else x) for k, x in d.items()}
            out[name] = {"frames": count, "keyframes": res["keyframes"], "inter_frames": inter,
                         "compress_s": round(t1 - t0, 3), "compress_fps": round(count / (t1 - t0), 1), "compress_mpixels_per_s": round(count * n / (t1 - t0) / 1e6, 1),
                         "gpu_busy_frac": round(tm.get("gpu_busy", 0.0) / (t1 - t0), 3),
                         "decompress_s": round(t2 - t1, 3), "decompress_fps": round(count / (t2 - t1), 1),
                         "decompress_gpu_busy_frac": round(td.get("gpu_busy", 0.0) / (t2 - t1), 3),
                         "compression_ratio": round(res["compression_ratio"], 4), "container": "BFV2",
                         "stages_s": rnd(tm), "decompress_stages_s": rnd(td),
                         "verify_bit_exact": {"success": v["success"], "exact_matches": v["exact_matches"], "frames_compared": v["frames_compared"]}}
            comp.close()
    out["what"] = ("ImprovedVideoCompressor.compress_video(frames, input_color_space='YUV') + decompress_video, host-resident %dx%d YUV444 uint8 frames, keyframe every %d, "
                   "blocks of %s frames alternating over %d GPU lanes (own context, stream and host thread each); "
                   "stages_s (summed over the blocks, which overlap): stack = np.stack of a block's frames (0 when they lie back to back), upload = pageable host -> HBM, gpu_encode = "
                   "enqueueing rbf_encode_runs, download_rows = rbf_pack_records + ONE exact-size download of the block's record (this is where the host waits for the kernels), "
                   "value_gather = rbf_gather_values_batch + download, gpu_phase = wall time until the last block's values were on the host, zlib_wait = what the host threads' "
                   "zlib-9 (keyframes: four jobs each; changed values: one job per frame) still owed after that, gpu_busy = wall time with a copy or kernel in flight on some lane; "
                   "decompress_stages_s likewise per run (mask_decode = rbf_bloom_decode_batch, apply_chain = device-side rebuild + download of the frames); "
                   "synthetic frames are incompressible noise, so the ratio says nothing about real video and zlib-9 of ~290 MB of it on the host's cores is the floor of compress_s"
                   % (W, H, I, block_frames or "2 x keyframe interval", gpu_lanes))
    return out


def decode_leg(torch, nat, coders, host_gops, n, pairs, G_res, reps=40):
    """A6 (improved_video_compressor.py:268-307) over the step's own records: every pipeline decodes the (filter, witness, k, l) rows its
    last encode left in HBM back into masks -- query + expansion, rbf_bloom_decode_batch -- `reps` times on its own
    stream; the decoded masks are compared with the masks the encoder saw AND with the CPU oracle's masks of the host frames."""
    import ctypes
    from oracle import oracle as orc
    L = nat.lib()
    outs = []
    for c in coders:
        c.encode(0)
        outs.append(c._alloc(c.mask_stride * pairs))
    for c in coders:
        c.ctx.sync()

    def decode(c, o):
        nat.check(L.rbf_bloom_decode_batch(c.ctx.handle, c.filters.ptr, c.filter_stride, c.witness.ptr, c.witness_stride,
                                           n, pairs, c.params, ctypes.byref(c.seeds), o.ptr, c.mask_stride))
    for c, o in zip(coders, outs):
        decode(c, o)
    for c in coders:
        c.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for c, o in zip(coders, outs):
            decode(c, o)
    for c in coders:
        c.ctx.sync()
    dt = (time.perf_counter() - t0) / (reps * len(coders))
    c0 = coders[0]
    c0.ctx.timing_reset()
    c0.ctx.timing(True)
    for _ in range(20):
        decode(c0, outs[0])
    c0.ctx.sync()
    c0.ctx.timing(False)
    alone = {k: round(v[0] / 20, 4) for k, v in c0.ctx.timing_read().items() if v[1]}
    frames = 0
    for k, (c, o) in enumerate(zip(coders, outs)):
        got = o.numpy(c.ctx)[:c.mask_stride * pairs].reshape(pairs, c.mask_stride)
        seen = c.masks.numpy(c.ctx)[:c.mask_stride * pairs].reshape(pairs, c.mask_stride)
        if not np.array_equal(got, seen):
            raise SystemExit("decode leg: pipeline %d's decoded masks differ from the encoder's" % k)
        gop = host_gops[k][0]
        for f in range(pairs):
            want = orc.residual_mask(np.ascontiguousarray(gop[f][..., 0]), np.ascontiguousarray(gop[f + 1][..., 0]), 0.0).reshape(-1)
            if not np.array_equal(np.unpackbits(got[f])[:n], want):
                raise SystemExit("decode leg: pipeline %d frame %d differs from the CPU oracle's mask" % (k, f))
            frames += 1
    return {"value": round(pairs * n / dt / 1e6, 1), "unit": "Mpixel/s", "ms_per_gop": round(dt * 1e3, 4), "pipelines": len(coders), "gops_timed": reps * len(coders),
            "kernels_ms_per_gop_alone": alone, "what": "rbf_bloom_decode_batch over the 29 records of a headline step (query + expand: two launches), records resident in HBM",
            "verified_vs_oracle": {"frames": frames, "fields": "decoded mask == encoder's mask == oracle's mask"}}


def pipelines_leg(torch, nat, device, local_rank, W, H, F, bits, planar, G_res, ncoders, density, ahead, steps, host_gops=None, verify=True, seed=3000, gops_per_call=1):
    """The headline's step on another geometry, layout or block size, shortened: `ncoders` pipelines, begin / finish in turn, `steps` timed
    steps, every kernel alone afterwards, every block checked against the CPU oracle; its own HBM roofline for the query kernel.
    gops_per_call > 1: a step is ONE rbf_encode_runs call over `gops_per_call` GOPs of F frames (a block cut at its keyframes)."""
    from new_bloom_filter_repo_amd.gop import GopCoder, TorchArena, torch_allocator
    from new_bloom_filter_repo_amd.synthetic import make_gop
    import collections
    GPC = max(1, gops_per_call)
    FB = F * GPC
    n, pairs, coded_pairs = W * H, FB - 1, GPC * (F - 1)
    dtype = np.uint8 if bits == 8 else np.uint16
    streams = [torch.cuda.Stream(device) for _ in range(ncoders)]
    ctxs = [nat.Context(local_rank, s.cuda_stream) for s in streams]
    arenas = [TorchArena(device, GopCoder.record_bytes(n, pairs)) for _ in range(ncoders)]
    coders = [GopCoder(ctxs[k], W, H, FB, channels=3, sample_bytes=bits // 8, allocator=torch_allocator(device), out_allocator=arenas[k],
                       planar_luma=planar, keep_interleaved=not planar, resident_gops=G_res, run_starts=[F * g for g in range(1, GPC)]) for k in range(ncoders)]
    if host_gops is None:
        host_gops = [[np.concatenate([np.stack(make_gop(seed + 16 * k + g + 4096 * j, W, H, F, p=density, dtype=dtype)) for j in range(GPC)])
                      for g in range(G_res)] for k in range(ncoders)]
    for k in range(ncoders):
        for g in range(G_res):
            coders[k].load_frames(host_gops[k][g], g)
    torch.cuda.synchronize(device)
    pending = collections.deque()

    def run(nsteps, first=0):
        for s in range(first, first + nsteps):
            k = s % ncoders
            coders[k].encode_begin((s // ncoders) % G_res)
            pending.append(k)
            if len(pending) > ahead:
                coders[pending.popleft()].encode_finish()
        while pending:
            coders[pending.popleft()].encode_finish()
    # warm-up until 30 ms have gone by (like the headline's settling), then one region of `steps` steps -- extended to >= 50 ms when shorter
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.030:
        run(ncoders)
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if dt < 0.050:
        steps = int(steps * 0.0625 / dt) + 1
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
    dt /= steps
    c0 = coders[0]
    c0.ctx.timing_reset()
    c0.ctx.timing(True)
    for i in range(20):
        c0.encode(i % G_res)
    c0.ctx.sync()
    c0.ctx.timing(False)
    alone = {k: round(v[0] / 20, 4) for k, v in c0.ctx.timing_read().items() if v[1]}
    out = {"value": round(coded_pairs * n / dt / 1e6, 1), "unit": "Mpixel/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "pipelines": ncoders,
           "workload": "%dx%d YUV444 %d-bit synthetic %d-frame GOP%s, k*=2.3, %s resident"
                       % (W, H, bits, F, "" if GPC == 1 else " x %d GOPs batched into ONE launch sequence per step (rbf_encode_runs: %d inter-frames per call)" % (GPC, coded_pairs),
                          "planar Y" if planar else "interleaved YUV444"),
           "gops_per_call": GPC, "inter_frames_per_step": coded_pairs,
           "resident_input_mb": round(sum(h.nbytes for hs in host_gops for h in hs) / (3 if planar else 1) / 1e6, 1),
           "kernels_ms_per_step_alone": alone}
    checked = []
    for g in range(G_res):
        for k in range(ncoders):
            coders[k].encode(g)
        for k in range(ncoders):
            checked.append((host_gops[k][g], coders[k].results()))
    res = checked[0][1]
    alg_bytes = coded_pairs * n / 8 + sum(r["l"] for r in res) / 8 + sum(r["witness_bits"] for r in res) / 8
    if alone.get("query"):
        ach = alg_bytes / (alone["query"] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_query_s64t" if n > 1920 * 1080 else "k_query_u64", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBPS, 5), "avg_launch_ms": alone["query"], "launches_averaged": 20,
                           "algorithmic_bytes_per_launch": int(alg_bytes), "frames_per_launch": coded_pairs}
        out["roofline"].update(measured_traffic(W, H, F, bits, False, GPC))
        out["roofline"]["issue"] = issue_roofline(W, H, F, bits, GPC, alone, dt * 1e3)
    if verify:
        out["verified_vs_oracle"] = verify_all([h for h, _ in checked], [r for _, r in checked], n, len(checked))
    for c in coders:
        c.close()
    for c in ctxs:
        c.close()
    del coders, ctxs, arenas
    torch.cuda.empty_cache()
    return out


TRAFFIC = "profiles/r06_traffic.json"             # written by tools/make_traffic.py from tools/r06_profile.sh's PMC passes


def replayed_traffic(W, H, F, bits, gpc):
    """HBM bytes per launch of a leg's query kernel from a committed rocprofv3 PMC pass (profiles/r06_traffic.json: FETCH_SIZE / WRITE_SIZE
    in separate --pmc runs, corrected as MI355X_MICROARCH.md prescribes), or None: a REPLAYED constant, tagged as such."""
    path = os.path.join(REPO, TRAFFIC)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f).get("%dx%dx%d_%dbit_gpc%d" % (W, H, F, bits, gpc))
    return None if t is None else {"hbm_bytes_per_launch": int(t["hbm_bytes_per_launch"]), "replayed": True, "source": "%s (%s)" % (TRAFFIC, t.get("source", "rocprofv3 --pmc"))}


def check_gathered(og, world, G, pairs, n, res_all):
    """What rank 0 holds after the last exchanges is complete and parses: every record of every rank has the frame
    count of a GOP and exactly its used size; rank 0's own records are the rows of one of its pipelines."""
    from new_bloom_filter_repo_amd.dist import unpack_device_record
    parsed = 0
    for ob in range(2):
        if og.sizes[ob] is None:
            continue
        for r in range(world):
            for rec in og.received(ob, r):
                rows = unpack_device_record(rec, n)
                if len(rows) != pairs:
                    raise SystemExit("gathered record of rank %d (outbox %d) holds %d frames, expected %d" % (r, ob, len(rows), pairs))
                parsed += 1
                if r == 0 and not any(all(g["l"] == w["l"] and g["witness_bits"] == w["witness_bits"] and np.array_equal(g["witness"], w["witness"])
                                          and (not w["l"] or np.array_equal(g["filter"], w["filter"])) for g, w in zip(rows, res)) for res in res_all):
                    raise SystemExit("a record rank 0 kept from itself matches none of its pipelines' rows")
    return parsed


ISSUE_MODEL = "profiles/r06_issue_model.json"     # written by `python tools/make_issue_model.py` from profiles/r06_counters.json + profiles/r04_opbench2.txt + the library's ISA


def issue_roofline(W, H, F, bits, gpc, breakdown, ms_per_step, steady_ms=None):
    """Companion to the HBM roofline ('expect the kernel to sit on the integer-ALU ceiling first; report both'): the
    instruction-issue bound of the WHOLE STEP and of each of its kernels.  The bound itself is a replayed constant -- wave-instructions
    per launch from rocprofv3 counters x the issue price of this chip's vector instructions, tools/make_issue_model.py -- the fractions
    are computed HERE from the times this run measured: every kernel alone (HIP events) and the four-pipeline step."""
    path = os.path.join(REPO, ISSUE_MODEL)
    if not os.path.exists(path) or not breakdown:
        return None
    with open(path) as f:
        model = json.load(f)
    shape = model["shapes"].get("%dx%dx%d_%dbit_gpc%d" % (W, H, F, bits, gpc))
    if shape is None:
        return None
    per = {}
    for name, k in shape["kernels"].items():
        alone = breakdown.get(k["role"])
        if k["role"] in ("hashtab", "pack", "expand") or not alone:
            continue
        b = k["valu_bound_us"] / 1e3
        per[name] = {"role": k["role"], "valu_wave_instructions": k["valu"], "mean_cycles_per_valu": k["mean_valu_cycles"], "issue_bound_ms": round(b, 5),
                     "issue_bound_with_salu_ms": round(k["valu_salu_bound_us"] / 1e3, 5), "alone_ms": alone, "frac_of_alone": round(b / alone, 3),
                     "remainder_ms": round(alone - b, 5), "remainder_is": k.get("remainder_is")}
    st = shape["step"]
    out = {"bound": "instruction issue: vector wave-instructions x their issue price on 1024 SIMDs (and the same + scalar instructions)", "replayed": True,
           "step_bound_ms": st["valu_bound_ms"], "step_bound_with_salu_ms": st["valu_salu_bound_ms"], "step_ms": round(ms_per_step, 5),
           "step_frac": round(st["valu_bound_ms"] / ms_per_step, 3), "step_frac_with_salu": round(st["valu_salu_bound_ms"] / ms_per_step, 3),
           "valu_wave_instructions_per_step": st["valu"], "vector_lane_instructions_per_pixel": st["valu_per_pixel"],
           "mpixels_per_s_at_the_bound": st["mpixels_per_s_at_the_valu_bound"], "per_kernel": per,
           "shader_clock_ghz": model["shader_clock_ghz"],
           "source": "%s (python tools/make_issue_model.py: rocprofv3 SQ_INSTS_VALU / SQ_INSTS_SALU per launch of this shape, profiles/r06_counters.json, x per-opcode issue prices, "
                     "profiles/r04_opbench2.txt, weighted by the opcode mix of each kernel's loops in the in-tree library's ISA) -- the bound is a replayed constant, the fractions "
                     "use this run's times" % ISSUE_MODEL}
    if steady_ms:
        out["steady_state_frac"] = round(st["valu_bound_ms"] / steady_ms, 3)
    return out


def measured_traffic(W, H, F, bits, custom_density, gpc=1):
    """`roofline.traffic` of the headline: HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (see
    replayed_traffic).  A REPLAYED constant -- `traffic_replayed` says so at the top level of `roofline` -- null for any other workload."""
    t = None if custom_density else replayed_traffic(W, H, F, bits, gpc)
    if t is None:
        return {"traffic": None, "traffic_replayed": False}
    return {"traffic": t["hbm_bytes_per_launch"], "traffic_replayed": True, "traffic_source": t["source"] + " -- replayed constant, not measured in this run"}


def pcie_inclusive(torch, nat, coder, gop, pixels, step_s):
    """Throughput if every GOP first had to cross PCIe: one pinned-host -> HBM upload of the GOP (measured here; the Y planes
    only when the coder keeps planar luma) plus one step, serialised.  Never the headline: `value` is measured with inputs
    resident in HBM."""
    data = np.ascontiguousarray(gop[..., 0]) if coder.planar_luma else gop
    dst = coder.luma.ptr if coder.planar_luma else coder.frames.ptr
    pinned = torch.from_numpy(data.reshape(-1).view(np.uint8)).pin_memory()
    best = None
    for _ in range(3):
        coder.ctx.sync()
        t0 = time.perf_counter()
        nat.check(nat.lib().rbf_memcpy_h2d(coder.ctx.handle, dst, pinned.data_ptr(), pinned.numel()))
        coder.ctx.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(pixels / (best + step_s) / 1e6, 1), "upload_ms": round(best * 1e3, 3), "upload_gbps": round(data.nbytes / best / 1e9, 1),
            "uploaded_mb_per_gop": round(data.nbytes / 1e6, 1), "note": "upload and step serialised; pinned host memory"}


def pcie_overlapped(torch, device, coder, stream, gops, pixels, reps=24):
    """The honest ceiling of a host-fed stream: the upload of GOP i+1 (Y planes, pinned host memory, its own copy stream) runs
    UNDER the step of GOP i (two resident slots alternate).  One pipeline; whole-GOP throughput in steady state."""
    planes = [torch.from_numpy(np.ascontiguousarray(g[..., 0]).reshape(-1).view(np.uint8)).pin_memory() for g in gops[:2]]
    nbytes = planes[0].numel()
    slot = [coder.luma.tensor.view(torch.uint8)[i * nbytes:(i + 1) * nbytes] for i in range(2)]
    copy_stream = torch.cuda.Stream(device)
    done = [torch.cuda.Event(), torch.cuda.Event()]          # step i has finished reading its slot
    ready = [torch.cuda.Event(), torch.cuda.Event()]         # slot holds its GOP

    def run(count):
        for i in range(count):
            s = i & 1
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[s])              # (recorded events only: the first two waits return at once)
                slot[s].copy_(planes[s], non_blocking=True)
                ready[s].record(copy_stream)
            with torch.cuda.stream(stream):
                stream.wait_event(ready[s])
                coder.encode(s)
                done[s].record(stream)
    run(4)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run(reps)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(pixels / dt / 1e6, 1), "ms_per_gop": round(dt * 1e3, 3), "upload_gbps": round(nbytes / dt / 1e9, 1),
            "note": "upload of the next GOP's Y planes on a copy stream under the current GOP's step; one pipeline, pinned host memory"}


def _oracle_frame(args):
    """(mask, k, l, bit_array, witness) of one inter-frame from the CPU oracle."""
    import ctypes
    from oracle import oracle as orc
    prev_y, curr_y, n = args
    mask = np.ascontiguousarray(orc.residual_mask(prev_y, curr_y, 0.0).reshape(-1), dtype=np.uint8)
    p = np.uint64(int(mask.sum())) / n
    k, l = orc.optimal_params(n, p)
    if p >= orc.P_STAR or l == 0 or l >= n:
        return mask, 0.0, 0, None, None
    bit_array = np.zeros(l, dtype=np.uint8)
    witness = np.zeros(n, dtype=np.uint8)
    seeds = (ctypes.c_uint64 * 3)(*orc.SEEDS_VIDEO)
    w = orc.lib().orc_compress(mask.ctypes.data, n, l, ctypes.c_double(k), seeds, bit_array.ctypes.data, witness.ctypes.data)
    return mask, k, l, bit_array, witness[:w].copy()


def verify_all(host_gops, res_all, n, npipes):
    """Every pipeline's every frame against the CPU oracle, from the HOST frames (mask, k, l, filter, witness):
    the launch shape that was just timed is the one that is checked.  The oracle is the checker only."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.lib()
    frames_checked = 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as pool:
        for k in range(npipes):
            gop, res = host_gops[k], res_all[k]
            jobs = [(np.ascontiguousarray(gop[f][..., 0]), np.ascontiguousarray(gop[f + (0 if res[f].get("skipped") else 1)][..., 0]), n) for f in range(len(res))]
            for f, (mask, kk, l, bit_array, witness) in enumerate(pool.map(_oracle_frame, jobs)):
                r = res[f]
                if r.get("skipped"):               # a pair across a keyframe of a multi-GOP block: not coded -- zero mask row, nothing else
                    if r["mask"].any() or r["witness_bits"] or r["ones"]:
                        raise SystemExit("pipeline %d pair %d is marked skipped but has outputs" % (k, f))
                    continue
                ok = np.array_equal(np.unpackbits(r["mask"])[:n], mask) and (r["k"], r["l"]) == (kk, l)
                if ok and l:
                    ok = (np.array_equal(np.unpackbits(r["filter"])[:l], bit_array) and r["witness_bits"] == len(witness)
                          and np.array_equal(np.unpackbits(r["witness"])[:len(witness)], witness))
                if not ok:
                    raise SystemExit("pipeline %d frame %d differs from the CPU oracle" % (k, f))
                frames_checked += 1
    return {"frames": frames_checked, "pipelines": npipes, "fields": "mask, k, l, filter, witness", "seconds": round(time.perf_counter() - t0, 2)}


def _cgroup_cpu_max():
    """The container's CPU quota as the kernel states it ("max 100000" = unlimited), or None: says why 256 threads may not mean 256 cores."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            pass
    return None


def cpu_baseline(res, n, nframes):
    """CPU oracle (scalar C port of the reference loops, oracle/rbf_oracle.c) on the step's own masks:
    residual mask is not included (numpy-trivial); insert + query/witness, one core, >= ~10 s of work."""
    import ctypes
    from oracle import oracle as orc
    L = orc.lib()
    seeds = (ctypes.c_uint64 * 3)(*orc.SEEDS_VIDEO)
    frames = [r for r in res[:nframes or len(res)] if r["l"] and not r.get("skipped")]
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
    # the same port with the frames spread over the host's cores (frames are independent; ctypes drops the GIL).  Every thread owns its
    # output buffers for the whole run and touches them beforehand: allocating 2.6 MB per call (round 4) had 256 threads fight over the
    # process's address-space lock (mmap / page faults / munmap) and capped the figure at ~10x one core.
    import threading
    # as many threads as the container may actually run: its cgroup CPU quota when there is one (the GPU boxes of this pool: 16 CPUs of a
    # 256-thread host -- 256 threads under that quota measured 11x one core, round 5), else the CPUs of the affinity mask
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_cpu_max()
    try:
        q, period = quota.split()[:2]
        if q != "max":
            usable = max(1, min(usable, int(-(-int(q) // int(period)))))
    except (AttributeError, ValueError):
        pass
    threads = max(1, usable)
    per_thread = max(2, min(16, 512 // threads))  # frames per thread (the step's masks, cyclically): ~0.1 s of work each
    jobs = threads * per_thread
    lmax = max(r["l"] for r in frames)
    start_evt = threading.Event()
    ready = threading.Barrier(threads + 1)
    done = [0.0] * threads

    def worker(t):
        bit_array = np.ones(lmax, dtype=np.uint8)
        witness = np.ones(n, dtype=np.uint8)      # (pages touched)
        ready.wait()
        start_evt.wait()
        for j in range(per_thread):
            r, mask = frames[(t * per_thread + j) % len(frames)], masks[(t * per_thread + j) % len(frames)]
            L.orc_compress(mask.ctypes.data, n, r["l"], ctypes.c_double(r["k"]), seeds, bit_array.ctypes.data, witness.ctypes.data)
        done[t] = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    ready.wait()
    t0 = time.perf_counter()
    start_evt.set()
    for th in ths:
        th.join()
    t_all = max(done) - t0
    return {"value": round(px / t_total / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d passes over the step's %d masks (%d pixels each), insert+query/witness in the scalar C oracle, %.1f s"
                      % (passes, len(frames), n, t_total),
            "all_cores": {"value": round(jobs * n / t_all / 1e6, 1), "unit": "Mpixel/s", "cores": threads, "host_cpus": os.cpu_count(),
                          "usable_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_cpu_max": _cgroup_cpu_max(),
                          "speedup_over_one_core": round(jobs * n / t_all / (px / t_total), 1), "threads_rule": "min(affinity, cgroup quota)",
                          "sample": "%d frames (the step's %d masks, cyclically) over %d threads with their own pre-touched buffers, %d frames each, one frame per call, %.2f s" % (jobs, len(frames), threads, per_thread, t_all)},
            "reference_python_mpixels_per_s": {"value": 0.38, "source": "BASELINE.md (recorded constant: the reference's own Python loops, build container, 1 core)"}}


# ------------------------------------------------------------------------------------------------------------
# strong-scaling clip mode: BASELINE configs[2] (1080p x 300 frames over 8 GPUs) and configs[4] (--bits 16)
# ------------------------------------------------------------------------------------------------------------
def clip_pieces(start, stop, interval):
    """The runs of inter-frames in [start, stop): list of (first_read_frame, nframes_read).  Frame t is a keyframe
    iff t % interval == 0 (not Bloom-coded); an inter-frame t is diffed against ORIGINAL frame t-1, so a run
    t0..t1 reads frames t0-1..t1 (the first one may be another rank's: the halo)."""
    pieces, t = [], start
    while t < stop:
        if t % interval == 0:
            t += 1
            continue
        end = min(stop, (t // interval + 1) * interval)
        pieces.append((t - 1, end - t + 1))
        t = end
    return pieces


def clip_blocks(start, stop, interval, block_gops, pipelines=4, force_groups=0):
    """The blocks a rank hands to the GPU, one rbf_encode_runs launch sequence each: list of (first_read_frame, nframes_read, run_starts)
    covering the inter-frames of [start, stop).  A block's first frame is only read (a keyframe, or the frame in front of the block's
    first inter-frame: a halo inside the rank's own shard costs one frame of extra reads); run_starts are the keyframes inside the block,
    relative to its first frame -- the pairs in front of them are not coded.
    block_gops = 1: clip_pieces' runs (one call per run, round 4); N > 1: blocks of N keyframe intervals, cut at keyframes;
    0 (auto): the rank's frames in min(pipelines, about one block per 72 inter-frames) contiguous ranges of EQUAL length (cut anywhere, not
    only at keyframes), at most 128 frames each -- every pipeline of the rank gets one block of the same size per pass.  A share of fewer
    blocks than pipelines (N >= 2) is NOT cut smaller -- every block pays the query kernel's hashing prologue once, and the insert and query
    kernels own their CUs, so small blocks only add prologues (profiles/r06_shard_proxy_sweep.txt) -- the idle pipelines take the NEXT pass
    instead (run_clip's pass slots).  force_groups > 0 overrides the number of ranges (--clip-groups: the sweep behind that statement)."""
    if block_gops == 1:
        return [(f0, cnt, []) for f0, cnt in clip_pieces(start, stop, interval)]
    first = start if start == 0 or start % interval == 0 else start - 1          # dist.halo_start
    if block_gops > 1:
        cuts = [first] + [t for t in range(first + 1, stop) if t % interval == 0][block_gops - 1::block_gops] + [stop]
        if first % interval:                                                     # a shard that starts inside an interval: its partial interval counts as one
            cuts = [first] + [t for t in range(first + 1, stop) if t % interval == 0][max(0, block_gops - 2)::block_gops] + [stop]
        ranges = list(zip(cuts[:-1], cuts[1:]))
    else:
        coded = sum(1 for t in range(start, stop) if t % interval)
        groups = max(1, min(pipelines, (coded + 36) // 72), (stop - start + 126) // 127)
        if force_groups > 0:
            groups = max(min(force_groups, max(1, coded)), (stop - start + 126) // 127)
        base, extra = divmod(stop - start, groups)
        ranges, a = [], start
        for g in range(groups):
            b = a + base + (1 if g < extra else 0)
            ranges.append((a, b))
            a = b
    blocks = []
    for a, b in ranges:                                                          # the range's inter-frames, read from their predecessor on
        lo = a if (a % interval == 0 or a == first) else a - 1
        lo = max(lo, first)
        if b - lo < 2 or not any(t % interval for t in range(max(lo + 1, a), b)):
            continue
        blocks.append((lo, b - lo, [t - lo for t in range(lo + 1, b) if t % interval == 0]))
    return blocks


def run_clip(args, env, T, I, bits, steps, warmup, verify=True, proxy=None, clip_cache=None, pool=None):
    """One clip of T frames (W x H YUV444, `bits` per sample), keyframe every I: the inter-frames shard over the ranks by
    CONTIGUOUS FRAME RANGE with one halo frame (dist.shard_range / halo_start), every rank codes its frames in blocks of one
    rbf_encode_runs launch sequence each (clip_blocks) and packs each block's record on the device; with N > 1 the records travel to
    rank 0 in one exact-size gather per pass (sizes all-gathered together with an error flag, then one point-to-point message per peer;
    rank 0's own records stay out of the collective).  A pass = the whole clip once; timed with and (N > 1) without the gather.  Strong
    scaling: total work is fixed as N grows.

    PASS SLOTS: a rank whose share is fewer blocks than it has pipelines (N >= 2: 145 / 73 / 37 inter-frames are 2 / 1 / 1 blocks) keeps
    `pipelines // blocks` SETS of coders -- own contexts, streams, masks, filters, witnesses -- and consecutive passes rotate over them,
    so that the mask stage (HBM), the host's parameter round trip, the compaction and the packing of pass p+1 run under the CU-exclusive
    insert and query kernels of pass p, exactly as the four pipelines of the N = 1 pass overlap each other.  Records are double-buffered
    and the gather of pass p is issued after pass p+1 has been enqueued, so the GPU works while the host sits in the size exchange.
    `pass_latency_ms` (one pass alone, synchronised before and after) is reported beside the sustained `ms_per_pass`.

    proxy = (N, r): this single process runs what rank r of an N-way split would run (the shard arithmetic, blocks, pass slots and the
    packed record of that rank; the gather is stubbed) -- bench.py's `shard_proxy` leg.  clip_cache: dict that keeps the whole synthetic
    clip of a bit depth (host and HBM) between calls; pool (default: clip_cache): dict that keeps the pipelines (streams, contexts) and the
    arena of the coders' buffers between calls.  Returns the result dict on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    world, rank, local_rank, device, use_dist = env
    from new_bloom_filter_repo_amd import _native as nat
    from new_bloom_filter_repo_amd.dist import shard_range, halo_start, gather_device_records, unpack_device_record
    from new_bloom_filter_repo_amd.gop import GopCoder, TorchArena, torch_allocator
    from new_bloom_filter_repo_amd.synthetic import make_clip_shard, P_KSTAR_2_3

    if proxy is not None and (world != 1 or use_dist):
        raise SystemExit("run_clip(proxy=...) is a single-process measurement")
    s_world, s_rank = proxy if proxy is not None else (world, rank)          # the split the shard arithmetic sees
    W, H = args.width, args.height
    n = W * H
    dtype = np.uint8 if bits == 8 else np.uint16
    start, stop = shard_range(T, s_world, s_rank)
    first = halo_start(start, I)
    density = args.density or P_KSTAR_2_3
    if clip_cache is not None:                    # frames first..stop-1 of the SAME clip on every rank (here: a view of the cached whole clip)
        key = (W, H, T, I, bits, density)
        if key not in clip_cache:
            for k2 in [k2 for k2 in clip_cache if k2[0] != "pipelines"]:
                del clip_cache[k2]                # one clip at a time: 300 x 1080p x 16 bit is 3.7 GB of host memory (the shared pipelines stay)
            clip_cache[key] = make_clip_shard(3000, W, H, 0, T, I, p=density, dtype=dtype)
        shard = clip_cache[key][first:stop]
    else:
        shard = make_clip_shard(3000, W, H, first, stop, I, p=density, dtype=dtype)
    BG = max(0, args.clip_block_gops)
    NP = max(1, args.streams)
    FG = max(0, args.clip_groups)
    pieces = clip_blocks(start, stop, I, BG, NP, FG)
    total_pairs = sum(max(0, min(T, (g + 1) * I) - g * I - 1) for g in range((T + I - 1) // I))
    my_pairs = sum(c - 1 - len(rs) for _, c, rs in pieces)
    max_pieces = max(len(clip_blocks(*shard_range(T, s_world, r), I, BG, NP, FG)) for r in range(s_world))     # every rank can compute every rank's count

    nblocks = len(pieces)
    nsets = 1 if not nblocks else max(1, min(args.clip_pass_slots or NP, NP // nblocks))
    nstreams = max(1, min(NP, nblocks * nsets))
    if pool is None:
        pool = clip_cache
    if pool is not None:
        # The calls that share a pool (the clip legs of one process, every rank of the proxies) share their pipelines -- streams, library contexts
        # (scratch, the process-wide hash table) -- and carve their coders' buffers out of ONE arena: where a call's buffers land in HBM
        # decided its speed by up to 15 % (the same calls of every run, fast in a process of their own), and a rank's work is what is measured.
        pk = ("pipelines", NP)
        if pk not in pool:
            st = [torch.cuda.Stream(device) for _ in range(NP)]
            pool[pk] = (st, [nat.Context(local_rank, x.cuda_stream) for x in st], TorchArena(device, 1536 << 20))
        all_streams, all_ctxs, arena = pool[pk]
        streams, ctxs = all_streams[:nstreams], all_ctxs[:nstreams]
        arena.used = 0
        alloc = arena
    else:
        streams = [torch.cuda.Stream(device) for _ in range(nstreams)]
        ctxs = [nat.Context(local_rank, s.cuda_stream) for s in streams]
        alloc = torch_allocator(device)
    if args.force_bits:
        for c in ctxs:
            c.force_generic(args.force_bits)
    planar = not args.interleaved
    frame_bytes = n * (1 if planar else 3) * (bits // 8)
    if clip_cache is not None:
        # ONE resident copy of the whole clip for every call that shares the cache (the N = 1 leg and all the proxies): a rank's share is a
        # view of it, as a rank's frames are views of what its loader left in HBM -- and every measurement sees the same placement in HBM
        # (re-uploading per call made single ranks 15 % slower, reproducibly the same calls of a run: the allocator's luck, not the rank's work)
        dkey = ("device",) + key + (planar,)
        if dkey not in clip_cache:
            whole = clip_cache[key]
            res = np.ascontiguousarray(whole[..., 0]) if planar else np.ascontiguousarray(whole)
            clip_cache[dkey] = torch.from_numpy(res.reshape(-1).view(np.uint8)).to(device)
            del res
        frames_t = clip_cache[dkey][first * frame_bytes:stop * frame_bytes]
    else:
        resident = np.ascontiguousarray(shard[..., 0]) if planar else np.ascontiguousarray(shard)       # planar: only the Y planes of the shard (+ halo) are resident in HBM
        frames_t = torch.from_numpy(resident.reshape(-1).view(np.uint8)).to(device)
        del resident

    class View:                                   # a run's frames inside the shard buffer
        def __init__(self, off, nbytes):
            self.ptr, self.nbytes = frames_t.data_ptr() + off, nbytes
    NREC = max(2, nsets)                          # record buffers per block: pass p packs into buffer p % NREC (the gather of pass p runs under pass p+1)
    sets, records = [], [[] for _ in range(NREC)]
    for s in range(nsets):
        coders = []
        for i, (f0, cnt, rs) in enumerate(pieces):
            view = View((f0 - first) * frame_bytes, cnt * frame_bytes)
            coders.append(GopCoder(ctxs[(s * nblocks + i) % nstreams], W, H, cnt, channels=3, sample_bytes=bits // 8, allocator=alloc,
                                   frames_block=None if planar else view, planar_luma=planar, keep_interleaved=False, luma_block=view if planar else None, run_starts=rs))
        sets.append(coders)
    for b in range(NREC):
        for i, (f0, cnt, rs) in enumerate(pieces):
            records[b].append(sets[0][i]._out_alloc(int(nat.lib().rbf_record_max_bytes(cnt - 1, n))))
    can_gather = use_dist and not args.no_gather
    state = {"p": 0, "pending": None, "free": [None] * NREC}

    def do_gather(pend):
        b, evs = pend
        cur = torch.cuda.current_stream(device)
        for ev in evs:
            cur.wait_event(ev)
        got = gather_device_records([r.tensor for r in records[b]], device, max_records=max_pieces)
        ev = torch.cuda.Event()
        ev.record(cur)                            # buffer b may be packed again once what the gather enqueued on this stream has read it
        state["free"][b] = ev
        return got

    def step(gather):
        p = state["p"]
        state["p"] += 1
        s, b = p % nsets, p % NREC
        evs = []
        for i, c in enumerate(sets[s]):
            st = streams[(s * nblocks + i) % nstreams]
            with torch.cuda.stream(st):
                c.encode()
                if gather and state["free"][b] is not None:
                    st.wait_event(state["free"][b])
                c.pack(records[b][i])
                if gather:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
        if not gather:
            return None
        prev, state["pending"] = state["pending"], (b, evs)
        return do_gather(prev) if prev is not None else None

    def finish_gather():
        prev, state["pending"] = state["pending"], None
        return do_gather(prev) if prev is not None else None

    def barrier():
        if use_dist:
            bounded_barrier(dist, device)
        torch.cuda.synchronize(device)

    def timed(gather):
        # warm-up: the passes asked for, then more until 30 ms have gone by (a pass is ~1 ms; the first ones after start-up run
        # 8-10 % slower than the steady state, like the headline's steps); every rank runs the same number
        got, spent, extra = None, 0.0, 0
        for _ in range(max(1, warmup)):
            step(gather)
        finish_gather()
        barrier()
        while extra < 400:
            t0 = time.perf_counter()
            for _ in range(nsets):
                step(gather)
            finish_gather()
            torch.cuda.synchronize(device)
            spent += time.perf_counter() - t0
            extra += 1
            done = spent >= 0.030
            if use_dist:
                flag = torch.tensor([1 if done else 0], dtype=torch.int64, device=COMM_DEVICE)
                all_reduce_b(dist, flag, dist.ReduceOp.MIN)
                done = bool(flag.item())
            if done:
                break
        # REGIONS regions of exactly `steps` passes, each between barrier + synchronize, max over ranks; the MEDIAN region counts (as in the
        # headline): something on these boxes stalls a process for 10-20 ms every second or two (a 45 ms region that catches it reads 15-50 %
        # slow, reproducibly for whichever call is running at that moment) and a median of five shrugs one hit off
        REGIONS = 5
        times, got = [], None
        gc.collect()
        for _ in range(REGIONS):
            barrier()
            gc.disable()                          # (a generation-2 collection of this process's heap -- the clip, the other legs' results -- is a millisecond the GPU starves)
            t0 = time.perf_counter()
            for _ in range(steps):
                step(gather)
            got = finish_gather()                 # the last pass's records (every pass was gathered inside the region: `steps` exchanges)
            torch.cuda.synchronize(device)
            barrier()
            elapsed = time.perf_counter() - t0
            gc.enable()
            if use_dist:
                te = torch.tensor([elapsed], dtype=torch.float64, device=COMM_DEVICE)
                all_reduce_b(dist, te, dist.ReduceOp.MAX)
                elapsed = float(te.item())
            times.append(elapsed)
        state["regions_ms"] = [round(t * 1e3, 3) for t in times]
        return sorted(times)[REGIONS // 2], got

    elapsed_plain, _ = timed(False)
    # one pass ALONE (nothing of a neighbouring pass under it): what a caller waits for who has only this one clip
    lat = []
    for _ in range(9):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        step(False)
        torch.cuda.synchronize(device)
        lat.append(time.perf_counter() - t0)
    latency = sorted(lat)[len(lat) // 2]
    elapsed_gather, got = timed(True) if can_gather else (None, None)

    verified = None
    if verify and not args.no_verify:
        # every rank checks ITS frames -- the rows of EVERY pass slot -- against the CPU oracle from the host frames; rank 0 also parses what it received
        for s in range(nsets):                    # (the slots' last passes may be older than the latency passes: run each once more)
            for c in sets[s]:
                c.encode()
        host = [np.stack([shard[f0 - first + j] for j in range(cnt)]) for f0, cnt, _ in pieces]
        res_sets = [[c.results() for c in coders] for coders in sets]
        frames_ok = 0
        for s in range(nsets):
            v = verify_all(host, res_sets[s], n, nblocks) if nblocks else {"frames": 0}
            if s and v["frames"] != frames_ok:
                raise SystemExit("clip: pass slot %d verified %d frames, slot 0 %d" % (s, v["frames"], frames_ok))
            frames_ok = v["frames"]
        res_all = res_sets[0] if nsets else []
        cnt_t = torch.tensor([frames_ok], dtype=torch.int64, device=COMM_DEVICE)
        if use_dist:
            all_reduce_b(dist, cnt_t)
        want = total_pairs if proxy is None else my_pairs
        verified = {"frames": int(cnt_t.item()), "of": want, "pass_slots_checked": nsets,
                    "fields": "mask, k, l, filter, witness" + ("; frames arrive on rank 0 in clip order" if proxy is None else "; the packed record parses to the same rows")}
        if verified["frames"] != want:
            raise SystemExit("clip: %d of %d inter-frames verified" % (verified["frames"], want))
        if proxy is not None and nblocks:
            # the gather is stubbed: what WOULD travel is the packed record of every block -- parse it and compare it with the rows
            b = (state["p"] - 1) % NREC
            torch.cuda.synchronize(device)
            parsed = 0
            for i in range(nblocks):
                rows = unpack_device_record(records[b][i].tensor.view(torch.uint8), n)
                if len(rows) != len(res_all[i]):
                    raise SystemExit("shard proxy: a packed record holds %d rows, its block %d" % (len(rows), len(res_all[i])))
                for a, w in zip(rows, res_all[i]):
                    if bool(a.get("skipped")) != bool(w.get("skipped")):
                        raise SystemExit("shard proxy: skipped flags differ")
                    if a.get("skipped"):
                        continue
                    if not (a["l"] == w["l"] and a["k"] == w["k"] and a["witness_bits"] == w["witness_bits"] and np.array_equal(a["witness"], w["witness"])
                            and (not w["l"] or np.array_equal(a["filter"], w["filter"]))):
                        raise SystemExit("shard proxy: a packed record differs from its block's rows")
                    parsed += 1
            verified["records_parsed"] = parsed
        if rank == 0 and got is not None:
            # every coded frame of the clip, in clip order, rebuilt from what ARRIVED: rank-major records, each a block's rows; the rows of
            # rank 0 are compared with its own results field by field, the others' frame indices with the shard arithmetic every rank can do
            parsed, expect = 0, []
            for r in range(world):
                for f0, cnt, rs in clip_blocks(*shard_range(T, world, r), I, BG, NP, FG):
                    expect.append([f0 + 1 + j for j in range(cnt - 1) if (j + 1) not in rs])
            if len(got) != len(expect):
                raise SystemExit("rank 0 holds %d records, the shards make %d blocks" % (len(got), len(expect)))
            order = []
            for b, frames_of_block in zip(got, expect):
                rows = [x for x in unpack_device_record(b, n) if not x.get("skipped")]
                if len(rows) != len(frames_of_block):
                    raise SystemExit("a gathered record holds %d coded frames, its block has %d" % (len(rows), len(frames_of_block)))
                order += frames_of_block
                parsed += len(rows)
            if order != [t for t in range(T) if t % I]:
                raise SystemExit("the gathered records do not cover the clip's inter-frames in order")
            mine = [x for res in res_all for x in res if not x.get("skipped")]
            theirs = [x for b in got[:nblocks] for x in unpack_device_record(b, n) if not x.get("skipped")]
            for a, w in zip(theirs, mine):
                if not (a["l"] == w["l"] and a["witness_bits"] == w["witness_bits"] and np.array_equal(a["witness"], w["witness"]) and (not w["l"] or np.array_equal(a["filter"], w["filter"]))):
                    raise SystemExit("a record rank 0 kept from itself differs from its own rows")
            verified["records_parsed_on_rank0"] = parsed
            verified["bytes_gathered_on_rank0"] = int(sum(b.numel() for b in got))
            if parsed != total_pairs:
                raise SystemExit("rank 0 received %d frame records, expected %d" % (parsed, total_pairs))
    out = None
    if rank == 0:
        e = elapsed_gather if elapsed_gather is not None else elapsed_plain
        pairs_timed = total_pairs if proxy is None else my_pairs
        out = {"value": round(pairs_timed * n * steps / e / 1e6, 2), "unit": "Mpixel/s", "ms_per_pass": round(e / steps * 1e3, 4),
               "pass_latency_ms": round(latency * 1e3, 4),
               "gather_to_rank0": elapsed_gather is not None,
               "without_gather": {"value": round(pairs_timed * n * steps / elapsed_plain / 1e6, 2), "ms_per_pass": round(elapsed_plain / steps * 1e3, 4)},
               "passes": steps, "regions_ms": state.get("regions_ms"), "timing": "median of 5 regions of `passes` passes each", "scaling": "strong", "n_gpus": world,
               "workload": "%dx%d YUV444 %d-bit synthetic clip of %d frames, keyframe every %d (%d inter-frames/pass over %d GPU%s), k*=2.3, threshold 0"
                           % (W, H, bits, T, I, total_pairs, s_world, "s" if s_world > 1 else ""),
               "sharding": "contiguous frame ranges + 1 halo frame (dist.shard_range)", "inter_frames_rank0": my_pairs,
               "gops_per_call": BG or "auto", "blocks_rank0": [cnt - 1 - len(rs) for _, cnt, rs in pieces], "calls_per_pass_rank0": nblocks,
               "pass_slots": nsets, "pipelines_used": nstreams,
               "backend": (args.backend if use_dist else None), "ranks_share_one_device": bool(args.one_device),
               "batching": "one rbf_encode_gop per run of inter-frames" if BG == 1 else "every rank hands blocks of several keyframe intervals to ONE rbf_encode_runs launch sequence each (cut at the keyframes; inter-frames per block: blocks_rank0); consecutive passes rotate over `pass_slots` sets of coders",
               "layout": "planar Y" if planar else "interleaved", "verified_vs_oracle": verified}
        if proxy is not None:
            out["proxy_of"] = {"world": s_world, "rank": s_rank, "frames": [start, stop], "first_frame_read": first, "gather": "stubbed (single process)"}
    # release the clip before the next leg
    for coders in sets:
        for c in coders:
            c.close()
    if pool is None:
        for c in ctxs:
            c.close()
    del sets, records, frames_t, ctxs
    torch.cuda.empty_cache()
    return out


def shard_proxy_leg(args, env, T, I, steps, clip_cache, bits, base):
    """What ONE rank of an N-way split of BASELINE configs[2] / [4] does, for N = 2, 4, 8 and EVERY rank of the split, on this one GPU:
    run_clip(proxy=(N, r)) -- the rank's frame range + halo, its blocks, its pass slots, its packed record; the gather is stubbed.
    The pass time of an N-GPU run is the slowest rank's (max over ranks, as the contract's timing takes it), so
    predicted speed-up = (N = 1 pass) / (slowest rank's pass) and efficiency = speed-up / N.  A PREDICTION from single-GPU runs: no
    xGMI transfer, no RCCL launch, no host of eight processes is in it.  The slowest rank's frames are verified against the oracle.
    base: the clip leg's result at N = 1 for this bit depth.  Returns {"N1": ..., "N2": ..., "N4": ..., "N8": ...}."""
    from new_bloom_filter_repo_amd.dist import shard_range
    leg = {}
    for N in (2, 4, 8):
        coded = [sum(1 for t in range(*shard_range(T, N, r)) if t % I) for r in range(N)]
        heavy = max(range(N), key=lambda r: (coded[r], -r))
        per_rank, checked = [], None
        for r in range(N):
            res = run_clip(args, env, T, I, bits, steps * N, 2, verify=(r == heavy), proxy=(N, r), clip_cache=clip_cache)      # (N x the passes: the timed region stays ~40 ms)
            per_rank.append(res)
            if r == heavy:
                checked = res["verified_vs_oracle"]
        slow = max(range(N), key=lambda r: per_rank[r]["ms_per_pass"])
        ms = per_rank[slow]["ms_per_pass"]
        row = {"inter_frames_per_rank": coded, "ms_per_pass_per_rank": [x["ms_per_pass"] for x in per_rank],
               "pass_latency_ms_per_rank": [x["pass_latency_ms"] for x in per_rank],
               "slowest_rank": slow, "ms_per_pass": ms, "pass_latency_ms": per_rank[slow]["pass_latency_ms"],
               "blocks_per_pass": per_rank[slow]["blocks_rank0"], "pass_slots": per_rank[slow]["pass_slots"],
               "mpixels_per_s_of_the_rank": per_rank[slow]["value"],
               "verified_vs_oracle": dict(checked or {}, rank=heavy)}
        if base:
            row["predicted_speedup"] = round(base["without_gather"]["ms_per_pass"] / ms, 2)
            row["predicted_efficiency"] = round(base["without_gather"]["ms_per_pass"] / ms / N, 3)
            row["predicted_clip_mpixels_per_s"] = round(base["without_gather"]["value"] * base["without_gather"]["ms_per_pass"] / ms, 1)
        leg["N%d" % N] = row
    if base:
        leg["N1"] = {"ms_per_pass": base["without_gather"]["ms_per_pass"], "pass_latency_ms": base.get("pass_latency_ms"), "blocks_per_pass": base["blocks_rank0"], "pass_slots": base.get("pass_slots")}
    return leg


SHARD_PROXY_WHAT = {"what": "one rank's share of the clip exactly as run_clip runs it at N = 2 / 4 / 8 -- shard range + halo frame, blocks, pass slots, record packed on the device, "
                            "gather stubbed -- timed for EVERY rank of the split on ONE GPU; predicted speed-up = N=1 pass / slowest rank's pass, efficiency = speed-up / N",
                    "prediction_note": "a PREDICTION from single-GPU runs: the exact-size gather over xGMI (~4.4 MB per 290 frames in total), RCCL's launches and eight host processes are not in it"}


def clip_main(args):
    """`--clip-frames F`: only the strong-scaling clip mode (BASELINE configs[2]; `--bits 16` = configs[4]), its own JSON line."""
    import torch.distributed as dist
    env = init_dist(args)
    world, rank, use_dist = env[0], env[1], env[4]
    if args.proxy:
        parts = [int(x) for x in args.proxy.split(",")]
        cache = {}
        rows = [run_clip(args, env, args.clip_frames, args.keyframe_interval, args.bits, args.steps, args.warmup, proxy=(parts[0], r), clip_cache=cache)
                for r in (parts[1:2] or range(parts[0]))]
        slow = max(rows, key=lambda x: x["ms_per_pass"])
        print(json.dumps({"metric": "one rank's share of the clip (shard proxy)", "proxy_world": parts[0], "slowest": slow,
                          "ms_per_pass_per_rank": [x["ms_per_pass"] for x in rows], "pass_latency_ms_per_rank": [x["pass_latency_ms"] for x in rows]}), flush=True)
        return
    set_phase("clip")
    r = run_clip(args, env, args.clip_frames, args.keyframe_interval, args.bits, args.steps, args.warmup)
    set_phase("shutdown")
    if use_dist:
        bounded_barrier(dist, env[3], "final barrier")
        dist.destroy_process_group()
    if rank == 0:
        out = {"metric": METRIC, "value": r["value"], "unit": "Mpixel/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_pass"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": r["workload"], "sharding": r["sharding"], "inter_frames_rank0": r["inter_frames_rank0"], "pass_slots": r.get("pass_slots"), "pass_latency_ms": r.get("pass_latency_ms"),
                          "gather_to_rank0": r["gather_to_rank0"], "without_gather": r["without_gather"],
                          "gather": "exact-size: all_gather of lengths + error flag, then grouped send/recv of payloads; rank 0's own records are not sent",
                          "launcher": os.environ.get("RBF_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "single process"), "rccl_ranks": world if use_dist else 0},
               "verified_vs_oracle": r["verified_vs_oracle"]}
        print_line(out)


if __name__ == "__main__":
    main()
