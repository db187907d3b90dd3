#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
( time python -m pytest tests/test_gpu_runs.py tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_gpu_surface.py -q -x ) > $O/tests.txt 2>&1; tail -15 $O/tests.txt
for b in 0 30 60 120; do for l in 1 2 3; do E2E_BRIEF=1 python tools/e2e_leg.py $b $l 2>&1 | tail -2; done; done > $O/e2e.txt 2>&1; cat $O/e2e.txt
python - > $O/zlib.txt 2>&1 <<'P'
import zlib, numpy as np, time, os
from concurrent.futures import ThreadPoolExecutor
a = np.random.default_rng(1).integers(0, 256, 1920*1080*3, dtype=np.uint8).tobytes()
t = time.perf_counter(); zlib.compress(a, 9); d1 = time.perf_counter() - t
print("one 6.2 MB zlib-9 job of noise: %.3f s = %.1f MB/s; os.cpu_count %d, affinity %d, cpu.max %s" % (d1, len(a) / d1 / 1e6, os.cpu_count(), len(os.sched_getaffinity(0)), open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None))
for th in (8, 16, 32, 64):
    with ThreadPoolExecutor(th) as pool:
        t = time.perf_counter(); list(pool.map(lambda _: zlib.compress(a, 9), range(64))); dt = time.perf_counter() - t
    print("64 jobs on %d threads: %.3f s = %.0f MB/s aggregate = %.1f cores' worth" % (th, dt, 64 * len(a) / dt / 1e6, 64 * d1 / dt))
P
cat $O/zlib.txt
NL="--no-cpu-baseline --no-legs --no-clips"
for bits in 8 16; do for g in 1 4; do python bench.py $NL --bits $bits --gops-per-call $g $( [ $g = 4 ] && echo --gops-per-pipeline 1 ) --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_u${bits}_g$g.json; done; done
python tools/show_bench.py $O/bench_u*.json
for bits in 8 16; do python bench.py --clip-frames 300 --steps 40 --warmup 2 --bits $bits 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('u$bits clip300 N=1: %.4f ms/pass (%.0f Mpixel/s) verified %s' % (d['ms_per_step'], d['value'], d['verified_vs_oracle']))"; done
SKIP_DEFAULT=1 SHAPES="u1 u4 g4" INST_SHAPES="g1 g4 u4" bash tools/r06_profile.sh > $O/profile.log 2>&1; grep -v "at::native\|rocclr" $O/profile.log | tail -70
