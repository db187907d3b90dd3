# round 4: how the host feeds the pipelines (begin-ahead depth, host threads), on the GPU box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_gpu_bench_shape.py -q -x -k "two_phase or config2 or fused_mask" > $O/tests_two_phase.txt 2>&1
q='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print("%-28s %8.0f Mpixel/s  %.4f ms/step  host %.4f ms/step  requested-region %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_ms_per_step"], (d.get("requested_region") or {}).get("ms_per_step")))'
for st in 4 6; do
for a in 0 1 2 3 5; do
  python bench.py --no-cpu-baseline --no-verify --no-clips --no-legs --no-kernel-timing --streams $st --begin-ahead $a 2>/dev/null | python -c "$q" "streams=$st ahead=$a"
done
python bench.py --no-cpu-baseline --no-verify --no-clips --no-legs --no-kernel-timing --streams $st --host-threads 1 2>/dev/null | python -c "$q" "streams=$st host-threads"
done > $O/feed_sweep.txt 2>&1
python bench.py --no-clips > $O/bench_default_noclips.json 2> $O/bench_default_noclips.err
