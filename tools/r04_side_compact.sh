# round 4: side-stream compaction A/B (default bench command minus the CPU leg / clips / legs), with and without begin-ahead
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
python -m pytest tests/test_gpu_bench_shape.py -q -x -k "side_stream or two_phase or config2" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
q='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print("%-34s %8.0f Mpixel/s  %.4f ms/step  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], (d.get("verified_vs_oracle") or {}).get("frames")))'
for rep in 1 2; do
for a in "" "--side-compact" "--begin-ahead 1" "--begin-ahead 3" "--streams 3" "--streams 6" "--streams 6 --begin-ahead 2"; do
  python bench.py --no-cpu-baseline --no-clips --no-legs --no-kernel-timing $a 2>/dev/null | python -c "$q" "${a:-side compact (default)}"
done; done > $O/side_compact.txt 2>&1
cat $O/side_compact.txt
