cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06final2; mkdir -p $O
python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
( time python -m pytest tests -m gpu -q ) > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
timeout 150 python tools/fuzz_surface.py 60 > $O/fuzz_surface.txt 2>&1; tail -1 $O/fuzz_surface.txt
python bench.py --steps 20 --warmup 5 --no-legs --no-clips --no-cpu-baseline > $O/bench_short.json 2>/dev/null; python tools/show_bench.py $O/bench_short.json
