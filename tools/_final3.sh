cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06final3; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python -m pytest tests/test_gpu_dist_shared.py -q -x 2>&1 | tail -2
