set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./build/bench_query > gpurun_out/r02_bench_query_b.txt 2>&1; head -30 gpurun_out/r02_bench_query_b.txt
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py tests/test_gpu_dist_nccl.py -x -q -m gpu > gpurun_out/r02_tests_c.txt 2>&1; tail -15 gpurun_out/r02_tests_c.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err; tail -c 1500 gpurun_out/r02_bench_b.json; tail -5 gpurun_out/r02_bench_b.err
