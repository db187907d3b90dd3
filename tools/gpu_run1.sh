set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./build/opbench > gpurun_out/r02_opbench.txt 2>&1; tail -5 gpurun_out/r02_opbench.txt
./build/bench_query > gpurun_out/r02_bench_query_a.txt 2>&1; head -12 gpurun_out/r02_bench_query_a.txt
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_dist_nccl.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02_tests_a.txt 2>&1; tail -15 gpurun_out/r02_tests_a.txt
timeout 600 python bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; tail -c 3000 gpurun_out/r02_bench_a.json; tail -5 gpurun_out/r02_bench_a.err
