set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./build/bench_query > gpurun_out/r02_bench_query_c.txt 2>&1; head -28 gpurun_out/r02_bench_query_c.txt
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02_tests_d.txt 2>&1; tail -5 gpurun_out/r02_tests_d.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_c.json 2> gpurun_out/r02_bench_c.err; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_c.json'));print(d['value'],d['ms_per_step'],d['kernels_ms_per_step_alone'],d['roofline']['frac'])"; tail -3 gpurun_out/r02_bench_c.err
