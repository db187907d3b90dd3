cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./build/bench_insert > gpurun_out/r02_bench_insert_a.txt 2>&1; cat gpurun_out/r02_bench_insert_a.txt
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02_tests_e.txt 2>&1; tail -5 gpurun_out/r02_tests_e.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_d.json 2> gpurun_out/r02_bench_d.err; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_d.json'));print(d['value'],d['ms_per_step'],d['kernels_ms_per_step_alone'],d['roofline']['frac'])"; tail -3 gpurun_out/r02_bench_d.err
