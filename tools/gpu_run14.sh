cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02_tests_f.txt 2>&1; tail -3 gpurun_out/r02_tests_f.txt
for extra in "" "--hash-cache"; do
timeout 600 python bench.py --no-cpu-baseline $extra > gpurun_out/r02_bench_e.json 2> gpurun_out/r02_bench_e.err; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_e.json'));print(d['value'],d['ms_per_step'],d['kernels_ms_per_step_alone'],d['roofline']['frac'])"; tail -2 gpurun_out/r02_bench_e.err
done
