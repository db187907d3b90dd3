cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./build/bench_query 2>&1 | head -6
./build/bench_insert 2>&1 | head -5
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -m gpu > gpurun_out/r02_tests_h.txt 2>&1; tail -3 gpurun_out/r02_tests_h.txt
for extra in "" "--rebuild-hash-table" ""; do
timeout 600 python bench.py --no-cpu-baseline $extra > gpurun_out/r02_bench_g.json 2> gpurun_out/r02_bench_g.err; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_g.json'));print(d['value'],d['ms_per_step'],d['kernels_ms_per_step_alone'],d['roofline']['frac'])"; tail -2 gpurun_out/r02_bench_g.err
done
