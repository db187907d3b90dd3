cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02_tests_g.txt 2>&1; tail -3 gpurun_out/r02_tests_g.txt
for args in "--width 3840 --height 2160 --frames 9 --steps 40" "--width 2560 --height 1440 --frames 30 --steps 60" "--width 5120 --height 2880 --frames 9 --steps 20"; do
timeout 600 python bench.py --no-cpu-baseline $args > gpurun_out/r02_bench_f.json 2> gpurun_out/r02_bench_f.err; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_f.json'));print(d['config']['workload'][:40], d['value'],d['ms_per_step'],d['kernels_ms_per_step_alone'],d.get('verified_vs_oracle'))"; tail -2 gpurun_out/r02_bench_f.err
done
