cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r02_tests_i.txt 2>&1; tail -3 gpurun_out/r02_tests_i.txt
for st in 1 4; do
timeout 600 python bench.py --no-cpu-baseline --streams $st > gpurun_out/r02_bench_h.json 2> gpurun_out/r02_bench_h.err; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_h.json'));print($st, d['value'],d['ms_per_step'],d['kernels_ms_per_step_alone'], d['verified_vs_oracle'])"; tail -1 gpurun_out/r02_bench_h.err
done
