cd $GRAFT_REPO_ROOT
for st in 1 2 3 4 6; do
timeout 600 python bench.py --no-cpu-baseline --no-verify --streams $st > gpurun_out/r02_bench_s.json 2> gpurun_out/r02_bench_s.err; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_s.json'));print($st, d['value'],d['ms_per_step'],d['kernels_ms_per_step_alone'])"
done
