# quick check after a kernel change: parity suite, then the headline (no legs) with one and four pipelines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
NL="--no-cpu-baseline --no-clips --no-legs"
for st in 4 1; do timeout 300 python bench.py $NL --streams $st 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('streams $st: %.0f Mpixel/s, %.4f ms/step, alone %s verified %s' % (d['value'], d['ms_per_step'], d['kernels_ms_per_step_alone'], d.get('verified_vs_oracle',{}).get('frames')))"; done | tee $O/bench.txt
