# the insert with straight-line probes / explicit LDS addresses / the decode's fast path against the library before (build/ablate/librbf_base.so), same box, alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04iab; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
NL="--no-cpu-baseline --no-clips --no-legs"
run() { tag="$1"; shift; "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-20s %.0f Mpixel/s, %.4f ms/step, alone %s verified %s' % ('$tag', d['value'], d['ms_per_step'], d['kernels_ms_per_step_alone'], d.get('verified_vs_oracle',{}).get('frames')))"; }
{
for rep in 1 2 3; do
run "before" env RBF_LIB_PATH=$GRAFT_REPO_ROOT/build/ablate/librbf_base.so python bench.py $NL
run "after" python bench.py $NL
done
run "after, p=0.05" python bench.py $NL --density 0.05
run "after, p=0.2" python bench.py $NL --density 0.2
} | tee $O/ab.txt
