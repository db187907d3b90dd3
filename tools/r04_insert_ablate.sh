# timing-only ablations of k_insert_tab (libraries built with -DRBF_INSERT_ABLATE=N: wrong results, --no-verify): what the gather costs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04abl; mkdir -p $O
NL="--no-cpu-baseline --no-clips --no-legs --no-verify"
for a in ${ABL:-0 2 3 4 6}; do
  lib=$GRAFT_REPO_ROOT/build/ablate/librbf_a$a.so; [ $a = 0 ] && lib=$GRAFT_REPO_ROOT/new_bloom_filter_repo_amd/librbf_hip.so
  for st in 1 4; do RBF_LIB_PATH=$lib timeout 300 python bench.py $NL --streams $st 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('ablate $a streams $st: %.0f Mpixel/s, %.4f ms/step, alone %s' % (d['value'], d['ms_per_step'], d['kernels_ms_per_step_alone']))"; done
done | tee $O/ablate.txt
