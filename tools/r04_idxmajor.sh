# pos / act arrays of the hash table index-major instead of slot-major (a build with -DRBF_TABLE_INDEX_MAJOR): correct results in both, so the STEP can be compared
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04idx; mkdir -p $O
NL="--no-cpu-baseline --no-clips --no-legs"
run() { tag="$1"; shift; "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-28s %.0f Mpixel/s, %.4f ms/step, alone %s verified %s' % ('$tag', d['value'], d['ms_per_step'], d['kernels_ms_per_step_alone'], d.get('verified_vs_oracle',{}).get('frames')))"; }
{
for rep in 1 2; do
run "slot-major (library)" python bench.py $NL
run "index-major" env RBF_LIB_PATH=$GRAFT_REPO_ROOT/build/ablate/librbf_idxmajor.so python bench.py $NL
done
run "slot-major, 1 pipeline" python bench.py $NL --streams 1
run "index-major, 1 pipeline" env RBF_LIB_PATH=$GRAFT_REPO_ROOT/build/ablate/librbf_idxmajor.so python bench.py $NL --streams 1
} | tee $O/idxmajor.txt
