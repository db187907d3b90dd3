# does the 26-byte-per-index table pay where round 2's 32-byte one did not?  1440p (118 MB allocation) and 2160p (265 MB): gather vs hashing in k_insert_positions
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04big; mkdir -p $O
NL="--no-cpu-baseline --no-clips --no-legs"
run() { tag="$1"; shift; "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-44s %.0f Mpixel/s, %.4f ms/step, alone %s verified %s' % ('$tag', d['value'], d['ms_per_step'], d['kernels_ms_per_step_alone'], d.get('verified_vs_oracle',{}).get('frames')))"; }
{
for lib in new_bloom_filter_repo_amd/librbf_hip.so build/ablate/librbf_tab130.so build/ablate/librbf_tab300.so; do
run "1440p x30 $lib" env RBF_LIB_PATH=$GRAFT_REPO_ROOT/$lib python bench.py $NL --width 2560 --height 1440 --frames 30 --steps 60
run "2160p x9  $lib" env RBF_LIB_PATH=$GRAFT_REPO_ROOT/$lib python bench.py $NL --width 3840 --height 2160 --frames 9 --steps 40
done
} | tee $O/bigtable.txt
