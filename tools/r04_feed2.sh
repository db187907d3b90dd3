# feed / kernel-choice sweep after the round-4 kernels: begin-ahead, host threads, 5 pipelines, always-wide query kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04feed2; mkdir -p $O
NL="--no-cpu-baseline --no-clips --no-legs --no-verify"
run() { tag="$1"; shift; "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-34s %.0f Mpixel/s, %.4f ms/step, host %.4f, alone %s' % ('$tag', d['value'], d['ms_per_step'], d.get('host_ms_per_step',0), d['kernels_ms_per_step_alone']))"; }
{
run "default (ahead 0)" python bench.py $NL
run "ahead 1" python bench.py $NL --begin-ahead 1
run "ahead 2" python bench.py $NL --begin-ahead 2
run "ahead 3" python bench.py $NL --begin-ahead 3
run "host threads 4" python bench.py $NL --host-threads 4
run "streams 5" python bench.py $NL --streams 5
run "streams 5 ahead 2" python bench.py $NL --streams 5 --begin-ahead 2
run "always k_query_u64w" env RBF_LIB_PATH=$GRAFT_REPO_ROOT/build/ablate/librbf_wide.so python bench.py $NL
run "always k_query_u64w, 1 stream" env RBF_LIB_PATH=$GRAFT_REPO_ROOT/build/ablate/librbf_wide.so python bench.py $NL --streams 1
run "default again" python bench.py $NL
} | tee $O/feed2.txt
