# round 4: does it pay that other kernels' waves fit next to k_query_u64?  The same library with the kernel's VGPR allocation padded to 120 / 128
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
q='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print("%-34s %8.0f Mpixel/s  %.4f ms/step  alone %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["kernels_ms_per_step_alone"]))'
for rep in 1 2; do
for v in "" build/variants/librbf_u64_pad119.so build/variants/librbf_u64_pad127.so; do
  RBF_LIB_PATH=${v:+$GRAFT_REPO_ROOT/$v} python bench.py --no-cpu-baseline --no-verify --no-clips --no-legs 2>/dev/null | python -c "$q" "${v:-default (112 VGPRs allocated)}"
done; done > $O/coresidency.txt 2>&1
cat $O/coresidency.txt
