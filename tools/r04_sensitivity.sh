# round 4: what does each kernel cost the overlapped four-pipeline step?  (kernels not launched: results wrong, timing only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
q='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print("%-28s %8.0f Mpixel/s  %.4f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))'
for sk in "" insert reduce query stitch "reduce,stitch" "insert,reduce" "insert,reduce,stitch" "insert,reduce,query,stitch"; do
  python bench.py --no-cpu-baseline --no-verify --no-clips --no-legs --no-kernel-timing ${sk:+--skip-kernels $sk} 2>/dev/null | python -c "$q" "skip: ${sk:-nothing}"
done > $O/sensitivity.txt 2>&1
cat $O/sensitivity.txt
