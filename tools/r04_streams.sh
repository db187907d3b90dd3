cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
q='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print("%-44s %8.0f Mpixel/s  %.4f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))'
for a in "--streams 1" "--streams 2" "--streams 3" "--streams 1 --side-compact" "--streams 2 --side-compact" "--streams 3 --side-compact" "--streams 2 --begin-ahead 1" "--streams 3 --begin-ahead 1" "--streams 3 --begin-ahead 2" "--streams 2 --host-threads 1" "--streams 3 --host-threads 1" "--streams 3 --gops-per-pipeline 4" "--streams 2 --gops-per-pipeline 6"; do
  python bench.py --no-cpu-baseline --no-clips --no-legs --no-kernel-timing --no-verify $a 2>/dev/null | python -c "$q" "$a"
done > $O/streams.txt 2>&1
cat $O/streams.txt
