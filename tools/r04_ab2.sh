cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
q='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print("%-44s %8.0f Mpixel/s  %.4f ms/step  requested-region %s" % (sys.argv[1], d["value"], d["ms_per_step"], (d.get("requested_region") or {}).get("ms_per_step")))'
for rep in 1 2 3; do
for a in "--streams 3" "--streams 4" "--streams 3 --side-compact" "--streams 4 --side-compact"; do
  python bench.py --no-cpu-baseline --no-clips --no-legs --no-verify $a 2>/dev/null | python -c "$q" "$a"
  python bench.py --no-cpu-baseline --no-clips --no-legs --no-verify --no-kernel-timing $a 2>/dev/null | python -c "$q" "$a --no-kernel-timing"
done; done > $O/ab2.txt 2>&1
sort $O/ab2.txt
