cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04m}; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
q='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print("%-44s %8.0f Mpixel/s  %.4f ms/step  alone %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("kernels_ms_per_step_alone")))'
for a in "--streams 3" "--streams 4" "--streams 3 --side-compact" "--streams 4 --side-compact" "--streams 3 --force-bits 32768" "--streams 3"; do
  python bench.py --no-cpu-baseline --no-clips --no-legs --no-verify $a 2>/dev/null | python -c "$q" "$a"
done > $O/ab.txt 2>&1
cat $O/ab.txt
