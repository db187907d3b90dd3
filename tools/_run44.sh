cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --steps 300 $* 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('kernels_ms_per_step_alone'), (d.get('verified_vs_oracle') or {}).get('frames'))"; }
echo "default:      $(run)"
echo "f64t 80 KiB:  $(run --lds-tile-kib 80)"
echo "default:      $(run)"
echo "f64t 80 KiB:  $(run --lds-tile-kib 80)"
