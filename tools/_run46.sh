cd $GRAFT_REPO_ROOT
BIG=1 timeout 120 ./build/bench_insert 2>&1 | grep HASHED
run() { timeout 300 python bench.py --no-cpu-baseline --steps 100 $* 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('kernels_ms_per_step_alone'), (d.get('verified_vs_oracle') or {}).get('frames'))"; }
echo "2160p: $(run --width 3840 --height 2160 --frames 9)"
echo "2160p gather: $(run --width 3840 --height 2160 --frames 9 --force-bits 0)"
echo "1440p hashed: $(run --width 2560 --height 1440 --force-bits 16384)"
echo "1440p: $(run --width 2560 --height 1440)"
echo "2880p: $(run --width 5120 --height 2880 --frames 9)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -x -q -m gpu 2>&1 | tail -3
