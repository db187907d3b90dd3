# round 4: kernel trace of the default four-pipeline timed region -> overlap report + per-kernel durations under overlap
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04e}; mkdir -p $O
ARGS="${@:2}"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktrace && rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-verify --no-kernel-timing --no-clips --no-legs --steps 400 --exact-steps $ARGS > /tmp/ktrace.log 2>&1 )
tail -1 /tmp/ktrace.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('traced run: %.0f Mpixel/s, %.4f ms/step' % (d['value'], d['ms_per_step']))" > $O/overlap.txt
python tools/overlap_report.py $(find /tmp/ktrace -name "*kernel_trace.csv" | head -1) >> $O/overlap.txt 2>&1
cat $O/overlap.txt
