#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command only (the pass whose average query duration
# must agree with bench.py's roofline.avg_launch_ms).  Usage: tools/profile_default.sh <tag>
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_default" -o stats -- python $ROOT/bench.py --no-cpu-baseline > "$OUT/stats_default.log" 2>&1
tail -1 "$OUT/stats_default.log"
python $ROOT/tools/summarize_profile.py "$OUT" | sed -n '/stats_default/,/== counters/p' | head -14
