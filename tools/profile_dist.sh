#!/bin/bash
# rocprofv3 kernel stats of the N > 1 code path run on one GPU (process group of one rank): shows the pack kernel
# and RCCL's gather next to the path's own kernels.  Usage: tools/profile_dist.sh <tag>
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_dist" -o stats -- python $ROOT/bench.py --force-dist --no-cpu-baseline > "$OUT/stats_dist.log" 2>&1
grep -a metric "$OUT/stats_dist.log" | cut -c1-200
python - <<PY
import csv, glob
for f in glob.glob("$OUT/stats_dist/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("  %-70s calls %5s  avg %10.1f ns  %6s%%" % (row["Name"].split("(")[0][:70], row["Calls"], float(row["AverageNs"]), row["Percentage"]))
PY
