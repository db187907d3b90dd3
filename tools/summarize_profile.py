#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile.sh: per-kernel time stats and PMC counters per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    for tok in ("void rbf::", "rbf::"):
        name = name.replace(tok, "")
    return name[:60]


for f in sorted(glob.glob(os.path.join(root, "stats*", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, root),
          "(default command: 4 GOP pipelines, kernels overlap)" if "stats_default" in f else
          "(--streams 1 --force-bits 32768: alone, the query kernel does not rewrite the shared hash table -- bench.py's `alone` condition)" if "stats_norewrite" in f else
          "(--streams 1: kernels run alone; the sole holder's query kernel also rewrites the 66 MB hash table)")
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %5s  avg %10.1f ns  total %12s ns  %5s%%" % (
            short(row.get("Name", "")), row.get("Calls"), float(row.get("AverageNs", 0)), row.get("TotalDurationNs"), row.get("Percentage")))

for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k].add(row.get("Dispatch_Id"))
        print("== counters (mean per launch):", os.path.relpath(f, root))
        for k in acc:
            n = max(1, len(cnt[k]))
            print("  %-60s launches %d" % (k, n))
            for c, v in sorted(acc[k].items()):
                print("      %-24s %16.1f" % (c, v / n))
