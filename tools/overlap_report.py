#!/usr/bin/env python3
"""overlap_report.py <kernel_trace.csv> -- how much do the kernels of the bench's pipelines overlap on the GPU?

Reads a rocprofv3 --kernel-trace CSV (Kernel_Name, Start_Timestamp, End_Timestamp, Queue_Id/Stream_Id ...), keeps the
timed region (the densest half of the query launches) and prints: the sum of the kernel durations, the
length of the union of their intervals (= GPU busy time), the idle time, and, per kernel, how much of its running time
it shared with at least one other kernel and with which."""
import csv
import sys
from collections import defaultdict


def short(name):
    for key in ("k_query_u64w", "k_query_u64", "k_query_s64t", "k_query_s64", "k_query_r64t", "k_query_r64", "k_query_f64t", "k_query_f64", "k_insert_tab", "k_insert_positions", "k_insert_records", "k_residual_mask_gop", "k_compact_witness",
                "k_filter_reduce", "k_finish_ones", "k_hash_table", "k_pack"):
        if key in name:
            return key
    return name.split("(")[0][:40]


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # the timed region = the densest run of query launches (the bench also launches kernels, sparsely, before and after it: warm-up,
    # PCIe legs): the window of half of all query launches, consecutive, that spans the least time
    q = [r for r in rows if r[2].startswith("k_query_")] or rows
    k = max(1, len(q) // 2)
    best = min(range(len(q) - k + 1), key=lambda i: q[i + k - 1][1] - q[i][0])
    lo, hi = q[best][0], q[best + k - 1][1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    total = sum(e - s for s, e, _ in rows)
    # sweep line
    ev = []
    for i, (s, e, _) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    last = ev[0][0]
    busy = 0
    depth_time = defaultdict(int)
    shared = defaultdict(int)                      # kernel -> ns with company
    alone = defaultdict(int)
    pair = defaultdict(int)
    for t, kind, i in ev:
        dt = t - last
        if dt and active:
            busy += dt
            depth_time[len(active)] += dt
            names = sorted(rows[j][2] for j in active)
            for j in active:
                (shared if len(active) > 1 else alone)[rows[j][2]] += dt
            if len(active) > 1:
                pair[" + ".join(names)] += dt
        last = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    span = rows[-1][1] - rows[0][0]
    print("kernels %d, span %.3f ms, sum of durations %.3f ms, busy (union) %.3f ms, idle %.3f ms" % (len(rows), span / 1e6, total / 1e6, busy / 1e6, max(0, span - busy) / 1e6))
    print("time with N kernels resident: " + ", ".join("%d: %.1f %%" % (d, 100.0 * v / span) for d, v in sorted(depth_time.items())))
    print("%-22s %10s %12s %12s" % ("kernel", "launches", "avg us", "shared %"))
    cnt = defaultdict(int)
    dur = defaultdict(int)
    for s, e, nme in rows:
        cnt[nme] += 1
        dur[nme] += e - s
    for nme in sorted(dur, key=lambda k: -dur[k]):
        print("%-22s %10d %12.1f %12.1f" % (nme, cnt[nme], dur[nme] / cnt[nme] / 1e3, 100.0 * shared[nme] / max(1, shared[nme] + alone[nme])))
    print("most common company (share of the span):")
    for k, v in sorted(pair.items(), key=lambda kv: -kv[1])[:12]:
        print("  %5.1f %%  %s" % (100.0 * v / span, k))


if __name__ == "__main__":
    main(sys.argv[1])
