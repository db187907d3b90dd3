#!/usr/bin/env python3
"""Summarise tools/r06_profile.sh: per-kernel time stats (rocprofv3 --stats), HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE reports half the bytes of
16-byte-per-lane reads: doubled; both counters are in KiB), and the instruction counters per launch of every shape that has a
pmc_insts_<shape> pass.  Writes <dir>/<round>_traffic.json and <dir>/<round>_counters.json (copied to profiles/; the second one feeds
tools/make_issue_model.py) and prints the summary (profiles/<round>_rocprofv3_summary.txt).
Usage: python tools/make_traffic.py gpurun_out/r06_profile [r06]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r06"


def short(name):
    name = name.split("(")[0]
    for tok in ("void rbf::", "rbf::"):
        name = name.replace(tok, "")
    return name[:64]


def stats(tag):
    rows = {}
    for f in glob.glob(os.path.join(root, "stats_" + tag, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r.get("Name", ""))] = (int(r.get("Calls", 0)), float(r.get("AverageNs", 0)), float(r.get("Percentage", 0)))
    return rows


def counter(tag, name):
    acc, cnt = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = short(r.get("Kernel_Name", ""))
            acc[k] += float(r["Counter_Value"])
            cnt[k].add(r.get("Dispatch_Id"))
    return {k: (acc[k] / max(1, len(cnt[k])), len(cnt[k])) for k in acc}


SHAPES = {"g1": ("1920x1080x30_8bit_gpc1", "1920x1080, 30-frame GOP, one GOP per call (29 inter-frames per launch)"),
          "g4": ("1920x1080x30_8bit_gpc4", "1920x1080, 4 x 30 frames per call (116 inter-frames per launch)"),
          "c4": ("3840x2160x30_8bit_gpc1", "3840x2160, 30-frame GOP (29 inter-frames per launch)"),
          "u1": ("1920x1080x30_16bit_gpc1", "1920x1080 16-bit samples, 30-frame GOP, one GOP per call (BASELINE config 5's single-GPU half)"),
          "u4": ("1920x1080x30_16bit_gpc4", "1920x1080 16-bit samples, 4 x 30 frames per call")}
print("# rocprofv3 (%s, tools/%s_profile.sh): kernel time stats and HBM traffic per launch; one MI355X" % (ROUND, ROUND))
st = stats("default")
print("\n== default command, four pipelines (python bench.py --steps 20 --warmup 5, without legs / clips / CPU baseline): kernels overlap")
for k, (calls, avg, pct) in sorted(st.items(), key=lambda kv: -kv[1][2]):
    print("  %-64s calls %6d  avg %10.1f ns  %5.1f%%" % (k, calls, avg, pct))
traffic = {}
for tag, (key, what) in SHAPES.items():
    st = stats(tag)
    if not st:
        continue
    fe, wr = counter("pmc_fetch_" + tag, "FETCH_SIZE"), counter("pmc_write_" + tag, "WRITE_SIZE")
    print("\n== %s: one pipeline alone (--streams 1 --force-bits 32768), %s" % (tag, what))
    for k, (calls, avg, pct) in sorted(st.items(), key=lambda kv: -kv[1][2]):
        f, w = fe.get(k, (0.0, 0)), wr.get(k, (0.0, 0))
        hbm = (2 * f[0] + w[0]) * 1024
        print("  %-64s calls %5d  avg %9.1f us  %5.1f%%   FETCH_SIZE %10.1f KiB (x2)  WRITE_SIZE %10.1f KiB  -> %8.2f MB/launch, %7.1f GB/s"
              % (k, calls, avg / 1e3, pct, f[0], w[0], hbm / 1e6, hbm / avg if avg else 0.0))
        if k.startswith("k_query_") and calls >= 5 and f[1]:
            traffic[key] = {"kernel": k, "hbm_bytes_per_launch": int(hbm), "fetch_size_kb": round(f[0], 1), "write_size_kb": round(w[0], 1), "avg_launch_us": round(avg / 1e3, 2),
                            "launches": calls, "source": "tools/%s_profile.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace only, one pipeline alone; reads doubled (16-byte-per-lane loads, MI355X_MICROARCH.md)" % ROUND}
        if k.startswith("k_residual_mask_gop") and calls >= 5 and f[1]:
            traffic[key + "_mask"] = {"kernel": k, "hbm_bytes_per_launch": int(hbm), "fetch_size_kb": round(f[0], 1), "write_size_kb": round(w[0], 1), "avg_launch_us": round(avg / 1e3, 2),
                                      "launches": calls, "hbm_gbps": round(hbm / avg, 1) if avg else None, "source": "tools/%s_profile.sh (as above)" % ROUND}
INSTS = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE")
counters = {}
for tag, (key, what) in SHAPES.items():
    ins = {}
    for sub in ("pmc_insts_" + tag, "pmc_insts2_" + tag):
        for name in INSTS:
            for k, (v, nl) in counter(sub, name).items():
                ins.setdefault(k, {})[name] = v
    if not ins:
        continue
    st = stats(tag)
    print("\n== %s: instruction counters per launch (one pipeline alone)" % tag)
    for k, d in ins.items():
        print("  %-64s %s" % (k, "  ".join("%s %.4g" % (a, b) for a, b in sorted(d.items()))))
        if k.startswith("k_"):
            d = dict(d)
            if k in st:
                d["avg_launch_us"], d["launches_timed"] = round(st[k][1] / 1e3, 2), st[k][0]
            counters.setdefault(key, {})[k] = d
with open(os.path.join(root, "%s_traffic.json" % ROUND), "w") as f:
    json.dump(traffic, f, indent=1)
with open(os.path.join(root, "%s_counters.json" % ROUND), "w") as f:
    json.dump({"source": "tools/%s_profile.sh: rocprofv3 --pmc <instruction counters> --kernel-trace, one pipeline alone; values are per launch (mean over the dispatches of the pass); "
                         "avg_launch_us from the --stats pass of the same shape" % ROUND, "shapes": counters}, f, indent=1)
print("\nwrote %s_traffic.json: %s" % (ROUND, sorted(traffic)))
print("wrote %s_counters.json: %s" % (ROUND, sorted(counters)))
