#!/bin/bash
# BASELINE config 4: 3840x2160 YUV444 synthetic residuals, LDS tile-size sweep with the HBM traffic of the query and insert
# kernels per tile size (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate --pmc passes with --kernel-trace only).  Runs on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tile_sweep; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
echo "3840x2160, 29 inter-frames per step (30-frame GOP); throughput with the default four pipelines, kernel times and traffic with one pipeline alone"
echo "(FETCH_SIZE doubled for the query kernel: 16-byte LDS-DMA reads are under-reported by half on gfx950, MI355X_MICROARCH.md)"
printf "%-9s %-10s %-9s %-10s %-10s %-12s %-12s %-12s \n" tile_KiB Gpixel/s ms/step insert_us query_us q_read_MB q_write_MB q_HBM_GB/s
for t in 16 32 64 96 128 0; do
  python $ROOT/bench.py --width 3840 --height 2160 --frames 30 --gops-per-pipeline 1 --steps 12 --no-cpu-baseline --no-verify --no-clips --no-legs --lds-tile-kib $t 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_$t.json
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${t}_$c
    rocprofv3 --output-format csv --pmc $c --kernel-trace -d $OUT/pmc_${t}_$c -o pmc -- python $ROOT/bench.py --width 3840 --height 2160 --frames 30 --gops-per-pipeline 1 --streams 1 --steps 6 --warmup 2 --exact-steps \
      --no-cpu-baseline --no-kernel-timing --no-verify --no-clips --no-legs --lds-tile-kib $t > $OUT/pmc_${t}_$c.log 2>&1
  done
  python - "$OUT" "$t" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, t = sys.argv[1], sys.argv[2]
d = json.load(open(os.path.join(out, "bench_%s.json" % t)))
k = d["kernels_ms_per_step_alone"]
val = defaultdict(lambda: defaultdict(list))      # counter -> kernel -> [KB per launch]
dur = defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "pmc_%s_%s" % (t, c), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = "query" if "k_query" in name else "insert" if "k_insert" in name else None
            if key:
                val[c][key].append(float(r["Counter_Value"]))
                if key == "query":
                    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
mean = lambda v: sum(v) / len(v) if v else 0.0
q_r = 2 * mean(val["FETCH_SIZE"]["query"]) / 1024
q_w = mean(val["WRITE_SIZE"]["query"]) / 1024
q_us = mean(dur["query"]) / 1e3
st = d.get("steady_state") or d
print("%-9s %-10.1f %-9.4f %-10.1f %-10.1f %-12.1f %-12.1f %-12.0f" % (t if t != "0" else "auto", st["value"] / 1e3, st["ms_per_step"], k["insert"] * 1e3, k["query"] * 1e3,
                                                                      q_r, q_w, (q_r + q_w) * 1.048576e6 / (q_us * 1e-6) / 1e9 if q_us else 0))
PY
done
