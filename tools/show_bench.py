#!/usr/bin/env python3
"""Print the essentials of bench.py JSON lines: tools/show_bench.py FILE..."""
import json, sys
for path in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        print("%-28s ERR %s" % (path.split("/")[-1], e)); continue
    k = d.get("kernels_ms_per_step_alone") or {}
    print("%-28s %9.0f Mpx/s  %.4f ms/step  req %s  alone: %s  ver %s" % (path.split("/")[-1], d["value"], d["ms_per_step"], (d.get("requested_region") or {}).get("ms_per_step"),
          " ".join("%s=%.1f" % (a, b * 1e3) for a, b in k.items()), (d.get("verified_vs_oracle") or {}).get("frames")))
