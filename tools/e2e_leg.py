#!/usr/bin/env python3
"""bench.py's e2e_surface leg on its own (GPU box): python tools/e2e_leg.py [block_frames [gpu_lanes [profile_stages]]]"""
import importlib.util
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.synthetic import P_KSTAR_2_3
bf = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) else None
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
out = b.e2e_surface_leg(nat, 0, 1920, 1080, P_KSTAR_2_3, block_frames=bf, gpu_lanes=lanes, profile_stages=len(sys.argv) > 3 and sys.argv[3] == "1")
out.pop("what")
if os.environ.get("E2E_BRIEF"):
    for k, v in out.items():
        print("block_frames %s lanes %d %s: compress %.3f s (gpu_busy_frac %.2f) decompress %.3f s | %s | %s" % (bf, lanes, k, v["compress_s"], v["gpu_busy_frac"], v["decompress_s"], v["stages_s"], v["decompress_stages_s"]))
else:
    print(json.dumps(out, indent=1))
