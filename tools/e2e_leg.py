#!/usr/bin/env python3
"""bench.py's e2e_surface leg on its own (GPU box): python tools/e2e_leg.py"""
import sys, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch; torch.cuda.init()
import importlib.util
spec = importlib.util.spec_from_file_location("b", __import__('os').path.join(sys.path[0], 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from new_bloom_filter_repo_amd import _native as nat
from new_bloom_filter_repo_amd.synthetic import P_KSTAR_2_3
out = b.e2e_surface_leg(nat, 0, 1920, 1080, P_KSTAR_2_3)
out.pop("what")
print(json.dumps(out, indent=1))
