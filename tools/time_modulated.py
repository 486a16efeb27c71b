#!/usr/bin/env python3
"""bench.py's C2_modulated leg on its own.   python tools/time_modulated.py [K]   (needs a GPU)"""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from signals_amd import runtime
runtime.set_device('cuda:0')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = bench.run_modulated(K=K, steps=20)
for k, v in out['voices'].items():
    print(k, v['modulation'], f"{v['value'] / 1e6:.2f} T", f"err {v['max_abs_error_blocks_0_1']:.2e} of {v['full_scale']:.3f}", v['launches_per_step'])
