#!/usr/bin/env python3
"""Time sig_fused_cascade_bus on BASELINE config 3 (1024-voice Saw -> LP -> LP -> x ADSR -> mono bus, N = 1024) over launch
geometries (voices per lane, blocks per lane):  python tools/time_cascade.py [K] [vpt:span ...]     (needs a GPU)"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import bench_configs as cfg
from signals_amd import _native as nat

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
geoms = [tuple(int(x) for x in g.split(':')) for g in sys.argv[2:]] or [(0, 0)]
V, N, RATE = 1024, 1024, 48000
p = cfg.c3_params(V)
dev = torch.device('cuda:0')
up = lambda a: torch.from_numpy(a).to(dev)
hz, ph, c1, c2 = up(p['hertz']), up(p['phase']), up(p['cut1']), up(p['cut2'])
env = {k: up(v) for k, v in p['env'].items()}
out = torch.empty(K * N, 1, dtype=torch.float32, device=dev)
ws = torch.empty(nat.lib().sig_fused_voice_bus_workspace(V, K * N, 1) // 8, dtype=torch.float64, device=dev)
ref = None
for vpt, span in geoms[:1] * 3 + geoms:        # (the first rounds only bring the clocks up)
    nat.set_fused_cascade_tuning(vpt, span)
    call = lambda pos: nat.fused_cascade_bus('Sawtooth', 'lp', 'lp', RATE, pos, pos - N if pos else 0, N, K, 100, V, hz, ph, c1, c2,
                                             None, env, None, out, workspace=ws)
    call(K * N)
    call(K * N)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    err = (out - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5):
        call(K * N * (i + 1))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f'vpt {vpt} span {span} geometry {nat.fused_cascade_geometry(V, K)}: {us:8.1f} us  {V * N * K / us / 1e6:.3f} T voice-samples/s   '
          f'(max |diff| to the first geometry {err:.2e})')
nat.set_fused_cascade_tuning(0, 0)
