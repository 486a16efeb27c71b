#!/usr/bin/env python3
"""sig_fused_voice_bus (C2, Sine closed form) at small batch lengths under forced launch geometries: HIP-event time per launch,
back to back (no host in the loop: 200 launches per measurement).   python tools/time_fused_geom.py   (needs a GPU)"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from signals_amd import _native, runtime
runtime.set_device('cuda:0')
V, N = 1024, 256
p = bench.synth_params(V)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')
order = np.argsort(p['cutoff'][0], kind='stable')
hz, ph, cut, g, pan = dev(p['hertz']), dev(p['phase']), dev(p['cutoff']), dev(p['gain']), dev(p['pan'])
def run(K, vpt, span, reps=300):
    _native.set_fused_tuning(vpt, span, 1, 0)
    out = torch.empty((N * K, 2), device='cuda')
    ws = torch.empty(_native.lib().sig_fused_voice_bus_workspace(V, N * K, 2) // 8, dtype=torch.float64, device='cuda')
    consts = torch.empty(_native.lib().sig_fused_voice_consts_size(V) // 8, dtype=torch.float64, device='cuda')
    call = _native.FusedVoiceBusCall('Sine', 'lp', 48000, N, K, 100, V, hz, ph, cut, g, pan, 2, ws, None, consts)
    call(N * K, out, False)
    for _ in range(50): call(N * K, out, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): call(N * K * (1 + i % 8), out, True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for K in (64, 128, 256, 512, 1024):
    row = []
    for vpt, span in ((0, 0), (8, 1), (4, 1), (2, 1), (1, 1), (4, 2), (2, 2), (8, 2)):
        try:
            us = run(K, vpt, span)
            row.append(f'({vpt},{span}) {us:6.1f}us {V * N * K / us / 1e6:5.2f}T')
        except Exception as e:
            row.append(f'({vpt},{span}) err')
    print(f'K={K:5d}: ' + '  '.join(row), flush=True)
