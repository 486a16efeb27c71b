#!/usr/bin/env python3
"""Marginal cost of the voice program's instructions: small hand-assembled programs, 1024 voices x 256 frames x 1024 blocks
(268 M voice-samples per launch), stereo bus sink.   python tools/time_vp_ops.py   (needs a GPU)"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from signals_amd import _native, runtime
runtime.set_device('cuda:0')
import os
if os.environ.get('VP_VPT'):
    _native.set_voice_program_tuning(int(os.environ['VP_VPT']), int(os.environ.get('VP_SPAN', '0')))
V, N, K = 1024, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')
hz, ph, cut, g = dev(rng.uniform(55, 1760, (1, V))), dev(rng.uniform(0, 1, (1, V))), dev(rng.uniform(200, 8000, (1, V))), dev(rng.uniform(0, 1, (1, V)))
th = rng.uniform(0, np.pi / 2, V); pan = dev(np.stack([np.cos(th), np.sin(th)]))
out = torch.empty((N * K, 2), device='cuda')
full = torch.empty((N * K, V), device='cuda')
ws = torch.empty(_native.lib().sig_fused_voice_bus_workspace(V, N * K, 2) // 8, dtype=torch.float64, device='cuda')
O = lambda k=2: ('Osc', k, 0, 0, 0)
F = lambda s=0: ('Filter', 0, s, 0, 0)
G = ('Gain', 0, 0, 0, 0)
progs = {
    'Osc(saw)': ([O()], 0), 'Osc(sine)': ([O(0)], 0), 'Osc,Gain': ([O(), G], 0), 'Osc,Gain x4': ([O(), G, G, G, G], 0),
    'Osc,Save,Load': ([O(), ('Save', 0, 0, 0, 0), ('Load', 0, 0, 0, 0)], 0),
    'Osc,Filter': ([O(), F()], 1), 'Osc,Filter,Gain': ([O(), F(), G], 1), 'Osc,Filter,Filter,Gain': ([O(), F(0), F(1), G], 2),
}
for name, (code, depth) in progs.items():
    filters = [(cut, 'lp', i + 1) for i in range(depth)]
    nt = 1 if any(op in ('Save', 'Load') for op, *_ in code) else 0
    hist = [] if depth < 2 else []
    for sink in ('bus', 'store'):
        def run():
            if sink == 'bus':
                _native.voice_program(code, [(hz, ph)], [g], filters, nt, depth, 48000, 0, N, K, 100, V, 1 + K, hist, out, bus_gains=pan, bus=True, workspace=ws)
            else:
                _native.voice_program(code, [(hz, ph)], [g], filters, nt, depth, 48000, 0, N, K, 100, V, 1 + K, hist, full)
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f'{name:28s} {sink:5s} {dt * 1e6:8.1f} us   {V * N * K / dt / 1e12:.3f} T/s', flush=True)
