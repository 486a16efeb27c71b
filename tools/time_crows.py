#!/usr/bin/env python3
"""The closed form with per-block cutoff rows (fused_steady_bus_kernel<.., CROWS>) against the block-invariant one on the SAME
voices (the K cutoff rows are copies of the one row): what deriving the filter constants per (block, voice) inside the launch costs.
   python tools/time_crows.py   (needs a GPU)"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from signals_amd import _native, runtime
runtime.set_device('cuda:0')
V, N = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = bench.synth_params(V)
order = np.argsort(p['cutoff'][0], kind='stable')
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')
def ordered(vpl):
    tile, tiles = 64 * vpl, V // (64 * vpl)
    q = np.arange(V); group, lane = q // 64, q % 64
    perm = np.empty(V, dtype=np.int64); perm[((group % tiles) * 64 + lane) * vpl + group // tiles] = order[q]
    return perm
for K in ((1024, 4096) if N == 256 else (256, 1024)):
    perm = ordered(8)
    hz, ph, cut, g, pan = (dev(p[k][:, perm]) for k in ('hertz', 'phase', 'cutoff', 'gain', 'pan'))
    cutK = cut.expand(K, V).contiguous()
    out = torch.empty((N * K, 2), device='cuda')
    ws = torch.empty(_native.lib().sig_fused_voice_bus_workspace(V, N * K, 2) // 8, dtype=torch.float64, device='cuda')
    def timeit(fn, reps=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    plain = timeit(lambda: _native.fused_voice_bus('Sine', 'lp', 48000, N * K, N, K, 100, V, hz, ph, cut, g, pan, out, workspace=ws))
    rows = timeit(lambda: _native.fused_rows('Sine', 'lp', 48000, N * K, N, K, 100, V, hz, ph, cutK, g, out, bus_gains=pan, bus=True, workspace=ws))
    a1 = out.clone()
    _native.fused_voice_bus('Sine', 'lp', 48000, N * K, N, K, 100, V, hz, ph, cut, g, pan, out, workspace=ws)
    print(f'K={K}: block-invariant {plain:7.1f} us   per-block cutoff rows {rows:7.1f} us   (+{rows - plain:6.1f} us, {(rows - plain) / (K * 2 / 1024):5.2f} us per block set-up and wave)   max |diff| {float((a1 - out).abs().max()):.2e}', flush=True)
