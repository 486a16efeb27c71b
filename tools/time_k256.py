#!/usr/bin/env python3
"""K = 256 (SURVEY 8d's throughput mode): wall time per batch against the kernel's own HIP-event time, and the host's share.
   python tools/time_k256.py [K]   (needs a GPU)"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import bench
from signals_amd import runtime
from signals_amd.engine import BatchRenderer, KernelTimer
runtime.set_device('cuda:0')
V, N = 1024, 256
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = bench.synth_params(V)
for label, timer in (('no events', None), ('events', KernelTimer())):
    r = BatchRenderer(bench.build_graph(p, 0, V), 2, 48000, timer=timer)
    pos, i = 0, 0
    wrap = max(1, min(1024, int(2.0 ** 26 / 1760.0 * 48000) // (N * K) - 1))     # stay inside the closed form's phase range (bench.py does the same)
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        for _ in range(50):
            if i % wrap == 0: pos = 0
            r.render(pos, N, K); pos += N * K; i += 1
        torch.cuda.synchronize()
    if timer: timer.reset()
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        if i % wrap == 0: pos = 0
        r.render(pos, N, K); pos += N * K; i += 1
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    line = f'{label:10s} K={K}: {dt / reps * 1e6:6.1f} us per batch wall, host enqueue {t_host / reps * 1e6:6.1f} us, {V * N * K * reps / dt / 1e12:.2f} T/s'
    if timer:
        s = timer.summary()
        line += '  kernels: ' + ', '.join(f"{k} {e['ms'] / e['calls'] * 1e3:.1f} us" for k, e in s.items())
    print(line, flush=True)
