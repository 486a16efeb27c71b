#!/usr/bin/env python3
"""Consecutive batches of the C2 graph alternated over 1 / 2 / 3 HIP streams (one renderer, and so one workspace, per stream): what
overlapping the tail of one launch with the head of the next buys.   python tools/time_two_streams.py [blocks per batch]   (needs a GPU)"""
import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import bench
from signals_amd import runtime
from signals_amd.engine import BatchRenderer
runtime.set_device('cuda:0')
V, N = 1024, 256
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = bench.synth_params(V)
def make():
    return BatchRenderer(bench.build_graph(p, 0, V), 2, 48000)
for nstreams in (1, 2, 3, 1, 2):
    rs = [make() for _ in range(nstreams)]
    ss = [torch.cuda.Stream() for _ in range(nstreams)]
    def run(steps, pos0=0):
        pos = pos0
        for i in range(steps):
            with torch.cuda.stream(ss[i % nstreams]):
                rs[i % nstreams].render(pos % (N * K * 1024), N, K)
            pos += N * K
        return pos
    t_end = time.perf_counter() + 1.5
    pos = 0
    while time.perf_counter() < t_end:
        pos = run(20, pos); torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 400 if K >= 1024 else 2000
    run(steps, pos)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'K={K} streams={nstreams}: {dt / steps * 1e6:.1f} us per batch, {V * N * K * steps / dt / 1e12:.2f} T/s', flush=True)
