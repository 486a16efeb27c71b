"""latency-mode timing (not part of the product): the one-launch block kernel alone, the engine's per-block render with
and without hipGraph replay; run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch
import bench
from signals_amd import _native, runtime
runtime.set_device('cuda:0')
V, N = 1024, 256
p = bench.synth_params(V)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')
hz, ph, cut, g, pan = dev(p['hertz']), dev(p['phase']), dev(p['cutoff']), dev(p['gain']), dev(p['pan'])
out = torch.empty((N, 2), device='cuda')
ws = _native.latency_voice_bus_workspace(V, N, 2, 'cuda')
pos = torch.zeros(1, dtype=torch.int64, device='cuda')
for _ in range(20):
    _native.latency_voice_bus('lp', 48000, pos, N, 100, V, hz, ph, cut, g, pan, out, ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    _native.latency_voice_bus('lp', 48000, pos, N, 100, V, hz, ph, cut, g, pan, out, ws)
torch.cuda.synchronize()
print(f'direct C-ABI calls, back to back: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per block')
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(50):
        _native.latency_voice_bus('lp', 48000, pos, N, 100, V, hz, ph, cut, g, pan, out, ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    graph.replay()
torch.cuda.synchronize()
print(f'hipGraph of 50 blocks, replayed: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us per block (GPU-side cost of a block)')
from signals_amd.engine import BatchRenderer
for replay in (False, True):
    r = BatchRenderer(bench.build_graph(p, 0, V), 2, 48000, graph_replay=replay)
    for i in range(20):
        r.render(i * N, N, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20, 520):
        r.render(i * N, N, 1)
    torch.cuda.synchronize()
    print(f'engine render, graph_replay={replay}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per block')
