"""Host-side cost of one latency-mode block through the engine, measured WITHOUT a GPU: the C ABI calls are replaced
by no-ops (this is a profiling harness for the Python path, not a compute fallback; nothing is rendered)."""
import cProfile
import pstats
import sys
import time
import pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import bench
from signals_amd import _native, runtime

runtime.set_device('cpu')
_native._gpu = lambda *a: None
_native._stream = lambda t: 0


class FakeLib:
    def __getattr__(self, name):
        if 'workspace' in name or 'size' in name:
            return lambda *a: 1 << 20
        return lambda *a: 0


_native._lib = FakeLib()
from signals_amd.engine import BatchRenderer
V, N = 1024, 256
p = bench.synth_params(V)
r = BatchRenderer(bench.build_graph(p, 0, V), 2, 48000)
for i in range(50):
    r.render(i * N, N, 1)
t0 = time.perf_counter()
n = 20000
for i in range(50, 50 + n):
    r.render(i * N, N, 1)
print(f'host path: {(time.perf_counter() - t0) / n * 1e6:.2f} us per block')
if len(sys.argv) > 1:
    pr = cProfile.Profile()
    pr.enable()
    for i in range(50, 5050):
        r.render(i * N, N, 1)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
