"""The path's only exchange step on real RCCL: `ShardedRenderer` reduces the stereo bus with
torch.distributed's "nccl" backend, which is RCCL on ROCm.  One GPU box = one rank, so the collective runs at world
size 1 (SIG_FORCE_DIST=1 makes `reduce_bus` issue it anyway): that exercises communicator creation, the async reduce /
all-reduce on RCCL's stream, `work.wait()` ordering against the render stream, and that the bus comes back bit-equal.
The multi-rank arithmetic is covered by tests/test_parallel_gloo.py (world size 2, CPU)."""
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent

SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r})
os.environ['SIG_FORCE_DIST'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ['MASTER_PORT'] = '29577'
import torch
import torch.distributed as dist
import bench
from signals_amd import parallel, runtime
from signals_amd.engine import BatchRenderer
runtime.set_device('cuda:0')
rank, world = parallel.init_process_group()
assert dist.is_initialized() and dist.get_backend() == 'nccl' and (rank, world) == (0, 1)
V, N, K = 1024, 256, 64
p = bench.synth_params(V)
local = BatchRenderer(bench.build_graph(p, 0, V), 2, 48000).render(0, N, K).clone()
r = parallel.ShardedRenderer(lambda lo, hi: bench.build_graph(p, lo, hi), V, bus_channels=2)
assert (r.lo, r.hi, r.world) == (0, V, 1)
bus, work = r.render_async(0, N, K, dst=0)              # dist.reduce(..., async_op=True) on RCCL's stream
assert work is not None, 'the RCCL reduce was not issued'
nxt, work2 = r.render_async(N * K, N, K, dst=0)         # the next batch renders while the first reduce is in flight
work.wait(); work2.wait()
torch.cuda.synchronize()
assert torch.equal(bus, local), float((bus - local).abs().max())
again = r.render(0, N, K)                               # dst=None: all-reduce, synchronous
assert torch.equal(again, local)
assert not torch.equal(nxt, local) and bool(torch.isfinite(nxt).all())
dist.destroy_process_group()
print('RCCL_OK', 'nccl', world)
'''


def test_sharded_renderer_reduces_the_bus_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, '-c', SCRIPT.format(root=str(ROOT))], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'RCCL_OK nccl 1' in out.stdout
