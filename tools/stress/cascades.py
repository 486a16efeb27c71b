import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from helpers import RATE, f32, maxerr
import test_gpu_fused_cascade as T
from oracle import chain_ref as R
from signals_amd.engine import BatchRenderer, KernelTimer
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    V = int(rng.choice([8, 64, 70, 200, 520]))
    N = int(rng.choice([112, 128, 256, 400, 1024]))
    kind = str(rng.choice(['Sine', 'Sawtooth', 'Square', 'Triangle']))
    t1, t2 = (str(rng.choice(['LowPass', 'HighPass'])) for _ in range(2))
    env, gain, stereo = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    start = int(rng.choice([0, 37, 100, 4096]))
    batches = [int(x) for x in rng.choice([1, 2, 3, 5], size=3)]
    p = T.params(V, int(rng.integers(0, 1000)))
    if rng.integers(0, 2): p['cut1'][0, : V // 4] = rng.uniform(20, 150, V // 4)
    pan = p['pan'] if stereo else None
    short = {'LowPass': 'lp', 'HighPass': 'hp'}
    timer = KernelTimer()
    r = BatchRenderer(T.graph(p, kind, t1, t2, env=env, gain=gain, pan=pan), 2 if stereo else 1, RATE, timer=timer)
    pos, parts = start, []
    for k in batches:
        parts.append(r.render(pos, N, k).cpu().numpy()); pos += N * k
    got = np.concatenate(parts)
    node, _ = T.oracle(p, kind, short[t1], short[t2], env=env, gain=gain, pan=pan)
    ref = R.sum_bus(R.render_stream(node, start, N, sum(batches), V), pan)
    scale = max(1.0, float(np.abs(ref).max()))
    err = maxerr(got, f32(ref))
    ok = err < 2e-6 * scale
    bad += not ok
    print('OK ' if ok else 'BAD', case, kind, t1, t2, 'V', V, 'N', N, 'start', start, 'batches', batches, 'env', env, 'gain', gain, 'stereo', stereo, 'err %.2e' % err, 'scale %.2f' % scale, sorted(n.split('[')[0] for n in timer.summary()))
print('bad', bad)
