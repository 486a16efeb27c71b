import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from helpers import RATE, f32, fix, maxerr, mkosc
from oracle import chain_ref as R
from signals_amd.chain import ext, fx
from signals_amd.engine import BatchRenderer, KernelTimer
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    V = int(rng.choice([1, 8, 64, 70, 130, 256, 512, 520, 1024, 1100]))
    N = int(rng.choice([16, 64, 100, 128, 256, 300, 1024]))
    kind = str(rng.choice(['Sine', 'Sine', 'Sawtooth', 'Square', 'Triangle']))
    btype = str(rng.choice(['lp', 'hp']))
    use_gain, pre_gain = bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0)
    bus = int(rng.choice([1, 2, 2, 4]))
    start = int(rng.choice([0, 37, 99, 4096, 48000 * 3600]))
    batches = [int(x) for x in rng.choice([1, 1, 2, 7, 33, 64], size=3)]
    hz = np.exp(rng.uniform(np.log(20), np.log(9000), (1, V))) * rng.choice([-1.0, 1.0], (1, V)); ph = rng.uniform(-1, 1, (1, V))
    cut, gain, pg = np.exp(rng.uniform(np.log(30), np.log(20000), (1, V))), rng.uniform(0.1, 1.0, (1, V)), rng.uniform(0.5, 2.0, (1, V))
    pan = rng.uniform(-1, 1, (bus, V))
    def build():
        o = mkosc(kind, hz, ph); src = o
        if pre_gain:
            g0 = fx.Gain(); g0.left = o; g0.right = fix(pg); src = g0
        f = getattr(fx, 'LowPass' if btype == 'lp' else 'HighPass')(); f.input = src; f.cutoff = fix(cut)
        top = f
        if use_gain:
            g = fx.Gain(); g.left = f; g.right = fix(gain); top = g
        b = ext.SumBus(); b.input = top
        if bus > 1: b.get_state().gains = np.ascontiguousarray(pan)
        return b
    node = R.Osc(kind, R.Fixed(hz), R.Fixed(ph))
    if pre_gain: node = R.Binary('Gain', node, R.Fixed(pg))
    node = R.Filter(btype, node, R.Fixed(cut))
    if use_gain: node = R.Binary('Gain', node, R.Fixed(gain))
    timer = KernelTimer()
    r = BatchRenderer(build(), bus, RATE, timer=timer)
    pos, parts = start, []
    for k in batches:
        parts.append(r.render(pos, N, k).cpu().numpy()); pos += N * k
    got = np.concatenate(parts)
    ref = R.sum_bus(R.render_stream(node, start, N, sum(batches), V), pan if bus > 1 else None)
    scale = max(1.0, float(np.abs(ref).max()))
    err = maxerr(got, f32(ref)) if np.isfinite(got).all() else float('inf')
    ok = err < 2e-6 * scale
    bad += not ok
    print('OK ' if ok else 'BAD', case, kind, btype, 'V', V, 'N', N, 'start', start, 'batches', batches, 'gain', use_gain, 'pre', pre_gain, 'bus', bus, 'err %.2e' % err, 'scale %.2f' % scale, sorted(set(n.split('[')[0] for n in timer.summary())))
print('bad', bad)
