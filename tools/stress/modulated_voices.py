import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from helpers import RATE, f32, fix, maxerr, mkosc
from oracle import chain_ref as R
from signals_amd.chain import ext, fx
from signals_amd.engine import BatchRenderer, KernelTimer
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
KINDS = ['Sine', 'Sawtooth', 'Square', 'Triangle']
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    V = int(rng.choice([8, 64, 70, 130, 256, 520]))
    N = int(rng.choice([64, 128, 256, 512]))
    kind = str(rng.choice(KINDS))
    btype = str(rng.choice(['lp', 'hp']))
    mods = {k: bool(rng.integers(0, 2)) for k in ('hertz', 'phase', 'cutoff', 'gain')}
    use_gain = mods['gain'] or bool(rng.integers(0, 2))
    bus = int(rng.choice([0, 1, 2]))
    start = int(rng.choice([0, 37, 4096, 48000 * 60]))
    batches = [int(x) for x in rng.choice([1, 2, 3, 5, 9], size=3)]
    hz, ph = rng.uniform(40, 3000, (1, V)), rng.uniform(0, 1, (1, V))
    cut, gain = rng.uniform(100, 9000, (1, V)), rng.uniform(0.1, 1.0, (1, V))
    th = rng.uniform(0, np.pi / 2, V); pan = np.stack([np.cos(th), np.sin(th)])
    lf = {k: (float(rng.uniform(0.3, 9.0)), str(rng.choice(KINDS)), float(rng.uniform(0.05, 0.4))) for k in mods}
    def lfo_e(k, centre):      # engine graph: centre * (1 + depth * osc)  as RingMod(Mix(osc, 1, depth'), centre)
        f_, kd, depth = lf[k]
        m = fx.Mix(); m.left = mkosc(kd, [[f_]]); m.right = fix([[1.0]]); m.mix = fix([[depth]])
        r_ = fx.RingMod(); r_.left = m; r_.right = fix(centre); return r_
    def lfo_o(k, centre):
        f_, kd, depth = lf[k]
        return R.Binary('RingMod', R.Binary('Mix', R.Osc(kd, R.Fixed([[f_]])), R.Fixed([[1.0]]), R.Fixed([[depth]])), R.Fixed(centre))
    def build():
        o = mkosc(kind, hz, ph)
        if mods['hertz']: o.hertz = lfo_e('hertz', hz)
        if mods['phase']: o.phase = lfo_e('phase', ph)
        f = getattr(fx, 'LowPass' if btype == 'lp' else 'HighPass')(); f.input = o
        f.cutoff = lfo_e('cutoff', cut) if mods['cutoff'] else fix(cut)
        top = f
        if use_gain:
            g = fx.Gain(); g.left = f; g.right = lfo_e('gain', gain) if mods['gain'] else fix(gain); top = g
        if bus:
            b = ext.SumBus(); b.input = top
            if bus == 2: b.get_state().gains = np.ascontiguousarray(pan)
            top = b
        return top
    def oracle():
        o = R.Osc(kind, lfo_o('hertz', hz) if mods['hertz'] else R.Fixed(hz), lfo_o('phase', ph) if mods['phase'] else R.Fixed(ph))
        node = R.Filter(btype, o, lfo_o('cutoff', cut) if mods['cutoff'] else R.Fixed(cut))
        if use_gain: node = R.Binary('Gain', node, lfo_o('gain', gain) if mods['gain'] else R.Fixed(gain))
        return node
    timer = KernelTimer()
    r = BatchRenderer(build(), bus if bus else V, RATE, timer=timer)
    pos, parts = start, []
    from signals_amd.engine import NotBatchable
    try:
        for k in batches:
            parts.append(r.render(pos, N, k).cpu().numpy()); pos += N * k
    except NotBatchable as e:
        print('SKIP', case, kind, 'N', N, 'mods', ''.join(k[0] if v else '-' for k, v in mods.items()), str(e)[:60]); continue
    got = np.concatenate(parts)
    ref = R.render_stream(oracle(), start, N, sum(batches), V)
    if bus: ref = R.sum_bus(ref, pan if bus == 2 else None)
    scale = max(1.0, float(np.abs(ref).max()))
    err = maxerr(got, f32(ref))
    names = sorted(timer.summary())
    ok = err < 2e-6 * scale
    bad += not ok
    print('OK ' if ok else 'BAD', case, kind, btype, 'V', V, 'N', N, 'start', start, 'batches', batches, 'mods', ''.join(k[0] if v else '-' for k, v in mods.items()), 'gain', use_gain, 'bus', bus, 'err %.2e' % err, 'scale %.2f' % scale, [n.split('[')[0] for n in names])
print('bad', bad)
