"""GPU parity, batched engine: K blocks per launch must equal K sequential eager requests (and the
reference's golden streams), including cascade cache-history semantics."""
import numpy as np
import pytest
import torch

from helpers import HOUR, RATE, f32, fix, maxerr, mkosc, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    assert torch.cuda.is_available()
    from signals_amd import _native, runtime
    runtime.set_device('cuda:0')
    _native.lib()


def batched(node, position, frames, blocks, channels, fuse=False, scan=True):
    """fuse=False: one kernel per node, bit-identical to the eager path; fuse=True: the engine's default"""
    from signals_amd.engine import BatchRenderer
    r = BatchRenderer(node, channels, RATE, fuse=fuse)
    if not scan:
        r.scan_max_chains = 0
    return r.render(position, frames, blocks).cpu().numpy()


def c2_graph(g, V=None, bus=False):
    from signals_amd.chain import ext, fx
    sl = slice(None) if V is None else slice(0, V)
    f = fx.LowPass(); f.input = mkosc('Sine', g['c2/hertz'][:, sl], g['c2/phase'][:, sl]); f.cutoff = fix(g['c2/cutoff'][:, sl])
    n = fx.Gain(); n.left = f; n.right = fix(g['c2/gain'][:, sl])
    if bus:
        b = ext.SumBus(); b.input = n
        return b
    return n


@pytest.mark.parametrize('tag,pos0', (('p0', 0), ('p1h', HOUR)))
def test_c2_batched_matches_golden_and_eager(golden, tag, pos0):
    g = golden('c2')
    got = batched(c2_graph(g), pos0, 256, 4, 32)
    assert maxerr(got, f32(g[f'c2/{tag}'])) < 2e-8
    assert np.array_equal(got, stream(c2_graph(g), pos0, 256, 4, 32))      # bitwise vs the eager path


def test_single_filter_any_block_size(golden):
    from signals_amd.chain import fx
    g = golden('filter')

    def build():
        f = fx.HighPass(); f.input = mkosc('Sawtooth', g['filt/hertz'], g['filt/phase']); f.cutoff = fix(g['filt/cutoff'])
        return f

    for pos, n, k in ((0, 64, 5), (37, 33, 7), (100, 256, 3), (HOUR, 1024, 2)):
        assert np.array_equal(batched(build(), pos, n, k, 16), stream(build(), pos, n, k, 16)), (pos, n, k)


@pytest.mark.parametrize('N', (256, 1024))
def test_cascade_batched(golden, N):
    from signals_amd.chain import fx
    c = golden('cascade')

    def build():
        f1 = fx.LowPass(); f1.input = mkosc('Sawtooth', c['casc/hertz'], c['casc/phase']); f1.cutoff = fix(c['casc/cut1'])
        f2 = fx.LowPass(); f2.input = f1; f2.cutoff = fix(c['casc/cut2'])
        return f2

    got = batched(build(), 0, N, 4, 8)
    assert maxerr(got, f32(c[f'casc/seq_n{N}'])) < 3e-7
    assert np.array_equal(got, stream(build(), 0, N, 4, 8))


def test_cascade_continuing_batches_and_fresh_start(golden):
    from signals_amd.chain import fx
    from signals_amd.engine import BatchRenderer
    c = golden('cascade')

    def build(second='LowPass'):
        f1 = fx.LowPass(); f1.input = mkosc('Sawtooth', c['casc/hertz'], c['casc/phase']); f1.cutoff = fix(c['casc/cut1'])
        f2 = getattr(fx, second)(); f2.input = f1; f2.cutoff = fix(c['casc/cut2'])
        return f2

    r = BatchRenderer(build(), 8, RATE, fuse=False)
    parts = [r.render(0, 256, 1), r.render(256, 256, 2), r.render(768, 256, 1)]      # tails carried across batches
    got = torch.cat(parts).cpu().numpy()
    assert maxerr(got, f32(c['casc/seq_n256'])) < 3e-7
    assert np.array_equal(got, stream(build(), 0, 256, 4, 8))
    # fresh graph straight at 768: inner filter's history cold-starts at 568 (casc/fresh_p768)
    fresh = batched(build('HighPass'), 768, 256, 1, 8)
    assert maxerr(fresh, f32(c['casc/fresh_p768'])) < 3e-7
    # three-deep cascade, fresh at a late position, vs a fresh eager graph
    def deep():
        f3 = fx.HighPass(); f3.input = build(); f3.cutoff = fix(c['casc/cut1'] * 0.5)
        return f3
    assert np.array_equal(batched(deep(), 5000, 256, 3, 8), stream(deep(), 5000, 256, 3, 8))


def test_mixed_graph_with_bus_and_merge(golden):
    from signals_amd.chain import ext, fx, shape
    g = golden('c2')

    def build():
        gain = c2_graph(g)
        rm = fx.RingMod(); rm.left = gain; rm.right = mkosc('Triangle', g['c2/hertz'] * 0.01, g['c2/phase'])
        mx = fx.Mix(); mx.left = rm; mx.right = gain; mx.mix = fix([[0.25]])
        pan = np.random.default_rng(3).uniform(0, np.pi / 2, 32)
        bus = ext.SumBus(); bus.input = mx; bus.get_state().gains = np.stack([np.cos(pan), np.sin(pan)])
        mono = ext.SumBus(); mono.input = gain
        m = shape.Merge(); m.left = bus; m.right = mono
        return m

    got = batched(build(), 512, 256, 3, 3)
    assert got.shape == (768, 3)
    assert np.array_equal(got, stream(build(), 512, 256, 3, 3))


def test_c2_bus_vs_oracle_float64(golden):
    from oracle import chain_ref as R
    g = golden('c2')
    ref = R.sum_bus(g['c2/p0'])
    got = batched(c2_graph(g, bus=True), 0, 256, 4, 1)
    assert maxerr(got, f32(ref)) < 1e-7          # 32 voices, each within 1.5e-7 * gain(<=1/32) + bus rounding


def test_not_batchable_falls_back(golden):
    from signals_amd.chain import fx
    from signals_amd.chain.driver import BlockDriver
    from signals_amd.engine import BatchRenderer, NotBatchable
    lfo = mkosc('Sine', [[2.0]])
    scaled = fx.Gain(); scaled.left = lfo; scaled.right = fix([[500.0]])
    off = fx.Mix(); off.left = scaled; off.right = fix([[4000.0]]); off.mix = fix([[0.5]])
    # an oscillator-driven cutoff IS batched (block-rate launches) ...
    f1 = fx.LowPass(); f1.input = mkosc('Sine', [[440.0]]); f1.cutoff = off
    assert np.array_equal(batched(f1, 0, 256, 3, 1), stream(f1, 0, 256, 3, 1))
    # ... a FILTER inside a control path is not: block-rate filtering needs the eager request pattern
    smooth = fx.LowPass(); smooth.input = off; smooth.cutoff = fix([[10.0]])
    f2 = fx.LowPass(); f2.input = mkosc('Sine', [[440.0]]); f2.cutoff = smooth
    with pytest.raises(NotBatchable):
        BatchRenderer(f2, 1, RATE).render(0, 256, 2)
    d = BlockDriver(); d.input = f2
    a = d.render(3)                                          # falls back to block-by-block eager pulls
    d2 = BlockDriver(); d2.input = f2
    b = np.concatenate([d2.pull(eager=True) for _ in range(3)])
    assert a.shape == (768, 1) and np.array_equal(a, b)


def test_driver_steps_like_the_callback(golden):
    from signals_amd.chain.driver import BlockDriver
    g = golden('c2')
    d = BlockDriver(blocksize=256); d.input = c2_graph(g, bus=True)
    out = d.render(4)
    assert d.frame_position == 1024 and d.tell() == 4 and out.shape == (1024, 1)
    d.seek(0)
    assert maxerr(out[:256], d.pull(eager=True)) < 2e-8   # render() may fuse Filter(Osc)xGain; eager pull does not
    assert maxerr(out[256:512], d.pull()) < 2e-8          # default pull(): the engine, one block per launch
    stereo = BlockDriver(); stereo.get_state().channels = 2; stereo.input = c2_graph(g, bus=True)
    s = stereo.render(1)
    assert s.shape == (256, 2) and np.array_equal(s[:, 0], s[:, 1])       # (N,1) reply broadcast to 2 channels


def test_white_noise_statistics():
    from signals_amd.chain.noise import White
    from helpers import render
    w = White(); w.get_state().channels = 64
    x = render(w, 0, 4096, 64).astype(np.float64)
    assert x.min() >= 0.0 and x.max() < 1.0
    assert abs(x.mean() - 0.5) < 5e-3 and abs(x.var() - 1 / 12) < 2e-3
    assert abs(np.corrcoef(x[:-1].ravel(), x[1:].ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(x[:, :-1].ravel(), x[:, 1:].ravel())[0, 1]) < 0.01
    w2 = White(); w2.get_state().channels = 64
    assert np.array_equal(render(w2, 1000, 96, 64), x[1000:1096].astype(np.float32))       # position-pure
    assert np.array_equal(batched(w2, 0, 256, 16, 64), x.astype(np.float32))


def adsr_rows(V, seed=11):
    rng = np.random.default_rng(seed)
    return dict(attack=rng.uniform(0.001, 0.02, (1, V)), decay=rng.uniform(0.001, 0.03, (1, V)),
                sustain=rng.uniform(0.2, 0.9, (1, V)), release=rng.uniform(0.001, 0.05, (1, V)),
                gate_on=rng.uniform(0.0, 0.01, (1, V)), gate_off=rng.uniform(0.03, 0.06, (1, V)))


def mk_adsr(rows):
    from signals_amd.chain.ext import ADSR
    a = ADSR()
    for k, v in rows.items():
        setattr(a, k, fix(v))
    return a


def test_adsr_bit_exact_vs_oracle():
    from oracle import chain_ref as R
    from helpers import render
    V = 24
    rows = adsr_rows(V)
    rows['attack'][0, 0] = 0.0; rows['decay'][0, 1] = 0.0; rows['release'][0, 2] = 0.0     # zero-length stages
    rows['gate_off'][0, 3] = rows['gate_on'][0, 3] + 0.5 * rows['attack'][0, 3]             # released mid-attack
    for pos, n in ((0, 8192), (1000, 777)):
        got = render(mk_adsr(rows), pos, n, V)
        ref = R.adsr(pos, n, RATE, **rows)
        assert got.shape == (n, V) and np.array_equal(got, f32(ref)), (pos, n)
    env = R.adsr(0, 8192, RATE, **rows)
    assert env.min() >= 0.0 and env.max() <= 1.0 and env[-1].max() == 0.0 and env.max() > 0.99
    ctrl = render(mk_adsr(rows), 960, 1, V)
    assert ctrl.dtype == np.float64 and np.array_equal(ctrl, R.adsr(960, 1, RATE, **rows))
    assert np.array_equal(batched(mk_adsr(rows), 0, 256, 32, V), f32(env))


def test_mix_matrix_mfma_layout_and_values():
    """A = identity-like probes with an ASYMMETRIC matrix catch row/col swaps in the MFMA fragment maps."""
    from oracle import chain_ref as R
    from signals_amd import _native
    rng = np.random.default_rng(5)
    M = rng.standard_normal((64, 64)) / 8
    for rows, V in ((32, 64), (77, 128), (256, 4096)):
        x = rng.standard_normal((rows, V)).astype(np.float32)
        xt = torch.from_numpy(x).cuda()
        out = _native.mix_matrix(xt, torch.from_numpy(M.astype(np.float32)).cuda(), torch.empty_like(xt)).cpu().numpy()
        ref = R.mix_matrix(x.astype(np.float64), M.astype(np.float32).astype(np.float64))
        assert maxerr(out, ref) < 1e-6 * max(1.0, float(np.abs(ref).max())), (rows, V)        # 64-term float32 accumulation, full scale ~4
    eye = np.zeros((64, 64), dtype=np.float32); eye[np.arange(64), np.arange(64)] = 1
    asym = (np.arange(64)[:, None] * 64 + np.arange(64)[None, :]).astype(np.float32)      # exact integers
    out = _native.mix_matrix(torch.from_numpy(eye).cuda(), torch.from_numpy(asym).cuda(),
                             torch.empty(64, 64, device='cuda')).cpu().numpy()
    assert np.array_equal(out, asym)


def c3_graph(V, seed=21):
    """BASELINE config 3: Saw -> LowPass -> LowPass -> (x ADSR) -> SumBus"""
    from signals_amd.chain import ext, fx
    rng = np.random.default_rng(seed)
    p = dict(hertz=rng.uniform(55, 1760, (1, V)), phase=rng.uniform(0, 1, (1, V)),
             cut1=rng.uniform(200, 8000, (1, V)), cut2=rng.uniform(200, 8000, (1, V)), env=adsr_rows(V, seed))
    f1 = fx.LowPass(); f1.input = mkosc('Sawtooth', p['hertz'], p['phase']); f1.cutoff = fix(p['cut1'])
    f2 = fx.LowPass(); f2.input = f1; f2.cutoff = fix(p['cut2'])
    rm = fx.RingMod(); rm.left = f2; rm.right = mk_adsr(p['env'])
    bus = ext.SumBus(); bus.input = rm
    return bus, p


def test_c3_config_vs_oracle_and_eager():
    from oracle import chain_ref as R
    V, N, K = 16, 1024, 3
    bus, p = c3_graph(V)
    got = batched(bus, 0, N, K, 1)
    o = R.Osc('Sawtooth', R.Fixed(p['hertz']), R.Fixed(p['phase']))
    f2 = R.Filter('lp', R.Filter('lp', o, R.Fixed(p['cut1'])), R.Fixed(p['cut2']))
    ref = R.sum_bus(R.render_stream(R.Binary('RingMod', f2, R.Adsr(**p['env'])), 0, N, K, V))
    assert maxerr(got, f32(ref)) < 1e-6
    assert np.array_equal(got, stream(c3_graph(V)[0], 0, N, K, 1))


def test_c5_config_vs_oracle():
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    V, N, K = 128, 256, 2
    rng = np.random.default_rng(31)
    hz, ph, cut = rng.uniform(55, 1760, (1, V)), rng.uniform(0, 1, (1, V)), rng.uniform(200, 8000, (1, V))
    M = np.linalg.qr(rng.standard_normal((64, 64)))[0]

    def build():
        f = fx.LowPass(); f.input = mkosc('Sine', hz, ph); f.cutoff = fix(cut)
        mm = ext.MixMatrix(); mm.input = f; mm.get_state().matrix = M
        return mm

    got = batched(build(), 0, N, K, V)
    lp = R.render_stream(R.Filter('lp', R.Osc('Sine', R.Fixed(hz), R.Fixed(ph)), R.Fixed(cut)), 0, N, K, V)
    ref = R.mix_matrix(lp, M.astype(np.float32).astype(np.float64))
    assert maxerr(got, f32(ref)) < min(2e-6, 1e-6 * float(np.abs(ref).max()))
    assert np.array_equal(got, stream(build(), 0, N, K, V))
    # engine default: the chain and the matrix in one launch (sig_fused_osc_biquad_mix), mid-stream too
    from signals_amd.engine import BatchRenderer, KernelTimer
    timer = KernelTimer()
    r = BatchRenderer(build(), V, RATE, timer=timer)
    fused = np.concatenate([r.render(0, N, K).cpu().numpy(), r.render(N * K, N, 3).cpu().numpy()])
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'fused_osc_biquad_mix[Sine,lp]'}, set(timer.summary())
    lp = R.render_stream(R.Filter('lp', R.Osc('Sine', R.Fixed(hz), R.Fixed(ph)), R.Fixed(cut)), 0, N, K + 3, V)
    ref = R.mix_matrix(lp, M.astype(np.float32).astype(np.float64))
    assert maxerr(fused, f32(ref)) < min(2e-6, 1e-6 * float(np.abs(ref).max()))


def test_fused_voice_chain_vs_golden_and_unfused(golden):
    """sig_fused_osc_biquad: Filter(Osc) [x Gain] in one launch; closer to the f64 reference than the
    materialised path (no f32 rounding between the stages), within 1e-6 of it either way."""
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    for tag, pos0 in (('p0', 0), ('p1h', HOUR)):
        timer = KernelTimer()
        got = BatchRenderer(c2_graph(g), 32, RATE, timer=timer, fuse=True).render(pos0, 256, 4).cpu().numpy()      # no bus on top
        torch.cuda.synchronize()
        assert list(timer.summary()) == ['fused_osc_biquad[Sine,lp,gain]']
        assert maxerr(got, f32(g[f'c2/{tag}'])) < 2e-8
        assert maxerr(got, batched(c2_graph(g), pos0, 256, 4, 32)) < 2e-8
    f = golden('filter')
    for fname, oname in (('LowPass', 'Sine'), ('HighPass', 'Sawtooth'), ('HighPass', 'Sine'), ('LowPass', 'Sawtooth')):
        for pos in (0, 50, 256, HOUR):
            flt = getattr(fx, fname)(); flt.input = mkosc(oname, f['filt/hertz'], f['filt/phase']); flt.cutoff = fix(f['filt/cutoff'])
            got = batched(flt, pos, 256, 1, 16, fuse=True)
            assert maxerr(got, f32(f[f'filt/{fname}/{oname}/p{pos}'])) < 3e-7, (fname, oname, pos)
    # odd widths / block sizes, Square and Triangle sources, scalar gain
    rng = np.random.default_rng(9)
    for V, N, K, kind in ((3, 33, 5, 'Square'), (130, 100, 3, 'Triangle'), (64, 512, 2, 'Sine')):
        hz, ph, cut = rng.uniform(55, 1760, (1, V)), rng.uniform(0, 1, (1, V)), rng.uniform(200, 8000, (1, V))

        def build():
            flt = fx.HighPass(); flt.input = mkosc(kind, hz, ph); flt.cutoff = fix(cut)
            gn = fx.Gain(); gn.left = flt; gn.right = fix([[0.5]])
            return gn
        assert maxerr(batched(build(), 77, N, K, V, fuse=True), batched(build(), 77, N, K, V)) < 3e-7, (V, N, K, kind)
    # a second consumer of the oscillator forbids fusing it away in the chain kernel: the whole graph is one voice program
    # (the oscillator kept in a temporary); without it, one kernel per node, bit-identical to the eager path
    o = mkosc('Sine', g['c2/hertz'], g['c2/phase'])
    flt = fx.LowPass(); flt.input = o; flt.cutoff = fix(g['c2/cutoff'])
    mx = fx.Mix(); mx.left = flt; mx.right = o; mx.mix = fix([[0.5]])
    timer = KernelTimer()
    got = BatchRenderer(mx, 32, RATE, timer=timer, fuse=True).render(0, 256, 2).cpu().numpy()
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'voice_program[Osc,Save,Filter,Mix]'}, set(timer.summary())
    assert maxerr(got, stream(mx, 0, 256, 2, 32)) < 3e-7
    timer = KernelTimer()
    got = BatchRenderer(mx, 32, RATE, timer=timer, fuse=True, fuse_program=False).render(0, 256, 2).cpu().numpy()
    torch.cuda.synchronize()
    assert not any(k.startswith(('fused', 'voice_program')) for k in timer.summary())
    assert np.array_equal(got, stream(mx, 0, 256, 2, 32))


def test_fused_voice_bus_vs_oracle_and_unfused(golden):
    """sig_fused_voice_bus: SumBus(Gain(Filter(Osc))) in one chain launch + a fixed-order tile sum"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    for tag, pos0 in (('p0', 0), ('p1h', HOUR)):
        timer = KernelTimer()
        r = BatchRenderer(c2_graph(g, bus=True), 1, RATE, timer=timer)
        r.scan_max_chains = 0                               # 32 voices x 4 blocks would take the latency path
        got = r.render(pos0, 256, 4).cpu().numpy()
        torch.cuda.synchronize()
        assert list(timer.summary()) == ['fused_voice_bus[Sine,lp,gain]']
        small = BatchRenderer(c2_graph(g, bus=True), 1, RATE).render(pos0, 256, 4).cpu().numpy()   # scan chain + sum_bus
        assert maxerr(small, got) < 1e-7                    # per-voice f32 rounding before the bus in the small path
        assert maxerr(got, f32(R.sum_bus(g[f'c2/{tag}']))) < 1e-7
        assert maxerr(got, batched(c2_graph(g, bus=True), pos0, 256, 4, 1)) < 1e-7
    rng = np.random.default_rng(13)
    for V, N, K, C, kind, with_gain in ((3, 33, 5, 2, 'Square', True), (130, 100, 3, 1, 'Triangle', False),
                                        (1024, 256, 3, 2, 'Sine', True), (200, 512, 2, 4, 'Sawtooth', True)):
        hz, ph, cut = rng.uniform(55, 1760, (1, V)), rng.uniform(0, 1, (1, V)), rng.uniform(200, 8000, (1, V))
        gains = rng.uniform(-1, 1, (C, V)) / V if C > 1 else None

        def build():
            node = fx.HighPass(); node.input = mkosc(kind, hz, ph); node.cutoff = fix(cut)
            if with_gain:
                gn = fx.Gain(); gn.left = node; gn.right = fix(rng.uniform(0, 1, (1, V)) if False else np.full((1, V), 0.5))
                node = gn
            bus = ext.SumBus(); bus.input = node
            if gains is not None:
                bus.get_state().gains = gains
            return bus
        fused = batched(build(), 77, N, K, C, fuse=True, scan=False)        # the bus-fused kernel itself
        plain = batched(build(), 77, N, K, C, fuse=False)
        scale = max(1.0, float(np.abs(plain).max()))
        assert fused.shape == (N * K, C) and maxerr(fused, plain) < 3e-7 * scale, (V, N, K, C, kind)
    # reproducible bit for bit, run to run
    a = batched(c2_graph(g, bus=True), 0, 256, 4, 1, fuse=True, scan=False)
    assert np.array_equal(a, batched(c2_graph(g, bus=True), 0, 256, 4, 1, fuse=True, scan=False))


def test_one_launch_replay_tracks_parameter_edits(golden):
    """latency mode: a graph that is ONE fused launch is replayed per block without re-walking it; edits to a
    Fixed's array (in place or replaced) and enable flags must still be seen, like the reference's shared arrays"""
    from signals_amd.chain import ext, fx
    from signals_amd.chain.fixed import Fixed
    from signals_amd.engine import BatchRenderer
    g = golden('c2')
    hz = g['c2/hertz'].copy()
    src = mkosc('Sine', hz, g['c2/phase'])
    f = fx.LowPass(); f.input = src; f.cutoff = fix(g['c2/cutoff'])
    gn = fx.Gain(); gn.left = f; gn.right = fix(g['c2/gain'])
    bus = ext.SumBus(); bus.input = gn
    r = BatchRenderer(bus, 1, RATE)
    a0 = r.render(0, 256, 1).cpu().numpy()
    assert r._replay is not None
    a1 = r.render(256, 256, 1).cpu().numpy()                        # replayed
    ref = batched(c2_graph(g, bus=True), 0, 256, 2, 1, fuse=True)
    assert maxerr(np.concatenate([a0, a1]), ref) < 2e-8                # one block per launch: sig_latency_voice_bus, float32 ulps of 0.1
    src.hertz.sig.get_state().value[0, :] *= 2.0                    # in-place edit of the shared array
    b = r.render(512, 256, 1).cpu().numpy()
    g2 = {k: g[k] for k in ('c2/hertz', 'c2/phase', 'c2/cutoff', 'c2/gain')}
    g2['c2/hertz'] = g['c2/hertz'] * 2.0
    assert maxerr(b, batched(c2_graph(g2, bus=True), 512, 256, 1, 1, fuse=True)) < 2e-8
    gn.get_state().enabled = False                                  # pattern broken -> re-plan -> zeros (1,1) into the bus
    with pytest.raises(Exception):
        r.render(768, 256, 1)                                       # SumBus over a one-row input is rejected, like eager would mis-shape


def test_latency_mode_prefix_scan_kernel(golden):
    """small launches run the voice chain as a wavefront prefix scan over time (fused_scan_kernel): same
    values as the serial kernels to f64 reassociation error, golden parity unchanged"""
    from signals_amd import _native
    from signals_amd.chain import fx
    f = golden('filter')
    for fname, oname in (('LowPass', 'Sine'), ('HighPass', 'Sawtooth')):
        for pos in (0, 50, 256, HOUR):                     # one block of 16 voices: scan path
            flt = getattr(fx, fname)(); flt.input = mkosc(oname, f['filt/hertz'], f['filt/phase']); flt.cutoff = fix(f['filt/cutoff'])
            got = batched(flt, pos, 256, 1, 16, fuse=True)
            assert maxerr(got, f32(f[f'filt/{fname}/{oname}/p{pos}'])) < 3e-7, (fname, oname, pos)
    # scan (V*K small) vs serial (forced by a big K) on the same voices; ragged sizes; gain stage
    rng = np.random.default_rng(23)
    mk = lambda lo, hi, V: torch.tensor(rng.uniform(lo, hi, (1, V)), device='cuda')
    for V, N, ctx_pos in ((5, 33, 7), (64, 256, 0), (130, 412, 5000), (1024, 256, HOUR)):
        hz, ph, cut, g = mk(55, 1760, V), mk(0, 1, V), mk(200, 8000, V), mk(0, 1, V)
        scan = torch.empty((N, V), device='cuda')
        _native.fused_osc_biquad('Triangle', 'lp', RATE, ctx_pos, N, 1, 100, hz, ph, cut, g, scan)
        K = 17000 // V + 1                                  # enough chains that the serial kernel is chosen
        serial = torch.empty((N * K, V), device='cuda')
        _native.fused_osc_biquad('Triangle', 'lp', RATE, ctx_pos, N, K, 100, hz, ph, cut, g, serial)
        assert float((scan - serial[:N]).abs().max()) < 2e-7, (V, N)


def modulated_graph(g, V=32):
    """FM at block rate + LFO-swept cutoff + tremolo: every control port fed by a computed block-rate signal"""
    from signals_amd.chain import ext, fx
    hz, ph, cut = g['c2/hertz'][:, :V], g['c2/phase'][:, :V], g['c2/cutoff'][:, :V]
    lfo = mkosc('Sine', [[3.0]])                                        # one channel, shared
    vib = fx.Gain(); vib.left = mkosc('Triangle', np.full((1, V), 5.0), ph); vib.right = fix([[12.0]])
    fm = fx.Mix(); fm.left = vib; fm.right = fix(hz * 2.0); fm.mix = fix([[0.5]])     # hertz = 0.5*vib + hz
    carrier = mkosc('Sawtooth', np.zeros((1, 1)), ph)
    carrier.hertz = fm
    sweep = fx.Gain(); sweep.left = lfo; sweep.right = fix([[0.4]])
    one = fix([[1.0]])
    depth = fx.Mix(); depth.left = sweep; depth.right = one; depth.mix = fix([[0.5]])   # 0.5*(0.4 lfo) + 0.5
    cutoff = fx.RingMod(); cutoff.left = depth; cutoff.right = fix(cut * 2.0)           # block-rate product (1,V)
    flt = fx.LowPass(); flt.input = carrier; flt.cutoff = cutoff
    trem = fx.Gain(); trem.left = flt; trem.right = depth
    bus = ext.SumBus(); bus.input = trem
    return bus


def test_per_block_control_inputs_batched_equals_eager(golden):
    """control ports driven by oscillators / effects (read once per block at the block's position): the
    engine evaluates them for all K blocks in block-rate launches; bitwise equal to the eager path, including
    a filter reading history rows of a frequency-modulated oscillator across batch boundaries"""
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    ref = stream(modulated_graph(g), 0, 256, 6, 1)
    timer = KernelTimer()
    r = BatchRenderer(modulated_graph(g), 1, RATE, fuse=False, timer=timer)
    got = torch.cat([r.render(0, 256, 2), r.render(512, 256, 3), r.render(1280, 256, 1)]).cpu().numpy()
    torch.cuda.synchronize()
    names = list(timer.summary())
    assert any('block-rate' in n for n in names) and any('per-block' in n for n in names), names
    assert np.array_equal(got, ref)
    assert np.abs(ref).max() > 1e-3
    # fresh start in the middle of the stream == a fresh eager graph asked for the same blocks
    assert np.array_equal(batched(modulated_graph(g), 4096, 256, 3, 1), stream(modulated_graph(g), 4096, 256, 3, 1))
    # default engine settings: the whole voice is one fused launch with per-block rows (block-rate FM included), equal to the
    # per-node result up to that path's float32 roundings between the nodes
    assert maxerr(batched(modulated_graph(g), 0, 256, 6, 1, fuse=True), ref) < 1e-6


def test_modulated_oscillator_vs_oracle(golden):
    """the same kind of patch against the CPU oracle's pull protocol (float64)"""
    from oracle import chain_ref as R
    from signals_amd.chain import fx
    g = golden('c2')
    V = 8
    hz, ph = g['c2/hertz'][:, :V], g['c2/phase'][:, :V]
    lfo = mkosc('Sine', [[2.0]])
    dev = fx.Gain(); dev.left = lfo; dev.right = fix([[30.0]])
    fm = fx.Mix(); fm.left = dev; fm.right = fix(hz * 2.0); fm.mix = fix([[0.5]])
    car = mkosc('Sine', np.zeros((1, 1)), ph); car.hertz = fm
    flt = fx.HighPass(); flt.input = car; flt.cutoff = fix(g['c2/cutoff'][:, :V])
    o_fm = R.Binary('Mix', R.Binary('Gain', R.Osc('Sine', R.Fixed([[2.0]])), R.Fixed([[30.0]])), R.Fixed(hz * 2.0), R.Fixed([[0.5]]))
    o_flt = R.Filter('hp', R.Osc('Sine', o_fm, R.Fixed(ph)), R.Fixed(g['c2/cutoff'][:, :V]))
    ref = R.render_stream(o_flt, 0, 256, 4, V)
    assert maxerr(batched(flt, 0, 256, 4, V), f32(ref)) < 3e-7


def test_fused_sine_fast_and_exact_phase_paths():
    """the fused kernels advance the Sine phase incrementally while every |t| of a wave is < 2^26 cycles and use
    the exact path (numpy's argument rounding tracked) beyond; both within 1e-6 of the reference arithmetic,
    including across the switch-over and at positions where float64 has ~1e-5 cycles of resolution left"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    rng = np.random.default_rng(29)
    V = 64
    hz, ph, cut = rng.uniform(55, 1760, (1, V)), rng.uniform(0, 1, (1, V)), rng.uniform(200, 8000, (1, V))
    hz[0, 0], hz[0, 1] = 23999.0, 0.01                       # near Nyquist / nearly DC

    def build():
        f = fx.LowPass(); f.input = mkosc('Sine', hz, ph); f.cutoff = fix(cut)
        return f
    switch = int(2 ** 26 / 1760 * RATE)                      # where the 1760 Hz voices cross 2^26 cycles
    for pos in (0, HOUR, 10 * HOUR, switch - 300, 40 * HOUR, 2 ** 40):
        got = batched(build(), pos, 256, 3, V, fuse=True, scan=False)
        ref = np.concatenate([R.filter_block('lp', lambda p, n: R.osc('Sine', p, n, RATE, hz, ph), pos + b * 256, 256,
                                             RATE, cut) for b in range(3)])
        assert maxerr(got, f32(ref)) < 4e-7, pos


def test_fused_first_stage_of_a_cascade(golden):
    """LowPass(LowPass(Osc)): the inner Filter(Osc) runs fused even though the outer filter needs its history
    rows (tail of the previous batch / fresh block); golden cascade parity and continuity across batches"""
    from signals_amd.chain import fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    c = golden('cascade')

    def build():
        f1 = fx.LowPass(); f1.input = mkosc('Sawtooth', c['casc/hertz'], c['casc/phase']); f1.cutoff = fix(c['casc/cut1'])
        f2 = fx.LowPass(); f2.input = f1; f2.cutoff = fix(c['casc/cut2'])
        return f2
    timer = KernelTimer()
    r = BatchRenderer(build(), 8, RATE, timer=timer, fuse_program=False)       # (by default the cascade is ONE interpreted launch, below)
    got = torch.cat([r.render(0, 256, 1), r.render(256, 256, 2), r.render(768, 256, 1)]).cpu().numpy()
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'fused_osc_biquad[Sawtooth,lp]', 'biquad_coldstart[lp]'}
    assert maxerr(got, f32(c['casc/seq_n256'])) < 3e-7
    timer = KernelTimer()
    r = BatchRenderer(build(), 8, RATE, timer=timer)
    got = torch.cat([r.render(0, 256, 1), r.render(256, 256, 2), r.render(768, 256, 1)]).cpu().numpy()
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'voice_program[Osc,Filter,Filter]'}, set(timer.summary())
    assert maxerr(got, f32(c['casc/seq_n256'])) < 3e-7
    assert maxerr(batched(build(), 768, 256, 1, 8, fuse=True), batched(build(), 768, 256, 1, 8)) < 3e-7     # fresh start


def test_gain_folded_into_bus_weights(golden):
    """SumBus(Gain(X, Fixed)) with no other consumer of the Gain: the engine folds the gain row into the bus
    weights instead of launching the Gain kernel (any X: here a cascade and a RingMod, mono and stereo)"""
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    V = 32
    rng = np.random.default_rng(31)
    pan = np.stack([np.cos(rng.uniform(0, 1.5, V)), np.sin(rng.uniform(0, 1.5, V))])

    def build(stereo):
        f1 = fx.LowPass(); f1.input = mkosc('Sawtooth', g['c2/hertz'], g['c2/phase']); f1.cutoff = fix(g['c2/cutoff'])
        f2 = fx.HighPass(); f2.input = f1; f2.cutoff = fix(g['c2/cutoff'] * 0.25)
        rm = fx.RingMod(); rm.left = f2; rm.right = mkosc('Sine', [[3.0]])
        gn = fx.Gain(); gn.left = rm; gn.right = fix(g['c2/gain'])
        bus = ext.SumBus(); bus.input = gn
        if stereo:
            bus.get_state().gains = pan
        return bus
    for stereo in (False, True):
        C = 2 if stereo else 1
        timer = KernelTimer()
        got = BatchRenderer(build(stereo), C, RATE, timer=timer, fuse_program=False).render(0, 256, 4).cpu().numpy()
        torch.cuda.synchronize()
        assert 'elementwise[Gain]' not in timer.summary() and 'sum_bus' in timer.summary()
        assert maxerr(got, batched(build(stereo), 0, 256, 4, C)) < 1e-7
        timer = KernelTimer()                                 # by default: the voice and the bus in one interpreted launch, the gain in the bus weights
        one = BatchRenderer(build(stereo), C, RATE, timer=timer).render(0, 256, 4).cpu().numpy()
        torch.cuda.synchronize()
        assert set(timer.summary()) == {'voice_program_bus[Osc,Filter,Filter,Save,Osc,Mul]'}, set(timer.summary())
        assert maxerr(one, got) < 1e-6 * max(1.0, float(np.abs(got).max()))


def test_hipgraph_replay_of_the_latency_loop(golden):
    """graph_replay=True: [scan chain(position on device) -> sum_bus -> position += N] captured once into a
    hipGraph and replayed per block; same bits as the uncaptured launches, across seeks and parameter edits"""
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer
    g = golden('c2')
    hz = g['c2/hertz'].copy()

    def build(hertz):
        f = fx.LowPass(); f.input = mkosc('Sine', hertz, g['c2/phase']); f.cutoff = fix(g['c2/cutoff'])
        gn = fx.Gain(); gn.left = f; gn.right = fix(g['c2/gain'])
        bus = ext.SumBus(); bus.input = gn
        return bus, f.input.sig.hertz.sig
    node, hz_fixed = build(hz)
    r = BatchRenderer(node, 1, RATE, graph_replay=True)
    r.latency_kernel = False            # (a Sine chain's one-launch block is NOT captured: a bound plain launch is faster)
    plain = BatchRenderer(build(hz)[0], 1, RATE)
    plain.latency_kernel = False
    positions = [0, 256, 512, 768, 4096, 4352, 256]                # sequential, a seek forward, a seek back
    for pos in positions:
        a = r.render(pos, 256, 1).clone()                           # graph-owned buffer: copy before the next call
        assert torch.equal(a, plain.render(pos, 256, 1)), pos
    assert r._captured is not None
    hz_fixed.get_state().value = hz * 1.5                           # new array -> new upload -> re-capture
    plain2 = BatchRenderer(build(hz * 1.5)[0], 1, RATE)
    plain2.latency_kernel = False
    for pos in (512, 768):
        assert torch.equal(r.render(pos, 256, 1).clone(), plain2.render(pos, 256, 1)), pos
    one = BatchRenderer(build(hz)[0], 1, RATE, graph_replay=True)          # default: sig_latency_voice_bus through a bound call
    blocks = [one.render(pos, 256, 1).clone() for pos in positions]
    assert one._captured is None
    ref = BatchRenderer(build(hz)[0], 1, RATE)
    for pos, blk in zip(positions, blocks):
        assert torch.equal(blk, ref.render(pos, 256, 1)), pos
    from signals_amd.chain.driver import BlockDriver
    d = BlockDriver(); d.input = build(hz)[0]
    e = BlockDriver(); e.input = build(hz)[0]
    pulled = np.concatenate([d.pull() for _ in range(6)])
    assert maxerr(pulled, np.concatenate([e.pull(eager=True) for _ in range(6)])) < 1e-7


def test_ringmod_with_adsr_in_one_pass():
    """The schedules of earlier rounds, kept behind fuse_program=False (the default runs these graphs as one interpreted launch):
    SumBus(RingMod(Filter, ADSR)) -> one pass over the filter's input, nothing per-voice stored
    (sig_biquad_coldstart_bus) when nothing else reads the three nodes; RingMod(Filter, ADSR) read by something
    else -> envelope in the filter's epilogue (sig_biquad_coldstart_env); filter read twice -> sig_adsr_apply; C3
    stays within 1e-6 of the oracle every way"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N, K = 16, 1024, 3
    o = lambda p: R.Osc('Sawtooth', R.Fixed(p['hertz']), R.Fixed(p['phase']))

    bus, p = c3_graph(V)
    timer = KernelTimer()
    r = BatchRenderer(bus, 1, RATE, timer=timer, fuse_program=False)
    r.fuse_cascade = False                       # (by default the whole voice is ONE launch: tests/test_gpu_fused_cascade.py; without the cascade kernel one interpreted launch: tests/test_gpu_program_engine.py)
    got = np.concatenate([r.render(0, N, K).cpu().numpy(), r.render(N * K, N, K).cpu().numpy()])
    torch.cuda.synchronize()
    names = set(timer.summary())
    assert 'biquad_bus[lp,env]' in names and not names & {'adsr_apply', 'adsr', 'sum_bus', 'biquad_coldstart[lp,env]'}, names
    assert not any(n.startswith('elementwise[RingMod') for n in names), names
    f2 = R.Filter('lp', R.Filter('lp', o(p), R.Fixed(p['cut1'])), R.Fixed(p['cut2']))
    ref_rm = R.render_stream(R.Binary('RingMod', f2, R.Adsr(**p['env'])), 0, N, 2 * K, V)
    assert maxerr(got, f32(R.sum_bus(ref_rm))) < 1e-6
    assert maxerr(got, batched(c3_graph(V)[0], 0, N, 2 * K, 1)) < 1e-6

    # the RingMod itself is the sink: its rows must exist, the envelope goes into the filter's epilogue
    bus, p = c3_graph(V)
    timer = KernelTimer()
    r = BatchRenderer(bus.input.sig, V, RATE, timer=timer, fuse_program=False)
    got = np.concatenate([r.render(0, N, K).cpu().numpy(), r.render(N * K, N, K).cpu().numpy()])
    torch.cuda.synchronize()
    names = set(timer.summary())
    assert 'biquad_coldstart[lp,env]' in names and not names & {'adsr_apply', 'adsr', 'biquad_bus[lp,env]'}, names
    assert maxerr(got, f32(ref_rm)) < 1e-6

    # a stereo bus with gains over a plain filter (no envelope), and a mono bus behind a Gain
    from signals_amd.chain import fx
    rng = np.random.default_rng(5)
    pan = rng.uniform(-1, 1, (2, V))
    for stereo in (True, False):
        f1 = fx.HighPass(); f1.input = mkosc('Triangle', p['hertz'], p['phase']); f1.cutoff = fix(p['cut1'])
        f2n = fx.LowPass(); f2n.input = f1; f2n.cutoff = fix(p['cut2'])
        top = f2n
        if not stereo:
            top = fx.Gain(); top.left = f2n; top.right = fix(p['cut1'] / 8000.0)
        b = ext.SumBus(); b.input = top
        if stereo:
            b.get_state().gains = pan
        timer = KernelTimer()
        rb = BatchRenderer(b, 2 if stereo else 1, RATE, timer=timer, fuse_program=False)
        rb.fuse_cascade = False
        got = rb.render(0, N, K).cpu().numpy()
        torch.cuda.synchronize()
        assert 'biquad_bus[lp]' in set(timer.summary()) and 'sum_bus' not in set(timer.summary()), set(timer.summary())
        chain = R.Filter('lp', R.Filter('hp', R.Osc('Triangle', R.Fixed(p['hertz']), R.Fixed(p['phase'])), R.Fixed(p['cut1'])),
                         R.Fixed(p['cut2']))
        ref = R.render_stream(chain, 0, N, K, V)
        want = ref @ pan.T if stereo else (ref * (p['cut1'] / 8000.0)).sum(axis=1, keepdims=True)
        assert maxerr(got, f32(want)) < min(2e-6, 1e-6 * max(1.0, float(np.abs(want).max()))), stereo

    # the filter has a second consumer: its rows must exist un-enveloped, the envelope is applied in one pass
    bus, p = c3_graph(V)
    rm = bus.input.sig
    both = fx_mix(rm, rm.left.sig)
    timer = KernelTimer()
    got = BatchRenderer(both, V, RATE, timer=timer, fuse_program=False).render(0, N, K).cpu().numpy()
    torch.cuda.synchronize()
    names = set(timer.summary())
    assert 'adsr_apply' in names and 'biquad_coldstart[lp,env]' not in names, names
    # the inner Saw -> LowPass pair runs fused (f64 oscillator samples): rounding-level agreement with the eager path
    assert maxerr(got, stream(fx_mix(*(lambda b: (b.input.sig, b.input.sig.left.sig))(c3_graph(V)[0])), 0, N, K, V)) < 1e-6


def test_closed_form_constants_are_kept_across_batches_and_follow_parameter_edits(golden):
    """the engine keeps the per-voice constants of sig_fused_voice_bus's closed form from batch to batch
    (sig_fused_voice_bus_prepared) and re-derives them when a parameter array is edited in place, when the stream
    leaves the short-context start, and when a node is swapped"""
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer
    g = golden('c2')
    N, K = 256, 8

    def build(scale_hz=1.0, scale_cut=1.0):
        f = fx.LowPass(); f.input = mkosc('Sine', g['c2/hertz'] * scale_hz, g['c2/phase']); f.cutoff = fix(g['c2/cutoff'] * scale_cut)
        gn = fx.Gain(); gn.left = f; gn.right = fix(g['c2/gain'])
        bus = ext.SumBus(); bus.input = gn
        return bus, f
    bus, f = build()
    def renderer(node):
        br = BatchRenderer(node, 1, RATE)
        br.scan_max_chains = 0                       # few voices here: force the batch kernels rather than the latency path
        return br
    r = renderer(bus)
    fresh = lambda pos, **kw: renderer(build(**kw)[0]).render(pos, N, K).cpu().numpy()
    assert maxerr(r.render(0, N, K).cpu().numpy(), fresh(0)) == 0.0                 # first block has no context: T0 != T
    assert maxerr(r.render(N * K, N, K).cpu().numpy(), fresh(N * K)) == 0.0         # constants re-derived (c0 changed) ...
    assert r._steady_consts is not None
    key = r._steady_consts[0]
    assert maxerr(r.render(2 * N * K, N, K).cpu().numpy(), fresh(2 * N * K)) == 0.0 # ... then reused
    assert r._steady_consts[0] == key
    f.cutoff.sig.get_state().value[0, :] *= 0.5                                      # in-place edit of the cutoff array
    assert maxerr(r.render(3 * N * K, N, K).cpu().numpy(), fresh(3 * N * K, scale_cut=0.5)) == 0.0
    assert r._steady_consts[0] != key
    f.input.sig.hertz = fix(g['c2/hertz'] * 2.0)                                     # another Fixed plugged in
    assert maxerr(r.render(4 * N * K, N, K).cpu().numpy(), fresh(4 * N * K, scale_hz=2.0, scale_cut=0.5)) == 0.0


def test_three_channel_bus_over_a_filter_takes_the_general_path():
    """bus widths other than 1, 2, 4 are not fused: filter launch + sum_bus, same values as eager"""
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N, K = 16, 256, 3
    rng = np.random.default_rng(9)
    hz, cut = rng.uniform(55, 1760, (1, V)), rng.uniform(200, 8000, (1, V))

    def build():
        inner = fx.HighPass(); inner.input = mkosc('Square', hz); inner.cutoff = fix(cut)
        f = fx.LowPass(); f.input = inner; f.cutoff = fix(cut[:, ::-1].copy())
        b = ext.SumBus(); b.input = f
        b.get_state().gains = rng.uniform(-1, 1, (3, V))
        return b
    rng = np.random.default_rng(10)
    bus = build()
    timer = KernelTimer()
    got = BatchRenderer(bus, 3, RATE, timer=timer).render(0, N, K).cpu().numpy()
    torch.cuda.synchronize()
    names = set(timer.summary())
    assert 'sum_bus' in names and not any(n.startswith('biquad_bus') for n in names), names
    rng = np.random.default_rng(10)
    assert maxerr(got, stream(build(), 0, N, K, 3)) < 1e-6


def fx_mix(a, b):
    from signals_amd.chain import fx
    m = fx.RingMod(); m.left = a; m.right = b
    return m


def test_gain_in_front_of_the_filter_is_folded_into_the_fused_chain(golden):
    """Filter(Gain(Osc)) = Gain(Filter(Osc)) for a block-invariant gain (the filter is linear and cold-starts every block):
    one fused launch, with a second Gain behind the filter multiplied in, with and without a bus; 1e-6 vs the oracle"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    hz, ph, cut, gain = g['c2/hertz'], g['c2/phase'], g['c2/cutoff'], g['c2/gain']
    V, N, K = hz.shape[1], 256, 4
    pre = np.random.default_rng(3).uniform(0.25, 2.0, (1, V))
    for kind in ('Sine', 'Square'):
        for post in (False, True):
            for bus in (False, True):
                gn = fx.Gain(); gn.left = mkosc(kind, hz, ph); gn.right = fix(pre)
                f = fx.HighPass(); f.input = gn; f.cutoff = fix(cut)
                top = f
                if post:
                    top = fx.Gain(); top.left = f; top.right = fix(gain)
                if bus:
                    b = ext.SumBus(); b.input = top
                    top = b
                timer = KernelTimer()
                r = BatchRenderer(top, 1 if bus else V, RATE, timer=timer)
                r.scan_max_chains = 0
                got = np.concatenate([r.render(512, N, K).cpu().numpy(), r.render(512 + N * K, N, 2).cpu().numpy()])
                torch.cuda.synchronize()
                names = set(timer.summary())
                assert len(names) == 1 and next(iter(names)).split('[')[0] in ('fused_osc_biquad', 'fused_voice_bus'), names
                node = R.Filter('hp', R.Binary('Gain', R.Osc(kind, R.Fixed(hz), R.Fixed(ph)), R.Fixed(pre)), R.Fixed(cut))
                if post:
                    node = R.Binary('Gain', node, R.Fixed(gain))
                ref = np.concatenate([R.render(node, 512 + i * N, N, V, RATE) for i in range(K + 2)])
                ref = R.sum_bus(ref) if bus else ref
                assert maxerr(got, f32(ref)) < 1e-6, (kind, post, bus)


def test_lfo_swept_cutoff_and_tremolo_run_in_the_fused_chain(golden):
    """cutoff and gain driven by block-rate signals (an LFO sweep, a tremolo: read once per block at the block's position,
    chain/__init__.py:305-306) over a position-pure oscillator: the control subgraph is evaluated for all K blocks in
    block-rate launches and the voice chain stays ONE launch with per-block parameter rows (sig_fused_osc_biquad_rows /
    sig_fused_voice_bus_rows): every block its own filter design, the next block's warm-up with the next block's design"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    V, N = 32, 256
    hz, ph, cut, gain = g['c2/hertz'][:, :V], g['c2/phase'][:, :V], g['c2/cutoff'][:, :V], g['c2/gain'][:, :V]
    assert hz.shape[1] == V

    def build(kind, bus, tremolo):
        lfo = mkosc('Sine', [[1.7]])
        sweep = fx.Mix(); sweep.left = lfo; sweep.right = fix([[1.0]]); sweep.mix = fix([[0.4]])      # 0.4 lfo + 0.6
        cutoff = fx.RingMod(); cutoff.left = sweep; cutoff.right = fix(cut)                           # (1, V) per block
        f = fx.LowPass(); f.input = mkosc(kind, hz, ph); f.cutoff = cutoff
        top = f
        if tremolo:
            trem = fx.Mix(); trem.left = mkosc('Triangle', [[3.1]]); trem.right = fix([[1.0]]); trem.mix = fix([[0.3]])
            depth = fx.RingMod(); depth.left = trem; depth.right = fix(gain)
            top = fx.Gain(); top.left = f; top.right = depth
        if bus:
            b = ext.SumBus(); b.input = top
            top = b
        return top

    def oracle(kind, bus, tremolo):
        sweep = R.Binary('Mix', R.Osc('Sine', R.Fixed([[1.7]])), R.Fixed([[1.0]]), R.Fixed([[0.4]]))
        node = R.Filter('lp', R.Osc(kind, R.Fixed(hz), R.Fixed(ph)), R.Binary('RingMod', sweep, R.Fixed(cut)))
        if tremolo:
            trem = R.Binary('Mix', R.Osc('Triangle', R.Fixed([[3.1]])), R.Fixed([[1.0]]), R.Fixed([[0.3]]))
            node = R.Binary('Gain', node, R.Binary('RingMod', trem, R.Fixed(gain)))
        return node

    for kind in ('Sine', 'Sawtooth'):
        for bus in (False, True):
            for tremolo in (False, True):
                timer = KernelTimer()
                r = BatchRenderer(build(kind, bus, tremolo), 1 if bus else V, RATE, timer=timer)
                got = np.concatenate([r.render(4096, N, 5).cpu().numpy(), r.render(4096 + 5 * N, N, 3).cpu().numpy(),
                                      r.render(4096 + 8 * N, N, 1).cpu().numpy()])
                torch.cuda.synchronize()
                names = set(timer.summary())
                fused = [n for n in names if n.startswith(('fused_osc_biquad[', 'fused_voice_bus['))]
                assert fused and all('per-block' in n for n in fused), names
                assert not names & {'biquad_coldstart[lp]', 'sum_bus', 'elementwise[Gain]', 'elementwise[Gain,per-block]'}, names
                ref = R.render_stream(oracle(kind, bus, tremolo), 4096, N, 9, V)
                ref = R.sum_bus(ref) if bus else ref
                assert maxerr(got, f32(ref)) < 1e-6, (kind, bus, tremolo)
                eager = stream(build(kind, bus, tremolo), 4096, N, 9, 1 if bus else V)
                assert maxerr(got, eager) < 1e-6, (kind, bus, tremolo)


def test_tremolo_only_sine_voice_keeps_the_closed_form(golden):
    """a Sine voice whose only block-rate parameter is its gain (a tremolo) runs the closed-form kernel with the bus weights
    rebuilt at every block's first row (fused_steady_bus_kernel<.., GROWS>) instead of the row walker: same answer as the
    walker (tuning hook) and the oracle, batches of 5 + 3 + 1 blocks, mono and stereo, voices that take the kernel's plain
    fallback among them"""
    from oracle import chain_ref as R
    from signals_amd import _native
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    V, N = 32, 256
    hz, ph, cut, gain = g['c2/hertz'][:, :V].copy(), g['c2/phase'][:, :V], g['c2/cutoff'][:, :V], g['c2/gain'][:, :V]
    hz[0, 3] = 2.0                                                      # below the closed form's range: that wave falls back
    th = np.random.default_rng(8).uniform(0, np.pi / 2, V)
    pan = np.stack([np.cos(th), np.sin(th)])

    def build(stereo):
        trem = fx.Mix(); trem.left = mkosc('Triangle', [[3.1]]); trem.right = fix([[1.0]]); trem.mix = fix([[0.3]])
        depth = fx.RingMod(); depth.left = trem; depth.right = fix(gain)
        f = fx.LowPass(); f.input = mkosc('Sine', hz, ph); f.cutoff = fix(cut)
        top = fx.Gain(); top.left = f; top.right = depth
        b = ext.SumBus(); b.input = top
        if stereo:
            b.get_state().gains = np.ascontiguousarray(pan)
        return b

    trem = R.Binary('Mix', R.Osc('Triangle', R.Fixed([[3.1]])), R.Fixed([[1.0]]), R.Fixed([[0.3]]))
    node = R.Binary('Gain', R.Filter('lp', R.Osc('Sine', R.Fixed(hz), R.Fixed(ph)), R.Fixed(cut)), R.Binary('RingMod', trem, R.Fixed(gain)))
    try:
        for stereo in (False, True):
            ref = R.sum_bus(R.render_stream(node, 4096, N, 9, V), pan if stereo else None)
            outs = {}
            for steady in (1, 0):
                _native.set_fused_tuning(0, 0, steady, 0)
                timer = KernelTimer()
                r = BatchRenderer(build(stereo), 2 if stereo else 1, RATE, timer=timer)
                outs[steady] = np.concatenate([r.render(4096, N, 5).cpu().numpy(), r.render(4096 + 5 * N, N, 3).cpu().numpy(),
                                               r.render(4096 + 8 * N, N, 1).cpu().numpy()])
                torch.cuda.synchronize()
                assert any(n.startswith('fused_voice_bus[Sine,lp,gain,per-block]') for n in timer.summary()), set(timer.summary())
                assert maxerr(outs[steady], f32(ref)) < 1e-6, (stereo, steady)
            assert maxerr(outs[1], outs[0]) < 2e-7, stereo
    finally:
        _native.set_fused_tuning()


def test_swept_cutoff_sine_voice_keeps_the_closed_form():
    """a Sine voice whose cutoff is driven at block rate (an LFO sweep, with or without a tremolo: chain/__init__.py:305-306,
    fx.py:127-129) stays on the closed-form kernel: the filter, its response at the voice's frequency, T_c and the decay bound
    per (block, voice) from a prep launch, the steady-state recurrence re-seeded at every block's first row
    (fused_steady_bus_kernel<.., CROWS>).  Against the row walker (tuning hook) and the oracle: batches of 5 + 3 + 1 blocks
    from 0 (a short first context) and mid-stream, mono and stereo, every geometry, voices outside the closed form's range,
    a cutoff LFO that is one column wide"""
    from oracle import chain_ref as R
    from signals_amd import _native
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N = 200, 256
    rng = np.random.default_rng(77)
    hz, ph = rng.uniform(55, 1760, (1, V)), rng.uniform(0, 1, (1, V))
    cut, gain = np.exp(rng.uniform(np.log(60.0), np.log(9000.0), (1, V))), rng.uniform(0.2, 1.0, (1, V))
    hz[0, 3], hz[0, 130] = 2.0, 20000.0                                 # below / above the closed form's range: those waves fall back
    th = rng.uniform(0, np.pi / 2, V)
    pan = np.stack([np.cos(th), np.sin(th)])

    def lfo_gpu(kind, f_, depth, centre):
        m = fx.Mix(); m.left = mkosc(kind, [[f_]]); m.right = fix([[1.0]]); m.mix = fix([[depth]])
        r_ = fx.RingMod(); r_.left = m; r_.right = fix(centre)
        return r_

    def lfo_ref(kind, f_, depth, centre):
        return R.Binary('RingMod', R.Binary('Mix', R.Osc(kind, R.Fixed([[f_]])), R.Fixed([[1.0]]), R.Fixed([[depth]])), R.Fixed(centre))

    def build(stereo, tremolo, narrow):
        f = fx.LowPass(); f.input = mkosc('Sine', hz, ph)
        f.cutoff = lfo_gpu('Sine', 1.7, 0.4, [[1500.0]] * 1 if narrow else cut)
        top = fx.Gain(); top.left = f; top.right = lfo_gpu('Triangle', 3.1, 0.3, gain) if tremolo else fix(gain)
        b = ext.SumBus(); b.input = top
        if stereo:
            b.get_state().gains = np.ascontiguousarray(pan)
        return b

    def oracle(tremolo, narrow):
        cutoff = lfo_ref('Sine', 1.7, 0.4, np.full((1, V), 1500.0) if narrow else cut)
        g_ = lfo_ref('Triangle', 3.1, 0.3, gain) if tremolo else R.Fixed(gain)
        return R.Binary('Gain', R.Filter('lp', R.Osc('Sine', R.Fixed(hz), R.Fixed(ph)), cutoff), g_)
    try:
        for stereo, tremolo, narrow, start in ((True, True, False, 0), (False, False, False, 4096), (True, False, True, 37)):
            if narrow:
                continue                                              # (the reference indexes cutoff[0, i] per channel: a one-column cutoff raises IndexError)
            ref = R.sum_bus(R.render_stream(oracle(tremolo, narrow), start, N, 9, V), pan if stereo else None)
            scale = max(1.0, float(np.abs(ref).max()))
            outs = {}
            for vpt, span, steady in ((0, 0, 1), (8, 4, 1), (2, 3, 1), (1, 1, 1), (0, 0, 0)):
                _native.set_fused_tuning(vpt, span, steady, 0)
                timer = KernelTimer()
                r = BatchRenderer(build(stereo, tremolo, narrow), 2 if stereo else 1, RATE, timer=timer)
                got = np.concatenate([r.render(start, N, 5).cpu().numpy(), r.render(start + 5 * N, N, 3).cpu().numpy(),
                                      r.render(start + 8 * N, N, 1).cpu().numpy()])
                torch.cuda.synchronize()
                assert any(n.startswith('fused_voice_bus[Sine,lp,gain,per-block]') for n in timer.summary()), set(timer.summary())
                assert np.isfinite(got).all()
                assert maxerr(got, f32(ref)) < 1e-6 * scale, (stereo, tremolo, start, vpt, span, steady)
                outs[(vpt, span, steady)] = got
            assert maxerr(outs[(0, 0, 1)], outs[(0, 0, 0)]) < 4e-7 * scale            # closed form vs row walker
    finally:
        _native.set_fused_tuning()


def test_control_program_equals_the_node_by_node_evaluation_bit_for_bit():
    """sig_control_program: a block-rate control subgraph (oscillators of every waveform, Gain / Mix / RingMod / Amp, Fixed rows
    one column or V wide, shared sub-expressions, an unplugged port, a disabled node) compiled into one launch gives the
    bits of the node-by-node block-rate launches (the same expressions under -ffp-contract=off), for several ports at
    once, K = 1 and K = 37, mid-stream positions"""
    from signals_amd.chain import Receiver, fx, port
    from signals_amd.engine import BatchRenderer, _Batch
    rng = np.random.default_rng(5)
    V = 70
    wide, wide2 = rng.uniform(0.2, 2.0, (1, V)), rng.uniform(-1.0, 1.0, (1, V))
    lfo = mkosc('Sine', [[1.3]])
    tri = mkosc('Triangle', rng.uniform(0.5, 9.0, (1, V)), rng.uniform(0, 1, (1, V)))           # a V-wide LFO bank
    sq = mkosc('Square', [[0.7]], [[0.1]])
    saw = mkosc('Sawtooth', [[2.9]])
    scaled = fx.Gain(); scaled.left = lfo; scaled.right = fix([[0.4]])
    offset = fx.Mix(); offset.left = scaled; offset.right = fix(wide); offset.mix = fix([[0.25]])
    prod = fx.RingMod(); prod.left = offset; prod.right = tri
    amp = fx.Amp(); amp.left = prod; amp.right = fix([[1.5]])
    shared = fx.Mix(); shared.left = scaled; shared.right = sq; shared.mix = fix(np.abs(wide2))   # `scaled` used twice
    off = fx.Gain(); off.left = saw; off.right = fix(wide2); off.get_state().enabled = False      # a disabled node answers zeros
    hole = fx.RingMod(); hole.left = saw                                                           # right unplugged

    class Ports(Receiver):
        a = port('a'); b = port('b'); c = port('c'); d = port('d'); e = port('e'); f = port('f')
        HOST_ARRAYS = False

        @classmethod
        def flags(cls):
            from signals_amd import SignalFlags
            return SignalFlags(0)
    host = Ports()
    host.a, host.b, host.c, host.d, host.e, host.f = amp, shared, off, hole, fix(wide), lfo
    ports = [host.a, host.b, host.c, host.d, host.e, host.f]
    r = BatchRenderer(lfo, 1, RATE)
    for pos, N, K in ((0, 256, 37), (48000 * 3 + 17, 128, 1), (999, 64, 5)):
        one = _Batch(r, pos, N, K, False)._control_many(ports)
        ref = [_Batch(r, pos, N, K, False)._control(p, p.name) for p in ports]
        torch.cuda.synchronize()
        for name, x, y in zip('abcdef', one, ref):
            assert x.shape[1] == y.shape[1] and np.array_equal(np.broadcast_to(x.cpu().numpy(), np.broadcast_shapes(x.shape, y.shape)),
                                                               np.broadcast_to(y.cpu().numpy(), np.broadcast_shapes(x.shape, y.shape)),
                                                               equal_nan=True), (name, pos, N, K)        # (Amp of a negative base: NaN either way)
    assert len(r._ctl_programs) == 3 and all(prog.n_outs == 4 for _, prog in r._ctl_programs.values())   # a, b, d, f computed; c is a disabled node, e a Fixed row


def test_block_rate_fm_runs_in_the_fused_chain(golden):
    """hertz (vibrato) and phase driven by block-rate signals (Osc reads both ports once per block, osc.py:28-30): the voice
    chain stays ONE launch with per-block hertz / phase rows (sig_fused_osc_biquad_fm / sig_fused_voice_bus_fm).  The
    reference's oscillators keep their previous block, so a block's filter context is the PREVIOUS block's samples (made
    with the previous block's hertz): the walker's warm-up chain runs on the samples at hand, and the block in front of a
    batch is the previous batch's last one on a contiguous stream, the context request answered on its own on a fresh
    graph -- batches of 5 + 3 + 1 blocks, a fresh start mid-stream, position 0, every waveform, with and without bus,
    together with a swept cutoff"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    V, N = 32, 256
    hz, ph, cut = g['c2/hertz'][:, :V], g['c2/phase'][:, :V], g['c2/cutoff'][:, :V]

    def build(kind, bus, phase_mod, sweep):
        vib = fx.Gain(); vib.left = mkosc('Sine', [[5.3]]); vib.right = fix([[9.0]])               # +-9 Hz, shared
        fm = fx.Mix(); fm.left = vib; fm.right = fix(hz * 2.0); fm.mix = fix([[0.5]])               # hertz = 0.5 vib + hz: (1, V) per block
        o = mkosc(kind, np.zeros((1, 1)), ph)
        o.hertz = fm
        if phase_mod:
            wob = fx.Gain(); wob.left = mkosc('Triangle', [[2.1]]); wob.right = fix([[0.05]])
            pm = fx.Mix(); pm.left = wob; pm.right = fix(ph * 2.0); pm.mix = fix([[0.5]])
            o.phase = pm
        f = fx.LowPass(); f.input = o
        if sweep:
            s_ = fx.Mix(); s_.left = mkosc('Sine', [[1.7]]); s_.right = fix([[1.0]]); s_.mix = fix([[0.4]])
            c = fx.RingMod(); c.left = s_; c.right = fix(cut)
            f.cutoff = c
        else:
            f.cutoff = fix(cut)
        if bus:
            b = ext.SumBus(); b.input = f
            return b
        return f

    def oracle(kind, phase_mod, sweep):
        fm = R.Binary('Mix', R.Binary('Gain', R.Osc('Sine', R.Fixed([[5.3]])), R.Fixed([[9.0]])), R.Fixed(hz * 2.0), R.Fixed([[0.5]]))
        phase = R.Fixed(ph)
        if phase_mod:
            phase = R.Binary('Mix', R.Binary('Gain', R.Osc('Triangle', R.Fixed([[2.1]])), R.Fixed([[0.05]])), R.Fixed(ph * 2.0), R.Fixed([[0.5]]))
        cutoff = R.Fixed(cut)
        if sweep:
            cutoff = R.Binary('RingMod', R.Binary('Mix', R.Osc('Sine', R.Fixed([[1.7]])), R.Fixed([[1.0]]), R.Fixed([[0.4]])), R.Fixed(cut))
        return R.Filter('lp', R.Osc(kind, fm, phase), cutoff)

    for kind, bus, phase_mod, sweep in (('Sawtooth', False, False, False), ('Sine', True, False, False), ('Triangle', True, True, True),
                                        ('Square', False, True, False), ('Sine', False, False, True)):
        for start in (4096, 0):
            timer = KernelTimer()
            r = BatchRenderer(build(kind, bus, phase_mod, sweep), 1 if bus else V, RATE, timer=timer)
            got = np.concatenate([r.render(start, N, 5).cpu().numpy(), r.render(start + 5 * N, N, 3).cpu().numpy(),
                                  r.render(start + 8 * N, N, 1).cpu().numpy()])
            torch.cuda.synchronize()
            names = set(timer.summary())
            fused = [n for n in names if n.startswith(('fused_osc_biquad[', 'fused_voice_bus['))]
            assert fused and all(',fm' in n for n in fused), names
            assert not any(n.startswith(('osc_bank_mod', 'biquad_coldstart', 'sum_bus')) for n in names), names
            ref = R.render_stream(oracle(kind, phase_mod, sweep), start, N, 9, V)
            ref = R.sum_bus(ref) if bus else ref
            assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), (kind, bus, phase_mod, sweep, start)
            # a fresh renderer asked for the stream's last blocks only: the context request answered as a block of its own
            fresh = BatchRenderer(build(kind, bus, phase_mod, sweep), 1 if bus else V, RATE).render(start + 7 * N, N, 2).cpu().numpy()
            ref2 = R.render_stream(oracle(kind, phase_mod, sweep), start + 7 * N, N, 2, V)
            ref2 = R.sum_bus(ref2) if bus else ref2
            assert maxerr(fresh, f32(ref2)) < 1e-6 * max(1.0, np.abs(ref2).max()), (kind, bus, phase_mod, sweep, start, 'fresh')


def test_two_oscillators_through_mix_or_ringmod_in_front_of_the_filter(golden):
    """Filter(Mix(Osc, Osc, m)) and Filter(RingMod(Osc, Osc)) (fx.py:35-46): both oscillators are evaluated per row inside
    the walker (sig_fused_osc_pair_biquad / sig_fused_voice_pair_bus) -- one launch instead of osc, osc, element-wise,
    filter [, gain, bus]; every waveform as the second oscillator, a Sine first oscillator on its incremental phase"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    g = golden('c2')
    V, N, K = 32, 256, 4
    hz, ph, cut, gain = g['c2/hertz'], g['c2/phase'], g['c2/cutoff'], g['c2/gain']
    rng = np.random.default_rng(77)
    hz2, ph2, m = rng.uniform(30, 900, (1, V)), rng.uniform(0, 1, (1, V)), rng.uniform(0, 1, (1, V))
    for op in ('Mix', 'RingMod'):
        for kind_a, kind_b in (('Sine', 'Sawtooth'), ('Triangle', 'Sine'), ('Square', 'Triangle'), ('Sawtooth', 'Square')):
            for bus in (False, True):
                e = getattr(fx, op)(); e.left = mkosc(kind_a, hz, ph); e.right = mkosc(kind_b, hz2, ph2)
                if op == 'Mix':
                    e.mix = fix(m)
                f = fx.LowPass(); f.input = e; f.cutoff = fix(cut)
                top = fx.Gain(); top.left = f; top.right = fix(gain)
                if bus:
                    b = ext.SumBus(); b.input = top
                    top = b
                timer = KernelTimer()
                r = BatchRenderer(top, 1 if bus else V, RATE, timer=timer)
                got = np.concatenate([r.render(300, N, K).cpu().numpy(), r.render(300 + N * K, N, 2).cpu().numpy()])
                torch.cuda.synchronize()
                names = set(timer.summary())
                want_name = f'{"fused_voice_bus" if bus else "fused_osc_biquad"}[{op}({kind_a},{kind_b}),lp,gain]'
                assert names == {want_name}, names
                src = R.Binary(op, R.Osc(kind_a, R.Fixed(hz), R.Fixed(ph)), R.Osc(kind_b, R.Fixed(hz2), R.Fixed(ph2)),
                               R.Fixed(m) if op == 'Mix' else None)
                node = R.Binary('Gain', R.Filter('lp', src, R.Fixed(cut)), R.Fixed(gain))
                ref = np.concatenate([R.render(node, 300 + i * N, N, V, RATE) for i in range(K + 2)])
                ref = R.sum_bus(ref) if bus else ref
                assert maxerr(got, f32(ref)) < 1e-6, (op, kind_a, kind_b, bus)


@pytest.mark.parametrize('pipeline', [2, 3])
def test_pipelined_batches_over_alternating_streams_equal_the_plain_render(pipeline):
    """BatchRenderer(pipeline=n): consecutive batches of a one-launch graph (C2: osc.py:26-62 -> fx.py:85-121 -> fx.py:49-52 ->
    bus) on alternating HIP streams, each with its own workspace, the caller's stream waiting for each -- the same bits as the
    plain render, for a stream of batches consumed with a lag, across a parameter edit (the constants are renewed, the streams
    re-bound) and for a first batch inside the first context (which the plain path takes)"""
    from signals_amd.chain import ext, fx
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N, K = 256, 256, 8
    rng = np.random.default_rng(31)
    hz, ph = rng.uniform(55, 1760, (1, V)), rng.uniform(0, 1, (1, V))
    cut, gain = rng.uniform(200, 8000, (1, V)), rng.uniform(0.1, 1, (1, V)) / V
    th = rng.uniform(0, np.pi / 2, V)
    pan = np.ascontiguousarray(np.stack([np.cos(th), np.sin(th)]))

    def build():
        f = fx.LowPass(); f.input = mkosc('Sine', hz, ph); f.cutoff = fix(cut)
        g = fx.Gain(); g.left = f; g.right = fix(gain)
        b = ext.SumBus(); b.input = g; b.get_state().gains = pan
        return b, g
    plain = BatchRenderer(build()[0], 2, RATE)
    bus, gain_node = build()
    timer = KernelTimer(region=True)
    piped = BatchRenderer(bus, 2, RATE, pipeline=pipeline, timer=timer)
    plain.scan_max_chains = piped.scan_max_chains = 0        # (not the latency regime's scan kernels: the batch launch)
    pos, held = 0, []
    for i in range(14):
        if i == 9:                                               # an in-place edit of a parameter array: seen at the next render (fixed.py:38-39)
            gain_node.right.sig.get_state().value[0, :7] *= 0.5
            plain.node.input.sig.right.sig.get_state().value[0, :7] *= 0.5
        got = piped.render(pos, N, K)
        want = plain.render(pos, N, K)
        held.append((got.clone(), want.clone()))                 # (the pipelined reply is borrowed: copied before it is overwritten)
        pos += N * K
    timer.close()
    torch.cuda.synchronize()
    for i, (got, want) in enumerate(held):
        assert torch.equal(got, want), i
    assert piped._pipe is not None and len(piped._pipe['streams']) == pipeline
    assert set(timer.summary()) == {'fused_voice_bus[Sine,lp,gain]'}
