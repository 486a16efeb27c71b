"""The batched engine on graphs no fused kernel covers -- post-filter Amp / Mix / RingMod, RingMod of two filtered voices, a
three-deep cascade, block-rate FM with a second oscillator, lowpass_test.sigs' Merge, and blocks NO LONGER than the filter
context (what a real-time sink pulls: 32, 64, 100 frames): each is ONE interpreted launch (sig_voice_program) per sink, checked
against the CPU oracle driven like the reference (sequential pulls, block caches) and against the eager pull path."""
import numpy as np
import pytest
import torch

from helpers import RATE, f32, fix, maxerr, mkosc, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    assert torch.cuda.is_available()
    from signals_amd import _native, runtime
    runtime.set_device('cuda:0')
    _native.lib()


def draw(V, seed):
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, np.pi / 2, V)
    return dict(hertz=rng.uniform(55, 1760, (1, V)), phase=rng.uniform(0, 1, (1, V)), hertz2=rng.uniform(55, 1760, (1, V)),
                cut1=rng.uniform(200, 8000, (1, V)), cut2=rng.uniform(200, 8000, (1, V)), cut3=rng.uniform(200, 8000, (1, V)),
                gain=rng.uniform(0.2, 1.0, (1, V)), mix=rng.uniform(0, 1, (1, V)), expo=rng.uniform(0.5, 2.0, (1, V)),
                pan=np.stack([np.cos(th), np.sin(th)]))


def render_batches(node, channels, position, N, batches, timer=None, **kw):
    from signals_amd.engine import BatchRenderer
    r = BatchRenderer(node, channels, RATE, timer=timer, **kw)
    parts, pos = [], position
    for k in batches:
        parts.append(r.render(pos, N, k).cpu().numpy())
        pos += N * k
    return np.concatenate(parts)


def launches(timer):
    torch.cuda.synchronize()
    return set(timer.summary())


def lfo(hz, depth, centre):
    """depth * sin + centre as Mix(Gain(Sine, 2 depth), 2 centre, 0.5), GPU nodes and oracle nodes"""
    from oracle import chain_ref as R
    from signals_amd.chain import fx
    s = mkosc('Sine', [[hz]])
    g = fx.Gain(); g.left = s; g.right = fix([[2.0 * depth]])
    m = fx.Mix(); m.left = g; m.right = fix(2.0 * np.asarray(centre)); m.mix = fix([[0.5]])
    ref = R.Binary('Mix', R.Binary('Gain', R.Osc('Sine', R.Fixed([[hz]])), R.Fixed([[2.0 * depth]])), R.Fixed(2.0 * np.asarray(centre)),
                   R.Fixed([[0.5]]))
    return m, ref


def shapes(p, which):
    """(GPU graph, oracle graph) of one voice shape"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    V = p['hertz'].shape[1]
    saw, rsaw = mkosc('Sawtooth', p['hertz'], p['phase']), R.Osc('Sawtooth', R.Fixed(p['hertz']), R.Fixed(p['phase']))
    tri, rtri = mkosc('Triangle', p['hertz2']), R.Osc('Triangle', R.Fixed(p['hertz2']))
    if which == 'amp_after_filter':                                            # fx.py:55-60 behind fx.py:85-106
        f = fx.LowPass(); f.input = saw; f.cutoff = fix(p['cut1'])
        a = fx.Amp(); a.left = f; a.right = fix(p['expo'])
        g = fx.Gain(); g.left = a; g.right = fix(p['gain'])
        return g, R.Binary('Gain', R.Binary('Amp', R.Filter('lp', rsaw, R.Fixed(p['cut1'])), R.Fixed(p['expo'])), R.Fixed(p['gain']))
    if which == 'ringmod_of_two_filtered':
        f1 = fx.LowPass(); f1.input = saw; f1.cutoff = fix(p['cut1'])
        f2 = fx.HighPass(); f2.input = tri; f2.cutoff = fix(p['cut2'])
        rm = fx.RingMod(); rm.left = f1; rm.right = f2
        return rm, R.Binary('RingMod', R.Filter('lp', rsaw, R.Fixed(p['cut1'])), R.Filter('hp', rtri, R.Fixed(p['cut2'])))
    if which == 'mix_after_filter':
        f = fx.HighPass(); f.input = saw; f.cutoff = fix(p['cut1'])
        m = fx.Mix(); m.left = f; m.right = tri; m.mix = fix(p['mix'])
        return m, R.Binary('Mix', R.Filter('hp', rsaw, R.Fixed(p['cut1'])), rtri, R.Fixed(p['mix']))
    if which == 'three_filters':
        f1 = fx.LowPass(); f1.input = saw; f1.cutoff = fix(p['cut1'])
        f2 = fx.HighPass(); f2.input = f1; f2.cutoff = fix(p['cut2'])
        f3 = fx.LowPass(); f3.input = f2; f3.cutoff = fix(p['cut3'])
        return f3, R.Filter('lp', R.Filter('hp', R.Filter('lp', rsaw, R.Fixed(p['cut1'])), R.Fixed(p['cut2'])), R.Fixed(p['cut3']))
    if which == 'fm_with_second_oscillator':                                   # osc.py:28-30 + fx.py:43-46
        hz, rhz = lfo(5.3, 4.5, p['hertz'])
        o = mkosc('Sawtooth', p['hertz'], p['phase']); o.hertz = hz
        rm = fx.RingMod(); rm.left = o; rm.right = tri
        f = fx.LowPass(); f.input = rm; f.cutoff = fix(p['cut1'])
        return f, R.Filter('lp', R.Binary('RingMod', R.Osc('Sawtooth', rhz, R.Fixed(p['phase'])), rtri), R.Fixed(p['cut1']))
    if which == 'modulated_cascade':                                           # a swept inner cutoff and a tremolo around two filters
        cut, rcut = lfo(1.7, 150.0, p['cut1'])
        trem, rtrem = lfo(3.1, 0.1, p['gain'])
        f1 = fx.LowPass(); f1.input = saw; f1.cutoff = cut
        f2 = fx.LowPass(); f2.input = f1; f2.cutoff = fix(p['cut2'])
        g = fx.Gain(); g.left = f2; g.right = trem
        return g, R.Binary('Gain', R.Filter('lp', R.Filter('lp', rsaw, rcut), R.Fixed(p['cut2'])), rtrem)
    raise KeyError(which)


SHAPES = ['amp_after_filter', 'ringmod_of_two_filtered', 'mix_after_filter', 'three_filters', 'fm_with_second_oscillator',
          'modulated_cascade']


@pytest.mark.parametrize('which', SHAPES)
def test_shapes_beyond_the_fused_kernels_are_one_launch_and_match_the_oracle(which):
    from oracle import chain_ref as R
    from signals_amd.chain import ext
    from signals_amd.engine import KernelTimer
    V, N = 96, 256
    p = draw(V, 13)
    p['cut1'][0, :8] = np.linspace(25.0, 140.0, 8)                           # slow filters: the block history matters
    batches = (3, 1, 4)
    node, ref_node = shapes(p, which)
    ref = R.render_stream(ref_node, 0, N, sum(batches), V)
    scale = max(1.0, np.nanmax(np.abs(ref)))
    timer = KernelTimer()
    got = render_batches(node, V, 0, N, batches, timer, fuse_program='always')     # (by default only where it beats one kernel per node)
    names = launches(timer)
    assert len([n for n in names if n.startswith('voice_program[')]) == 1, names
    assert all(n.startswith(('voice_program[', 'control_program')) for n in names), names
    assert maxerr(got, f32(ref)) < 1e-6 * scale, which
    # the same voices under a stereo bus: graph and bus in one launch
    node, ref_node = shapes(p, which)
    bus = ext.SumBus(); bus.input = node; bus.get_state().gains = np.ascontiguousarray(p['pan'])
    timer = KernelTimer()
    got = render_batches(bus, 2, 0, N, batches, timer, fuse_program='always')
    names = launches(timer)
    assert any(n.startswith('voice_program_bus[') for n in names) and not any(n.startswith(('sum_bus', 'biquad', 'osc_bank')) for n in names), names
    want = np.nan_to_num(R.render_stream(ref_node, 0, N, sum(batches), V)) @ p['pan'].T if which == 'amp_after_filter' else ref @ p['pan'].T
    if which != 'amp_after_filter':                                            # (Amp's NaN voices poison a bus: checked per voice above)
        assert maxerr(got, f32(want)) < 1e-6 * max(1.0, np.abs(want).max()), which
    # a fresh renderer mid-stream answers what a fresh reference graph answers
    node, ref_node = shapes(p, which)
    fresh_ref = R.render_stream(ref_node, 5 * N, N, 2, V)
    fresh = render_batches(node, V, 5 * N, N, (2,), fuse_program='always')
    assert maxerr(fresh, f32(fresh_ref)) < 1e-6 * scale, which
    # the default policy: the interpreter where it beats one kernel per node (programs that fit its small register file)
    node, _ = shapes(p, which)
    timer = KernelTimer()
    default = render_batches(node, V, 0, N, batches, timer)
    small = which not in ('amp_after_filter', 'three_filters')
    if which != 'three_filters':                          # (there the inner two filters' history block IS a small program)
        assert any(n.startswith('voice_program[') for n in launches(timer)) == small, (which, launches(timer))
    assert maxerr(default, f32(ref)) < 1e-6 * scale, which


@pytest.mark.parametrize('kind', ['Sawtooth', 'Sine', 'two'])
def test_voices_without_a_filter_under_a_bus_are_one_launch(kind):
    """SumBus(Gain(Osc)) / SumBus(Mix(Osc, Osc)): oscillator(s) and bus in one interpreted launch instead of an oscillator kernel
    writing 4 B per voice-sample for the bus kernel to read back (osc.py:26-62, fx.py:35-46)"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    from signals_amd.engine import KernelTimer
    V, N = 200, 256
    p = draw(V, 29)
    first = 'Sawtooth' if kind == 'two' else kind
    o, ro = mkosc(first, p['hertz'], p['phase']), R.Osc(first, R.Fixed(p['hertz']), R.Fixed(p['phase']))
    if kind == 'two':
        m = fx.Mix(); m.left = o; m.right = mkosc('Triangle', p['hertz2']); m.mix = fix(p['mix'])
        o, ro = m, R.Binary('Mix', ro, R.Osc('Triangle', R.Fixed(p['hertz2'])), R.Fixed(p['mix']))
    g = fx.Gain(); g.left = o; g.right = fix(p['gain'])
    bus = ext.SumBus(); bus.input = g; bus.get_state().gains = np.ascontiguousarray(p['pan'])
    batches = (2, 5)
    want = (R.render_stream(R.Binary('Gain', ro, R.Fixed(p['gain'])), 7 * N, N, sum(batches), V)) @ p['pan'].T
    timer = KernelTimer()
    got = render_batches(bus, 2, 7 * N, N, batches, timer)
    names = launches(timer)
    assert all(n.startswith('voice_program_bus[') for n in names), names
    assert maxerr(got, f32(want)) < 1e-6 * max(1.0, np.abs(want).max())
    per_node = render_batches(bus, 2, 7 * N, N, batches, fuse_program=False)
    assert maxerr(got, per_node) < 1e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('N', [32, 64, 100])
def test_blocks_no_longer_than_the_context_are_batched(N):
    """cascades and block-rate FM with 32-, 64- and 100-frame blocks (dev.py:139-141: the sink takes whatever block size PortAudio
    hands it): no NotBatchable any more, whatever the batching of the stream -- against the oracle's sequential stream and the
    eager pull path"""
    from oracle import chain_ref as R
    from signals_amd.chain import ext, fx
    from signals_amd.engine import KernelTimer
    V = 40
    p = draw(V, 17 + N)
    p['cut1'][0, :8] = np.linspace(25.0, 140.0, 8)
    batches = (1, 3, 2, 6)

    def cascade():
        f1 = fx.LowPass(); f1.input = mkosc('Triangle', p['hertz'], p['phase']); f1.cutoff = fix(p['cut1'])
        f2 = fx.HighPass(); f2.input = f1; f2.cutoff = fix(p['cut2'])
        g = fx.Gain(); g.left = f2; g.right = fix(p['gain'])
        return g
    ref_cascade = R.Binary('Gain', R.Filter('hp', R.Filter('lp', R.Osc('Triangle', R.Fixed(p['hertz']), R.Fixed(p['phase'])), R.Fixed(p['cut1'])),
                                            R.Fixed(p['cut2'])), R.Fixed(p['gain']))
    ref = R.render_stream(ref_cascade, 0, N, sum(batches), V)
    timer = KernelTimer()
    got = render_batches(cascade(), V, 0, N, batches, timer)
    assert all(n.startswith('voice_program[') for n in launches(timer))
    assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), N
    eager = stream(cascade(), 0, N, sum(batches), V)
    assert maxerr(got, eager) < 1e-6 * max(1.0, np.abs(ref).max()), N       # (the eager path rounds every edge to float32)
    bus = ext.SumBus(); bus.input = cascade(); bus.get_state().gains = np.ascontiguousarray(p['pan'])
    got_bus = render_batches(bus, 2, 0, N, batches)
    want = ref @ p['pan'].T
    assert maxerr(got_bus, f32(want)) < 1e-6 * max(1.0, np.abs(want).max()), N

    def fm_voice():
        hz, rhz = lfo(5.3, 4.5, p['hertz'])
        cut, rcut = lfo(1.7, 150.0, p['cut2'])
        trem, rtrem = lfo(3.1, 0.1, p['gain'])
        o = mkosc('Sawtooth', p['hertz'], p['phase']); o.hertz = hz
        f = fx.LowPass(); f.input = o; f.cutoff = cut
        g = fx.Gain(); g.left = f; g.right = trem
        return g, R.Binary('Gain', R.Filter('lp', R.Osc('Sawtooth', rhz, R.Fixed(p['phase'])), rcut), rtrem)
    node, ref_node = fm_voice()
    ref = R.render_stream(ref_node, 0, N, sum(batches), V)
    got = render_batches(node, V, 0, N, batches)
    assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), ('fm', N)
    node, _ = fm_voice()
    assert maxerr(got, stream(node, 0, N, sum(batches), V)) < 1e-6 * max(1.0, np.abs(ref).max()), ('fm eager', N)


def test_lowpass_test_patch_merge_is_two_launches():
    """src/signals/lowpass_test.sigs: Triangle -> Gain -> LowPass -> Merge(., .) (shape.py:69-74): each side of the Merge is one
    launch, the concatenation is buffer plumbing"""
    from oracle import chain_ref as R
    from signals_amd.chain import fx, shape
    from signals_amd.engine import KernelTimer
    V, N, K = 8, 256, 4
    p = draw(V, 23)

    def side(hz):
        g = fx.Gain(); g.left = mkosc('Triangle', hz); g.right = fix([[0.5]])
        f = fx.LowPass(); f.input = g; f.cutoff = fix(p['cut1'])
        a = fx.Amp(); a.left = f; a.right = fix([[1.0]])
        return a
    m = shape.Merge(); m.left = side(p['hertz']); m.right = side(p['hertz2'])
    rside = lambda hz: R.Binary('Amp', R.Filter('lp', R.Binary('Gain', R.Osc('Triangle', R.Fixed(hz)), R.Fixed([[0.5]])), R.Fixed(p['cut1'])),
                                R.Fixed([[1.0]]))
    ref = R.render_stream(R.Merge(rside(p['hertz']), rside(p['hertz2']), V, V), 0, N, K, 2 * V)
    timer = KernelTimer()
    got = render_batches(m, 2 * V, 0, N, (K,), timer, fuse_program='always')
    names = launches(timer)
    assert names == {'voice_program[Osc,Gain,Filter,Amp]'}, names
    assert got.shape == (N * K, 2 * V) and maxerr(got, f32(ref)) < 1e-6


@pytest.mark.parametrize('N', [32, 50, 64, 100])
def test_short_block_fixtures_of_the_reference(golden, N):
    """tests/golden/small.npz -- OUTPUTS OF THE REFERENCE ITSELF: a LowPass -> HighPass cascade and a vibrato + phase wobble +
    cutoff sweep + tremolo voice rendered sequentially in 32-, 50-, 64- and 100-frame blocks from 0 and from 4096; the engine's
    default schedule (one interpreted launch per batch, whatever the batching) within 1e-6; three filters in series at 256"""
    from signals_amd.chain import fx
    from signals_amd.engine import KernelTimer
    g = golden('small')
    V = g['small/hertz'].shape[1]

    def mod(kind, hz, depth, centre):
        m = fx.Mix(); m.left = mkosc(kind, [[hz]]); m.right = fix([[1.0]]); m.mix = fix([[depth]])
        r = fx.RingMod(); r.left = m; r.right = fix(centre)
        return r

    def cascade(depth=2):
        f1 = fx.LowPass(); f1.input = mkosc('Triangle', g['small/hertz'], g['small/phase']); f1.cutoff = fix(g['small/cut1'])
        f2 = fx.HighPass(); f2.input = f1; f2.cutoff = fix(g['small/cut2'])
        top = f2
        if depth == 3:
            top = fx.LowPass(); top.input = f2; top.cutoff = fix(g['small/cut3'])
        gn = fx.Gain(); gn.left = top; gn.right = fix(g['small/gain'])
        return gn

    def fm():
        o = mkosc('Sawtooth', g['small/hertz'], g['small/phase'])
        o.hertz = mod('Sine', 5.3, 0.02, g['small/hertz']); o.phase = mod('Triangle', 2.1, 0.1, g['small/phase'])
        f = fx.LowPass(); f.input = o; f.cutoff = mod('Sine', 1.7, 0.4, g['small/cut2'])
        gn = fx.Gain(); gn.left = f; gn.right = mod('Triangle', 3.1, 0.3, g['small/gain'])
        return gn
    for start in (0, 4096):
        for name, build in (('cascade', cascade), ('fm', fm)):
            ref = g[f'small/{name}/n{N}_p{start}']
            timer = KernelTimer()
            got = render_batches(build(), V, start, N, (1, 4, 2, 5), timer)
            want = 'fused_osc_biquad[' if (name == 'fm' and N >= 100) else 'voice_program['      # (100-frame FM: the walker's own block-rate rows)
            assert any(n.startswith(want) for n in launches(timer))
            assert maxerr(got, f32(ref)) < 1e-6, (name, N, start)
    if N == 32:
        for mode in (True, 'always'):                        # three filters in series: per node by default, one launch on request
            got = render_batches(cascade(3), V, 0, 256, (2, 1, 3), fuse_program=mode)
            assert maxerr(got, f32(g['small/cascade3/n256_p0'])) < 1e-6, mode
            assert maxerr(render_batches(cascade(3), V, 1000, 256, (1,), fuse_program=mode), f32(g['small/cascade3/fresh_p1000'])) < 1e-6, mode


@pytest.mark.parametrize('name', ['ringmod', 'mix', 'amp', 'fanout', 'swept_cascade'])
def test_voice_graph_fixtures_of_the_reference(golden, name):
    """tests/golden/shapes.npz -- OUTPUTS OF THE REFERENCE ITSELF for graphs beyond the fused kernels' patterns (RingMod of two
    filtered oscillators, Mix and Amp behind a filter, a node with two readers, a swept cascade under a tremolo; six 256-frame
    blocks from 0 and a fresh graph at 1000): the engine's schedules -- default policy, every graph forced through the
    interpreted launch, and the kernel specialised for the program -- within 1e-6, whatever the batching"""
    from signals_amd import specialise
    from signals_amd.chain import fx
    from signals_amd.engine import KernelTimer
    g = golden('shapes')
    F = lambda k: fix(g[f'shapes/{k}'])
    V = g['shapes/hertz'].shape[1]
    saw = lambda: mkosc('Sawtooth', g['shapes/hertz'], g['shapes/phase'])
    tri = lambda: mkosc('Triangle', g['shapes/hertz2'])

    def filt(cls, src, cut):
        f = cls(); f.input = src; f.cutoff = cut
        return f

    def lfo(hz, depth, centre):
        m = fx.Mix(); m.left = mkosc('Sine', [[hz]]); m.right = fix([[1.0]]); m.mix = fix([[depth]])
        r = fx.RingMod(); r.left = m; r.right = fix(centre)
        return r

    def build():
        if name == 'ringmod':
            n = fx.RingMod(); n.left = filt(fx.LowPass, saw(), F('cut1')); n.right = filt(fx.HighPass, tri(), F('cut2'))
        elif name == 'mix':
            n = fx.Mix(); n.left = filt(fx.HighPass, saw(), F('cut1')); n.right = tri(); n.mix = F('mix')
        elif name == 'amp':
            a = fx.Amp(); a.left = filt(fx.LowPass, saw(), F('cut2')); a.right = F('expo')
            n = fx.Gain(); n.left = a; n.right = F('gain')
        elif name == 'fanout':
            shared = filt(fx.LowPass, saw(), F('cut1'))
            rm = fx.RingMod(); rm.left = shared; rm.right = tri()
            n = fx.Mix(); n.left = shared; n.right = rm; n.mix = F('mix')
        else:
            inner = filt(fx.LowPass, saw(), lfo(1.7, 0.4, g['shapes/cut2']))
            outer = filt(fx.LowPass, inner, fix(g['shapes/cut1'] * 4.0))
            n = fx.Gain(); n.left = outer; n.right = lfo(3.1, 0.3, g['shapes/gain'])
        return n
    ref, fresh = g[f'shapes/{name}/n256_p0'], g[f'shapes/{name}/fresh_p1000']
    scale = max(1.0, float(np.nanmax(np.abs(ref))))
    modes = [dict(), dict(fuse_program='always'), dict(fuse_program=False)]
    if specialise.hipcc() is not None:
        modes.append(dict(fuse_program='always', specialise=True))
    for kw in modes:
        timer = KernelTimer()
        got = render_batches(build(), V, 0, 256, (2, 3, 1), timer, **kw)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), (name, kw)           # (Amp: NaN where the reference has NaN)
        assert maxerr(got, f32(ref)) < 1e-6 * scale, (name, kw)
        if kw.get('fuse_program') == 'always':
            assert any(n.startswith('voice_program[') for n in launches(timer)), (name, launches(timer))
        if kw.get('specialise'):
            assert any(n.endswith('*specialised') for n in launches(timer)), (name, launches(timer))
        got = render_batches(build(), V, 1000, 256, (1,), **kw)
        assert np.array_equal(np.isnan(got), np.isnan(fresh)) and maxerr(got, f32(fresh)) < 1e-6 * scale, (name, kw)
