"""Pin the CPU oracle (oracle/chain_ref.py) against fixtures produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.signal

from oracle import chain_ref as R

RATE = 48000
HOUR = 172_800_000
KINDS = ('Sine', 'Square', 'Sawtooth', 'Triangle')


def same(a, b):
    """bit-for-bit, NaNs in the same places"""
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


# ------------------------------------------------------------------ oscillators (A6)
@pytest.mark.parametrize('kind', KINDS)
def test_osc_bit_exact(golden, kind):
    g = golden('osc')
    for pos in g['osc/positions']:
        ref = g[f'osc/{kind}/p{int(pos)}']
        got = R.osc(kind, int(pos), int(g['osc/frames']), RATE, g['osc/hertz'], g['osc/phase'])
        assert same(got, ref), (kind, pos)
    assert same(R.osc(kind, 256, 256, RATE, g['osc/hertz'], None), g[f'osc/{kind}/nophase/p256'])
    assert same(R.osc(kind, 0, 64, RATE, g['osc/edge/hertz'], g['osc/edge/phase']), g[f'osc/edge/{kind}'])
    assert same(R.osc(kind, 1000, 128, RATE, g['osc/neg/hertz'], g['osc/neg/phase']), g[f'osc/neg/{kind}'])


def test_osc_int_hertz_and_ctrl(golden):
    g = golden('osc')
    assert same(R.osc('Sine', 0, 256, RATE, np.array([[220]])), g['osc/int_hertz/Sine'])
    assert same(R.osc('Sine', 512, 1, RATE, g['osc/hertz'], g['osc/phase']), g['osc/ctrl/Sine/p512'])


def test_square_is_zero_at_half(golden):
    # sign(0.5 - mod(t,1)) is exactly 0 where mod == 0.5
    assert (golden('osc')['osc/edge/Square'] == 0).any()


# ------------------------------------------------------------------ filters (A7)
@pytest.mark.parametrize('fname,btype', (('LowPass', 'lp'), ('HighPass', 'hp')))
@pytest.mark.parametrize('oname', ('Sine', 'Sawtooth'))
def test_single_filter_bit_exact(golden, fname, btype, oname):
    g = golden('filter')

    def src(p, n):
        return R.osc(oname, p, n, RATE, g['filt/hertz'], g['filt/phase'])

    for pos in g['filt/positions']:
        ref = g[f'filt/{fname}/{oname}/p{int(pos)}']
        got = R.filter_block(btype, src, int(pos), 256, RATE, g['filt/cutoff'])
        assert same(got, ref), (fname, oname, pos)


def test_filter_ragged_and_short_context(golden):
    g = golden('filter')

    def src(p, n):
        return R.osc('Triangle', p, n, RATE, g['filt/hertz'], g['filt/phase'])

    for pos, n in ((7, 33), (99, 101), (100, 64), (101, 1000)):
        got = R.filter_block('lp', src, pos, n, RATE, g['filt/cutoff'])
        assert same(got, g[f'filt/ragged/p{pos}_n{n}']), (pos, n)


def test_closed_form_design_matches_scipy():
    for btype in ('lp', 'hp'):
        for hz in np.geomspace(1.0, 23999.0, 200):
            wn = hz / 24000
            ref = scipy.signal.butter(2, wn, btype, output='sos')
            assert np.max(np.abs(R.butter2_sos(wn, btype) - ref)) < 4e-15
    with pytest.raises(ValueError):
        R.butter2_sos(0.0, 'lp')
    with pytest.raises(ValueError):
        R.butter2_sos(1.0, 'hp')


def test_df2t_loop_is_sosfilt_bitwise():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(400)
    for hz in (50, 200, 1000, 8000, 20000):
        for btype in ('lp', 'hp'):
            sos = scipy.signal.butter(2, hz / 24000, btype, output='sos')
            assert np.array_equal(R.sosfilt_df2t(sos, x), scipy.signal.sosfilt(sos, x))


def test_closed_form_filter_within_1e12(golden):
    """what the HIP kernel computes (closed-form design + DF2T) vs the reference output"""
    g = golden('filter')

    def src(p, n):
        return R.osc('Sine', p, n, RATE, g['filt/hertz'], g['filt/phase'])

    for pos in (0, 50, HOUR):
        got = R.filter_block('lp', src, pos, 256, RATE, g['filt/cutoff'], closed_form=True, loop=True)
        assert np.max(np.abs(got - g[f'filt/LowPass/Sine/p{pos}'])) < 1e-12


# ------------------------------------------------------------------ graph-level: cache + cascade (A3/A4/A9)
def test_sequential_single_filter(golden):
    g = golden('filter')
    o = R.Osc('Sine', R.Fixed(g['filt/hertz']), R.Fixed(g['filt/phase']))
    f = R.Filter('lp', o, R.Fixed(g['filt/seq/cutoff']))
    assert same(R.render_stream(f, 0, 256, 4, 16), g['filt/seq/LowPass'])


@pytest.mark.parametrize('N', (256, 1024))
def test_cascade_sequential(golden, N):
    g = golden('cascade')
    o = R.Osc('Sawtooth', R.Fixed(g['casc/hertz']), R.Fixed(g['casc/phase']))
    f1 = R.Filter('lp', o, R.Fixed(g['casc/cut1']))
    f2 = R.Filter('lp', f1, R.Fixed(g['casc/cut2']))
    assert same(R.render_stream(f2, 0, N, 4, 8), g[f'casc/seq_n{N}'])


def test_cascade_fresh_graph(golden):
    g = golden('cascade')
    o = R.Osc('Sawtooth', R.Fixed(g['casc/hertz']), R.Fixed(g['casc/phase']))
    f1 = R.Filter('lp', o, R.Fixed(g['casc/cut1']))
    f2 = R.Filter('hp', f1, R.Fixed(g['casc/cut2']))
    assert same(R.render(f2, 768, 256, 8), g['casc/fresh_p768'])


def test_cascade_is_history_dependent(golden):
    """A9: block 3 rendered on a fresh graph differs from the sequential render."""
    g = golden('cascade')

    def build():
        o = R.Osc('Sawtooth', R.Fixed(g['casc/hertz']), R.Fixed(g['casc/phase']))
        return R.Filter('lp', R.Filter('lp', o, R.Fixed(g['casc/cut1'])), R.Fixed(g['casc/cut2']))

    fresh = R.render(build(), 768, 256, 8)
    assert np.max(np.abs(fresh - g['casc/seq_n256'][768:1024])) > 1e-6


# ------------------------------------------------------------------ effects (A8) and protocol corners
def test_effects(golden):
    g = golden('effects')
    V, N, pos = 8, 128, 300
    hz, ph = g['fxs/hertz'], g['fxs/phase']

    def o(kind, s=1.0):
        return R.Osc(kind, R.Fixed(hz * s), R.Fixed(ph))

    assert same(R.render(R.Binary('Gain', o('Sine'), R.Fixed(g['fxs/gain'])), pos, N, V), g['fxs/Gain'])
    assert same(R.render(R.Binary('Gain', o('Sine'), R.Fixed([[0.2]])), pos, N, V), g['fxs/Gain_scalar'])
    assert same(R.render(R.Binary('Mix', o('Sine'), o('Sawtooth', 0.5), R.Fixed(g['fxs/gain'])), pos, N, V),
                g['fxs/Mix'])
    assert same(R.render(R.Binary('RingMod', o('Sine'), o('Triangle', 0.25)), pos, N, V), g['fxs/RingMod'])
    e = g['fxs/amp_exp']
    assert same(R.render(R.Binary('Amp', o('Sawtooth'), R.Fixed(np.round(e))), pos, N, V), g['fxs/Amp_int'])
    frac = R.render(R.Binary('Amp', o('Sawtooth'), R.Fixed(e)), pos, N, V)
    assert same(frac, g['fxs/Amp_frac']) and np.isnan(frac).any()
    assert same(R.render(R.Merge(o('Sine'), o('Square', 0.5), V, V), pos, N, 2 * V), g['fxs/Merge'])


def test_protocol_corners(golden):
    g = golden('effects')
    V, N, pos = 8, 128, 300
    o = R.Osc('Sine', R.Fixed(g['fxs/hertz']), R.Fixed(g['fxs/phase']))
    o.enabled = False
    assert same(R.render(o, pos, N, V), g['fxs/disabled']) and g['fxs/disabled'].shape == (1, 1)
    assert same(R.render(R.Binary('Gain', None, R.Fixed(g['fxs/gain'])), pos, N, V), g['fxs/unplugged_left'])
    b = R.render(R.Osc('Sine', R.Fixed([[440.0]])), 0, N, 2)
    assert same(b, g['fxs/broadcast_1to2']) and b.shape == (N, 1)


def test_sigs_topologies(golden):
    g = golden('sigs')
    assert same(R.render_stream(R.Osc('Sine', R.Fixed(np.array([[220]]))), 0, 256, 3, 1), g['sigs/vis_test'])
    tri = R.Osc('Triangle', R.Fixed(np.array([[440]])))
    gn = R.Binary('Gain', tri, R.Fixed(np.array([[0.2]])))
    lp = R.Filter('lp', gn, R.Fixed(np.array([[600]])))
    m = R.Merge(lp, gn, 1, 1)
    assert same(R.render_stream(m, 0, 256, 3, 2), g['sigs/lowpass_test'])


@pytest.mark.parametrize('tag,pos0', (('p0', 0), ('p1h', HOUR)))
def test_c2_reduced(golden, tag, pos0):
    g = golden('c2')
    o = R.Osc('Sine', R.Fixed(g['c2/hertz']), R.Fixed(g['c2/phase']))
    f = R.Filter('lp', o, R.Fixed(g['c2/cutoff']))
    n = R.Binary('Gain', f, R.Fixed(g['c2/gain']))
    assert same(R.render_stream(n, pos0, 256, 4, 32), g[f'c2/{tag}'])


MOD_CASES = {'fm': ('Sawtooth', True, False, False, False), 'fm_pm_sine': ('Sine', True, True, False, False),
             'sweep_trem': ('Square', False, False, True, True), 'all': ('Triangle', True, True, True, True),
             'trem_sine': ('Sine', False, False, False, True)}


def modulated_voice(g, spec):
    """the oracle graph of tests/golden/gen_golden.py: gen_modulated"""
    kind, fm, pm, sweep, trem = spec

    def lfo(k, hz, depth, centre):
        return R.Binary('RingMod', R.Binary('Mix', R.Osc(k, R.Fixed([[hz]])), R.Fixed([[1.0]]), R.Fixed([[depth]])), R.Fixed(centre))
    hertz = lfo('Sine', 5.3, 0.02, g['mod/hertz']) if fm else R.Fixed(g['mod/hertz'])
    phase = lfo('Triangle', 2.1, 0.1, g['mod/phase']) if pm else R.Fixed(g['mod/phase'])
    cutoff = lfo('Sine', 1.7, 0.4, g['mod/cutoff']) if sweep else R.Fixed(g['mod/cutoff'])
    gain = lfo('Triangle', 3.1, 0.3, g['mod/gain']) if trem else R.Fixed(g['mod/gain'])
    return R.Binary('Gain', R.Filter('lp', R.Osc(kind, hertz, phase), cutoff), gain)


@pytest.mark.parametrize('name', sorted(MOD_CASES))
def test_block_rate_modulated_voices(golden, name):
    """control ports driven at block rate, rendered sequentially by the REFERENCE: vibrato, phase wobble, cutoff sweep,
    tremolo.  Pins what a filter's context rows are under a modulated oscillator -- the oscillator's cached previous block
    when the block is at least as long as the context (N = 256), a block of its own read at p - 100 when it is shorter
    (N = 64: the request is contained in no single cached block) -- which is what the fused FM path builds on"""
    g = golden('modulated')
    for N, blocks, start in ((256, 5, 0), (256, 4, 4096), (64, 6, 4096)):
        ref = g[f'mod/{name}/n{N}_p{start}']
        got = R.render_stream(modulated_voice(g, MOD_CASES[name]), start, N, blocks, ref.shape[1])
        assert same(got, ref), (name, N, start, float(np.abs(got - ref).max()))


PAIR_CASES = (('Mix', 'Sine', 'Sawtooth'), ('RingMod', 'Triangle', 'Square'), ('Mix', 'Sawtooth', 'Sine'))


def pair_voice(g, op, ka, kb):
    a = R.Osc(ka, R.Fixed(g['pair/hertz']), R.Fixed(g['pair/phase']))
    b = R.Osc(kb, R.Fixed(g['pair/hertz2']), R.Fixed(g['pair/phase2']))
    e = R.Binary('Mix', a, b, R.Fixed(g['pair/mix'])) if op == 'Mix' else R.Binary('RingMod', a, b)
    return R.Filter('lp', e, R.Fixed(g['pair/cutoff']))


def test_two_oscillator_and_pre_gain_voices(golden):
    """a filter reading Mix / RingMod of two oscillators, and Gain in front of a filter, rendered sequentially by the
    REFERENCE: the topologies the fuser folds into one launch"""
    g = golden('pairs')
    for op, ka, kb in PAIR_CASES:
        ref = g[f'pair/{op}_{ka}_{kb}']
        assert same(R.render_stream(pair_voice(g, op, ka, kb), 4096, 256, 3, ref.shape[1]), ref), (op, ka, kb)
    pre = R.Filter('hp', R.Binary('Gain', R.Osc('Triangle', R.Fixed(g['pair/hertz']), R.Fixed(g['pair/phase'])), R.Fixed(g['pair/mix'])),
                   R.Fixed(g['pair/cutoff']))
    ref = g['pair/pre_gain_Triangle_hp']
    assert same(R.render_stream(pre, 0, 256, 3, ref.shape[1]), ref)


def small_block_voices(g):
    """oracle graphs of tests/golden/small.npz (gen_golden.py: gen_small_blocks)"""
    def lfo(kind, hz, depth, centre):
        return R.Binary('RingMod', R.Binary('Mix', R.Osc(kind, R.Fixed([[hz]])), R.Fixed([[1.0]]), R.Fixed([[depth]])), R.Fixed(centre))

    def cascade(depth=2):
        top = R.Filter('hp', R.Filter('lp', R.Osc('Triangle', R.Fixed(g['small/hertz']), R.Fixed(g['small/phase'])), R.Fixed(g['small/cut1'])),
                       R.Fixed(g['small/cut2']))
        if depth == 3:
            top = R.Filter('lp', top, R.Fixed(g['small/cut3']))
        return R.Binary('Gain', top, R.Fixed(g['small/gain']))

    def fm():
        o = R.Osc('Sawtooth', lfo('Sine', 5.3, 0.02, g['small/hertz']), lfo('Triangle', 2.1, 0.1, g['small/phase']))
        return R.Binary('Gain', R.Filter('lp', o, lfo('Sine', 1.7, 0.4, g['small/cut2'])), lfo('Triangle', 3.1, 0.3, g['small/gain']))
    return cascade, fm


@pytest.mark.parametrize('N', [32, 50, 64, 100])
def test_blocks_no_longer_than_the_filter_context(golden, N):
    """two filters in series and a block-rate modulated voice rendered by the REFERENCE in 32-, 50-, 64- and 100-frame blocks
    (what a real-time sink pulls, dev.py:139-141): the context request is answered as a block of its own, the block itself by
    the oldest cached `after` reply containing it (chain/__init__.py:431-442).  The oracle's cache restatement reproduces
    every array bit for bit; also three filters in series at 256 frames"""
    g = golden('small')
    cascade, fm = small_block_voices(g)
    for start in (0, 4096):
        for name, build in (('cascade', cascade), ('fm', fm)):
            ref = g[f'small/{name}/n{N}_p{start}']
            got = R.render_stream(build(), start, N, 12, ref.shape[1])
            assert same(got, ref), (name, N, start, float(np.abs(got - ref).max()))
    if N == 32:
        ref = g['small/cascade3/n256_p0']
        assert same(R.render_stream(cascade(3), 0, 256, 6, ref.shape[1]), ref)
        assert same(R.render(cascade(3), 1000, 256, ref.shape[1]), g['small/cascade3/fresh_p1000'])


def shape_voices(g):
    """oracle graphs of tests/golden/shapes.npz (gen_golden.py: gen_shapes), by name"""
    F = lambda k: R.Fixed(g[f'shapes/{k}'])
    saw = lambda: R.Osc('Sawtooth', F('hertz'), F('phase'))
    tri = lambda: R.Osc('Triangle', F('hertz2'))

    def lfo(hz, depth, centre):
        return R.Binary('RingMod', R.Binary('Mix', R.Osc('Sine', R.Fixed([[hz]])), R.Fixed([[1.0]]), R.Fixed([[depth]])), R.Fixed(centre))

    def fanout():
        shared = R.Filter('lp', saw(), F('cut1'))                             # ONE node, two readers: the second is served by its block cache
        return R.Binary('Mix', shared, R.Binary('RingMod', shared, tri()), F('mix'))
    return {
        'ringmod': lambda: R.Binary('RingMod', R.Filter('lp', saw(), F('cut1')), R.Filter('hp', tri(), F('cut2'))),
        'mix': lambda: R.Binary('Mix', R.Filter('hp', saw(), F('cut1')), tri(), F('mix')),
        'amp': lambda: R.Binary('Gain', R.Binary('Amp', R.Filter('lp', saw(), F('cut2')), F('expo')), F('gain')),
        'fanout': fanout,
        'swept_cascade': lambda: R.Binary('Gain', R.Filter('lp', R.Filter('lp', saw(), lfo(1.7, 0.4, g['shapes/cut2'])),
                                                           R.Fixed(g['shapes/cut1'] * 4.0)), lfo(3.1, 0.3, g['shapes/gain'])),
    }


@pytest.mark.parametrize('name', ['ringmod', 'mix', 'amp', 'fanout', 'swept_cascade'])
def test_voice_graphs_beyond_the_fused_patterns(golden, name):
    """RingMod of two filtered oscillators, a Mix and an Amp behind a filter (fx.py:35-60 over fx.py:85-121), a node with two
    readers (chain/__init__.py:424-457), a swept cascade under a tremolo -- six 256-frame blocks from 0 and a fresh graph at
    1000, as the REFERENCE rendered them: the oracle reproduces every array bit for bit (Amp's NaN pattern included)"""
    g = golden('shapes')
    build = shape_voices(g)[name]
    with np.errstate(invalid='ignore'):
        ref = g[f'shapes/{name}/n256_p0']
        assert same(R.render_stream(build(), 0, 256, 6, ref.shape[1]), ref), name
        assert same(R.render(build(), 1000, 256, ref.shape[1]), g[f'shapes/{name}/fresh_p1000']), name
    if name == 'amp':
        assert np.isnan(ref).any() and np.isfinite(ref).any()


def test_blockloc_table(golden):
    for pos, n, bp, bf, ap, af, fr0, fr1, b_le, l_le, r_le in golden('blockloc')['blockloc/table']:
        assert R.before(int(pos), int(n), 100) == (bp, bf)
        assert R.after(int(pos), int(n), 100) == (ap, af)
        fr = R.frame_range(int(pos), int(n))
        assert fr.dtype == np.int64 and fr[0, 0] == fr0 and fr[-1, 0] == fr1 and fr.shape == (n, 1)


def test_shape_le_doctest_rows():
    # chain/__init__.py:26-51 (the only behaviour the reference's own doctests pin on this path)
    s = (10, 2)
    assert R.shape_le(s, s) and R.shape_le((1, 1), s) and R.shape_le((10, 1), s) and R.shape_le((1, 2), s)
    assert not R.shape_le((0, 0), s) and not R.shape_le((3, 2), s) and not R.shape_le((10, 0), s)
