"""sig_voice_program (signals_amd/csrc/voice_program.hip) through the C ABI, hand-assembled programs against the CPU oracle
driven like the reference (sequential pulls, block caches): the block-sequence machine -- current / next filter chains,
history blocks in front of a launch, virtual blocks on a fresh graph and behind every block shorter than the context --
for chains, cascades two and three filters deep, DAGs with temporaries, per-block control rows, every launch geometry."""
import numpy as np
import pytest
import torch

from helpers import RATE, f32, maxerr

pytestmark = pytest.mark.gpu
CTX = 100
KINDS = {'Sine': 0, 'Square': 1, 'Sawtooth': 2, 'Triangle': 3}


@pytest.fixture(scope='module', autouse=True)
def _device():
    assert torch.cuda.is_available()
    from signals_amd import _native, runtime
    runtime.set_device('cuda:0')
    yield
    _native.set_voice_program_tuning()


def dev(a):
    return torch.tensor(np.ascontiguousarray(np.array(a, ndmin=2, dtype=np.float64)), dtype=torch.float64, device='cuda')


def hist_for(position, depth, recent=()):
    """history block starts in front of `position` for a cascade `depth` filters deep: the given real block starts, then
    virtual 100-row blocks (a fresh graph's context requests), clipped at 0"""
    want = max(depth - 1, 0)
    starts = [p for p in recent if p < position][-want:] if want else []
    first = starts[0] if starts else position
    while len(starts) < want and first > 0:
        first = max(first - CTX, 0)
        starts.insert(0, first)
    return starts


def launch(code, oscs, params, filters, n_temps, depth, pos, N, K, V, bus=None, recent=(), rows=None, adsr=None, seeds=(0, 0), before=None):
    """out (K N, V) or, with bus = pan (C, V) | 'mono', the bus"""
    from signals_amd import _native
    hist = [] if N < CTX and depth > 0 else hist_for(pos, depth, recent)
    control_rows = rows if rows is not None else (2 * K if (N < CTX and depth > 0) else len(hist) + 1 + K)
    before = len([q for q in recent if q < pos]) if before is None else before      # blocks of this size rendered in front of the launch
    if bus is None:
        out = torch.full((K * N, V), float('nan'), device='cuda')
        _native.voice_program(code, oscs, params, filters, n_temps, depth, RATE, pos, N, K, CTX, V, control_rows, hist, out,
                              adsr=adsr, noise_seeds=seeds, blocks_before=before)
    else:
        pan = None if isinstance(bus, str) else dev(bus)
        C = 1 if pan is None else pan.shape[0]
        out = torch.full((K * N, C), float('nan'), device='cuda')
        _native.voice_program(code, oscs, params, filters, n_temps, depth, RATE, pos, N, K, CTX, V, control_rows, hist, out,
                              bus_gains=pan, bus=True, adsr=adsr, noise_seeds=seeds, blocks_before=before)
    return out.cpu().numpy()


def draw(V, seed):
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, np.pi / 2, V)
    return dict(hertz=rng.uniform(55, 1760, (1, V)), phase=rng.uniform(0, 1, (1, V)), hertz2=rng.uniform(55, 1760, (1, V)),
                phase2=rng.uniform(0, 1, (1, V)), cut1=rng.uniform(200, 8000, (1, V)), cut2=rng.uniform(200, 8000, (1, V)),
                cut3=rng.uniform(200, 8000, (1, V)), gain=rng.uniform(0.2, 1.0, (1, V)), mix=rng.uniform(0, 1, (1, V)),
                pan=np.stack([np.cos(th), np.sin(th)]))


@pytest.mark.parametrize('kind', list(KINDS))
def test_chain_store_and_bus_every_geometry(kind):
    """Osc -> Filter -> Gain: the shape the walkers cover, here through the interpreter -- ragged voice counts, a batch that is
    not a multiple of the span, a short first context, one and two voices per lane"""
    from oracle import chain_ref as R
    from signals_amd import _native
    V, N, K, pos = 200, 256, 5, 37
    p = draw(V, 3)
    btype = 'hp' if kind in ('Square', 'Triangle') else 'lp'
    src = lambda q, n: R.osc(kind, q, n, RATE, p['hertz'], p['phase'])
    ref = np.concatenate([R.gain(R.filter_block(btype, src, pos + b * N, N, RATE, p['cut1']), p['gain']) for b in range(K)])
    code = [('Osc', KINDS[kind], 0, 0, 0), ('Filter', 0, 0, 0, 0), ('Gain', 0, 0, 0, 0)]
    args = (code, [(dev(p['hertz']), dev(p['phase']))], [dev(p['gain'])], [(dev(p['cut1']), btype, 1)], 0, 1)
    for vpt, span in ((1, 1), (2, 1), (2, 2), (1, 4), (2, 16)):
        _native.set_voice_program_tuning(vpt, span)
        got = launch(*args, pos, N, K, V)
        assert np.isfinite(got).all() and maxerr(got, f32(ref)) < 2e-7, (kind, vpt, span)
        bus = launch(*args, pos, N, K, V, bus=p['pan'])
        want = ref @ p['pan'].T
        assert maxerr(bus, f32(want)) < 1e-6 * max(1.0, np.abs(want).max()), (kind, vpt, span)
        mono = launch(*args, pos, N, K, V, bus='mono')
        assert maxerr(mono, f32(ref.sum(axis=1, keepdims=True))) < 1e-6 * max(1.0, np.abs(ref.sum(axis=1)).max()), (kind, vpt, span)


def cascade_oracle(p, kind, types, gain=True):
    from oracle import chain_ref as R
    node = R.Osc(kind, R.Fixed(p['hertz']), R.Fixed(p['phase']))
    for t, key in zip(types, ('cut1', 'cut2', 'cut3')):
        node = R.Filter(t, node, R.Fixed(p[key]))
    return R.Binary('Gain', node, R.Fixed(p['gain'])) if gain else node


def cascade_program(p, kind, types):
    code = [('Osc', KINDS[kind], 0, 0, 0)] + [('Filter', 0, i, 0, 0) for i in range(len(types))] + [('Gain', 0, 0, 0, 0)]
    filters = [(dev(p[key]), t, i + 1) for i, (t, key) in enumerate(zip(types, ('cut1', 'cut2', 'cut3')))]
    return code, [(dev(p['hertz']), dev(p['phase']))], [dev(p['gain'])], filters, 0, len(types)


@pytest.mark.parametrize('types', [('lp', 'lp'), ('hp', 'lp', 'lp')])
def test_cascades_follow_the_block_cache_history(types):
    """two and three filters in series, slow inner filters (the history matters): batches of 3 + 2 + 4 blocks equal the oracle's
    9 sequential blocks -- the launch re-walks the blocks in front of it from where the reference cold-started them -- and a
    fresh launch mid-stream equals a fresh reference graph (virtual 100-row blocks), which is NOT the continuing stream"""
    from oracle import chain_ref as R
    from signals_amd import _native
    V, N = 70, 256
    p = draw(V, 4)
    p['cut1'][0, :8] = np.linspace(20.0, 120.0, 8)
    p['cut2'][0, :8] = np.linspace(150.0, 30.0, 8)
    ref = R.render_stream(cascade_oracle(p, 'Sawtooth', types), 0, N, 9, V)
    prog = cascade_program(p, 'Sawtooth', types)
    scale = np.abs(ref).max()
    for vpt, span in ((0, 0), (1, 1), (2, 3), (1, 16)):
        _native.set_voice_program_tuning(vpt, span)
        starts = [b * N for b in range(9)]
        got = np.concatenate([launch(*prog, 0, N, 3, V), launch(*prog, 3 * N, N, 2, V, recent=starts[:3]),
                              launch(*prog, 5 * N, N, 4, V, recent=starts[:5])])
        assert maxerr(got, f32(ref)) < 1e-6 * scale, (types, vpt, span)
        whole = launch(*prog, 0, N, 9, V, bus='mono')
        assert maxerr(whole, f32(ref.sum(axis=1, keepdims=True))) < 1e-6 * np.abs(ref.sum(axis=1)).max(), (types, vpt, span)
    _native.set_voice_program_tuning()
    fresh_ref = R.render_stream(cascade_oracle(p, 'Sawtooth', types), 5 * N, N, 2, V)
    fresh = launch(*prog, 5 * N, N, 2, V)
    assert maxerr(fresh, f32(fresh_ref)) < 1e-6 * scale
    assert maxerr(fresh, f32(ref[5 * N:7 * N])) > 1e-5 * scale
    early = R.render_stream(cascade_oracle(p, 'Sawtooth', types), 150, N, 2, V)       # virtual blocks clipped at 0
    assert maxerr(launch(*prog, 150, N, 2, V), f32(early)) < 1e-6 * scale


@pytest.mark.parametrize('N', [32, 64, 100, 50, 99])
def test_blocks_no_longer_than_the_context(N):
    """what a real-time sink asks for (dev.py:139-141: PortAudio picks the block size): with N <= 100 the context request of a
    filter lies in no single cached block of its input, so the reference answers it as a block of its own for every block
    (chain/__init__.py:431-442) -- each block is independent of how the stream was batched"""
    from oracle import chain_ref as R
    V = 40
    p = draw(V, 5 + N)
    p['cut1'][0, :8] = np.linspace(20.0, 120.0, 8)
    for types in (('lp',), ('lp', 'hp')):
        nblocks = 11
        ref = R.render_stream(cascade_oracle(p, 'Triangle', types), 0, N, nblocks, V)
        prog = cascade_program(p, 'Triangle', types)
        starts = [b * N for b in range(nblocks)]
        got = np.concatenate([launch(*prog, 0, N, 4, V), launch(*prog, 4 * N, N, 1, V, recent=starts[:4]),
                              launch(*prog, 5 * N, N, 6, V, recent=starts[:5])])
        assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), (N, types)


def test_dag_with_temporaries_mix_ringmod_and_a_shared_node():
    """RingMod(LowPass(Saw), HighPass(Mix(Tri, Sine, m))) x gain, the Saw also feeding a second product: temporaries, two
    filters side by side (depth 1), a node with two readers"""
    from oracle import chain_ref as R
    V, N, K = 96, 256, 4
    p = draw(V, 8)
    saw = R.Osc('Sawtooth', R.Fixed(p['hertz']), R.Fixed(p['phase']))
    tri = R.Osc('Triangle', R.Fixed(p['hertz2']), R.Fixed(p['phase2']))
    sine = R.Osc('Sine', R.Fixed(p['hertz'] * 0.5))
    mixed = R.Binary('Mix', tri, sine, R.Fixed(p['mix']))
    rm = R.Binary('RingMod', R.Filter('lp', saw, R.Fixed(p['cut1'])), R.Filter('hp', mixed, R.Fixed(p['cut2'])))
    top = R.Binary('Gain', R.Binary('RingMod', rm, saw), R.Fixed(p['gain']))
    ref = R.render_stream(top, 512, N, K, V)
    code = [('Osc', KINDS['Sawtooth'], 0, 0, 0), ('Save', 0, 0, 0, 0),          # T0 = saw (two readers)
            ('Filter', 0, 0, 0, 0), ('Save', 0, 1, 0, 0),                        # T1 = LowPass(saw)
            ('Osc', KINDS['Triangle'], 1, 0, 0), ('Save', 0, 2, 0, 0),           # T2 = tri
            ('Osc', KINDS['Sine'], 2, 0, 0), ('Mix', 0, 2, 1, 0),               # acc = m T2 + (1 - m) acc
            ('Filter', 0, 1, 0, 0), ('Mul', 0, 1, 0, 0), ('Mul', 0, 0, 0, 0), ('Gain', 0, 0, 0, 0)]
    oscs = [(dev(p['hertz']), dev(p['phase'])), (dev(p['hertz2']), dev(p['phase2'])), (dev(p['hertz'] * 0.5), None)]
    got = launch(code, oscs, [dev(p['gain']), dev(p['mix'])], [(dev(p['cut1']), 'lp', 1), (dev(p['cut2']), 'hp', 1)], 3, 1, 512, N, K, V)
    assert maxerr(got, f32(ref)) < 1e-6


def test_amp_adsr_and_noise_instructions():
    """Amp behind a filter (fx.py:55-60, NaN pattern included), an ADSR envelope multiplied in (build-defined), White noise as a
    source (the same counter hash as sig_white_noise)"""
    from oracle import chain_ref as R
    from signals_amd import _native
    V, N, K = 64, 256, 3
    p = draw(V, 9)
    rng = np.random.default_rng(10)
    env = dict(attack=rng.uniform(0.001, 0.01, (1, V)), decay=rng.uniform(0.002, 0.01, (1, V)), sustain=rng.uniform(0.2, 0.9, (1, V)),
               release=rng.uniform(0.002, 0.01, (1, V)), gate_on=rng.uniform(0.0, 0.002, (1, V)), gate_off=rng.uniform(0.008, 0.012, (1, V)))
    expo = rng.uniform(0.5, 2.0, (1, V))
    expo[0, :8] = [1.0, 2.0, 3.0, 0.5, 1.5, 2.0, 1.0, 3.0]
    flt = R.Filter('lp', R.Osc('Sine', R.Fixed(p['hertz']), R.Fixed(p['phase'])), R.Fixed(p['cut1']))
    top = R.Binary('RingMod', R.Binary('Amp', flt, R.Fixed(expo)), R.Adsr(**env))
    ref = R.render_stream(top, 0, N, K, V)
    code = [('Osc', KINDS['Sine'], 0, 0, 0), ('Filter', 0, 0, 0, 0), ('Amp', 0, 0, 0, 0), ('Save', 0, 0, 0, 0), ('Adsr', 0, 0, 0, 0),
            ('Mul', 0, 0, 0, 0)]
    got = launch(code, [(dev(p['hertz']), dev(p['phase']))], [dev(expo)], [(dev(p['cut1']), 'lp', 1)], 1, 1, 0, N, K, V,
                 adsr={k: dev(v) for k, v in env.items()})
    assert maxerr(got, f32(ref)) < 1e-6                      # (maxerr checks the NaN pattern: negative base, fractional exponent)
    assert np.isnan(got).any()
    noise = launch([('Noise', 0, 0, 0, 0), ('Gain', 0, 0, 0, 0)], [], [dev(p['gain'])], [], 0, 0, 777, N, 2, V, seeds=(12345, 0))
    want = torch.empty((2 * N, V), device='cuda')
    _native.white_noise(12345, 777, want)
    assert np.array_equal(noise, (want.double().cpu().numpy() * p['gain']).astype(np.float32))


def lfo_rows(positions, base, depth, hz):
    """(len(positions), V) rows: base + depth sin(2 pi hz t), evaluated by the oracle at block rate"""
    from oracle import chain_ref as R
    node = R.Binary('Mix', R.Binary('Gain', R.Osc('Sine', R.Fixed([[hz]])), R.Fixed([[2.0 * depth]])), R.Fixed(2.0 * base), R.Fixed([[0.5]]))
    return node, np.concatenate([R.render(node, q, 1, base.shape[1], RATE) for q in positions])


@pytest.mark.parametrize('N', [256, 64, 32])
def test_per_block_rows_fm_sweep_and_tremolo(N):
    """hertz, cutoff and gain driven at block rate: every node reads its controls at the position of the request that
    evaluates it (chain/__init__.py:305-306).  N >= 100: the block's own position, the rows in front of a launch the previous
    block's.  N < 100: the virtual block's controls are read at max(p - 100, 0); over the block itself, whatever feeds the LAST
    filter was evaluated by the oldest cached `after` request containing the block -- at q = p - m N, m = min((100 - N) / N,
    blocks rendered before - 1) -- and read its controls there; the last filter and what follows read theirs at p.  Two
    filters in series keep block-invariant oscillator controls when N < 100"""
    from oracle import chain_ref as R
    V = 48
    p = draw(V, 11)
    K1, K2 = 5, 4
    for types in (('lp',), ('lp', 'lp')):
        depth = len(types)
        fm = not (N < CTX and depth > 1)
        hz_node, _ = lfo_rows([0], p['hertz'], 4.5, 5.3)
        cut_node, _ = lfo_rows([0], p['cut1'], 150.0, 1.7)
        g_node, _ = lfo_rows([0], p['gain'], 0.1, 3.1)
        node = R.Osc('Sawtooth', hz_node if fm else R.Fixed(p['hertz']), R.Fixed(p['phase']))
        node = R.Filter(types[0], node, cut_node)
        if depth > 1:
            node = R.Filter(types[1], node, R.Fixed(p['cut2']))
        top = R.Binary('Gain', node, g_node)
        ref = R.render_stream(top, 0, N, K1 + K2, V)
        got = []
        for start, K, recent in ((0, K1, []), (K1 * N, K2, [b * N for b in range(K1)])):
            def positions(inner):
                """control positions of a port; `inner`: the port's node lies in front of the last filter"""
                own = [start + b * N for b in range(K)]
                if N >= CTX:
                    hist = hist_for(start, depth, recent)
                    front = [max((hist[0] if hist else start) - (N if recent else CTX), 0)]
                    return front + hist + own
                virtual = [max(q - CTX, 0) for q in own]
                if inner:
                    mmax = (CTX - N) // N
                    own = [q - max(min(mmax, len(recent) + b - 1), 0) * N for b, q in enumerate(own)]
                return virtual + own
            rows = lambda base, d, f, inner: dev(lfo_rows(positions(inner), base, d, f)[1])
            code = [('Osc', KINDS['Sawtooth'], 0, 0, 0)] + [('Filter', 0, i, 0, 0) for i in range(depth)] + [('Gain', 0, 0, 0, 0)]
            filters = [(rows(p['cut1'], 150.0, 1.7, depth > 1), types[0], 1)] + ([(dev(p['cut2']), types[1], 2)] if depth > 1 else [])
            oscs = [(rows(p['hertz'], 4.5, 5.3, True) if fm else dev(p['hertz']), dev(p['phase']))]
            got.append(launch(code, oscs, [rows(p['gain'], 0.1, 3.1, False)], filters, 0, depth, start, N, K, V, recent=recent,
                              rows=len(positions(False))))
        got = np.concatenate(got)
        assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), (N, types)
