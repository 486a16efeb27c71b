"""The fused span walker (sig_fused_osc_biquad / sig_fused_voice_bus, signals_amd/csrc/fused_voice.hip) under every
launch geometry, called through the C ABI and checked against the CPU oracle: voices per lane x blocks per lane,
ragged voice counts, batches that are not a multiple of the span, short first contexts, block sizes at and below
the context length, the Sine recurrence and its two fall-backs to the exact phase (positions beyond 2^26 cycles,
voices that advance more than a quarter turn per row), all four waveforms, both filter types, both sinks."""
import os

import numpy as np
import pytest
import torch

from helpers import RATE, f32, maxerr

pytestmark = pytest.mark.gpu
CTX = 100


@pytest.fixture(scope='module', autouse=True)
def _device():
    assert torch.cuda.is_available()
    from signals_amd import runtime
    runtime.set_device('cuda:0')
    yield
    from signals_amd import _native
    _native.set_fused_tuning()                               # back to the launch heuristics


def geometry(vpt, span, steady=1):
    """force the launch geometry (sig_fused_set_tuning): voices per lane, blocks per lane, the serial walker rather than
    the latency-mode scan kernel, and whether Sine + bus uses the closed-form kernel for the waves that qualify"""
    from signals_amd import _native
    _native.set_fused_tuning(vpt, span, steady, 0)


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')


def bar(ref, cap=None):
    """BASELINE's bar for a bus: 1e-6 of its full scale (never below 1: a float32 bus of |x| < 1 is compared absolutely),
    and never looser than `cap` where an absolute tolerance was already met"""
    tol = 1e-6 * max(1.0, float(np.abs(ref).max()))
    return tol if cap is None else min(tol, cap)


def params(V, seed, hz_hi=1760.0):
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, np.pi / 2, V)
    return dict(hertz=rng.uniform(55, hz_hi, (1, V)), phase=rng.uniform(0, 1, (1, V)),
                cutoff=rng.uniform(200, 8000, (1, V)), gain=rng.uniform(0.1, 1, (1, V)),
                pan=np.stack([np.cos(th), np.sin(th)]))


def oracle_chain(kind, btype, p, pos, N, K):
    """float64 (K*N, V): gain * Filter(Osc), every block cold-started c rows early (fx.py:85-121)"""
    from oracle import chain_ref as R
    src = lambda q, n: R.osc(kind, q, n, RATE, p['hertz'], p['phase'])
    return np.concatenate([R.gain(R.filter_block(btype, src, pos + b * N, N, RATE, p['cutoff']), p['gain'])
                           for b in range(K)])


def run_chain(kind, btype, p, pos, N, K, gain=True):
    from signals_amd import _native
    V = p['hertz'].shape[1]
    out = torch.full((K * N, V), float('nan'), device='cuda')
    _native.fused_osc_biquad(kind, btype, RATE, pos, N, K, CTX, dev(p['hertz']), dev(p['phase']), dev(p['cutoff']),
                             dev(p['gain']) if gain else None, out)
    return out.cpu().numpy()


def run_bus(kind, btype, p, pos, N, K, C=2):
    from signals_amd import _native
    V = p['hertz'].shape[1]
    out = torch.full((K * N, C), float('nan'), device='cuda')
    _native.fused_voice_bus(kind, btype, RATE, pos, N, K, CTX, V, dev(p['hertz']), dev(p['phase']), dev(p['cutoff']),
                            dev(p['gain']), dev(p['pan'][:C]) if C > 1 else None, out)
    return out.cpu().numpy()


GEOMETRIES = [(1, 1), (2, 1), (4, 1), (4, 2), (2, 4), (1, 8), (4, 8), (2, 3), (8, 2)]   # 8 per lane: closed-form kernel only


@pytest.mark.parametrize('kind', ['Sine', 'Sawtooth', 'Square', 'Triangle'])
def test_every_geometry_matches_the_oracle(kind):
    """V = 200 (ragged for every voices-per-lane), K = 5 (ragged for every span), first block with a 37-row context"""
    V, N, K, pos = 200, 256, 5, 37
    p = params(V, 3)
    btype = 'hp' if kind == 'Square' else 'lp'
    ref = oracle_chain(kind, btype, p, pos, N, K)
    ref_bus = ref @ p['pan'].T
    # Square / Sawtooth jump by 2: one f64 rounding of t at a discontinuity moves a sample by 2 -- the fused path
    # reproduces the reference's t operation for operation on these waveforms, so it does not happen
    for vpt, span in GEOMETRIES:
        geometry(vpt, span)
        got = run_chain(kind, btype, p, pos, N, K)
        assert np.isfinite(got).all(), (vpt, span)
        assert maxerr(got, f32(ref)) < 2e-7, (kind, vpt, span)
        bus = run_bus(kind, btype, p, pos, N, K)
        assert maxerr(bus, f32(ref_bus)) < bar(ref_bus, 2e-6), (kind, vpt, span)          # sums of 200 voices of O(1)


def test_geometries_agree_with_each_other_to_rounding():
    """the same chains whatever the geometry: the Sine recurrence is seeded per span, so agreement is to 1e-12
    of the f64 result, i.e. at most one float32 ulp after the store"""
    V, N, K, pos = 256, 256, 8, 0
    p = params(V, 4)
    geometry(1, 1)
    base = run_chain('Sine', 'lp', p, pos, N, K)
    for vpt, span in GEOMETRIES[1:]:
        geometry(vpt, span)
        assert maxerr(run_chain('Sine', 'lp', p, pos, N, K), base) < 1.2e-7, (vpt, span)
    geometry(1, 1)
    base = run_chain('Sawtooth', 'lp', p, pos, N, K)
    for vpt, span in GEOMETRIES[1:]:
        geometry(vpt, span)
        assert np.array_equal(run_chain('Sawtooth', 'lp', p, pos, N, K), base), (vpt, span)   # exact phase: bit for bit


@pytest.mark.parametrize('N', [100, 64, 101, 356])
def test_block_sizes_around_the_context_length(N):
    """N == ctx: a block is all tail; N < ctx: the walker must not span blocks (falls back to one block per lane)"""
    V, K, pos = 64, 6, 0
    p = params(V, 5)
    ref = oracle_chain('Sine', 'lp', p, pos, N, K)
    for vpt, span in [(1, 1), (1, 4), (1, 8)]:
        geometry(vpt, span)
        assert maxerr(run_chain('Sine', 'lp', p, pos, N, K), f32(ref)) < 2e-7, (N, vpt, span)
        assert maxerr(run_bus('Sine', 'lp', p, pos, N, K), f32(ref @ p['pan'].T)) < 1e-6, (N, vpt, span)


def test_sine_falls_back_to_the_exact_phase():
    """(a) four hours into the stream a 6-9 kHz voice is past 2^26 cycles; (b) voices above rate/4 advance more than
    a quarter turn per row.  Both are wave-uniform fall-backs; mixed with ordinary voices in other waves."""
    V, N, K = 192, 256, 4
    p = params(V, 6)
    p['hertz'][0, :64] = np.random.default_rng(7).uniform(6000, 9000, 64)           # wave 0 (at vpt=1): large t at 4 h
    hour = 4 * 172_800_000
    ref = oracle_chain('Sine', 'lp', p, hour, N, K)
    for vpt, span in [(1, 1), (1, 4), (2, 2), (4, 4)]:
        geometry(vpt, span)
        assert maxerr(run_chain('Sine', 'lp', p, hour, N, K), f32(ref)) < 3e-7, (vpt, span)
    p = params(V, 8)
    p['hertz'][0, 70:75] = [12500.0, 15000.0, 23000.0, -14000.0, 30000.0]           # beyond rate/4, beyond Nyquist
    ref = oracle_chain('Sine', 'hp', p, 512, N, K)
    for vpt, span in [(1, 1), (1, 4), (2, 2), (4, 4)]:
        geometry(vpt, span)
        assert maxerr(run_chain('Sine', 'hp', p, 512, N, K), f32(ref)) < 3e-7, (vpt, span)
        assert maxerr(run_bus('Sine', 'hp', p, 512, N, K), f32(ref @ p['pan'].T)) < bar(ref @ p['pan'].T, 2e-6), (vpt, span)


def test_low_and_negative_frequencies_and_long_spans():
    """the difference-form recurrence has no 1/theta error growth: 0.05 Hz .. 20 Hz voices, negative hertz, DC,
    over a span of 8 blocks of 1024 rows"""
    V, N, K = 64, 1024, 8
    p = params(V, 9)
    p['hertz'][0, :] = np.concatenate([np.geomspace(0.05, 20, 40), -np.geomspace(0.05, 2000, 20), [0.0, 0.0, 1e-9, 11999.0]])
    ref = oracle_chain('Sine', 'lp', p, 0, N, K)
    geometry(1, 8)
    got = run_chain('Sine', 'lp', p, 0, N, K)
    assert maxerr(got, f32(ref)) < 2e-7


def test_mono_and_quad_bus_and_missing_gain():
    V, N, K = 130, 256, 3
    p = params(V, 10)
    rng = np.random.default_rng(11)
    p['pan'] = rng.uniform(-1, 1, (4, V))
    ref = oracle_chain('Triangle', 'lp', p, 0, N, K)
    geometry(2, 2)
    assert maxerr(run_bus('Triangle', 'lp', p, 0, N, K, C=4), f32(ref @ p['pan'].T)) < bar(ref @ p['pan'].T, 2e-6)
    assert maxerr(run_bus('Triangle', 'lp', p, 0, N, K, C=1), f32(ref.sum(axis=1, keepdims=True))) < bar(ref.sum(axis=1, keepdims=True), 2e-6)
    nogain = dict(p, gain=np.ones((1, V)))
    assert maxerr(run_chain('Triangle', 'lp', p, 0, N, K, gain=False), f32(oracle_chain('Triangle', 'lp', nogain, 0, N, K))) < 2e-7


def test_bad_cutoff_is_nan_and_flagged_in_every_geometry():
    from signals_amd import _native
    V, N, K = 128, 256, 4
    p = params(V, 12)
    p['cutoff'][0, 5] = 0.0
    p['cutoff'][0, 100] = 30000.0
    for vpt, span in [(1, 1), (4, 2)]:
        geometry(vpt, span)
        status = torch.zeros(1, dtype=torch.int32, device='cuda')
        out = torch.empty((K * N, V), device='cuda')
        _native.fused_osc_biquad('Sine', 'lp', RATE, 0, N, K, CTX, dev(p['hertz']), dev(p['phase']), dev(p['cutoff']), None,
                                 out, status=status)
        got = out.cpu().numpy()
        bad = np.isnan(got).all(axis=0)
        assert bad[5] and bad[100] and bad.sum() == 2
        assert int(status.item()) & _native.STATUS_BAD_CUTOFF


@pytest.mark.parametrize('btype', ['lp', 'hp'])
@pytest.mark.parametrize('pos', [0, 37, 100, 5000, 172_800_000 // 64])
def test_closed_form_sine_kernel_matches_the_walker_and_the_oracle(btype, pos):
    """sig_fused_voice_bus on Sine voices: steady-state sinusoid + homogeneous transient per block (no warm-up rows)
    against the row-by-row walker (SIG_FUSED_STEADY=0) and the oracle; first contexts of 0, 37 and 100 rows"""
    V, N, K = 200, 256, 5
    p = params(V, 20 + pos % 7)
    ref_bus = oracle_chain('Sine', btype, p, pos, N, K) @ p['pan'].T
    for vpt, span in GEOMETRIES:
        geometry(vpt, span, steady=0)
        walker = run_bus('Sine', btype, p, pos, N, K)
        geometry(vpt, span, steady=1)
        steady = run_bus('Sine', btype, p, pos, N, K)
        assert np.isfinite(steady).all()
        assert maxerr(steady, walker) < 5e-7, (vpt, span)                # sums of 200 voices, up to +-2: two float32 ulps
        assert maxerr(steady, f32(ref_bus)) < bar(ref_bus, 2e-6), (vpt, span)


def test_closed_form_kernel_leaves_unqualified_waves_to_the_walker():
    """per-wave choice: voices below ~8 Hz (sin(theta) < 1e-3), above rate/4, or past 2^26 cycles keep their whole
    wave on the walker; the bus is the sum of both kernels' partial tiles"""
    V, N, K = 256, 256, 6
    p = params(V, 30)
    p['hertz'][0, 3] = 2.0                                   # wave 0 at vpt=1: walker
    p['hertz'][0, 70] = 20000.0                              # wave 1: walker (exact phase)
    p['hertz'][0, 130] = 0.0                                 # wave 2: walker
    ref_bus = oracle_chain('Sine', 'lp', p, 512, N, K) @ p['pan'].T
    for vpt, span in [(1, 1), (1, 4), (2, 2), (4, 8), (8, 2), (8, 1)]:
        geometry(vpt, span, steady=1)
        assert maxerr(run_bus('Sine', 'lp', p, 512, N, K), f32(ref_bus)) < bar(ref_bus, 2e-6), (vpt, span)
    hour = 172_800_000
    ref_bus = oracle_chain('Sine', 'lp', p, hour, N, 2) @ p['pan'].T
    geometry(1, 2, steady=1)
    assert maxerr(run_bus('Sine', 'lp', p, hour, N, 2), f32(ref_bus)) < bar(ref_bus, 2e-6)


def test_closed_form_kernel_block_sizes_and_bus_widths():
    V = 64
    p = params(V, 31)
    p['pan'] = np.random.default_rng(32).uniform(-1, 1, (4, V))
    for N, K in [(100, 6), (64, 5), (1024, 3), (17, 9)]:
        ref = oracle_chain('Sine', 'lp', p, 0, N, K)
        for C in (1, 2, 4):
            want = f32(ref @ p['pan'][:C].T) if C > 1 else f32(ref.sum(axis=1, keepdims=True))
            for span in (1, 4):
                geometry(1, span, steady=1)
                assert maxerr(run_bus('Sine', 'lp', p, 0, N, K, C=C), want) < 1e-6, (N, K, C, span)


def test_closed_form_kernel_one_second_blocks_and_extreme_cutoffs():
    """N = 48000 (the homogeneous part decays to nothing long before the block ends; the steady recurrence runs
    96 000 rows from one seed) and cutoffs from 5 Hz to 23.9 kHz (poles next to the unit circle / next to -1)"""
    V, N, K = 64, 48000, 2
    p = params(V, 50)
    p['cutoff'][0, :] = np.geomspace(5.0, 23900.0, V)
    ref = oracle_chain('Sine', 'lp', p, 0, N, K)
    ref_bus = ref @ p['pan'].T
    for steady in (1, 0):
        geometry(1, 2, steady=steady)
        got = run_bus('Sine', 'lp', p, 0, N, K)
        assert np.isfinite(got).all()
        assert maxerr(got, f32(ref_bus)) < 1e-6, steady
    ref = oracle_chain('Sine', 'hp', p, 4800, 256, 4) @ p['pan'].T
    geometry(2, 2, steady=1)
    assert maxerr(run_bus('Sine', 'hp', p, 4800, 256, 4), f32(ref)) < 1e-6


@pytest.mark.parametrize('kind', ['Sine', 'Sawtooth'])
def test_mix_matrix_sink_equals_mix_matrix_over_the_stored_chain(kind):
    """sig_fused_osc_biquad_mix against sig_mix_matrix(sig_fused_osc_biquad) over the same float32 rows, for spans, ragged
    batch ends (N*K not a multiple of the 32-row MFMA tile) and short first contexts.  Both contract exact products of the
    float32 rows and the float32 matrix in float32 accumulators -- the per-node kernel with v_mfma_f32_32x32x2_f32, the
    fused sink with each float32 as three bfloat16 (sig_mix_tile.h) -- so they agree to the accumulation order: a few
    float32 ulps of the rows' scale, and each is within that of the f64 product of the same rows"""
    from signals_amd import _native
    rng = np.random.default_rng(60)
    M = torch.tensor(rng.standard_normal((64, 64)), dtype=torch.float32, device='cuda')
    for V, N, K, pos, span in [(128, 256, 5, 37, 1), (128, 256, 5, 0, 4), (64, 100, 3, 512, 2), (192, 50, 7, 0, 1), (64, 17, 5, 3, 8)]:
        p = params(V, 61 + V + N)
        geometry(1, span, steady=0)                          # the row walker feeds the sink: the same float32 rows as the stored chain
        chain = torch.tensor(run_chain(kind, 'lp', p, pos, N, K), device='cuda')
        want = _native.mix_matrix(chain, M, torch.empty_like(chain)).cpu().numpy()
        exact = (chain.double().reshape(K * N, V // 64, 64) @ M.double()).reshape(K * N, V).cpu().numpy()
        got = torch.full((K * N, V), float('nan'), device='cuda')
        _native.fused_osc_biquad_mix(kind, 'lp', RATE, pos, N, K, CTX, dev(p['hertz']), dev(p['phase']), dev(p['cutoff']),
                                     dev(p['gain']), M, got)
        got = got.cpu().numpy()
        ulp = float(np.spacing(np.float32(np.abs(exact).max())))
        # measured (ulps of the mixed rows' scale): sink 2.3-4.8, per-node kernel 2.8-5.3, between them 4-7 -- 64-term float32
        # accumulations either way, the sink's slightly closer (16 exact products enter its accumulator at a time)
        assert np.isfinite(got).all() and maxerr(got, want) < 10 * ulp, (V, N, K, pos, span)
        assert maxerr(got, exact) < 6 * ulp and maxerr(want, exact) < 7 * ulp, (V, N, K, pos, span)
        assert maxerr(got, exact) <= maxerr(want, exact) + ulp, (V, N, K, pos, span)
        if kind == 'Sine':
            # by default a Sine chain reaches the sink through the closed form (fused_steady_mix_kernel): same values to
            # the rounding of the rows, and within 1e-6 (of the mixed rows' scale) of the oracle's chain times the matrix
            ref = oracle_chain(kind, 'lp', p, pos, N, K)
            ref = (ref.reshape(K * N, V // 64, 64) @ M.cpu().numpy().astype(np.float64)).reshape(K * N, V)
            scale = float(np.abs(want).max())
            for steady in (1, 3):                            # 3: the same kernel with its sink on the float32 MFMA (tuning hook)
                geometry(1, span, steady=steady)
                closed = torch.full((K * N, V), float('nan'), device='cuda')
                _native.fused_osc_biquad_mix(kind, 'lp', RATE, pos, N, K, CTX, dev(p['hertz']), dev(p['phase']), dev(p['cutoff']),
                                             dev(p['gain']), M, closed)
                closed = closed.cpu().numpy()
                assert np.isfinite(closed).all() and maxerr(closed, want) < 1e-6 * scale, (V, N, K, pos, span, steady)
                assert maxerr(closed, f32(ref)) < 1e-6 * max(1.0, scale), (V, N, K, pos, span, steady)


def test_tile_sum_inside_the_kernel_equals_the_second_launch_bit_for_bit():
    """the closed-form bus kernel adds its voice tiles itself when they are 1, 2 or 4 (the waves of a span group then sit
    in one workgroup; sig_bus::sum_tiles_in_workgroup) and leaves them to partials_kernel otherwise: the same additions
    in the same order, so the same bits -- for every tile count, batch lengths that are not a multiple of the span
    (workgroups with idle waves), mono and stereo, short first contexts"""
    from signals_amd import _native
    try:
        for V, vpt, tiles in ((64, 1, 1), (100, 1, 2), (256, 1, 4), (1024, 8, 2), (520, 2, 5), (192, 1, 3)):
            assert -(-V // (64 * vpt)) == tiles
            for N, K, span, pos, C in ((256, 7, 4, 0, 2), (128, 5, 2, 37, 1), (256, 9, 8, 4096, 2), (64, 3, 1, 100, 1)):
                p = params(V, 70 + V + N)
                for kind in ('Sine', 'Sawtooth'):            # the closed form; the row walker (its voices per lane stop at 4)
                    geometry(vpt, span, steady=2)            # tiles added by partials_kernel
                    two = run_bus(kind, 'lp', p, pos, N, K, C=C)
                    geometry(vpt, span, steady=1)            # ... by the kernel itself where it can
                    one = run_bus(kind, 'lp', p, pos, N, K, C=C)
                    assert np.isfinite(one).all() and np.array_equal(one, two), (kind, V, vpt, N, K, span, pos, C)
        ref = oracle_chain(kind, 'lp', p, pos, N, K).sum(axis=1, keepdims=True)        # (the last case: mono, Sawtooth)
        assert C == 1 and maxerr(one, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max())
    finally:
        geometry(0, 0, steady=-1)


def test_closed_form_vs_walker_over_random_parameter_draws():
    """40 random draws of (V, N, K, position, filter type, bus width, geometry) with log-uniform oscillator
    frequencies (8 Hz .. 11.9 kHz, either sign) and cutoffs (20 Hz .. 23 kHz): the closed form and the row-by-row
    walker agree to a few float32 ulps of the bus; every fifth draw is also checked against the oracle"""
    rng = np.random.default_rng(2026)
    for draw in range(40):
        V = int(rng.choice([64, 100, 256, 320]))
        N = int(rng.choice([100, 128, 256, 300, 1024]))
        K = int(rng.integers(1, 7))
        pos = int(rng.choice([0, 1, 99, 100, 4096, 48000 * 600]))
        btype = str(rng.choice(['lp', 'hp']))
        C = int(rng.choice([1, 2, 4]))
        vpt, span = int(rng.choice([1, 2, 4, 8, 16])), int(rng.choice([1, 2, 4, 8]))     # (16: the steady kernel only; the walker clamps to 4)
        p = dict(hertz=np.exp(rng.uniform(np.log(8.0), np.log(11900.0), (1, V))) * rng.choice([-1.0, 1.0], (1, V)),
                 phase=rng.uniform(-2, 2, (1, V)), cutoff=np.exp(rng.uniform(np.log(20.0), np.log(23000.0), (1, V))),
                 gain=rng.uniform(0.1, 1, (1, V)) / np.sqrt(V), pan=rng.uniform(-1, 1, (4, V)))
        geometry(vpt, span, steady=0)
        walker = run_bus('Sine', btype, p, pos, N, K, C=C)
        geometry(vpt, span, steady=1)
        steady = run_bus('Sine', btype, p, pos, N, K, C=C)
        scale = max(1.0, float(np.abs(walker).max()))
        assert np.isfinite(steady).all(), draw
        assert maxerr(steady, walker) < 4e-7 * scale, (draw, V, N, K, pos, btype, C, vpt, span)
        if draw % 5 == 0:
            ref = oracle_chain('Sine', btype, p, pos, N, K)
            want = ref @ p['pan'][:C].T if C > 1 else ref.sum(axis=1, keepdims=True)
            assert maxerr(steady, f32(want)) < 1e-6 * scale, (draw, V, N, K, pos, btype, C)


def run_latency(btype, p, pos, N, C=2, device_pos=False, ws=None):
    from signals_amd import _native
    V = p['hertz'].shape[1]
    out = torch.full((N, C), float('nan'), device='cuda')
    ws = ws if ws is not None else _native.latency_voice_bus_workspace(V, N, C, 'cuda')
    position = torch.tensor([pos], dtype=torch.int64, device='cuda') if device_pos else pos
    _native.latency_voice_bus(btype, RATE, position, N, CTX, V, dev(p['hertz']), dev(p['phase']), dev(p['cutoff']),
                              dev(p['gain']), dev(p['pan'][:C]) if C > 1 else None, out, ws)
    if device_pos:
        assert int(position.item()) == pos + N                    # the launch advanced the device-side position
    return out.cpu().numpy()


@pytest.mark.parametrize('btype', ['lp', 'hp'])
def test_one_launch_latency_kernel_matches_the_oracle(btype):
    """sig_latency_voice_bus: one block per launch, 16-row chunks seeded in closed form, tiles added by the last
    workgroup; ragged voice counts and block lengths, short first contexts, one hour in, every bus width, voices
    the closed form does not cover (walked the plain way by their lane), the position read from and advanced in
    device memory, the same workspace reused launch after launch"""
    from signals_amd import _native
    for V, N, pos, C in [(200, 256, 0, 2), (200, 256, 37, 2), (64, 100, 5000, 1), (1024, 256, 172_800_000, 2), (130, 250, 100, 4),
                         (320, 17, 3, 2), (96, 1024, 48000, 1)]:
        p = params(V, 70 + V + N)
        if C == 4:
            p['pan'] = np.random.default_rng(71).uniform(-1, 1, (4, V))
        if V >= 200:
            p['hertz'][0, 3], p['hertz'][0, 70], p['hertz'][0, 140] = 2.0, 20000.0, 0.0       # not covered by the closed form
        ref = oracle_chain('Sine', btype, p, pos, N, 1)
        want = f32(ref @ p['pan'][:C].T) if C > 1 else f32(ref.sum(axis=1, keepdims=True))
        ws = _native.latency_voice_bus_workspace(V, N, C, 'cuda')
        for device_pos in (False, True, False):                       # three launches on one workspace: the counter re-arms
            got = run_latency(btype, p, pos, N, C=C, device_pos=device_pos, ws=ws)
            assert np.isfinite(got).all()
            assert maxerr(got, want) < 1e-6 * max(1.0, float(np.abs(want).max())), (V, N, pos, C, device_pos)


def test_engine_latency_mode_uses_one_launch_and_follows_a_stream():
    """BatchRenderer with one block per render on a Sine chain: sig_latency_voice_bus (plain and under hipGraph replay,
    seeks included) against the batched engine"""
    import bench
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N = 1024, 256
    prm = bench.synth_params(V)
    want = BatchRenderer(bench.build_graph(prm, 0, V), 2, RATE).render(0, N, 12).cpu().numpy()
    timer = KernelTimer()
    plain = BatchRenderer(bench.build_graph(prm, 0, V), 2, RATE, timer=timer)
    got = np.concatenate([plain.render(i * N, N, 1).cpu().numpy() for i in range(12)])
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'latency_voice_bus[Sine,lp,gain]'}, set(timer.summary())
    assert maxerr(got, want) < 4e-9
    replay = BatchRenderer(bench.build_graph(prm, 0, V), 2, RATE, graph_replay=True)
    got = np.concatenate([replay.render(i * N, N, 1).cpu().numpy().copy() for i in range(12)])
    assert maxerr(got, want) < 4e-9
    assert maxerr(replay.render(5 * N, N, 1).cpu().numpy(), want[5 * N:6 * N]) < 4e-9          # seek back
    old = BatchRenderer(bench.build_graph(prm, 0, V), 2, RATE)
    old.latency_kernel = False                                          # the scan kernel + sum_bus path stays available
    assert maxerr(np.concatenate([old.render(i * N, N, 1).cpu().numpy() for i in range(3)]), want[:3 * N]) < 4e-9
