"""The fused cascade (sig_fused_cascade_bus, signals_amd/csrc/fused_cascade.hip): Osc -> Filter -> Filter [-> x ADSR]
-> SumBus in one launch, against the CPU oracle driven like the reference (sequential pulls, block caches: the outer
filter's context is the inner filter's PREVIOUS block), the reference's own golden cascades, the
per-node engine schedule and the eager path; continuing streams, fresh starts mid-stream, every launch geometry."""
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


def params(V, seed):
    rng = np.random.default_rng(seed)
    env = dict(attack=rng.uniform(0.001, 0.05, (1, V)), decay=rng.uniform(0.01, 0.2, (1, V)), sustain=rng.uniform(0.2, 0.9, (1, V)),
               release=rng.uniform(0.02, 0.3, (1, V)), gate_on=rng.uniform(0.0, 0.05, (1, V)), gate_off=rng.uniform(0.06, 0.12, (1, V)))
    th = rng.uniform(0, np.pi / 2, V)
    return dict(hertz=rng.uniform(55, 1760, (1, V)), phase=rng.uniform(0, 1, (1, V)), cut1=rng.uniform(200, 8000, (1, V)),
                cut2=rng.uniform(200, 8000, (1, V)), gain=rng.uniform(0.2, 1.0, (1, V)), env=env, pan=np.stack([np.cos(th), np.sin(th)]))


def graph(p, kind='Sawtooth', t1='LowPass', t2='LowPass', env=True, gain=False, pan=None):
    from signals_amd.chain import ext, fx
    f1 = getattr(fx, t1)(); f1.input = mkosc(kind, p['hertz'], p['phase']); f1.cutoff = fix(p['cut1'])
    f2 = getattr(fx, t2)(); f2.input = f1; f2.cutoff = fix(p['cut2'])
    top = f2
    if env:
        a = ext.ADSR()
        for k, v in p['env'].items():
            setattr(a, k, fix(v))
        rm = fx.RingMod(); rm.left = f2; rm.right = a
        top = rm
    if gain:
        g = fx.Gain(); g.left = top; g.right = fix(p['gain'])
        top = g
    bus = ext.SumBus(); bus.input = top
    if pan is not None:
        bus.get_state().gains = np.ascontiguousarray(pan)
    return bus


def oracle(p, kind='Sawtooth', t1='lp', t2='lp', env=True, gain=False, pan=None):
    from oracle import chain_ref as R
    node = R.Filter(t2, R.Filter(t1, R.Osc(kind, R.Fixed(p['hertz']), R.Fixed(p['phase'])), R.Fixed(p['cut1'])), R.Fixed(p['cut2']))
    if env:
        node = R.Binary('RingMod', node, R.Adsr(**p['env']))
    if gain:
        node = R.Binary('Gain', node, R.Fixed(p['gain']))
    return node, pan


def oracle_stream(p, pos, N, K, V, **kw):
    from oracle import chain_ref as R
    node, pan = oracle(p, **kw)
    return R.sum_bus(R.render_stream(node, pos, N, K, V), pan)


def fused(node, channels, timer=None, **kw):
    from signals_amd.engine import BatchRenderer
    return BatchRenderer(node, channels, RATE, timer=timer, **kw)


def test_c3_graph_is_one_launch_and_matches_the_oracle():
    from signals_amd.engine import KernelTimer
    V, N, K = 48, 1024, 5
    p = params(V, 3)
    timer = KernelTimer()
    r = fused(graph(p), 1, timer)
    got = r.render(0, N, K).cpu().numpy()
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'fused_cascade_bus[Sawtooth,lp,lp,env]'}, set(timer.summary())
    ref = oracle_stream(p, 0, N, K, V)
    scale = np.abs(ref).max()
    assert scale > 1.0 and maxerr(got, f32(ref)) < 1e-6 * scale
    # the per-node schedule and the eager pull path (bit-identical to each other) agree to their own float32 roundings
    from signals_amd.engine import BatchRenderer
    plain = BatchRenderer(graph(p), 1, RATE, fuse=False).render(0, N, K).cpu().numpy()
    assert maxerr(plain, f32(ref)) < 1e-6 * scale                    # (each schedule against the oracle, not against each other)


def test_continuing_stream_equals_one_long_batch_and_the_oracle():
    """batches of 3 + 2 + 4 blocks == the oracle's 9 sequential blocks: block 0 of a later batch takes its outer context
    from the inner filter's previous block (cold-started N + 100 rows earlier), not from a fresh cold start"""
    V, N = 32, 256
    p = params(V, 4)
    p['cut1'][0, :8] = np.linspace(20.0, 120.0, 8)                   # slow filters: the history really matters
    ref = oracle_stream(p, 0, N, 9, V)
    r = fused(graph(p), 1)
    got = np.concatenate([r.render(0, N, 3).cpu().numpy(), r.render(3 * N, N, 2).cpu().numpy(), r.render(5 * N, N, 4).cpu().numpy()])
    scale = np.abs(ref).max()
    assert maxerr(got, f32(ref)) < 1e-6 * scale
    whole = fused(graph(p), 1).render(0, N, 9).cpu().numpy()
    assert maxerr(whole, got) < 1e-6 * scale
    # a FRESH renderer started mid-stream answers what a fresh reference graph answers (history block = [p - 100, p))
    from oracle import chain_ref as R
    node, _ = oracle(p)
    fresh_ref = R.sum_bus(R.render_stream(node, 5 * N, N, 2, V))
    fresh = fused(graph(p), 1).render(5 * N, N, 2).cpu().numpy()
    assert maxerr(fresh, f32(fresh_ref)) < 1e-6 * scale
    assert maxerr(fresh, got[5 * N:7 * N]) > 1e-5 * scale            # ... which is NOT what the continuing stream rendered


@pytest.mark.parametrize('kind,t1,t2', [('Sawtooth', 'LowPass', 'HighPass'), ('Square', 'HighPass', 'LowPass'),
                                        ('Triangle', 'LowPass', 'LowPass'), ('Sine', 'HighPass', 'HighPass')])
def test_waveforms_filter_types_and_bus_widths(kind, t1, t2):
    V, N, K = 40, 256, 4
    p = params(V, 7)
    short = {'LowPass': 'lp', 'HighPass': 'hp'}
    pan4 = np.random.default_rng(8).uniform(-1, 1, (4, V))
    for pan in (None, p['pan'], pan4):
        C = 1 if pan is None else pan.shape[0]
        for env, gain in ((True, True), (False, False)):
            got = fused(graph(p, kind, t1, t2, env=env, gain=gain, pan=pan), C).render(0, N, K).cpu().numpy()
            ref = oracle_stream(p, 0, N, K, V, kind=kind, t1=short[t1], t2=short[t2], env=env, gain=gain, pan=pan)
            scale = max(1.0, np.abs(ref).max())
            assert got.shape == (N * K, C) and maxerr(got, f32(ref)) < 1e-6 * scale, (kind, C, env)


def test_golden_cascades_of_the_reference(golden):
    """the reference's own 2 x LowPass cascade rendered sequentially (tests/golden/cascade.npz, generated by importing the
    reference): per-voice outputs are not exposed by the bus kernel, so the check is on the voice sum"""
    from signals_amd.chain import ext, fx
    g = golden('cascade')

    def build(t2):
        f1 = fx.LowPass(); f1.input = mkosc('Sawtooth', g['casc/hertz'], g['casc/phase']); f1.cutoff = fix(g['casc/cut1'])
        f2 = getattr(fx, t2)(); f2.input = f1; f2.cutoff = fix(g['casc/cut2'])
        bus = ext.SumBus(); bus.input = f2
        return bus
    for N, key in ((256, 'casc/seq_n256'), (1024, 'casc/seq_n1024')):
        ref = g[key]
        K = ref.shape[0] // N
        got = fused(build('LowPass'), 1).render(0, N, K).cpu().numpy()
        want = ref.sum(axis=1, keepdims=True)
        assert maxerr(got, f32(want)) < 1e-6 * max(1.0, np.abs(want).max()), key
    # a fresh graph asked for a late block: the inner filter's history block is [p - 100, p), cold-started at p - 200
    want = g['casc/fresh_p768'].sum(axis=1, keepdims=True)
    got = fused(build('HighPass'), 1).render(768, 256, 1).cpu().numpy()
    assert maxerr(got, f32(want)) < 1e-6 * max(1.0, np.abs(want).max())


def test_every_launch_geometry():
    """voices per lane 1, 2, 4 x spans of 1 .. 5 blocks (forced through the tuning hook: at test sizes the heuristic
    always spreads the voices thin), ragged voice counts, batch lengths that are not a multiple of the span, mono /
    stereo, continuing streams"""
    from signals_amd import _native
    try:
        for V, N, K in ((70, 256, 7), (130, 128, 11)):
            p = params(V, 20 + V)
            ref = oracle_stream(p, 0, N, K + 3, V, pan=p['pan'])
            for vpt in (1, 2, 4):
                for span in (1, 2, 3, 5, 16):
                    _native.set_fused_cascade_tuning(vpt, span)
                    assert _native.fused_cascade_geometry(V, K) == (vpt, span)
                    r = fused(graph(p, pan=p['pan']), 2)
                    got = np.concatenate([r.render(0, N, K).cpu().numpy(), r.render(K * N, N, 3).cpu().numpy()])
                    assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), (V, N, K, vpt, span)
    finally:
        _native.set_fused_cascade_tuning()
    assert _native.fused_cascade_geometry(1024, 1024) == (4, 4) and _native.fused_cascade_geometry(1024, 256) == (2, 2)
    assert _native.fused_cascade_geometry(1024, 4096) == (4, 16) and _native.fused_cascade_geometry(1024, 64) == (1, 1)
    assert _native.fused_cascade_geometry(5, 3) == (1, 1)


def test_tile_sum_inside_the_kernel_equals_the_second_launch_bit_for_bit():
    """1, 2 or 4 voice tiles: the cascade kernel adds them itself (sig_bus::sum_tiles_in_workgroup); a negative span through
    the tuning hook keeps the second launch -- same additions in the same order, same bits, also when the batch is not a
    multiple of the span and when the last workgroup has idle waves"""
    from signals_amd import _native
    try:
        for V, vpt in ((40, 1), (200, 2), (200, 1), (520, 4), (700, 4)):      # 1, 2, 4, 3 (second launch either way), 3 tiles
            p = params(V, 30 + V)
            for N, K, span, C in ((256, 5, 2, 2), (128, 7, 4, 1)):
                outs = []
                for sign in (-1, 1):
                    _native.set_fused_cascade_tuning(vpt, sign * span)
                    r = fused(graph(p, pan=p['pan'] if C == 2 else None), C)
                    outs.append(np.concatenate([r.render(0, N, K).cpu().numpy(), r.render(K * N, N, 2).cpu().numpy()]))
                assert np.isfinite(outs[1]).all() and np.array_equal(outs[0], outs[1]), (V, vpt, N, K, span, C)
    finally:
        _native.set_fused_cascade_tuning()


def test_other_context_lengths_through_the_c_abi(monkeypatch):
    """the reference's context is 100 frames (fx.py:82-83); the entry point takes it as an argument, and the kernel's
    restarts (running chain minus A^ctx times its state ctx rows ago) must hold for any: 1, 37, 255 of a 256-frame block
    against the oracle with its constant patched; 0 (no context: every block from zero state) against the same voices
    rendered one block per launch"""
    from oracle import chain_ref as R
    from signals_amd import _native
    V, N, K = 70, 256, 6
    p = params(V, 9)
    p['cut1'][0, :6] = np.linspace(30.0, 150.0, 6)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    hz, ph, c1, c2 = up(p['hertz']), up(p['phase']), up(p['cut1']), up(p['cut2'])
    env = {k: up(v) for k, v in p['env'].items()}

    def launch(ctx, pos, k, history):
        out = torch.full((k * N, 1), float('nan'), device='cuda')
        _native.fused_cascade_bus('Sawtooth', 'lp', 'lp', RATE, pos, history, N, k, ctx, V, hz, ph, c1, c2, None, env, None, out)
        return out.cpu().numpy()

    try:
        for ctx in (1, 37, 255):
            monkeypatch.setattr(R, 'CONTEXT_FRAMES', ctx)
            node = R.Binary('RingMod', R.Filter('lp', R.Filter('lp', R.Osc('Sawtooth', R.Fixed(p['hertz']), R.Fixed(p['phase'])),
                                                               R.Fixed(p['cut1']), ctx=ctx), R.Fixed(p['cut2']), ctx=ctx), R.Adsr(**p['env']))
            ref = R.sum_bus(R.render_stream(node, 0, N, K, V), None)
            for vpt, span in ((0, 0), (2, 3), (4, 1)):
                _native.set_fused_cascade_tuning(vpt, span)
                got = np.concatenate([launch(ctx, 0, 4, 0), launch(ctx, 4 * N, K - 4, 3 * N)])
                assert maxerr(got, f32(ref)) < 1e-6 * np.abs(ref).max(), (ctx, vpt, span)
        _native.set_fused_cascade_tuning(2, 4)
        whole = launch(0, 0, K, 0)
        _native.set_fused_cascade_tuning(1, 1)
        single = np.concatenate([launch(0, b * N, 1, b * N) if b == 0 else launch(0, b * N, 1, (b - 1) * N) for b in range(K)])
        assert np.isfinite(whole).all() and maxerr(whole, single) < 1e-6 * np.abs(single).max()
    finally:
        _native.set_fused_cascade_tuning()


def test_bad_cutoff_in_either_filter_is_reported():
    from signals_amd import runtime
    V, N = 16, 256
    p = params(V, 9)
    p['cut2'][0, 3] = 30000.0                                          # Wn >= 1: scipy raises ValueError (fx.py:99-121)
    r = fused(graph(p), 1)                                             # (the renderer owns the device status words)
    r.render(0, N, 2)
    with pytest.raises(ValueError):
        runtime.check_status()
    p['cut2'][0, 3], p['cut1'][0, 5] = 3000.0, -1.0
    r2 = fused(graph(p), 1)
    r2.render(0, N, 2)
    with pytest.raises(ValueError):
        runtime.check_status()


def test_block_sizes_the_cascade_kernel_does_not_take_fall_back_to_the_older_schedule():
    """N must be a whole number of row groups (16 / bus channels) and > 100: otherwise the voice runs as one interpreted
    launch (sig_voice_program), or -- without it -- the two-launch schedule of round 1 (fused Saw + LowPass, then filter +
    envelope + bus); the oracle's stream either way"""
    from signals_amd.engine import KernelTimer
    V, K = 24, 3
    p = params(V, 31)
    for N in (1000, 250):
        ref = oracle_stream(p, 0, N, K, V)
        for program, prefix in (('always', 'voice_program_bus'), (True, 'biquad_bus'), (False, 'biquad_bus')):    # (an envelope: the interpreter's full register file, not chosen by default)
            timer = KernelTimer()
            got = fused(graph(p), 1, timer, fuse_program=program).render(0, N, K).cpu().numpy()
            torch.cuda.synchronize()
            names = set(timer.summary())
            assert not any(n.startswith('fused_cascade_bus') for n in names) and any(n.startswith(prefix) for n in names), names
            assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), (N, program)


def test_a_stream_may_switch_between_the_per_node_schedule_and_the_cascade_kernel():
    """the cascade kernel re-walks the previous block itself, the per-node schedule keeps rounded float32 tails: a batch
    that continues a per-node batch stays per-node (its history is those tails), a fresh stream or a stream of cascade
    batches uses the kernel; every combination renders the oracle's sequential stream"""
    from signals_amd.engine import KernelTimer
    V, N = 24, 256
    p = params(V, 41)
    ref = oracle_stream(p, 0, N, 6, V)
    scale = max(1.0, np.abs(ref).max())
    timer = KernelTimer()
    r = fused(graph(p), 1, timer, fuse_program=False)
    r.fuse_cascade = False
    a = r.render(0, N, 2).cpu().numpy()                       # per-node: fused Saw + LowPass, then filter + envelope + bus
    r.fuse_cascade = True
    b = r.render(2 * N, N, 2).cpu().numpy()                   # continues the per-node batch: tails
    torch.cuda.synchronize()
    assert not any(n.startswith('fused_cascade_bus') for n in timer.summary())
    r.reset()
    c = r.render(4 * N, N, 2).cpu().numpy()                   # a FRESH start mid-stream: the cascade kernel, fresh-graph history
    torch.cuda.synchronize()
    assert any(n.startswith('fused_cascade_bus') for n in timer.summary())
    assert maxerr(np.concatenate([a, b]), f32(ref[:4 * N])) < 1e-6 * scale
    from oracle import chain_ref as R
    node, _ = oracle(p)
    assert maxerr(c, f32(R.sum_bus(R.render_stream(node, 4 * N, N, 2, V)))) < 1e-6 * scale


def test_a_cascade_batch_followed_by_a_batch_the_kernel_does_not_take():
    """the cascade kernel leaves no tails; when the NEXT contiguous batch needs the per-node schedule (a block size that is
    not a whole number of row groups, `fuse_cascade` switched off) the engine first re-renders the previous block per node
    -- the inner filter cold-started where the reference cold-started the block it keeps cached -- so the outer filter's
    context is that block's last 100 rows, not a fresh block cold-started at p - 200.  Slow inner filters: the difference
    between the two is far above the bar."""
    from signals_amd.engine import KernelTimer
    V, N = 24, 256
    p = params(V, 43)
    p['cut1'][0, :8] = np.linspace(20.0, 120.0, 8)
    from oracle import chain_ref as R
    node, _ = oracle(p)
    ref = R.sum_bus(np.concatenate([R.render_stream(node, 0, N, 3, V), R.render_stream(node, 3 * N, 250, 2, V)]))
    scale = max(1.0, np.abs(ref).max())
    timer = KernelTimer()
    r = fused(graph(p), 1, timer, fuse_program=False)
    a = r.render(0, N, 3).cpu().numpy()                       # the cascade kernel
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'fused_cascade_bus[Sawtooth,lp,lp,env]'}
    timer.reset()
    b = r.render(3 * N, 250, 2).cpu().numpy()                 # 250 frames: not a whole number of 16-row groups -> per node
    torch.cuda.synchronize()
    assert not any(n.startswith(('fused_cascade_bus', 'voice_program')) for n in timer.summary())
    assert maxerr(np.concatenate([a, b]), f32(ref)) < 1e-6 * scale
    fresh = fused(graph(p), 1).render(3 * N, 250, 2).cpu().numpy()
    assert maxerr(fresh, b) > 1e-5 * scale                    # ... which a fresh start at 3 N does NOT render
    # by default the 250-frame batch is one interpreted launch, which re-walks the previous block itself: the same stream
    rd = fused(graph(p), 1)
    d = np.concatenate([rd.render(0, N, 3).cpu().numpy(), rd.render(3 * N, 250, 2).cpu().numpy()])
    assert maxerr(d, f32(ref)) < 1e-6 * scale
    # the same with the kernel switched off mid-stream, and back on: every batch the oracle's sequential stream
    ref2 = R.sum_bus(R.render_stream(oracle(p)[0], 0, N, 7, V))
    r2 = fused(graph(p), 1, fuse_program=False)
    parts = [r2.render(0, N, 2)]
    r2.fuse_cascade = False
    parts.append(r2.render(2 * N, N, 2))
    r2.fuse_cascade = True
    parts.append(r2.render(4 * N, N, 3))                      # continues per-node tails: stays per node
    got = torch.cat(parts).cpu().numpy()
    assert maxerr(got, f32(ref2)) < 1e-6 * max(1.0, np.abs(ref2).max())
