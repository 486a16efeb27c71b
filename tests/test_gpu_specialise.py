"""Specialised voice-program kernels (signals_amd/specialise.py, sig_voice_program_attach): voice_program.hip built for ONE
program must give what the interpreter gives -- the same source, the same arithmetic -- and both what the oracle gives."""
import numpy as np
import pytest
import torch

from helpers import RATE, f32, maxerr
import test_gpu_program_engine as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _device(tmp_path_factory):
    assert torch.cuda.is_available()
    from signals_amd import _native, runtime, specialise
    runtime.set_device('cuda:0')
    if specialise.hipcc() is None:
        pytest.skip('no hipcc on this machine: nothing to specialise with')
    yield
    torch.cuda.synchronize()
    specialise.forget()
    _native.voice_program_use_attached(True)
    _native.set_voice_program_tuning()


def names(timer):
    torch.cuda.synchronize()
    return set(timer.summary())


@pytest.mark.parametrize('which', E.SHAPES)
def test_specialised_kernel_equals_the_interpreter_and_the_oracle(which):
    """the shapes of tests/test_gpu_program_engine.py (Amp / RingMod / Mix behind filters, three filters in series, FM with a
    second oscillator, a swept cascade with a tremolo: fx.py:35-60, :85-121, osc.py:26-62), stored and under a stereo bus,
    continuing batches, two launch geometries"""
    from oracle import chain_ref as R
    from signals_amd import _native
    from signals_amd.chain import ext
    from signals_amd.engine import KernelTimer
    V, N, batches = 96, 256, (3, 1, 4)
    p = E.draw(V, 13)
    p['cut1'][0, :8] = np.linspace(25.0, 140.0, 8)
    node, ref_node = E.shapes(p, which)
    ref = R.render_stream(ref_node, 0, N, sum(batches), V)
    scale = max(1.0, np.nanmax(np.abs(ref)))
    for vpl in (1, 2):
        _native.set_voice_program_tuning(vpl, 2)
        node, _ = E.shapes(p, which)
        _native.voice_program_use_attached(False)                             # (attached kernels are process-wide)
        plain = E.render_batches(node, V, 0, N, batches, fuse_program='always')
        _native.voice_program_use_attached(True)
        node, _ = E.shapes(p, which)
        timer = KernelTimer()
        got = E.render_batches(node, V, 0, N, batches, timer, fuse_program='always', specialise=True)
        assert any(n.startswith('voice_program[') and n.endswith('*specialised') for n in names(timer)), names(timer)
        assert maxerr(got, f32(ref)) < 1e-6 * scale, (which, vpl)
        assert maxerr(got, plain) <= 2.5e-7 * scale, (which, vpl)             # (the same operations; the compiler may order a sum differently)
        # under a bus
        node, _ = E.shapes(p, which)
        bus = ext.SumBus(); bus.input = node; bus.get_state().gains = np.ascontiguousarray(p['pan'])
        _native.voice_program_use_attached(False)
        plain = E.render_batches(bus, 2, 0, N, batches, fuse_program='always')
        _native.voice_program_use_attached(True)
        node, _ = E.shapes(p, which)
        bus = ext.SumBus(); bus.input = node; bus.get_state().gains = np.ascontiguousarray(p['pan'])
        timer = KernelTimer()
        got = E.render_batches(bus, 2, 0, N, batches, timer, fuse_program='always', specialise=True)
        assert any(n.startswith('voice_program_bus[') and n.endswith('*specialised') for n in names(timer)), names(timer)
        if which != 'amp_after_filter':
            want = ref @ p['pan'].T
            assert maxerr(got, f32(want)) < 1e-6 * max(1.0, np.abs(want).max()), (which, vpl)
        ok = np.isfinite(plain) & np.isfinite(got)
        assert np.array_equal(np.isfinite(plain), np.isfinite(got))
        if ok.any():                                                          # (Amp's NaN voices poison every row of a bus)
            assert np.abs(got[ok].astype(np.float64) - plain[ok]).max() <= 2.5e-7 * max(1.0, np.abs(plain[ok]).max()), (which, vpl)
    _native.set_voice_program_tuning()


@pytest.mark.parametrize('N', [32, 64, 100])
def test_specialised_kernel_on_blocks_no_longer_than_the_context(N):
    """the four-step form for blocks shorter than the filter context (chain/__init__.py:266-303) is part of the same source"""
    from oracle import chain_ref as R
    from signals_amd.engine import KernelTimer
    V = 64
    p = E.draw(V, 17)
    p['cut1'][0, :6] = np.linspace(30.0, 150.0, 6)
    batches = (3, 2, 4)
    node, ref_node = E.shapes(p, 'modulated_cascade')
    ref = R.render_stream(ref_node, 0, N, sum(batches), V)
    timer = KernelTimer()
    got = E.render_batches(node, V, 0, N, batches, timer, fuse_program='always', specialise=True)
    assert any(n.endswith('*specialised') for n in names(timer)), names(timer)
    assert maxerr(got, f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max()), N


@pytest.mark.parametrize('which', ['envelope_left', 'envelope_right', 'noise_gate'])
def test_specialised_kernel_with_the_extended_handlers(which):
    """programs with an ADSR or White instruction need the extended handlers compiled in (SIG_VP_S_EXT): RingMod of an envelope
    and a filtered oscillator, either way round, against the eager pull path from position 0 and on a fresh graph mid-stream;
    White (noise.py:22-23, unseeded in the reference: statistics only) against the interpreter"""
    from helpers import fix, mkosc, stream
    from signals_amd import _native
    from signals_amd.chain import ext, fx, noise
    from signals_amd.engine import BatchRenderer, KernelTimer
    V = 8
    rng = np.random.default_rng(5)
    row = lambda lo, hi, wide=True: rng.uniform(lo, hi, (1, V if wide else 1))
    P = dict(hz=row(55, 1760), cut=row(300, 6000), attack=row(0.001, 0.008), decay=row(0.002, 0.008), sustain=row(0.2, 0.9),
             release=row(0.002, 0.01), gate_on=row(0.0, 0.006), gate_off=row(0.012, 0.03, wide=False), g=row(0.2, 1.0))

    def build():
        o = mkosc('Sawtooth', P['hz'])
        f = fx.HighPass(); f.input = o; f.cutoff = fix(P['cut'])
        if which == 'noise_gate':
            w = noise.White(); w.get_state().channels = V
            m = fx.RingMod(); m.left = f; m.right = w
            g = fx.Gain(); g.left = m; g.right = fix(P['g'])
            return g
        env = ext.ADSR()
        for name in ('attack', 'decay', 'sustain', 'release', 'gate_on', 'gate_off'):
            setattr(env, name, fix(P[name]))
        n = fx.RingMod()
        n.left, n.right = (env, f) if which == 'envelope_left' else (f, env)
        return n
    for pos, N, K in ((0, 128, 3), (1000, 256, 2), (0, 64, 4)):
        for vpl in (1, 2):
            _native.set_voice_program_tuning(vpl, 1)
            _native.voice_program_use_attached(False)
            plain = BatchRenderer(build(), V, RATE, fuse_program='always').render(pos, N, K).cpu().numpy()
            _native.voice_program_use_attached(True)
            timer = KernelTimer()
            got = BatchRenderer(build(), V, RATE, fuse_program='always', specialise=True, timer=timer).render(pos, N, K).cpu().numpy()
            assert any(n.endswith('*specialised') for n in names(timer)), names(timer)
            if which == 'noise_gate':
                per_node = BatchRenderer(build(), V, RATE, fuse_program=False).render(pos, N, K).cpu().numpy()
                assert maxerr(got, plain) < 1e-6 and maxerr(got, per_node) < 1e-6     # one counter-based generator (noise.hip) everywhere: the same samples
                assert np.abs(got).max() > 1e-3
            else:
                want = stream(build(), pos, N, K, V)
                assert maxerr(plain, want) < 1e-6, (which, pos, N, vpl)
                assert maxerr(got, want) < 1e-6, (which, pos, N, vpl)
    _native.set_voice_program_tuning()


def test_specialised_control_program_gives_the_interpreters_bits():
    """control_program.hip built for one program's structure (registers in VGPRs instead of an LDS file behind an interpretive
    loop): a subgraph with every waveform, Gain / Mix / RingMod / Amp, one-column and V-wide rows, a shared sub-expression --
    several ports at once, K = 1 and K = 37, mid-stream positions, the block in front of a batch -- bit for bit what
    sig_control_program writes (forward_at_block_rate, chain/__init__.py:305-306; osc.py:26-62; fx.py:35-60)"""
    from helpers import fix, mkosc
    from signals_amd import SignalFlags
    from signals_amd.chain import Receiver, fx, port
    from signals_amd.engine import BatchRenderer, KernelTimer, _Batch
    rng = np.random.default_rng(5)
    V = 300
    wide, wide2 = rng.uniform(0.2, 2.0, (1, V)), rng.uniform(-1.0, 1.0, (1, V))
    lfo = mkosc('Sine', [[1.3]])
    tri = mkosc('Triangle', rng.uniform(0.5, 9.0, (1, V)), rng.uniform(0, 1, (1, V)))
    sq = mkosc('Square', [[0.7]], [[0.1]])
    saw = mkosc('Sawtooth', [[2.9]])
    scaled = fx.Gain(); scaled.left = lfo; scaled.right = fix([[0.4]])
    offset = fx.Mix(); offset.left = scaled; offset.right = fix(wide); offset.mix = fix([[0.25]])
    prod = fx.RingMod(); prod.left = offset; prod.right = tri
    amp = fx.Amp(); amp.left = prod; amp.right = fix([[1.5]])
    shared = fx.Mix(); shared.left = scaled; shared.right = sq; shared.mix = fix(np.abs(wide2))
    hole = fx.RingMod(); hole.left = saw

    class Ports(Receiver):
        a = port('a'); b = port('b'); c = port('c'); d = port('d')
        HOST_ARRAYS = False

        @classmethod
        def flags(cls):
            return SignalFlags(0)
    host = Ports()
    host.a, host.b, host.c, host.d = amp, shared, hole, lfo
    ports = [host.a, host.b, host.c, host.d]
    plain, timer = BatchRenderer(lfo, 1, RATE), KernelTimer()
    special = BatchRenderer(lfo, 1, RATE, specialise=True, timer=timer)
    for pos, N, K in ((0, 256, 37), (48000 * 3 + 17, 128, 1), (999, 64, 5)):
        for front in (-1, max(pos - 100, 0)):
            want = _Batch(plain, pos, N, K, False)._control_many(ports, front)
            got = _Batch(special, pos, N, K, False)._control_many(ports, front)
            torch.cuda.synchronize()
            for group_w, group_g in (((want, got),) if front < 0 else zip(want, got)):
                for name, x, y in zip('abcd', group_w, group_g):
                    assert x.shape == y.shape and np.array_equal(x.cpu().numpy(), y.cpu().numpy(), equal_nan=True), (name, pos, N, K, front)
    assert any(n == 'control_program[block-rate]*specialised' for n in names(timer)), names(timer)


def test_background_specialisation_never_blocks_a_render():
    """specialise='background': the first renders run the interpreter while a worker thread builds the kernel; once it is
    attached the same renderer's launches use it -- same stream of blocks either way (dev.py:167-179: a sink cannot wait)"""
    import time
    from oracle import chain_ref as R
    from signals_amd import specialise
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N = 64, 256
    p = E.draw(V, 23)
    torch.cuda.synchronize()
    specialise.forget()                                                       # (attached kernels are process-wide: start from none)
    node, ref_node = E.shapes(p, 'mix_after_filter')
    ref = R.render_stream(ref_node, 0, N, 6, V)
    timer = KernelTimer()
    r = BatchRenderer(node, V, RATE, fuse_program='always', specialise='background', timer=timer)
    t0 = time.perf_counter()
    first = r.render(0, N, 3).cpu().numpy()
    took = time.perf_counter() - t0
    assert not any(n.endswith('*specialised') for n in names(timer)), (names(timer), took)     # the interpreter rendered these
    specialise.wait()
    timer.reset()
    second = r.render(3 * N, N, 3).cpu().numpy()
    assert any(n.endswith('*specialised') for n in names(timer)), names(timer)
    assert maxerr(np.concatenate([first, second]), f32(ref)) < 1e-6 * max(1.0, np.abs(ref).max())


def test_switching_attached_kernels_off_and_refusing_a_foreign_image():
    from signals_amd import _native, specialise
    code_a = [('Osc', 2, 0, 0, 0), ('Gain', 0, 0, 0, 0)]
    code_b = [('Osc', 3, 0, 0, 0), ('Gain', 0, 0, 0, 0)]
    image_a = specialise.build(code_a, 1, 1, 0, 0, 1, 0)
    with pytest.raises(_native.NativeError):                                  # built for another program: hipErrorInvalidImage
        _native.voice_program_attach(code_b, 1, 1, 0, 0, 1, 0, image_a)
    with pytest.raises(_native.NativeError):                                  # ... another geometry
        _native.voice_program_attach(code_a, 1, 1, 0, 0, 2, 0, image_a)
    with pytest.raises(_native.NativeError):                                  # not a code object at all
        _native.voice_program_attach(code_a, 1, 1, 0, 0, 1, 0, b'\x7fELF' + bytes(4096))
    assert specialise.ensure(code_a, 1, 1, 0, 0, 1, 0)
    V, N, K = 64, 256, 3
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')
    rng = np.random.default_rng(3)
    hz, ph, g = rng.uniform(55, 1760, (1, V)), rng.uniform(0, 1, (1, V)), rng.uniform(0.1, 1, (1, V))
    _native.set_voice_program_tuning(1, 1)

    def run():
        out = torch.full((N * K, V), float('nan'), device='cuda')
        _native.voice_program(code_a, [(dev(hz), dev(ph))], [dev(g)], [], 0, 0, RATE, 4800, N, K, 100, V, 1 + K, [], out)
        return out.cpu().numpy()
    special = run()
    _native.voice_program_use_attached(False)
    interpreted = run()
    _native.voice_program_use_attached(True)
    from oracle import chain_ref as R
    ref = R.gain(R.osc('Sawtooth', 4800, N * K, RATE, hz, ph), g)
    assert np.array_equal(special, interpreted)                               # Sawtooth and a product: the same bits either way
    assert np.array_equal(special, f32(ref))
