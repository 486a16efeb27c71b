"""Plugin interop on the GPU (reference chain/__init__.py:245-247: every `_eval` takes and returns numpy arrays): a
generator and an effect written against the REFERENCE's API, inside a graph whose filter runs as a HIP kernel -- eager
pull and batched engine -- against the CPU oracle of the same graph."""
import numpy as np
import pytest
import torch

from helpers import RATE, Probe, f32, fix, loc, maxerr, stream

pytestmark = pytest.mark.gpu
V, N = 24, 256


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    assert torch.cuda.is_available()
    from signals_amd import _native, runtime
    runtime.set_device('cuda:0')
    _native.lib()


def pulse(position, frames, channels):
    """a band-limited-ish pulse train per channel, float64 numpy -- what a reference plugin would compute"""
    t = np.arange(position, position + frames, dtype=np.int64).reshape(-1, 1) / RATE
    f = 110.0 * np.arange(1, channels + 1).reshape(1, -1)
    return np.tanh(4.0 * np.sin(2 * np.pi * f * t)) * (1.0 + 0.25 * np.cos(2 * np.pi * 3.0 * t))


def make_nodes():
    from signals_amd import SignalFlags
    from signals_amd.chain import ExplicitChannelsEmitter, ImplicitChannels, port

    class NumpyPulse(ExplicitChannelsEmitter):
        @classmethod
        def flags(cls):
            return SignalFlags.GENERATOR

        def _eval(self, request):
            return pulse(request.loc.position, request.loc.shape.frames, request.loc.shape.channels)

    class NumpyClip(ImplicitChannels):
        input = port('input')
        HOST_ARRAYS = True

        @classmethod
        def flags(cls):
            return SignalFlags.EFFECT

        def _eval(self, request):
            x = self.input.forward(request)
            assert isinstance(x, np.ndarray) and x.dtype == np.float64
            return np.clip(1.5 * x, -1.0, 1.0)
    return NumpyPulse, NumpyClip


def graph(cutoff):
    from signals_amd.chain import fx
    NumpyPulse, NumpyClip = make_nodes()
    src = NumpyPulse(); src.get_state().channels = V
    lp = fx.LowPass(); lp.input = src; lp.cutoff = fix(cutoff)
    clip = NumpyClip(); clip.input = lp
    return src, lp, clip


def test_numpy_generator_feeds_a_gpu_filter_and_a_numpy_effect():
    from oracle import chain_ref as R
    cutoff = np.random.default_rng(5).uniform(300, 6000, (1, V))
    src, lp, clip = graph(cutoff)
    K = 3
    got_lp = stream(lp, 0, N, K, V)
    ref_lp = np.concatenate([R.filter_block('lp', lambda p, n: pulse(p, n, V).astype(np.float32).astype(np.float64),
                                            b * N, N, RATE, cutoff) for b in range(K)])
    assert maxerr(got_lp, f32(ref_lp)) < 3e-7                 # the filter sees the float32 upload of the plugin's reply
    src, lp, clip = graph(cutoff)
    got = stream(clip, 0, N, K, V)
    assert maxerr(got, f32(np.clip(1.5 * f32(ref_lp).astype(np.float64), -1.0, 1.0))) < 3e-7
    # a (N, 1) numpy reply broadcasts at the port like any reply; a wrong shape raises BadShape
    from signals_amd.chain import BadShape
    p = Probe(); p.input = src
    assert tuple(p.input.request(loc(0, N, 1)).shape) == (N, 1)
    src._eval = lambda request: np.zeros((3, 2))
    with pytest.raises(BadShape):
        p.input.request(loc(0, N, V))


def test_batched_engine_schedules_around_plugin_nodes():
    """the engine pulls a plugin node block by block and batches everything downstream: same bits as the eager path"""
    from signals_amd.engine import BatchRenderer, KernelTimer
    cutoff = np.random.default_rng(6).uniform(300, 6000, (1, V))
    K = 5
    src, lp, clip = graph(cutoff)
    eager = stream(lp, 0, N, K, V)
    src, lp, clip = graph(cutoff)
    timer = KernelTimer()
    r = BatchRenderer(lp, V, RATE, timer=timer)
    batched = np.concatenate([r.render(0, N, 3).cpu().numpy(), r.render(3 * N, N, 2).cpu().numpy()])   # a continued stream
    torch.cuda.synchronize()
    assert {k.split('[')[0] for k in timer.summary()} == {'biquad_coldstart'} and len(timer.records) == 2
    assert np.array_equal(batched, eager)
    # a plugin effect DOWNSTREAM of kernels: its input arrives through its own (eager) port requests
    src, lp, clip = graph(cutoff)
    eager_clip = stream(clip, 0, N, K, V)
    src, lp, clip = graph(cutoff)
    assert np.array_equal(BatchRenderer(clip, V, RATE).render(0, N, K).cpu().numpy(), eager_clip)


def test_sink_device_pulls_a_gpu_graph_block_by_block(golden):
    """signals.chain.dev.SinkDevice (= the headless BlockDriver): one fused launch per pulled 256-frame block, like the
    reference's callback (dev.py:167-179) would request them; equal to a batch render of the same graph to rounding"""
    from signals_amd.chain import dev, ext, fx
    from signals_amd.chain.driver import BlockDriver
    from helpers import mkosc
    g = golden('c2')

    def build():
        f = fx.LowPass(); f.input = mkosc('Sine', g['c2/hertz'], g['c2/phase']); f.cutoff = fix(g['c2/cutoff'])
        gn = fx.Gain(); gn.left = f; gn.right = fix(g['c2/gain'])
        bus = ext.SumBus(); bus.input = gn
        return bus
    sink = dev.SinkDevice(blocksize=256)
    sink.input = build()
    n = 12
    played = np.concatenate([sink.pull() for _ in range(n)])
    assert sink.tell() == n and sink.is_active
    batch = BlockDriver(rate=48000, blocksize=256); batch.input = build()
    assert np.abs(batch.render(n) - played).max() < 1e-7
    sink.destroy()
