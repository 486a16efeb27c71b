"""Host-side planning logic of the batched engine and the block cache, on CPU (no kernels)."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import chain_ref as R
from signals_amd import SignalFlags
from signals_amd.chain import BlockCachingEmitter, BlockLoc, ExplicitChannelsEmitter, Receiver, Request, Shape, port
from signals_amd.chain import ext, fixed, fx, osc
from signals_amd.engine import _audio_ports, _control_ports, _ctl_const, _is_pure, _modulated


@pytest.fixture(autouse=True)
def _cpu_device():
    from signals_amd import runtime
    old = runtime._device
    runtime.set_device('cpu')
    yield
    runtime._device = old


def fix(v):
    f = fixed.Fixed()
    f.get_state().value = np.array(v, ndmin=2, dtype=float)
    return f


def test_purity_and_modulation_classification():
    o = osc.Sine(); o.hertz = fix([[440.0]])
    assert _ctl_const(o.hertz) and _ctl_const(o.phase) and not _modulated(o) and _is_pure(o, {})
    g = fx.Gain(); g.left = o; g.right = fix([[0.5]])
    assert _is_pure(g, {}) and _control_ports(g) == [g.right] and _audio_ports(g) == [g.left]
    lp = fx.LowPass(); lp.input = g; lp.cutoff = fix([[1000.0]])
    assert not _is_pure(lp, {}) and not _modulated(lp)            # a filter is request-dependent by itself
    bus = ext.SumBus(); bus.input = lp
    assert not _is_pure(bus, {})                                   # ... and so is everything downstream of it
    lfo = osc.Sine(); lfo.hertz = fix([[2.0]])
    fm = osc.Sawtooth(); fm.hertz = lfo                            # block-rate FM: hertz re-read every block
    assert _modulated(fm) and not _is_pure(fm, {}) and not _ctl_const(fm.hertz)
    trem = fx.Gain(); trem.left = o; trem.right = lfo
    assert _modulated(trem) and not _is_pure(trem, {})
    lfo.get_state().enabled = False                                # a disabled emitter answers zeros((1,1)): constant
    assert _ctl_const(fm.hertz) and not _modulated(fm) and _is_pure(fm, {})
    rm = fx.RingMod(); rm.left = o; rm.right = g
    assert _is_pure(rm, {}) and _control_ports(rm) == [] and len(_audio_ports(rm)) == 2
    env = ext.ADSR()
    assert len(_control_ports(env)) == 6 and _audio_ports(env) == []


class Ramp(BlockCachingEmitter, ExplicitChannelsEmitter):
    def __init__(self):
        super().__init__()
        self.evals = 0

    @classmethod
    def flags(cls):
        return SignalFlags.GENERATOR

    def _eval(self, request: Request) -> torch.Tensor:
        self.evals += 1
        n = torch.from_numpy(request.loc.frame_range.astype(np.float64))
        return (n * 1000 + self.evals).expand(-1, request.loc.shape.channels).clone()     # value encodes WHICH evaluation made it


class OracleRamp(R.Node):
    def __init__(self, channels):
        super().__init__()
        self.channels, self.evals = channels, 0

    def eval(self, position, frames, channels, rate):
        self.evals += 1
        return np.broadcast_to(R.frame_range(position, frames) * 1000.0 + self.evals, (frames, channels)).copy()


class Probe(Receiver):
    input = port('input')
    HOST_ARRAYS = False

    @classmethod
    def flags(cls):
        return SignalFlags(0)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 40), st.integers(1, 12), st.integers(1, 3)), min_size=1, max_size=40))
def test_block_cache_matches_the_oracle_cache_on_random_request_sequences(requests):
    """exact hits, containment slices (first containing block in insertion order), FIFO eviction at 16 entries:
    the served VALUES encode which evaluation produced them, so any divergence in cache policy shows"""
    node = Ramp(); node.get_state().channels = 3
    p = Probe(); p.input = node
    ref = OracleRamp(3)
    for position, frames, channels in requests:
        got = p.input.request(BlockLoc(position=position * 4, rate=48000, shape=Shape(frames * 4, channels)))
        want = ref.respond(position * 4, frames * 4, channels, 48000)
        assert tuple(got.shape) == want.shape and np.array_equal(got.numpy(), want)
    assert node.evals == ref.evals and len(node._block_cache) == len(ref._cache) <= 16
