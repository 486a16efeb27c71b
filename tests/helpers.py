"""Graph-building helpers for the GPU parity tests: the same topologies gen_golden.py drove
through the reference, built through signals_amd's node API."""
import numpy as np

from signals_amd import SignalFlags
from signals_amd.chain import BlockLoc, Receiver, Shape, port
from signals_amd.chain import fixed, fx, osc, shape

RATE = 48000
HOUR = 172_800_000
OSC = {'Sine': osc.Sine, 'Square': osc.Square, 'Sawtooth': osc.Sawtooth, 'Triangle': osc.Triangle}


class Probe(Receiver):
    input = port('input')
    HOST_ARRAYS = False           # written against signals_amd: takes device tensors

    @classmethod
    def flags(cls):
        return SignalFlags(0)


def fix(value) -> fixed.Fixed:
    f = fixed.Fixed()
    f.get_state().value = np.array(value, ndmin=2)
    return f


def loc(position, frames, channels, rate=RATE):
    return BlockLoc(position=position, rate=rate, shape=Shape(frames=frames, channels=channels))


def render(node, position, frames, channels):
    p = Probe()
    p.input = node
    out = p.input.request(loc(position, frames, channels))
    del p.input
    return out.cpu().numpy()


def stream(node, position, frames, blocks, channels):
    p = Probe()
    p.input = node
    return np.concatenate([p.input.request(loc(position + b * frames, frames, channels)).cpu().numpy()
                           for b in range(blocks)])


def mkosc(kind, hertz, phase=None):
    o = OSC[kind]()
    o.hertz = fix(hertz)
    if phase is not None:
        o.phase = fix(phase)
    return o


def f32(a):
    """the reference's float64 output as the float32 the GPU path stores"""
    return np.asarray(a, dtype=np.float64).astype(np.float32)


def maxerr(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), 'NaN pattern differs'
    if nan_g.all():
        return 0.0
    return float(np.max(np.abs(got[~nan_g] - ref[~nan_g])))
