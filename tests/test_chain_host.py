"""Host-side node API on CPU (no kernels): shapes, block addresses, ports, state, cache, errors --
the contract rows of SURVEY.md §8b, checked against the reference's documented behaviour
(doctests at chain/__init__.py:26-51, :67-83) and the golden BlockLoc table."""
import doctest
import re

import attr
import numpy as np
import pytest
import torch

import signals_amd
import signals_amd.chain as chain
from signals_amd import SignalFlags
from signals_amd.chain import BadShape, BadStateSchema, BadStateValue, BlockLoc, Request, Shape, port
from signals_amd.chain import ext, fixed, fx, noise, osc, shape


@pytest.fixture(autouse=True)
def _cpu_device():
    from signals_amd import runtime
    old = runtime._device
    runtime.set_device('cpu')
    yield
    runtime._device = old


class Probe(chain.Receiver):
    input = port('input')
    HOST_ARRAYS = False           # written against signals_amd: takes device tensors

    @classmethod
    def flags(cls):
        return SignalFlags(0)


class Ramp(chain.BlockCachingEmitter, chain.ExplicitChannelsEmitter):
    """test emitter: value = absolute frame index (+ channel/1000); counts evaluations"""

    def __init__(self):
        super().__init__()
        self.evals = []

    @classmethod
    def flags(cls):
        return SignalFlags.GENERATOR

    def _eval(self, request: Request) -> torch.Tensor:
        self.evals.append(request.loc)
        n = torch.from_numpy(request.loc.frame_range.astype(np.float64))
        return n + torch.arange(self.channels, dtype=torch.float64)[None, :] / 1000


def loc(position, frames, channels, rate=48000):
    return BlockLoc(position=position, rate=rate, shape=Shape(frames=frames, channels=channels))


def test_doctests():
    import signals_amd.chain.blocks as blocks
    res = doctest.testmod(blocks)
    assert res.failed == 0 and res.attempted >= 6


def test_shape_semantics_match_reference_doctests():
    s = Shape(frames=10, channels=2)
    assert s == (10, 2) and s <= (10, 2) and s >= (10, 2) and not (s == (1, 1))
    assert (1, 1) <= Shape(frames=10, channels=1) <= s
    assert (1, 1) <= Shape(frames=1, channels=2) <= s
    assert not ((0, 0) <= s) and not (Shape(frames=3, channels=2) <= s) and not (Shape(frames=10, channels=0) <= s)
    assert Shape.of_array(np.array([[1, 2, 3]])) == (1, 3) and Shape.of_array(np.array([[1], [2], [2]])) == (3, 1)
    with pytest.raises(TypeError):
        Shape.of_array(np.array([]))
    with pytest.raises(TypeError):
        Shape.of_array(np.array([[[]]]))
    assert Shape.unit() == (1, 1)


def test_blockloc_integer_rows_bit_exact(golden):
    for pos, n, bp, bf, ap, af, fr0, fr1, b_le, l_le, r_le in golden('blockloc')['blockloc/table']:
        l = loc(int(pos), int(n), 4)
        b, a = l.before(100), l.after(100)
        assert (b.position, b.shape.frames, a.position, a.shape.frames) == (bp, bf, ap, af)
        fr = l.frame_range
        assert fr.dtype == np.int64 and fr.shape == (n, 1) and fr[0, 0] == fr0 and fr[-1, 0] == fr1
        assert not fr.flags.writeable
        assert (int(b <= l), int(l <= l), int(l.resize(1) <= l)) == (b_le, l_le, r_le)
    l = loc(512, 256, 4)
    assert l.end_position == 768 and l.timestamp == 512 / 48000
    assert l.resize(256) is l and l.reslice(4) is l and l.reslice(2).shape == (256, 2)
    assert hash(l) == hash(loc(512, 256, 4)) and l != loc(512, 256, 4, rate=44100)
    assert not (loc(512, 256, 4) <= loc(512, 256, 4, rate=44100))
    with pytest.raises(attr.exceptions.FrozenInstanceError):
        l.position = 3


def test_ports_connect_disconnect_and_names():
    assert osc.Sine.port_names() == ['hertz', 'phase']
    assert fx.Mix.port_names() == ['left', 'mix', 'right']
    assert fx.Gain.port_names() == fx.RingMod.port_names() == fx.Amp.port_names() == ['left', 'right']
    assert fx.LowPass.port_names() == fx.HighPass.port_names() == ['cutoff', 'input']
    assert fx.BandPass.port_names() == fx.BandStop.port_names() == ['high', 'input', 'low']
    assert shape.Merge.port_names() == ['left', 'right'] and shape.Flatten.port_names() == ['input']
    s, f = osc.Sine(), fixed.Fixed()
    assert not s.hertz and s.hertz.channels is None
    s.hertz = f
    assert s.hertz and s.hertz.sig is f and ('hertz', s) in f.outputs_with_ports and s.inputs_by_port == {'hertz': f}
    g = fixed.Fixed()
    s.hertz = g                       # re-assign expels the old emitter
    assert not f.outputs_with_ports and ('hertz', s) in g.outputs_with_ports
    del s.hertz
    assert not s.hertz and not g.outputs_with_ports
    p = Probe(); p.input = s
    s.destroy()                       # emitter.destroy unplugs every consumer
    assert not p.input


def test_qualified_names_and_flags():
    assert osc.Sine.cls_name() == 'signals.chain.osc.Sine' and fx.LowPass.cls_name() == 'signals.chain.fx.LowPass'
    assert fixed.Fixed.cls_name() == 'signals.chain.fixed.Fixed' and noise.White.cls_name() == 'signals.chain.noise.White'
    assert osc.Sine.flags() == SignalFlags.GENERATOR and fx.Gain.flags() == SignalFlags.EFFECT
    assert SignalFlags.AUDIO == SignalFlags.GENERATOR | SignalFlags.EFFECT | SignalFlags.SOURCE_DEVICE
    from signals_amd.chain.driver import load_signal
    assert load_signal('signals.chain.osc.Triangle') is osc.Triangle
    assert load_signal('signals_amd.chain.ext.SumBus') is ext.SumBus
    with pytest.raises(TypeError):
        load_signal('signals.chain.osc.Osc')        # abstract
    signals_amd.install_as_signals(host_plugins=False)
    import signals.chain.fx
    assert signals.chain.fx.Gain is fx.Gain


def test_state_schema_and_validation():
    f = fixed.Fixed()
    assert f.state_attrs() == {'enabled', 'value'} and f.get_state().value.shape == (1, 1)
    f.get_state().value = np.array([[330]])
    assert f.channels == 1
    with pytest.raises(BadStateValue):
        f.get_state().value = np.array([1.0, 2.0])
    with pytest.raises(BadStateValue):
        f.get_state().value = [[1.0]]
    with pytest.raises(BadStateSchema):
        f.set_state(osc.Sine.State())
    with pytest.raises(TypeError):
        fixed.Fixed.State(enabled='yes')
    w = noise.White()
    assert w.state_attrs() == {'enabled', 'channels', 'seed'} and w.channels == 1
    with pytest.raises(ValueError):
        noise.White.State(channels=0)
    assert str(BadShape(f, (3, 2), (10, 2))).startswith("BadShape Invalid response from 'signals.chain.fixed.Fixed'")


def test_fixed_resident_copy_tracks_the_array():
    f = fixed.Fixed()
    v = np.array([[220]])                     # int64, like a .sigs value
    f.get_state().value = v
    t = f.resident()
    assert t.dtype == torch.float64 and t.shape == (1, 1) and t[0, 0] == 220.0
    assert f.resident() is t                  # no re-upload
    v[0, 0] = 440                             # in-place edit is seen, like the reference's shared array
    assert f.resident()[0, 0] == 440.0
    f.get_state().value = np.ones((5, 2))
    assert f.resident().dtype == torch.float32 and f.resident().shape == (5, 2)


def test_fixed_sees_in_place_edits_of_arrays_of_every_size():
    """the reference's Fixed._eval returns the live array (fixed.py:38-39): an in-place edit of ANY array is rendered
    at the next reply -- also beyond the 65 536 elements an earlier version stopped watching at, through strided views,
    and after an in-place reshape; NaNs do not force a re-upload per reply"""
    for n in (10, 1024, 70000, 1 << 18):
        f = fixed.Fixed()
        v = np.zeros((1, n))
        f.get_state().value = v
        t = f.resident()
        assert f.resident() is t
        v[0, n - 3] = 3.5
        assert f.resident() is not t and f.resident()[0, n - 3] == 3.5, n
        assert f.resident() is f.resident()
    base = np.zeros((4, 8))
    f = fixed.Fixed()
    f.get_state().value = base[:, ::2]            # a view: the edit goes through the base array
    f.resident()
    base[1, 2] = 7.0
    assert f.resident()[1, 1] == 7.0
    f = fixed.Fixed()
    f.get_state().value = np.full((1, 4), np.nan)
    assert f.resident() is f.resident()
    v = np.arange(8.0).reshape(1, 8)
    f.get_state().value = v
    assert f.resident().shape == (1, 8)
    v.shape = (2, 4)                               # same bytes, another layout
    assert f.resident().shape == (2, 4)


def test_unplugged_and_disabled_answer_unit_zero():
    p = Probe()
    z = p.input.request(loc(0, 256, 2))
    assert z.shape == (1, 1) and z.dtype == torch.float64 and z[0, 0] == 0
    r = Ramp(); r.get_state().channels = 2; r.get_state().enabled = False
    p.input = r
    assert p.input.request(loc(0, 256, 2)).shape == (1, 1) and r.evals == []


def test_bad_shape_is_raised_at_the_port():
    r = Ramp(); r.get_state().channels = 3
    p = Probe(); p.input = r
    with pytest.raises(BadShape):
        p.input.request(loc(0, 16, 2))
    assert p.input.request(loc(0, 16, 3)).shape == (16, 3)


def test_block_cache_exact_hit_containment_slice_and_fifo():
    r = Ramp(); r.get_state().channels = 2
    p = Probe(); p.input = r
    a = p.input.request(loc(0, 256, 2))
    assert p.input.request(loc(0, 256, 2)) is a and len(r.evals) == 1          # exact hit: same object
    s = p.input.request(loc(100, 50, 1))                                        # contained: sliced view
    assert len(r.evals) == 1 and s.shape == (50, 1) and s.data_ptr() == a[100:150, :1].data_ptr()
    assert s[0, 0] == 100.0
    p.input.request(loc(200, 100, 2))                                           # overlaps the end: miss
    assert len(r.evals) == 2
    for k in range(1, 17):                                                      # 16 more -> first key evicted
        p.input.request(loc(1000 * k, 8, 2))
    assert len(r._block_cache) == 16 and loc(0, 256, 2) not in r._block_cache
    p.input.request(loc(0, 256, 2))
    assert len(r.evals) == 19
    assert not (loc(0, 256, 2) <= loc(0, 256, 2, rate=44100))


def test_forward_with_context_request_pattern():
    """before / block / after, clamped at 0, like chain/__init__.py:308-315 + :149-159"""
    r = Ramp(); r.get_state().channels = 1
    p = Probe(); p.input = r
    w = p.input.forward_with_context(Request(requestor=p, port='input', loc=loc(0, 64, 1)), 100)
    assert [(l.position, l.shape.frames) for l in r.evals] == [(0, 64), (64, 100)] and w.shape == (164, 1)
    r2 = Ramp(); r2.get_state().channels = 1
    p.input = r2
    w = p.input.forward_with_context(Request(requestor=p, port='input', loc=loc(30, 64, 1)), 100)
    assert [(l.position, l.shape.frames) for l in r2.evals] == [(0, 30), (30, 64), (94, 100)]
    assert torch.equal(w[:, 0], torch.arange(0, 194, dtype=torch.float64))
    ctl = p.input.forward_at_block_rate(Request(requestor=p, port='input', loc=loc(500, 64, 1)))
    assert ctl.shape == (1, 1) and ctl[0, 0] == 500.0


def test_upstream_order_and_cycle_detection():
    o = osc.Sine(); g = fx.Gain(); lp = fx.LowPass(); m = shape.Merge()
    g.left = o; lp.input = g; m.left = lp; m.right = g
    up = list(m.upstream())
    assert up[-1] is m and up.index(o) < up.index(g) < up.index(lp)
    a, b = fx.Gain(), fx.Gain()
    a.left = b; b.left = a
    with pytest.raises(AssertionError, match='Cycle'):
        a.upstream()


def test_implicit_channels():
    g = fx.Gain()
    a = fixed.Fixed(); a.get_state().value = np.zeros((1, 8))
    b = fixed.Fixed()
    g.left = a; g.right = b
    assert g.channels == 8                                  # the one non-1 width
    c = fixed.Fixed(); c.get_state().value = np.zeros((1, 4))
    g.right = c
    with pytest.raises(ValueError):
        g.channels
    m = shape.Merge(); m.left = a; m.right = c
    assert m.channels == 12


def test_kernels_refuse_cpu_tensors_loudly():
    from signals_amd._native import NativeError
    s = osc.Sine()
    f = fixed.Fixed(); f.get_state().value = np.array([[440.0]])
    s.hertz = f
    p = Probe(); p.input = s
    with pytest.raises(NativeError, match='no CPU fallback'):
        p.input.request(loc(0, 64, 1))


def test_every_native_wrapper_refuses_cpu_tensors():
    """the product path has no CPU fallback: each ctypes wrapper raises before touching the library"""
    import torch
    from signals_amd import _native
    from signals_amd._native import NativeError
    row = torch.zeros((1, 64), dtype=torch.float64)
    audio = torch.zeros((256, 64), dtype=torch.float32)
    bus = torch.zeros((256, 2), dtype=torch.float32)
    env = {k: row for k in _native.ADSR_PARAMS}
    calls = {
        'osc_bank': lambda: _native.osc_bank('Sine', 0, 48000, row, row, audio),
        'biquad_coldstart': lambda: _native.biquad_coldstart('lp', 48000, 0, 256, 1, 100, row, audio, 0, audio.clone()),
        'biquad_coldstart_bus': lambda: _native.biquad_coldstart_bus('lp', 48000, 0, 256, 1, 100, row, audio, 0, None, bus[:, :1],
                                                                     envelope=env),
        'fused_osc_biquad': lambda: _native.fused_osc_biquad('Sine', 'lp', 48000, 0, 256, 1, 100, row, row, row, None, audio),
        'fused_voice_bus': lambda: _native.fused_voice_bus('Sine', 'lp', 48000, 0, 256, 1, 100, 64, row, row, row, row,
                                                           torch.zeros((2, 64), dtype=torch.float64), bus),
        'fused_osc_biquad_mix': lambda: _native.fused_osc_biquad_mix('Sine', 'lp', 48000, 0, 256, 1, 100, row, row, row, None,
                                                                     torch.zeros((64, 64)), audio),
        'sum_bus': lambda: _native.sum_bus(audio, None, bus[:, :1]),
        'adsr': lambda: _native.adsr(0, 48000, env, audio),
        'mix_matrix': lambda: _native.mix_matrix(audio, torch.zeros((64, 64)), audio.clone()),
    }
    for name, call in calls.items():
        with pytest.raises(NativeError, match='no CPU fallback'):
            call()


def test_bench_closed_form_predicate_mirrors_the_kernel():
    """bench.steady_applies (host) restates fused_voice.hip:steady_voice_ok; it decides which f64 operation count the
    roofline leg uses"""
    import bench
    p = bench.synth_params(1024)
    assert bench.steady_applies(p, 0, 1024, 0, 110 * 4096 * 256, 256)                 # the bench's own stream
    assert bench.steady_applies(p, 0, 1024, 172_800_000, 172_800_000 + 4096 * 256, 256)
    assert not bench.steady_applies(p, 0, 1024, 50, 50 + 64 * 17, 17)                  # N < ctx with a short first context
    slow = {k: v.copy() for k, v in p.items()}
    slow['hertz'][0, 5] = 2.0                                                          # |sin theta| < 1e-3
    assert not bench.steady_applies(slow, 0, 1024, 0, 1000, 256)
    assert bench.steady_applies(slow, 6, 1024, 0, 1000, 256)                           # ... in another shard
    fast = {k: v.copy() for k, v in p.items()}
    fast['hertz'][0, 5] = 13000.0                                                      # more than a quarter turn per row
    assert not bench.steady_applies(fast, 0, 1024, 0, 1000, 256)
    far = 2 ** 26 * 48000 // 55 + 48000                                                # 55 Hz voice past 2^26 cycles
    assert not bench.steady_applies(p, 0, 1024, far * 40, far * 40 + 1000, 256)
    assert bench.steady_applies(p, 0, 1024, 10 * 172_800_000, 10 * 172_800_000 + 1000, 256)   # ten hours in: 1760 Hz is at 2^25.9


def test_graph_version_bumps_on_mutation():
    v0 = chain.graph_clock.version
    s = osc.Sine(); f = fixed.Fixed()
    s.hertz = f
    assert chain.graph_clock.version > v0
    v1 = chain.graph_clock.version
    s.set_state(osc.Sine.State(enabled=False))
    assert chain.graph_clock.version > v1


class NumpyRamp(chain.ExplicitChannelsEmitter):
    """a generator written against the REFERENCE's API: `_eval` returns a float64 numpy array (chain/__init__.py:245-247)"""

    @classmethod
    def flags(cls):
        return SignalFlags.GENERATOR

    def _eval(self, request):
        loc = request.loc
        return loc.frame_range / loc.rate * np.arange(1, loc.shape.channels + 1).reshape(1, -1)


class NumpySquarer(chain.ImplicitChannels):
    """an effect written against the reference's API: numpy in, numpy out"""
    input = port('input')
    HOST_ARRAYS = True

    @classmethod
    def flags(cls):
        return SignalFlags.EFFECT

    def _eval(self, request):
        x = self.input.forward(request)
        assert isinstance(x, np.ndarray) and x.dtype == np.float64
        return np.square(x)


def test_numpy_plugin_nodes_interoperate_at_the_ports():
    """replies of reference-style nodes are uploaded at the port (float64 for one row, float32 audio otherwise); a
    receiver that asks for host arrays gets float64 numpy; ranks other than 2 raise TypeError like Shape.of_array"""
    src = NumpyRamp(); src.get_state().channels = 3
    p = Probe(); p.input = src
    got = p.input.request(chain.BlockLoc(position=48000, rate=48000, shape=chain.Shape(4, 3)))
    assert isinstance(got, torch.Tensor) and got.dtype == torch.float32 and tuple(got.shape) == (4, 3)
    want = (np.arange(48000, 48004).reshape(-1, 1) / 48000) * np.array([[1, 2, 3]])
    assert np.array_equal(got.numpy(), want.astype(np.float32))
    row = p.input.request(chain.BlockLoc(position=7, rate=48000, shape=chain.Shape(1, 3)))
    assert row.dtype == torch.float64 and np.array_equal(row.numpy(), (np.array([[7]]) / 48000) * np.array([[1, 2, 3]]))
    sq = NumpySquarer(); sq.input = src
    p2 = Probe(); p2.input = sq
    got2 = p2.input.request(chain.BlockLoc(position=48000, rate=48000, shape=chain.Shape(4, 3)))
    assert np.array_equal(got2.numpy(), np.square(want.astype(np.float32).astype(np.float64)).astype(np.float32))
    del sq.input
    assert np.array_equal(p2.input.request(chain.BlockLoc(position=0, rate=48000, shape=chain.Shape(4, 3))).numpy(),
                          np.zeros((1, 1)))                      # unplugged: zeros((1, 1)) as numpy, squared, uploaded

    class Flat(NumpyRamp):
        def _eval(self, request):
            return np.zeros(request.loc.shape.frames)              # 1-D: the reference raises TypeError (chain/__init__.py:84)
    bad = Flat(); p.input = bad
    with pytest.raises(TypeError):
        p.input.request(chain.BlockLoc(position=0, rate=48000, shape=chain.Shape(4, 1)))


def test_numpy_plugin_filter_reads_its_input_with_context():
    """a reference-style effect that pulls `forward_with_context` (the CritFilter pattern, fx.py:93-94) gets ONE float64 numpy
    window [<=100 before | block | 100 after] (chain/__init__.py:308-315)"""

    class NumpyDiff(chain.ImplicitChannels):
        input = port('input')
        HOST_ARRAYS = True

        @classmethod
        def flags(cls):
            return SignalFlags.EFFECT

        def _eval(self, request):
            w = self.input.forward_with_context(request, 100)
            assert isinstance(w, np.ndarray) and w.dtype == np.float64
            n, c = request.loc.shape.frames, min(100, request.loc.position)
            assert w.shape[0] == c + n + 100
            return np.diff(w, axis=0, prepend=0.0)[c:c + n]

    src = NumpyRamp(); src.get_state().channels = 2
    fx_ = NumpyDiff(); fx_.input = src
    p = Probe(); p.input = fx_
    for pos in (0, 37, 4800):
        got = p.input.request(chain.BlockLoc(position=pos, rate=48000, shape=chain.Shape(8, 2))).numpy()
        ramp = lambda a, b: (np.arange(a, b).reshape(-1, 1) / 48000) * np.array([[1, 2]])
        lo = pos - min(100, pos)
        window = np.concatenate([ramp(lo, pos), ramp(pos, pos + 8), ramp(pos + 8, pos + 108)]).astype(np.float32).astype(np.float64)
        want = np.diff(window, axis=0, prepend=0.0)[pos - lo:pos - lo + 8]
        assert np.array_equal(got, want.astype(np.float32)), pos


def test_install_as_signals_aliases_every_chain_module_scripts_import():
    import importlib
    import signals_amd
    from signals_amd.chain import nodes
    try:
        signals_amd.install_as_signals()
        for name in ('signals.chain.vis', 'signals.chain.dev', 'signals.chain.discovery', 'signals.chain.files'):
            assert importlib.import_module(name).__name__.startswith('signals_amd.chain.')
        import signals.chain.discovery as d
        import signals.chain.vis as vis
        import signals.chain.dev as dev
        assert d.load_signal('signals.chain.vis.Wave') is vis.Wave and vis.Wave.cls_name() == 'signals.chain.vis.Wave'
        assert vis.Wave().state_attrs() == {'enabled', 'min_amp', 'max_amp'}
        assert vis.Spec().state_attrs() == {'enabled', 'min_freq', 'max_freq', 'bands'}
        with pytest.raises(d.BadSyntax):
            d.load_signal('not a name')
        with pytest.raises(d.BadPath):
            d.load_signal('signals.chain.osc.Nope')
        with pytest.raises(d.InvalidObject):
            d.load_signal('signals.chain.osc.Osc')             # abstract
        sink = dev.SinkDevice()
        assert sink.flags() & SignalFlags.SINK_DEVICE and sink.port_names() == ['input'] and sink.tell() == 0
        # a node class defined outside the package is a reference-style plugin once a script installed the alias
        class Foreign(chain.Receiver):
            input = port('input')
            @classmethod
            def flags(cls):
                return SignalFlags(0)
        assert nodes.wants_host_arrays(Foreign()) is True and nodes.wants_host_arrays(Probe()) is False
    finally:
        nodes.host_plugins(False)


def test_library_scan_finds_plugin_nodes(tmp_path):
    """chain/discovery.py:71-93: any concrete Signal subclass defined in a scanned module is a node"""
    from signals_amd.chain import discovery
    (tmp_path / 'my_plugin.py').write_text(
        'import numpy as np\n'
        'from signals_amd import SignalFlags\n'
        'from signals_amd.chain import ExplicitChannelsEmitter\n'
        'class Dc(ExplicitChannelsEmitter):\n'
        '    @classmethod\n'
        '    def flags(cls):\n'
        '        return SignalFlags.GENERATOR\n'
        '    def _eval(self, request):\n'
        '        return np.ones(request.loc.shape)\n'
        'class _Hidden(Dc):\n'
        '    pass\n')
    lib = discovery.Library([tmp_path / 'my_plugin.py'])
    lib.scan()
    assert 'my_plugin.Dc' in lib.names and not any('_Hidden' in n for n in lib.names)
    for name in ('signals.chain.osc.Sine', 'signals.chain.fx.LowPass', 'signals.chain.fixed.Fixed', 'signals.chain.vis.Wave'):
        assert name in lib.names
    assert not any(n.endswith('.Osc') or 'SinkDevice' in n or 'BlockDriver' in n for n in lib.names)   # abstract / devices
    cls = discovery.load_signal('my_plugin.Dc')
    node = cls(); node.get_state().channels = 2
    p = Probe(); p.input = node
    assert np.array_equal(p.input.request(chain.BlockLoc(position=0, rate=48000, shape=chain.Shape(3, 2))).numpy(), np.ones((3, 2)))


def test_sink_device_name_is_the_headless_block_driver():
    """audio devices are out of scope: `signals.chain.dev.SinkDevice` is the headless BlockDriver under the
    name graph scripts import -- callers pull blocks themselves; an exception in the graph marks the stream inactive
    (reference dev.py:174-176)"""
    from signals_amd.chain import dev
    from signals_amd.chain.driver import BlockDriver
    sink = dev.SinkDevice(blocksize=64)
    assert isinstance(sink, BlockDriver) and sink.flags() & SignalFlags.SINK_DEVICE and sink.port_names() == ['input']
    src = fixed.Fixed(); src.get_state().value = np.array([[0.25, -0.5]])
    sink.input = src
    sink.set_state(sink.State(channels=2))
    blocks = [sink.pull(eager=True) for _ in range(3)]
    assert sink.tell() == 3 and sink.frame_position == 192
    assert blocks[0].shape == (64, 2) and np.all(blocks[0] == np.array([[0.25, -0.5]], dtype=np.float32))
    sink.seek(100)
    assert sink.tell() == 100

    class Broken(chain.ExplicitChannelsEmitter):
        @classmethod
        def flags(cls):
            return SignalFlags.GENERATOR

        def _eval(self, request):
            raise RuntimeError('boom')
    bad = dev.SinkDevice(blocksize=32)
    bad.input = Broken()
    with pytest.raises(RuntimeError):
        bad.pull(eager=True)
    assert not bad.is_active and bad.tell() == 0
    bad.destroy()
    assert not bad.input


def test_reference_script_imports_resolve():
    """scripts/edited_sine.py:5-9 of the reference imports these five modules"""
    import importlib
    from signals_amd.chain import nodes
    try:
        signals_amd.install_as_signals()
        for name in ('signals.chain.dev', 'signals.chain.discovery', 'signals.chain.fixed', 'signals.chain.osc', 'signals.map.control'):
            importlib.import_module(name)
        import signals.chain.dev
        sink = signals.chain.dev.SinkDevice()
        sine = signals.chain.osc.Sine(); sink.input = sine
        assert sink.input.sig is sine
    finally:
        nodes.host_plugins(False)
