#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE.

Runs ONLY in the build container (needs /root/reference).  The GPU box never
sees the reference; it sees the .npz files this script wrote.  Nothing from the
reference's source text is stored -- fixtures are inputs (per-voice parameter
rows, positions, block sizes) and float64 outputs.

The reference targets Python 3.11 / numpy 1.23 and imports its Qt GUI from the
package root, so it is imported under the in-process shim SURVEY.md §8c
describes:
  * typing.Self, enum.StrEnum            (3.11 names, absent from 3.10)
  * np.float = float                     (removed numpy alias, fx.py:99)
  * more_itertools.one                   (absent package; chain/__init__.py:406)
  * PyQt5 / signals.ui / signals.ui.theme  (GUI, imported by signals/__init__.py:7-13,
                                            never executed on the render path)
None of these carries arithmetic: osc/fx/fixed/shape run their own numpy/scipy
code unchanged.

Usage:  python tests/golden/gen_golden.py            # writes tests/golden/*.npz
"""
import enum
import importlib.abc
import importlib.machinery
import json
import pathlib
import sys
import types
import typing

sys.dont_write_bytecode = True

import numpy as np
import scipy

HERE = pathlib.Path(__file__).resolve().parent
REF_SRC = '/root/reference/src'
RATE = 48000
HOUR = 172_800_000


def _install_shim():
    if not hasattr(typing, 'Self'):
        typing.Self = typing.TypeVar('Self')
    if not hasattr(enum, 'StrEnum'):
        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)
        enum.StrEnum = StrEnum
    if not hasattr(np, 'float'):
        np.float = float

    mi = types.ModuleType('more_itertools')

    def one(iterable):
        it = iter(iterable)
        try:
            first = next(it)
        except StopIteration:
            raise ValueError('too few items in iterable (expected 1)')
        try:
            next(it)
        except StopIteration:
            return first
        raise ValueError('Expected exactly one item in iterable')
    mi.one = one
    sys.modules.setdefault('more_itertools', mi)

    qt = types.ModuleType('PyQt5')
    qtw = types.ModuleType('PyQt5.QtWidgets')
    qtw.QApplication = type('QApplication', (), {})
    qt.QtWidgets = qtw
    sys.modules.setdefault('PyQt5', qt)
    sys.modules.setdefault('PyQt5.QtWidgets', qtw)

    class _UiStubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        names = ('signals.ui', 'signals.ui.theme')

        def find_spec(self, name, path, target=None):
            if name in self.names:
                return importlib.machinery.ModuleSpec(name, self, is_package=(name == 'signals.ui'))
            return None

        def create_module(self, spec):
            m = types.ModuleType(spec.name)
            if spec.name == 'signals.ui':
                m.__path__ = []
            else:
                m.Theme = object
            return m

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, _UiStubFinder())
    sys.path.insert(0, REF_SRC)


_install_shim()

import signals.chain as chain          # noqa: E402
import signals.chain.fixed as fixed    # noqa: E402
import signals.chain.fx as fx          # noqa: E402
import signals.chain.osc as osc        # noqa: E402
import signals.chain.shape as shape    # noqa: E402
from signals import SignalFlags        # noqa: E402


class Probe(chain.Receiver):
    input = chain.port('input')

    @classmethod
    def flags(cls):
        return SignalFlags(0)


def fix(value) -> fixed.Fixed:
    f = fixed.Fixed()
    f.get_state().value = np.array(value, ndmin=2)
    return f


def loc(position, frames, channels):
    return chain.BlockLoc(position=position, rate=RATE,
                          shape=chain.Shape(frames=frames, channels=channels))


def render(node, position, frames, channels):
    p = Probe()
    p.input = node
    out = p.input.request(loc(position, frames, channels))
    del p.input
    return np.array(out, dtype=np.float64)


def voice_params(V, seed=0):
    rng = np.random.default_rng(seed)
    return dict(
        hertz=rng.uniform(55, 1760, size=(1, V)),
        phase=rng.uniform(0, 1, size=(1, V)),
        cutoff=rng.uniform(200, 8000, size=(1, V)),
        gain=rng.uniform(0, 1, size=(1, V)) / V,
    )


POSITIONS = [0, 50, 256, 512, HOUR]
OSC = {'Sine': osc.Sine, 'Square': osc.Square, 'Sawtooth': osc.Sawtooth, 'Triangle': osc.Triangle}


def gen_osc(out):
    V, N = 16, 256
    vp = voice_params(V)
    out['osc/hertz'] = vp['hertz']
    out['osc/phase'] = vp['phase']
    out['osc/positions'] = np.array(POSITIONS, dtype=np.int64)
    out['osc/frames'] = np.int64(N)
    for name, cls in OSC.items():
        for pos in POSITIONS:
            # fresh node per position: the block cache must not slice an older block
            o = cls()
            o.hertz = fix(vp['hertz'])
            o.phase = fix(vp['phase'])
            out[f'osc/{name}/p{pos}'] = render(o, pos, N, V)
        # unplugged phase -> (1,1) zeros (chain/__init__.py:297-298)
        o = cls()
        o.hertz = fix(vp['hertz'])
        out[f'osc/{name}/nophase/p256'] = render(o, 256, N, V)
    # exact-discontinuity case: hertz that lands mod(t,1) on 0.5 / 0.0 exactly
    hz = np.array([[12000.0, 6000.0, 24000.0, 3000.0]])
    ph = np.array([[0.0, 0.25, 0.5, 0.75]])
    out['osc/edge/hertz'] = hz
    out['osc/edge/phase'] = ph
    for name, cls in OSC.items():
        o = cls()
        o.hertz = fix(hz)
        o.phase = fix(ph)
        out[f'osc/edge/{name}'] = render(o, 0, 64, 4)
    # negative phase / negative hertz: numpy mod sign semantics
    hz = np.array([[-440.0, 440.0, -1.5, 1e-3]])
    ph = np.array([[0.0, -0.3, -1e-20, -2.75]])
    out['osc/neg/hertz'] = hz
    out['osc/neg/phase'] = ph
    for name, cls in OSC.items():
        o = cls()
        o.hertz = fix(hz)
        o.phase = fix(ph)
        out[f'osc/neg/{name}'] = render(o, 1000, 128, 4)
    # integer-valued Fixed (json `value=[[220]]` -> int64 array, map/__init__.py:131-139)
    o = osc.Sine()
    o.hertz = fix(np.array([[220]]))
    out['osc/int_hertz/Sine'] = render(o, 0, 256, 1)
    # block-rate request (frames=1) -- what a control port sees
    o = osc.Sine()
    o.hertz = fix(vp['hertz'])
    o.phase = fix(vp['phase'])
    out['osc/ctrl/Sine/p512'] = render(o, 512, 1, V)


def gen_filter(out):
    V, N = 16, 256
    vp = voice_params(V)
    cut = np.geomspace(50, 20000, V).reshape(1, V)
    out['filt/hertz'] = vp['hertz']
    out['filt/phase'] = vp['phase']
    out['filt/cutoff'] = cut
    out['filt/positions'] = np.array(POSITIONS, dtype=np.int64)
    for fname, fcls in (('LowPass', fx.LowPass), ('HighPass', fx.HighPass)):
        for oname in ('Sine', 'Sawtooth'):
            for pos in POSITIONS:
                o = OSC[oname]()
                o.hertz = fix(vp['hertz'])
                o.phase = fix(vp['phase'])
                f = fcls()
                f.input = o
                f.cutoff = fix(cut)
                out[f'filt/{fname}/{oname}/p{pos}'] = render(f, pos, N, V)
    # odd block size / short context (position < 100)
    for pos, n in ((7, 33), (99, 101), (100, 64), (101, 1000)):
        o = osc.Triangle()
        o.hertz = fix(vp['hertz'])
        o.phase = fix(vp['phase'])
        f = fx.LowPass()
        f.input = o
        f.cutoff = fix(cut)
        out[f'filt/ragged/p{pos}_n{n}'] = render(f, pos, n, V)
    # sequential render from 0 (cache populated like the device callback does)
    o = osc.Sine()
    o.hertz = fix(vp['hertz'])
    o.phase = fix(vp['phase'])
    f = fx.LowPass()
    f.input = o
    f.cutoff = fix(vp['cutoff'])
    out['filt/seq/cutoff'] = vp['cutoff']
    p = Probe()
    p.input = f
    out['filt/seq/LowPass'] = np.concatenate(
        [np.array(p.input.request(loc(b * N, N, V))) for b in range(4)])


def gen_cascade(out):
    """A9: two LowPass in series, rendered sequentially from 0 -- cache-history dependent."""
    V = 8
    vp = voice_params(V, seed=1)
    cut1 = np.geomspace(150, 6000, V).reshape(1, V)
    cut2 = np.geomspace(300, 9000, V).reshape(1, V)
    out['casc/hertz'] = vp['hertz']
    out['casc/phase'] = vp['phase']
    out['casc/cut1'] = cut1
    out['casc/cut2'] = cut2
    for N in (256, 1024):
        o = osc.Sawtooth()
        o.hertz = fix(vp['hertz'])
        o.phase = fix(vp['phase'])
        f1 = fx.LowPass()
        f1.input = o
        f1.cutoff = fix(cut1)
        f2 = fx.LowPass()
        f2.input = f1
        f2.cutoff = fix(cut2)
        p = Probe()
        p.input = f2
        out[f'casc/seq_n{N}'] = np.concatenate(
            [np.array(p.input.request(loc(b * N, N, V))) for b in range(4)])
    # fresh graph straight at a late position: inner 'before' window cold-starts at p-200
    o = osc.Sawtooth()
    o.hertz = fix(vp['hertz'])
    o.phase = fix(vp['phase'])
    f1 = fx.LowPass()
    f1.input = o
    f1.cutoff = fix(cut1)
    f2 = fx.HighPass()
    f2.input = f1
    f2.cutoff = fix(cut2)
    out['casc/fresh_p768'] = render(f2, 768, 256, V)


def gen_modulated(out):
    """Control ports driven at block rate (forward_at_block_rate, chain/__init__.py:305-306): vibrato on an oscillator's
    hertz, a wobble on its phase, an LFO sweep on a filter's cutoff, a tremolo on a gain -- rendered sequentially, so that
    a filter's context rows are what the oscillator's block cache answers (the previous block's samples when the block is
    at least as long as the context; a block of its own at p - 100 when it is shorter: N = 64)."""
    V = 8
    vp = voice_params(V, seed=2)
    cut = np.geomspace(200, 7000, V).reshape(1, V)
    gain = np.linspace(0.2, 1.0, V).reshape(1, V)
    for k, v in (('hertz', vp['hertz']), ('phase', vp['phase']), ('cutoff', cut), ('gain', gain)):
        out[f'mod/{k}'] = v

    def lfo(kind, hz, depth, centre):
        """centre * (1 - depth + depth * osc)  as  RingMod(Mix(osc, 1, depth), centre)"""
        o = OSC[kind]()
        o.hertz = fix([[hz]])
        m = fx.Mix()
        m.left = o
        m.right = fix([[1.0]])
        m.mix = fix([[depth]])
        r = fx.RingMod()
        r.left = m
        r.right = fix(centre)
        return r

    def voice(kind, fm, pm, sweep, trem):
        o = OSC[kind]()
        o.hertz = lfo('Sine', 5.3, 0.02, vp['hertz']) if fm else fix(vp['hertz'])
        o.phase = lfo('Triangle', 2.1, 0.1, vp['phase']) if pm else fix(vp['phase'])
        f = fx.LowPass()
        f.input = o
        f.cutoff = lfo('Sine', 1.7, 0.4, cut) if sweep else fix(cut)
        g = fx.Gain()
        g.left = f
        g.right = lfo('Triangle', 3.1, 0.3, gain) if trem else fix(gain)
        return g

    cases = {'fm': ('Sawtooth', True, False, False, False), 'fm_pm_sine': ('Sine', True, True, False, False),
             'sweep_trem': ('Square', False, False, True, True), 'all': ('Triangle', True, True, True, True),
             'trem_sine': ('Sine', False, False, False, True)}
    for name, spec in cases.items():
        for N, blocks, start in ((256, 5, 0), (256, 4, 4096), (64, 6, 4096)):
            p = Probe()
            p.input = voice(*spec)
            out[f'mod/{name}/n{N}_p{start}'] = np.concatenate(
                [np.array(p.input.request(loc(start + b * N, N, V))) for b in range(blocks)])


def gen_small_blocks(out):
    """Blocks no longer than the filter context (dev.py:139-141: the sink takes whatever block size PortAudio hands it; 32- and
    64-frame callbacks are the normal real-time case): two filters in series and a block-rate modulated voice rendered
    sequentially in 32-, 50-, 64- and 100-frame blocks.  A filter's context request [p - 100, p) then lies in no single cached
    block of its input and is answered as a block of its own, while the block itself is served from the OLDEST cached reply to
    an `after` request that contains it (chain/__init__.py:431-442) -- what the engine's voice program reproduces.  Also three
    filters in series at 256 frames (the block history reaches two blocks back)."""
    V = 8
    vp = voice_params(V, seed=3)
    cut1 = np.geomspace(25, 4000, V).reshape(1, V)            # slow inner filters: where a chain was cold-started matters
    cut2 = np.geomspace(300, 9000, V).reshape(1, V)
    cut3 = np.geomspace(9000, 40, V).reshape(1, V)
    gain = np.linspace(0.2, 1.0, V).reshape(1, V)
    for k, v in (('hertz', vp['hertz']), ('phase', vp['phase']), ('cut1', cut1), ('cut2', cut2), ('cut3', cut3), ('gain', gain)):
        out[f'small/{k}'] = v

    def cascade(depth=2):
        o = osc.Triangle()
        o.hertz = fix(vp['hertz'])
        o.phase = fix(vp['phase'])
        f1 = fx.LowPass()
        f1.input = o
        f1.cutoff = fix(cut1)
        f2 = fx.HighPass()
        f2.input = f1
        f2.cutoff = fix(cut2)
        top = f2
        if depth == 3:
            f3 = fx.LowPass()
            f3.input = f2
            f3.cutoff = fix(cut3)
            top = f3
        g = fx.Gain()
        g.left = top
        g.right = fix(gain)
        return g

    def lfo(kind, hz, depth, centre):
        o = OSC[kind]()
        o.hertz = fix([[hz]])
        m = fx.Mix()
        m.left = o
        m.right = fix([[1.0]])
        m.mix = fix([[depth]])
        r = fx.RingMod()
        r.left = m
        r.right = fix(centre)
        return r

    def fm_voice():
        o = osc.Sawtooth()
        o.hertz = lfo('Sine', 5.3, 0.02, vp['hertz'])
        o.phase = lfo('Triangle', 2.1, 0.1, vp['phase'])
        f = fx.LowPass()
        f.input = o
        f.cutoff = lfo('Sine', 1.7, 0.4, cut2)
        g = fx.Gain()
        g.left = f
        g.right = lfo('Triangle', 3.1, 0.3, gain)
        return g

    for N in (32, 50, 64, 100):
        for start in (0, 4096):
            for name, build in (('cascade', cascade), ('fm', fm_voice)):
                p = Probe()
                p.input = build()
                out[f'small/{name}/n{N}_p{start}'] = np.concatenate(
                    [np.array(p.input.request(loc(start + b * N, N, V))) for b in range(12)])
    p = Probe()
    p.input = cascade(3)
    out['small/cascade3/n256_p0'] = np.concatenate([np.array(p.input.request(loc(b * 256, 256, V))) for b in range(6)])
    out['small/cascade3/fresh_p1000'] = render(cascade(3), 1000, 256, V)


def gen_shapes(out):
    """Voice graphs beyond the fused kernels' patterns, rendered sequentially in 256-frame blocks from 0 and once on a fresh graph
    mid-stream: RingMod of two filtered oscillators, a Mix behind a filter, an Amp behind a filter (with its NaN pattern,
    fx.py:55-60), a node with TWO readers inside the voice (its block cache serves the second, chain/__init__.py:424-457), a
    cascade with a swept inner cutoff under a tremolo.  What the engine's voice program (one launch per batch) must reproduce."""
    V = 8
    vp = voice_params(V, seed=11)
    rng = np.random.default_rng(12)
    hz2 = rng.uniform(55, 1760, (1, V))
    cut1 = np.geomspace(30, 5000, V).reshape(1, V)
    cut2 = np.geomspace(8000, 200, V).reshape(1, V)
    mix = np.linspace(0.1, 0.9, V).reshape(1, V)
    gain = np.linspace(0.3, 1.0, V).reshape(1, V)
    expo = np.array([[1.0, 2.0, 3.0, 1.5, 0.5, 2.0, 2.5, 1.0]])
    for k, v in (('hertz', vp['hertz']), ('phase', vp['phase']), ('hertz2', hz2), ('cut1', cut1), ('cut2', cut2), ('mix', mix),
                 ('gain', gain), ('expo', expo)):
        out[f'shapes/{k}'] = v

    def saw():
        o = osc.Sawtooth(); o.hertz = fix(vp['hertz']); o.phase = fix(vp['phase'])
        return o

    def tri():
        o = osc.Triangle(); o.hertz = fix(hz2)
        return o

    def filt(cls, src, cut):
        f = cls(); f.input = src; f.cutoff = cut if isinstance(cut, chain.Emitter) else fix(cut)
        return f

    def lfo(hz, depth, centre):
        o = osc.Sine(); o.hertz = fix([[hz]])
        m = fx.Mix(); m.left = o; m.right = fix([[1.0]]); m.mix = fix([[depth]])
        r = fx.RingMod(); r.left = m; r.right = fix(centre)
        return r

    def ringmod():
        n = fx.RingMod(); n.left = filt(fx.LowPass, saw(), cut1); n.right = filt(fx.HighPass, tri(), cut2)
        return n

    def mix_after_filter():
        n = fx.Mix(); n.left = filt(fx.HighPass, saw(), cut1); n.right = tri(); n.mix = fix(mix)
        return n

    def amp():
        a = fx.Amp(); a.left = filt(fx.LowPass, saw(), cut2); a.right = fix(expo)
        g = fx.Gain(); g.left = a; g.right = fix(gain)
        return g

    def fanout():
        shared = filt(fx.LowPass, saw(), cut1)
        rm = fx.RingMod(); rm.left = shared; rm.right = tri()
        n = fx.Mix(); n.left = shared; n.right = rm; n.mix = fix(mix)
        return n

    def swept_cascade():
        inner = filt(fx.LowPass, saw(), lfo(1.7, 0.4, cut2))
        outer = filt(fx.LowPass, inner, cut1 * 4.0)
        g = fx.Gain(); g.left = outer; g.right = lfo(3.1, 0.3, gain)
        return g

    for name, build in (('ringmod', ringmod), ('mix', mix_after_filter), ('amp', amp), ('fanout', fanout), ('swept_cascade', swept_cascade)):
        p = Probe()
        p.input = build()
        with np.errstate(invalid='ignore'):
            out[f'shapes/{name}/n256_p0'] = np.concatenate([np.array(p.input.request(loc(b * 256, 256, V))) for b in range(6)])
            out[f'shapes/{name}/fresh_p1000'] = render(build(), 1000, 256, V)


def gen_pairs(out):
    """A filter reading TWO oscillators through Mix / RingMod (fx.py:35-46), and a Gain in front of a filter: the
    topologies round 2's fuser folds into one launch; rendered sequentially."""
    V, N = 8, 256
    vp, vq = voice_params(V, seed=3), voice_params(V, seed=4)
    cut = np.geomspace(200, 7000, V).reshape(1, V)
    m = np.linspace(0.1, 0.9, V).reshape(1, V)
    for k, v in (('hertz', vp['hertz']), ('phase', vp['phase']), ('hertz2', vq['hertz'] * 0.5), ('phase2', vq['phase']),
                 ('cutoff', cut), ('mix', m)):
        out[f'pair/{k}'] = v

    def two(ka, kb):
        a = OSC[ka]()
        a.hertz = fix(vp['hertz'])
        a.phase = fix(vp['phase'])
        b = OSC[kb]()
        b.hertz = fix(vq['hertz'] * 0.5)
        b.phase = fix(vq['phase'])
        return a, b

    for op, ka, kb in (('Mix', 'Sine', 'Sawtooth'), ('RingMod', 'Triangle', 'Square'), ('Mix', 'Sawtooth', 'Sine')):
        a, b = two(ka, kb)
        e = getattr(fx, op)()
        e.left = a
        e.right = b
        if op == 'Mix':
            e.mix = fix(m)
        f = fx.LowPass()
        f.input = e
        f.cutoff = fix(cut)
        p = Probe()
        p.input = f
        out[f'pair/{op}_{ka}_{kb}'] = np.concatenate([np.array(p.input.request(loc(4096 + b_ * N, N, V))) for b_ in range(3)])
    a, _ = two('Triangle', 'Sine')
    g = fx.Gain()
    g.left = a
    g.right = fix(m)
    f = fx.HighPass()
    f.input = g
    f.cutoff = fix(cut)
    p = Probe()
    p.input = f
    out['pair/pre_gain_Triangle_hp'] = np.concatenate([np.array(p.input.request(loc(b_ * N, N, V))) for b_ in range(3)])


def gen_effects(out):
    V, N, pos = 8, 128, 300
    vp = voice_params(V, seed=2)
    out['fxs/hertz'] = vp['hertz']
    out['fxs/phase'] = vp['phase']
    rng = np.random.default_rng(3)
    g = rng.uniform(0, 1, size=(1, V))
    out['fxs/gain'] = g

    def mk(cls, hz_scale=1.0):
        o = cls()
        o.hertz = fix(vp['hertz'] * hz_scale)
        o.phase = fix(vp['phase'])
        return o

    n = fx.Gain()
    n.left = mk(osc.Sine)
    n.right = fix(g)
    out['fxs/Gain'] = render(n, pos, N, V)
    n = fx.Gain()                       # scalar (1,1) gain broadcast
    n.left = mk(osc.Sine)
    n.right = fix([[0.2]])
    out['fxs/Gain_scalar'] = render(n, pos, N, V)
    n = fx.Mix()
    n.left = mk(osc.Sine)
    n.right = mk(osc.Sawtooth, 0.5)
    n.mix = fix(g)
    out['fxs/Mix'] = render(n, pos, N, V)
    n = fx.RingMod()
    n.left = mk(osc.Sine)
    n.right = mk(osc.Triangle, 0.25)
    out['fxs/RingMod'] = render(n, pos, N, V)
    e = rng.uniform(0.5, 3.0, size=(1, V))
    out['fxs/amp_exp'] = e
    n = fx.Amp()
    n.left = mk(osc.Sawtooth)
    n.right = fix(np.round(e))          # integer exponents: defined for negative input
    out['fxs/Amp_int'] = render(n, pos, N, V)
    with np.errstate(invalid='ignore'):
        n = fx.Amp()
        n.left = mk(osc.Sawtooth)
        n.right = fix(e)                # fractional exponent: NaN where input < 0 (fx.py:60)
        out['fxs/Amp_frac'] = render(n, pos, N, V)
    n = shape.Merge()
    n.left = mk(osc.Sine)
    n.right = mk(osc.Square, 0.5)
    out['fxs/Merge'] = render(n, pos, N, 2 * V)
    # disabled emitter -> (1,1) zero (chain/__init__.py:253-254); unplugged port likewise
    o = mk(osc.Sine)
    o.get_state().enabled = False
    out['fxs/disabled'] = render(o, pos, N, V)
    n = fx.Gain()
    n.right = fix(g)
    out['fxs/unplugged_left'] = render(n, pos, N, V)
    # broadcast: 1-channel osc answering a 2-channel request (Shape.__le__)
    o = osc.Sine()
    o.hertz = fix([[440.0]])
    out['fxs/broadcast_1to2'] = render(o, 0, N, 2)


def gen_sigs_topologies(out):
    """The two patch fixtures (src/signals/*.sigs) rebuilt without device/file/vis nodes."""
    N = 256
    # vis_test.sigs: Fixed(220) -> Sine
    o = osc.Sine()
    o.hertz = fix(np.array([[220]]))
    p = Probe()
    p.input = o
    out['sigs/vis_test'] = np.concatenate(
        [np.array(p.input.request(loc(b * N, N, 1))) for b in range(3)])
    # lowpass_test.sigs: Fixed(440)->Triangle->Gain(0.2)->LowPass(600); Merge(left=LowPass, right=Gain)
    tri = osc.Triangle()
    tri.hertz = fix(np.array([[440]]))
    g = fx.Gain()
    g.left = tri
    g.right = fix(np.array([[0.2]]))
    lp = fx.LowPass()
    lp.input = g
    lp.cutoff = fix(np.array([[600]]))
    m = shape.Merge()
    m.left = lp
    m.right = g
    p = Probe()
    p.input = m
    out['sigs/lowpass_test'] = np.concatenate(
        [np.array(p.input.request(loc(b * N, N, 2))) for b in range(3)])


def gen_c2(out):
    """BASELINE config 2 at reduced width: Fixed->Sine->LowPass->Gain, sequential from 0 and from 1 h."""
    V, N = 32, 256
    vp = voice_params(V)
    for k in ('hertz', 'phase', 'cutoff', 'gain'):
        out[f'c2/{k}'] = vp[k]
    for tag, pos0 in (('p0', 0), ('p1h', HOUR)):
        o = osc.Sine()
        o.hertz = fix(vp['hertz'])
        o.phase = fix(vp['phase'])
        f = fx.LowPass()
        f.input = o
        f.cutoff = fix(vp['cutoff'])
        g = fx.Gain()
        g.left = f
        g.right = fix(vp['gain'])
        p = Probe()
        p.input = g
        out[f'c2/{tag}'] = np.concatenate(
            [np.array(p.input.request(loc(pos0 + b * N, N, V))) for b in range(4)])


def gen_blockloc(out):
    """Integer rows: frame_range / before / after / containment (chain/__init__.py:107-159)."""
    rows = []
    for pos, n in ((0, 256), (50, 256), (100, 256), (101, 7), (HOUR, 1024)):
        l = loc(pos, n, 4)
        b, a = l.before(100), l.after(100)
        rows.append([pos, n, b.position, b.shape.frames, a.position, a.shape.frames,
                     int(l.frame_range[0, 0]), int(l.frame_range[-1, 0]),
                     int(b <= l), int(l <= l), int(l.resize(1) <= l)])
    out['blockloc/table'] = np.array(rows, dtype=np.int64)


def main():
    groups = {
        'osc': gen_osc, 'filter': gen_filter, 'cascade': gen_cascade, 'effects': gen_effects,
        'sigs': gen_sigs_topologies, 'c2': gen_c2, 'blockloc': gen_blockloc, 'modulated': gen_modulated, 'pairs': gen_pairs,
        'small': gen_small_blocks, 'shapes': gen_shapes,
    }
    only = sys.argv[1:]                                    # python tests/golden/gen_golden.py [group ...]: only those groups
    if only:
        groups = {k: v for k, v in groups.items() if k in only}
    meta = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                rate=RATE, reference='/root/reference (noah-aviel-dove/signals @ v1)',
                note='outputs are float64 exactly as the reference returned them')
    for name, fn in groups.items():
        out = {}
        fn(out)
        path = HERE / f'{name}.npz'
        np.savez_compressed(path, **{k.replace('/', '__'): v for k, v in out.items()})
        print(f'{path.name}: {len(out)} arrays, {path.stat().st_size / 1024:.0f} KiB')
    if not only:
        (HERE / 'META.json').write_text(json.dumps(meta, indent=1) + '\n')


if __name__ == '__main__':
    main()
