"""Differential test on random graphs: the batched engine (one kernel per node) must equal the eager pull path
bit for bit -- shared sub-graphs, cascades, modulated controls, mixed widths -- and the fusing engine must
stay within the float32 rounding it removes."""
import random

import numpy as np
import pytest
import torch

from helpers import RATE, fix, maxerr, mkosc, stream

pytestmark = pytest.mark.gpu
V = 8


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    assert torch.cuda.is_available()
    from signals_amd import runtime
    runtime.set_device('cuda:0')


class Builder:
    """builds the SAME random graph every time it is called with the same seed (fresh node objects)"""

    def __init__(self, seed):
        self.seed = seed

    def build(self):
        from signals_amd.chain import ext, fx
        self.fx, self.ext = fx, ext
        self.rng = random.Random(self.seed)
        self.np = np.random.default_rng(self.seed)
        self.pool = []                                   # sub-graphs available for sharing (diamonds)
        node = self.audio(3)
        if self.rng.random() < 0.5:
            bus = ext.SumBus(); bus.input = node
            if self.rng.random() < 0.5:
                bus.get_state().gains = self.np.uniform(-1, 1, (2, V))
            return bus, bus.channels
        return node, V

    def row(self, lo, hi, wide=True):
        return self.np.uniform(lo, hi, (1, V if wide else 1))

    def control(self, lo, hi):
        """a control input: usually Fixed, sometimes an LFO-driven block-rate signal"""
        fx = self.fx
        if self.rng.random() < 0.3:
            lfo = mkosc(self.rng.choice(['Sine', 'Triangle']), self.row(0.5, 8.0, wide=self.rng.random() < 0.5))
            depth = fx.Gain(); depth.left = lfo; depth.right = fix([[0.4 * (hi - lo) / 2]])
            mid = fx.Mix(); mid.left = depth; mid.right = fix(self.row(lo + 0.3 * (hi - lo), hi - 0.3 * (hi - lo)) * 2)
            mid.mix = fix([[0.5]])
            return mid
        return fix(self.row(lo, hi))

    def audio(self, depth):
        fx, r = self.fx, self.rng
        if depth == 0 or r.random() < 0.2:
            if self.pool and r.random() < 0.3:
                return r.choice(self.pool)
            o = mkosc(r.choice(['Sine', 'Square', 'Sawtooth', 'Triangle']), self.row(55, 1760),
                      self.row(0, 1) if r.random() < 0.7 else None)
            if r.random() < 0.25:
                o.hertz = self.control(100, 900)
            self.pool.append(o)
            return o
        kind = r.choice(['lp', 'hp', 'gain', 'mix', 'ring', 'amp', 'bp', 'env'])
        if kind in ('lp', 'hp'):
            n = (fx.LowPass if kind == 'lp' else fx.HighPass)()
            n.input = self.audio(depth - 1)
            n.cutoff = self.control(300, 6000)
        elif kind == 'bp':
            n = r.choice([fx.BandPass, fx.BandStop])()
            n.input = self.audio(depth - 1)
            lo = self.row(100, 2000)
            n.low = fix(lo); n.high = fix(lo * self.np.uniform(1.5, 4.0, (1, V)))
        elif kind == 'gain':
            n = fx.Gain(); n.left = self.audio(depth - 1); n.right = self.control(0.1, 1.0)
        elif kind == 'amp':
            n = fx.Amp(); n.left = self.audio(depth - 1); n.right = fix(np.round(self.row(1, 3)))
        elif kind == 'mix':
            n = fx.Mix(); n.left = self.audio(depth - 1); n.right = self.audio(depth - 1); n.mix = self.control(0.1, 0.9)
        elif kind == 'env':
            # an ADSR whose every stage boundary falls inside the rendered windows (frames 0..1512 at 48 kHz)
            env = self.ext.ADSR()
            for name, (lo, hi) in dict(attack=(0.001, 0.008), decay=(0.002, 0.008), sustain=(0.2, 0.9), release=(0.002, 0.01),
                                       gate_on=(0.0, 0.006), gate_off=(0.012, 0.03)).items():
                setattr(env, name, fix(self.row(lo, hi, wide=r.random() < 0.8)))
            n = fx.RingMod()
            if r.random() < 0.5:
                n.left = self.audio(depth - 1); n.right = env
            else:
                n.left = env; n.right = self.audio(depth - 1)
        else:
            n = fx.RingMod(); n.left = self.audio(depth - 1); n.right = self.audio(depth - 1)
        self.pool.append(n)
        return n


@pytest.mark.parametrize('seed', range(48))
def test_random_graph_eager_vs_batched(seed):
    from signals_amd.engine import BatchRenderer
    b = Builder(seed)
    for pos, N, K in ((0, 128, 3), (1000, 256, 2)):
        node, channels = b.build()
        want = stream(node, pos, N, K, channels)
        node, channels = b.build()
        plain = BatchRenderer(node, channels, RATE, fuse=False)
        got = torch.cat([plain.render(pos, N, 1), plain.render(pos + N, N, K - 1)]).cpu().numpy()    # two batches: tails
        assert got.shape == want.shape
        assert np.array_equal(got, want, equal_nan=True), (seed, pos)
        node, channels = b.build()
        fused = BatchRenderer(node, channels, RATE).render(pos, N, K).cpu().numpy()
        scale = max(1.0, float(np.nanmax(np.abs(want)))) if np.isfinite(want).any() else 1.0
        assert maxerr(fused, want) < 1e-6 * scale, (seed, pos)


PROGRAM_RUNS = {'renders': 0, 'with_program': 0, 'short_blocks': 0}


@pytest.mark.parametrize('seed', range(48))
def test_random_graph_voice_program_vs_eager(seed):
    """the same random graphs with every per-voice sub-graph the compiler can express forced through the interpreted
    launch (sig_voice_program; by default only where it is faster), continuing batches and a fresh mid-stream start, blocks
    longer and SHORTER than the filter context -- against the eager pull path, the literal mirror of the reference's request
    protocol and block cache (chain/__init__.py:266-303, :424-457)"""
    from signals_amd import _native
    from signals_amd.engine import BatchRenderer, KernelTimer, NotBatchable
    b = Builder(seed)
    ran = 0
    # launch geometry: the heuristic (one voice per lane, one block per lane at this size), two voices per lane with spans of
    # two blocks, one voice per lane with spans of three
    _native.set_voice_program_tuning(*((0, 0), (2, 2), (1, 3))[seed % 3])
    try:
        _random_graph_program_case(b, seed, BatchRenderer, KernelTimer, NotBatchable)
    finally:
        _native.set_voice_program_tuning(0, 0)


def _random_graph_program_case(b, seed, BatchRenderer, KernelTimer, NotBatchable):
    ran = 0
    for pos, N, batches in ((0, 128, (1, 2)), (1000, 256, (2, 1)), (0, 64, (3, 2, 1)), (640, 32, (4, 3))):
        node, channels = b.build()
        K = sum(batches)
        want = stream(node, pos, N, K, channels)
        node, channels = b.build()
        timer = KernelTimer()
        r = BatchRenderer(node, channels, RATE, fuse_program='always', timer=timer)
        try:
            parts, p = [], pos
            for k in batches:
                parts.append(r.render(p, N, k)); p += N * k
        except NotBatchable:
            assert N < 100, (seed, pos, N)                   # only short blocks may be refused (deep cascades, band filters: the eager path)
            continue
        got = torch.cat(parts).cpu().numpy()
        ran += 1
        names = set(timer.summary())
        PROGRAM_RUNS['renders'] += 1
        PROGRAM_RUNS['with_program'] += any(n.startswith('voice_program') for n in names)
        PROGRAM_RUNS['short_blocks'] += N < 100
        scale = max(1.0, float(np.nanmax(np.abs(want)))) if np.isfinite(want).any() else 1.0
        assert got.shape == want.shape
        assert maxerr(got, want) < 1e-6 * scale, (seed, pos, N)
    assert ran >= 2


def test_the_random_graphs_did_exercise_the_voice_program():
    """(runs after the 48 seeds above) most of those graphs contain a sub-graph the interpreter takes, and a good share of
    the short-block renders stayed in the batched engine"""
    if PROGRAM_RUNS['renders'] == 0:
        pytest.skip('the seeds did not run in this session')
    print(PROGRAM_RUNS)
    assert PROGRAM_RUNS['with_program'] >= PROGRAM_RUNS['renders'] // 2, PROGRAM_RUNS
    assert PROGRAM_RUNS['short_blocks'] >= 24, PROGRAM_RUNS
