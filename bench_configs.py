"""BASELINE.json's GPU configurations as node graphs with the CPU-oracle graph of each, shared by
bench.py, tools/measure_configs.py and the full-size tests.  Parameters: numpy default_rng(0), drawn once per config.

  C2  1024-voice Fixed -> Sine -> LowPass -> Gain -> SumBus(stereo), N = 256          (the headline)
  C3  1024-voice Sawtooth -> LowPass -> LowPass -> x ADSR -> SumBus(mono), N = 1024
  C5  4096-voice Sine -> LowPass -> MixMatrix(64 x 64), N = 256
  C2m C2's voice with block-rate modulation (not a BASELINE configuration: what the reference's control ports allow on top of
      it, chain/__init__.py:305-306): any oscillator, vibrato on hertz, an LFO sweep on the cutoff, a tremolo on the gain
"""
import numpy as np

RATE = 48000


def fixed(v):
    from signals_amd.chain.fixed import Fixed
    f = Fixed()
    f.get_state().value = np.ascontiguousarray(np.array(v, ndmin=2, dtype=float))
    return f


# ----------------------------------------------------------------------------------------------- C2
def c2_params(total_voices: int) -> dict:
    rng = np.random.default_rng(0)
    hertz = rng.uniform(55, 1760, size=(1, total_voices))
    phase = rng.uniform(0, 1, size=(1, total_voices))
    cutoff = rng.uniform(200, 8000, size=(1, total_voices))
    gain = rng.uniform(0, 1, size=(1, total_voices)) / total_voices
    theta = rng.uniform(0, np.pi / 2, size=total_voices)
    pan = np.stack([np.cos(theta), np.sin(theta)])
    return dict(hertz=hertz, phase=phase, cutoff=cutoff, gain=gain, pan=pan)


def c2_graph(p: dict, lo: int, hi: int):
    from signals_amd.chain.ext import SumBus
    from signals_amd.chain.fx import Gain, LowPass
    from signals_amd.chain.osc import Sine
    osc = Sine()
    osc.hertz = fixed(p['hertz'][:, lo:hi])
    osc.phase = fixed(p['phase'][:, lo:hi])
    lp = LowPass()
    lp.input = osc
    lp.cutoff = fixed(p['cutoff'][:, lo:hi])
    g = Gain()
    g.left = lp
    g.right = fixed(p['gain'][:, lo:hi])
    bus = SumBus()
    bus.input = g
    bus.get_state().gains = np.ascontiguousarray(p['pan'][:, lo:hi])
    return bus


def c2_oracle(p: dict, lo: int, hi: int):
    """(oracle node answering the per-voice Gain output, pan) -- bus = R.sum_bus(R.render(node, ...), pan)"""
    from oracle import chain_ref as R
    sl = slice(lo, hi)
    node = R.Binary('Gain', R.Filter('lp', R.Osc('Sine', R.Fixed(p['hertz'][:, sl]), R.Fixed(p['phase'][:, sl])),
                                     R.Fixed(p['cutoff'][:, sl])), R.Fixed(p['gain'][:, sl]))
    return node, p['pan'][:, sl]


# ----------------------------------------------------------------------------------------------- C3
C3_ENV = dict(attack=(0.001, 0.05), decay=(0.01, 0.2), sustain=(0.2, 0.9), release=(0.05, 0.5),
              gate_on=(0.0, 0.5), gate_off=(1.0, 4.0))


def c3_params(V: int) -> dict:
    rng = np.random.default_rng(0)
    p = dict(hertz=rng.uniform(55, 1760, (1, V)), phase=rng.uniform(0, 1, (1, V)),
             cut1=rng.uniform(200, 8000, (1, V)), cut2=rng.uniform(200, 8000, (1, V)))
    p['env'] = {name: rng.uniform(lo, hi, (1, V)) for name, (lo, hi) in C3_ENV.items()}
    return p


def c3_graph(p: dict):
    """Saw -> LowPass -> LowPass -> (x ADSR) -> SumBus, mono"""
    from signals_amd.chain import ext, fx, osc
    o = osc.Sawtooth(); o.hertz = fixed(p['hertz']); o.phase = fixed(p['phase'])
    f1 = fx.LowPass(); f1.input = o; f1.cutoff = fixed(p['cut1'])
    f2 = fx.LowPass(); f2.input = f1; f2.cutoff = fixed(p['cut2'])
    env = ext.ADSR()
    for name in C3_ENV:
        setattr(env, name, fixed(p['env'][name]))
    rm = fx.RingMod(); rm.left = f2; rm.right = env
    bus = ext.SumBus(); bus.input = rm
    return bus


def c3_oracle(p: dict):
    from oracle import chain_ref as R
    o = R.Osc('Sawtooth', R.Fixed(p['hertz']), R.Fixed(p['phase']))
    f2 = R.Filter('lp', R.Filter('lp', o, R.Fixed(p['cut1'])), R.Fixed(p['cut2']))
    return R.Binary('RingMod', f2, R.Adsr(**p['env']))


# ----------------------------------------------------------------------------------------------- C5
def c5_params(V: int) -> dict:
    rng = np.random.default_rng(0)
    p = dict(hertz=rng.uniform(55, 1760, (1, V)), phase=rng.uniform(0, 1, (1, V)), cutoff=rng.uniform(200, 8000, (1, V)))
    p['matrix'] = np.linalg.qr(rng.standard_normal((64, 64)))[0]
    return p


def c5_graph(p: dict):
    """Sine -> LowPass -> MixMatrix(64 x 64)"""
    from signals_amd.chain import ext, fx, osc
    o = osc.Sine(); o.hertz = fixed(p['hertz']); o.phase = fixed(p['phase'])
    f = fx.LowPass(); f.input = o; f.cutoff = fixed(p['cutoff'])
    mm = ext.MixMatrix(); mm.input = f
    mm.get_state().matrix = p['matrix']
    return mm


def c5_oracle(p: dict):
    from oracle import chain_ref as R
    lp = R.Filter('lp', R.Osc('Sine', R.Fixed(p['hertz']), R.Fixed(p['phase'])), R.Fixed(p['cutoff']))
    return R.MixMatrix(lp, p['matrix'].astype(np.float32).astype(np.float64))     # the GPU multiplies by the float32 matrix


# ----------------------------------------------------------------------------------------------- C2 with modulated controls
def c2_modulated_graph(p: dict, kind: str = 'Sawtooth', vibrato: bool = True, sweep: bool = True, tremolo: bool = True):
    """C2's voice with its control ports driven at block rate: hertz = p.hertz +- 4.5 Hz at 5.3 Hz, cutoff = p.cutoff x
    (0.6 + 0.4 sin 1.7 Hz), gain = p.gain x (0.7 + 0.3 sin 3.1 Hz)"""
    from signals_amd.chain import ext, fx, osc

    def lfo(hz, depth, centre_row):
        """depth * sin + centre  as  Mix(Gain(Sine, 2 depth), 2 centre, 0.5)"""
        s_ = osc.Sine(); s_.hertz = fixed([[hz]])
        g = fx.Gain(); g.left = s_; g.right = fixed([[2.0 * depth]])
        m = fx.Mix(); m.left = g; m.right = fixed(2.0 * np.asarray(centre_row)); m.mix = fixed([[0.5]])
        return m

    o = getattr(osc, kind)()
    o.hertz = lfo(5.3, 4.5, p['hertz']) if vibrato else fixed(p['hertz'])
    o.phase = fixed(p['phase'])
    f = fx.LowPass(); f.input = o
    if sweep:
        c = fx.RingMod(); c.left = lfo(1.7, 0.4, [[0.6]]); c.right = fixed(p['cutoff'])
        f.cutoff = c
    else:
        f.cutoff = fixed(p['cutoff'])
    g = fx.Gain(); g.left = f
    if tremolo:
        d = fx.RingMod(); d.left = lfo(3.1, 0.3, [[0.7]]); d.right = fixed(p['gain'])
        g.right = d
    else:
        g.right = fixed(p['gain'])
    bus = ext.SumBus(); bus.input = g
    bus.get_state().gains = np.ascontiguousarray(p['pan'])
    return bus


def c2_modulated_oracle(p: dict, kind: str = 'Sawtooth', vibrato: bool = True, sweep: bool = True, tremolo: bool = True):
    from oracle import chain_ref as R

    def lfo(hz, depth, centre_row):
        return R.Binary('Mix', R.Binary('Gain', R.Osc('Sine', R.Fixed([[hz]])), R.Fixed([[2.0 * depth]])),
                        R.Fixed(2.0 * np.asarray(centre_row)), R.Fixed([[0.5]]))
    hertz = lfo(5.3, 4.5, p['hertz']) if vibrato else R.Fixed(p['hertz'])
    cutoff = R.Binary('RingMod', lfo(1.7, 0.4, [[0.6]]), R.Fixed(p['cutoff'])) if sweep else R.Fixed(p['cutoff'])
    gain = R.Binary('RingMod', lfo(3.1, 0.3, [[0.7]]), R.Fixed(p['gain'])) if tremolo else R.Fixed(p['gain'])
    return R.Binary('Gain', R.Filter('lp', R.Osc(kind, hertz, R.Fixed(p['phase'])), cutoff), gain), p['pan']
