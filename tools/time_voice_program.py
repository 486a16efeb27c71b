#!/usr/bin/env python3
"""Throughput of graphs that run as ONE interpreted launch (sig_voice_program) against the same graph one kernel per node:
1024 voices, 48 kHz, under a stereo bus.  Prints one JSON object per shape.

    python tools/time_voice_program.py [blocks per batch] [block frames]        (needs a GPU)
"""
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import bench_configs as cfg

RATE = 48000


def fixed(v):
    return cfg.fixed(v)


def lfo(hz, depth, centre):
    from signals_amd.chain import fx, osc
    s = osc.Sine(); s.hertz = fixed([[hz]])
    g = fx.Gain(); g.left = s; g.right = fixed([[2.0 * depth]])
    m = fx.Mix(); m.left = g; m.right = fixed(2.0 * np.asarray(centre)); m.mix = fixed([[0.5]])
    return m


def shapes(V):
    from signals_amd.chain import ext, fx, osc
    p = cfg.c2_params(V)
    rng = np.random.default_rng(1)
    cut2, cut3 = rng.uniform(200, 8000, (1, V)), rng.uniform(200, 8000, (1, V))
    hz2 = rng.uniform(55, 1760, (1, V))

    def o(kind, hz=None):
        n = getattr(osc, kind)(); n.hertz = fixed(p['hertz'] if hz is None else hz); n.phase = fixed(p['phase'])
        return n

    def filt(kind, src, cut):
        f = getattr(fx, kind)(); f.input = src; f.cutoff = cut if not isinstance(cut, np.ndarray) else fixed(cut)
        return f

    def gain(src, row):
        g = fx.Gain(); g.left = src; g.right = row if not isinstance(row, np.ndarray) else fixed(row)
        return g

    def bus(top):
        b = ext.SumBus(); b.input = top; b.get_state().gains = np.ascontiguousarray(p['pan'])
        return b

    def amp_after_filter():
        a = fx.Amp(); a.left = filt('LowPass', o('Sawtooth'), p['cutoff']); a.right = fixed(np.full((1, V), 1.0))
        return bus(gain(a, p['gain']))

    def ringmod_of_two_filtered():
        rm = fx.RingMod(); rm.left = filt('LowPass', o('Sawtooth'), p['cutoff']); rm.right = filt('HighPass', o('Triangle', hz2), cut2)
        return bus(gain(rm, p['gain']))

    def mix_after_filter():
        m = fx.Mix(); m.left = filt('LowPass', o('Sawtooth'), p['cutoff']); m.right = o('Sine', hz2); m.mix = fixed(np.full((1, V), 0.7))
        return bus(gain(m, p['gain']))

    def three_filters():
        return bus(gain(filt('LowPass', filt('HighPass', filt('LowPass', o('Sawtooth'), p['cutoff']), cut2), cut3), p['gain']))

    def modulated_cascade():
        f1 = filt('LowPass', o('Sawtooth'), lfo(1.7, 150.0, p['cutoff']))
        return bus(gain(filt('LowPass', f1, cut2), lfo(3.1, 0.3, p['gain'])))

    def two_filters_tremolo():
        return bus(gain(filt('LowPass', filt('LowPass', o('Sawtooth'), p['cutoff']), cut2), lfo(3.1, 0.3, p['gain'])))

    def osc_gain_only():
        return bus(gain(o('Sawtooth'), p['gain']))

    return {f.__name__: f for f in (amp_after_filter, ringmod_of_two_filtered, mix_after_filter, three_filters, modulated_cascade,
                                    two_filters_tremolo, osc_gain_only)}


def rate(build, V, N, K, program, steps=10, specialise=False):
    from signals_amd import _native
    from signals_amd.engine import BatchRenderer, KernelTimer
    _native.voice_program_use_attached(bool(specialise))    # (attached kernels are process-wide: an interpreter run must not pick one up)
    timer = KernelTimer(sample_every=4)
    r = BatchRenderer(build(), 2, RATE, timer=timer, fuse_program=program, specialise=specialise)
    pos = 0
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        r.render(pos, N, K); pos += N * K
        torch.cuda.synchronize()
    timer.reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(pos, N, K); pos += N * K
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return V * N * K / dt / 1e12, {k: round(e['ms'] / e['calls'] * 1e3, 1) for k, e in timer.summary().items()}


if __name__ == '__main__':
    from signals_amd import runtime
    runtime.set_device('cuda:0')
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    V = 1024
    for name, build in shapes(V).items():
        fast, launches = rate(build, V, N, K, True)
        spec, spec_launches = rate(build, V, N, K, 'always', specialise=True)
        slow, per_node = rate(build, V, N, K, False, steps=4)
        print(json.dumps({'shape': name, 'voices': V, 'block_frames': N, 'blocks_per_batch': K, 'T_voice_samples_per_s': round(fast, 3),
                          'launches_us': launches, 'specialised_T': round(spec, 3), 'specialised_launches_us': spec_launches,
                          'per_node_T': round(slow, 3), 'per_node_launches_us': per_node}), flush=True)
