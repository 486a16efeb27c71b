#!/usr/bin/env python3
"""Non-headline configurations of BASELINE.json (C3, C5) on one MI355X: Msamples/s and per-kernel algorithmic
GB/s through the batched engine.  Parity for these graphs is in tests/test_gpu_engine.py; this only times them.

    python tools/measure_configs.py            (needs a GPU)
"""
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

RATE = 48000


def fixed(v):
    from signals_amd.chain.fixed import Fixed
    f = Fixed()
    f.get_state().value = np.ascontiguousarray(np.array(v, ndmin=2, dtype=float))
    return f


def c3(V):
    """Saw -> LowPass -> LowPass -> (x ADSR) -> SumBus, N = 1024"""
    from signals_amd.chain import ext, fx, osc
    rng = np.random.default_rng(0)
    o = osc.Sawtooth(); o.hertz = fixed(rng.uniform(55, 1760, (1, V))); o.phase = fixed(rng.uniform(0, 1, (1, V)))
    f1 = fx.LowPass(); f1.input = o; f1.cutoff = fixed(rng.uniform(200, 8000, (1, V)))
    f2 = fx.LowPass(); f2.input = f1; f2.cutoff = fixed(rng.uniform(200, 8000, (1, V)))
    env = ext.ADSR()
    for name, (lo, hi) in dict(attack=(0.001, 0.05), decay=(0.01, 0.2), sustain=(0.2, 0.9), release=(0.05, 0.5),
                               gate_on=(0.0, 0.5), gate_off=(1.0, 4.0)).items():
        setattr(env, name, fixed(rng.uniform(lo, hi, (1, V))))
    rm = fx.RingMod(); rm.left = f2; rm.right = env
    bus = ext.SumBus(); bus.input = rm
    return bus, 1, 1024, 256, {'osc_bank': 4, 'biquad_coldstart': 8, 'adsr': 4, 'adsr_apply': 8, 'fused_osc_biquad': 4, 'elementwise': 12, 'sum_bus': 4, 'biquad_bus': 4}


def c5(V):
    """Sine -> LowPass -> MixMatrix(64x64), V = 4096, N = 256"""
    from signals_amd.chain import ext, fx, osc
    rng = np.random.default_rng(0)
    o = osc.Sine(); o.hertz = fixed(rng.uniform(55, 1760, (1, V))); o.phase = fixed(rng.uniform(0, 1, (1, V)))
    f = fx.LowPass(); f.input = o; f.cutoff = fixed(rng.uniform(200, 8000, (1, V)))
    mm = ext.MixMatrix(); mm.input = f
    mm.get_state().matrix = np.linalg.qr(rng.standard_normal((64, 64)))[0]
    return mm, V, 256, 64, {'fused_osc_biquad': 4, 'mix_matrix': 8, 'osc_bank': 4, 'biquad_coldstart': 8, 'fused_osc_biquad_mix': 4}


def run(name, build, V, steps=10):
    from signals_amd.engine import BatchRenderer, KernelTimer
    node, channels, N, K, algo = build(V)
    timer = KernelTimer()
    r = BatchRenderer(node, channels, RATE, timer=timer)
    pos = 0
    for _ in range(2):
        r.render(pos, N, K); pos += N * K
    torch.cuda.synchronize(); timer.reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(pos, N, K); pos += N * K
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {'config': name, 'voices': V, 'block_frames': N, 'blocks_per_batch': K,
           'Msamples_per_s': V * N * K * steps / dt / 1e6, 'ms_per_batch': dt / steps * 1e3, 'kernels': {}}
    for k, e in timer.summary().items():
        avg = e['ms'] / e['calls']
        bpu = algo.get(k.split('[')[0], 0)
        out['kernels'][k] = {'avg_us': avg * 1e3, 'algo_GBs': bpu * (e['units'] / e['calls']) / (avg * 1e-3) / 1e9}
    print(json.dumps(out))


if __name__ == '__main__':
    from signals_amd import runtime
    runtime.set_device('cuda:0')
    run('C3 saw->LP->LP->xADSR->bus', c3, 1024)
    run('C5 sine->LP->MixMatrix', c5, 4096)
