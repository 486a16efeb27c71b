#!/usr/bin/env python3
"""Non-headline configurations of BASELINE.json (C3, C5) and a modulated C2 voice on one MI355X: Msamples/s and per-kernel algorithmic
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


import bench_configs as cfg


def c3(V):
    """Saw -> LowPass -> LowPass -> (x ADSR) -> SumBus, N = 1024"""
    return cfg.c3_graph(cfg.c3_params(V)), 1, 1024, 1024, ALGO


def c2m(V, K=1024):
    """C2's voices with block-rate vibrato + cutoff sweep + tremolo (Sawtooth), N = 256"""
    return cfg.c2_modulated_graph(cfg.c2_params(V), 'Sawtooth'), 2, 256, K, ALGO


def c2s(V, K=4096):
    """C2's Sine voices with a block-rate cutoff sweep + tremolo: the closed form with per-block filter constants, N = 256"""
    return cfg.c2_modulated_graph(cfg.c2_params(V), 'Sine', vibrato=False), 2, 256, K, ALGO


def vp(V, K=1024):
    """a shape no fused kernel covers (RingMod of two filtered oscillators -> Gain -> SumBus): one interpreted launch, N = 256"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('time_voice_program', ROOT / 'tools' / 'time_voice_program.py')
    tvp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tvp)
    return tvp.shapes(V)['ringmod_of_two_filtered'](), 2, 256, K, ALGO


def c5(V, K=256):
    """Sine -> LowPass -> MixMatrix(64x64), V = 4096, N = 256"""
    return cfg.c5_graph(cfg.c5_params(V)), V, 256, K, ALGO


ALGO = {'osc_bank': 4, 'biquad_coldstart': 8, 'adsr': 4, 'adsr_apply': 8, 'fused_osc_biquad': 4, 'elementwise': 12, 'sum_bus': 4,
        'biquad_bus': 4, 'mix_matrix': 8, 'fused_osc_biquad_mix': 4}


def run(name, build, V, steps=10, **engine_options):
    from signals_amd.engine import BatchRenderer, KernelTimer
    node, channels, N, K, algo = build(V)
    timer = KernelTimer()
    r = BatchRenderer(node, channels, RATE, timer=timer, **engine_options)
    pos = 0
    t_end, warm = time.perf_counter() + 0.2, 0            # clocks up (they take ~30 ms of load to settle); the first render of a
    while time.perf_counter() < t_end or warm < 20:      # process also loads the code objects, which can take longer than that
        r.render(pos, N, K); pos += N * K; warm += 1
        torch.cuda.synchronize()
    timer.reset()
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    run('C3 saw->LP->LP->xADSR->bus', c3, 1024, steps)
    run('C5 sine->LP->MixMatrix', c5, 4096, steps)
    run('C2 voices with vibrato + cutoff sweep + tremolo', c2m, 1024, steps)
    run('C2 Sine voices with cutoff sweep + tremolo', c2s, 1024, steps)
    run('RingMod of two filtered oscillators (voice program)', vp, 1024, steps)
    run('RingMod of two filtered oscillators (kernel specialised for the program)', vp, 1024, steps, specialise=True)
    run('C2 voices with vibrato + cutoff sweep + tremolo (control program specialised)', c2m, 1024, steps, specialise=True)
