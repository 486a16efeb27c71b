#!/usr/bin/env python3
"""Timing harness behind the round-2 statements in DESIGN.md §7 (not part of the product; needs a GPU):

    python tools/time_fused_kernels.py steady     closed-form bus kernel: every slot dropped / none dropped / k slots live,
                                                  and the launch geometry (voices per lane x blocks per lane) on the bench's voices
    python tools/time_fused_kernels.py walker     the row walker per waveform (closed form off), K = 1024 and 4096
    python tools/time_fused_kernels.py mix        config 5: closed form into the MixMatrix sink, per span

Tuning builds of one kernel file (`FILE=x.hip tools/build_variant.sh name -D...`) are loaded with SIG_LIB_PATH.
"""
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import bench
import bench_configs as cfg
from signals_amd import _native, runtime

runtime.set_device('cuda:0')
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')


def timed(f, label, units, warm=20, n=60):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{label}: {dt * 1e6:.1f} us  {units / dt / 1e12:.2f} T voice-samples/s', flush=True)


def ordered(cut, V, vpt):
    """engine.py: ordered_by_cutoff"""
    order = np.argsort(cut, kind='stable'); tile = 64 * vpt; tiles = V // tile
    perm = order.copy(); q = np.arange(tiles * tile); group, lane = q // 64, q % 64
    perm[((group % tiles) * 64 + lane) * vpt + group // tiles] = order[q]
    return perm


def steady():
    V, N, K = 1024, 256, 4096
    p = bench.synth_params(V)
    out = torch.empty((N * K, 2), device='cuda')
    ws = torch.empty(_native.lib().sig_fused_voice_bus_workspace(V, N * K, 2) // 8, dtype=torch.float64, device='cuda')
    consts = torch.empty(_native.lib().sig_fused_voice_consts_size(V) // 8, dtype=torch.float64, device='cuda')

    def run(perm, cut, label):
        hz, ph, g, pan = (dev(p[k][:, perm]) for k in ('hertz', 'phase', 'gain', 'pan'))
        c = dev(cut)
        call = lambda ready: _native.fused_voice_bus('Sine', 'lp', 48000, N * K, N, K, 100, V, hz, ph, c, g, pan, out, workspace=ws,
                                                     consts=consts, consts_ready=ready)
        call(False)
        timed(lambda: call(True), label, V * N * K)
    ident = np.arange(V)
    run(ident, np.full((1, V), 6000.0), 'every slot dropped from row 0 (all cutoffs 6 kHz)')
    run(ident, np.full((1, V), 300.0), 'no slot ever dropped (all cutoffs 300 Hz)')
    run(ident, p['cutoff'], 'bench cutoffs, voices as drawn')
    for m in range(1, 8):
        cut = np.full((1, V), 6000.0).reshape(2, 64, 8); cut[:, :, :m] = 300.0
        run(ident, cut.reshape(1, V), f'{m} slots live')
    for vpt in (16, 8, 4):
        perm = ordered(p['cutoff'].reshape(-1), V, vpt)
        for span in (8, 4, 2, 1):
            _native.set_fused_tuning(vpt, span, 1, 0)
            run(perm, p['cutoff'][:, perm], f'bench cutoffs ordered, {vpt} voices x {span} blocks per lane')
    _native.set_fused_tuning()


def walker():
    V, N = 1024, 256
    p = bench.synth_params(V)
    hz, ph, g, pan, cut = (dev(p[k]) for k in ('hertz', 'phase', 'gain', 'pan', 'cutoff'))
    for K in (1024, 4096):
        out = torch.empty((N * K, 2), device='cuda')
        ws = torch.empty(_native.lib().sig_fused_voice_bus_workspace(V, N * K, 2) // 8, dtype=torch.float64, device='cuda')
        for kind in ('Sine', 'Sawtooth', 'Square', 'Triangle'):
            _native.set_fused_tuning(0, 0, 0, 0)
            timed(lambda: _native.fused_voice_bus(kind, 'lp', 48000, N * K, N, K, 100, V, hz, ph, cut, g, pan, out, workspace=ws),
                  f'{kind} walker K={K}', V * N * K)
    _native.set_fused_tuning()


def mix():
    V, N, K = 4096, 256, 64
    p = cfg.c5_params(V)
    hz, ph = dev(p['hertz']), dev(p['phase'])
    M = torch.tensor(p['matrix'], dtype=torch.float32, device='cuda')
    out = torch.empty((N * K, V), device='cuda')
    for label, cut in (('bench cutoffs', p['cutoff']), ('all 6 kHz (nothing live)', np.full((1, V), 6000.0))):
        c = dev(cut)
        for span in (0, 1, 2, 4, 8):
            _native.set_fused_tuning(0, span, -1, 0)
            timed(lambda: _native.fused_osc_biquad_mix('Sine', 'lp', 48000, N * K, N, K, 100, hz, ph, c, None, M, out),
                  f'{label}, span {span or "default"}', V * N * K)
    _native.set_fused_tuning(0, 0, 0, 0)
    c = dev(p['cutoff'])
    timed(lambda: _native.fused_osc_biquad_mix('Sine', 'lp', 48000, N * K, N, K, 100, hz, ph, c, None, M, out),
          'row walker into the sink (closed form off)', V * N * K)
    _native.set_fused_tuning()


if __name__ == '__main__':
    print('lib', os.environ.get('SIG_LIB_PATH', 'product'))
    {'steady': steady, 'walker': walker, 'mix': mix}[sys.argv[1] if len(sys.argv) > 1 else 'steady']()
