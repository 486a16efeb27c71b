"""kernel tuning harness (not part of the product): times kernel variants on the C2 shapes through the C ABI.
   TUNE={osc,bus,fused,fusedbus,walk} [SIG_BIQUAD_VARIANT=<vpt><ring>] [SIG_FUSED_VPT=n] [SIG_BIQUAD_WALK=n] python tools/tune_kernels.py"""
import os, sys, subprocess, json
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from signals_amd import _native, runtime
runtime.set_device('cuda:0')
V, N = 1024, 256
def time_biquad(K, reps=20):
    x = torch.rand((K * N, V), device='cuda') * 2 - 1
    out = torch.empty_like(x)
    cut = torch.tensor(np.random.default_rng(0).uniform(200, 8000, (1, V)), device='cuda')
    for _ in range(3):
        _native.biquad_coldstart('lp', 48000, 0, N, K, 100, cut, x, 0, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        _native.biquad_coldstart('lp', 48000, 0, N, K, 100, cut, x, 0, out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms, 8 * K * N * V / ms / 1e6
if __name__ == '__main__':
    for K in (256, 1024):
        ms, gbs = time_biquad(K)
        print(f'variant={os.environ.get("SIG_BIQUAD_VARIANT","default")} K={K}: {ms*1e3:.1f} us  {gbs:.0f} GB/s algorithmic', flush=True)

def time_osc(K, kind='Sine', reps=20):
    rng = np.random.default_rng(0)
    hz = torch.tensor(rng.uniform(55, 1760, (1, V)), device='cuda'); ph = torch.tensor(rng.uniform(0, 1, (1, V)), device='cuda')
    out = torch.empty((K * N, V), device='cuda')
    for _ in range(3): _native.osc_bank(kind, 0, 48000, hz, ph, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): _native.osc_bank(kind, 0, 48000, hz, ph, out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms, 4 * K * N * V / ms / 1e6

if __name__ == '__main__' and os.environ.get('TUNE') == 'osc':
    for kind in ('Sine', 'Square', 'Sawtooth', 'Triangle'):
        ms, gbs = time_osc(256, kind)
        print(f'osc {kind}: {ms*1e3:.1f} us {gbs:.0f} GB/s', flush=True)

def time_bus(K, reps=20, C=2):
    x = torch.rand((K * N, V), device='cuda') * 2 - 1
    rng = np.random.default_rng(0)
    th = rng.uniform(0, np.pi / 2, V)
    g = torch.tensor(np.stack([np.cos(th), np.sin(th)])[:C], device='cuda') if C else None
    out = torch.empty((K * N, max(C, 1)), device='cuda')
    for _ in range(3): _native.sum_bus(x, g, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): _native.sum_bus(x, g, out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms, 4 * K * N * V / ms / 1e6

if __name__ == '__main__' and os.environ.get('TUNE') == 'bus':
    for C in (2, 0):
        ms, gbs = time_bus(256, C=C)
        print(f'bus C={C}: {ms*1e3:.1f} us {gbs:.0f} GB/s', flush=True)


def time_fused(K, reps=20, gain=True):
    rng = np.random.default_rng(0)
    mk = lambda lo, hi: torch.tensor(rng.uniform(lo, hi, (1, V)), device='cuda')
    hz, ph, cut, g = mk(55, 1760), mk(0, 1), mk(200, 8000), mk(0, 1)
    out = torch.empty((K * N, V), device='cuda')
    f = lambda: _native.fused_osc_biquad('Sine', 'lp', 48000, 0, N, K, 100, hz, ph, cut, g if gain else None, out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

if __name__ == '__main__' and os.environ.get('TUNE') == 'fused':
    for K in (256, 1024):
        ms = time_fused(K)
        print(f'fused vpt={os.environ.get("SIG_FUSED_VPT","default")} K={K}: {ms*1e3:.1f} us  {K*N*V/ms/1e3:.0f} Msamples/s', flush=True)


def time_fused_bus(K, reps=20):
    rng = np.random.default_rng(0)
    mk = lambda lo, hi: torch.tensor(rng.uniform(lo, hi, (1, V)), device='cuda')
    hz, ph, cut, g = mk(55, 1760), mk(0, 1), mk(200, 8000), mk(0, 1)
    th = rng.uniform(0, np.pi / 2, V)
    pan = torch.tensor(np.stack([np.cos(th), np.sin(th)]), device='cuda')
    out = torch.empty((K * N, 2), device='cuda')
    ws = torch.empty(_native.lib().sig_fused_voice_bus_workspace(V, K * N, 2) // 8, dtype=torch.float64, device='cuda')
    f = lambda: _native.fused_voice_bus('Sine', 'lp', 48000, 0, N, K, 100, V, hz, ph, cut, g, pan, out, workspace=ws)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

if __name__ == '__main__' and os.environ.get('TUNE') == 'fusedbus':
    for K in (256, 1024):
        ms = time_fused_bus(K)
        print(f'fused+bus vpt={os.environ.get("SIG_FUSED_VPT","default")} K={K}: {ms*1e3:.1f} us  {K*N*V/ms/1e3:.0f} Msamples/s', flush=True)

if __name__ == '__main__' and os.environ.get('TUNE') == 'walk':
    for K in (256, 1024):
        ms, gbs = time_biquad(K)
        print(f'walk={os.environ.get("SIG_BIQUAD_WALK","auto")} K={K}: {ms*1e3:.1f} us  {gbs:.0f} GB/s algorithmic', flush=True)


def time_gain(K, reps=20):
    x = torch.rand((K * N, V), device='cuda') * 2 - 1
    g = torch.tensor(np.random.default_rng(0).uniform(0, 1, (1, V)), device='cuda')
    out = torch.empty_like(x)
    for _ in range(3): _native.elementwise('Gain', x, g, None, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): _native.elementwise('Gain', x, g, None, out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms, 8 * K * N * V / ms / 1e6

if __name__ == '__main__' and os.environ.get('TUNE') == 'gain':
    for K in (256, 1024):
        ms, gbs = time_gain(K)
        print(f'gain K={K}: {ms*1e3:.1f} us {gbs:.0f} GB/s', flush=True)
