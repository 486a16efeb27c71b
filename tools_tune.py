"""scratch tuning harness (not part of the product): time kernel variants on the C2 shapes"""
import os, sys, subprocess, json
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
