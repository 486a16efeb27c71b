import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_gpu_fused_walker as T
from signals_amd import _native
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    V = 64 * int(rng.integers(1, 6))
    N = int(rng.choice([17, 32, 50, 96, 100, 128, 256, 300]))
    K = int(rng.integers(1, 12))
    pos = int(rng.choice([0, 3, 99, 100, 512, 4096, 48000 * 30]))
    kind = str(rng.choice(['Sine', 'Sine', 'Sawtooth', 'Square', 'Triangle']))
    btype = str(rng.choice(['lp', 'hp']))
    gain = bool(rng.integers(0, 2))
    steady = int(rng.choice([-1, 1, 0, 3]))
    span = int(rng.choice([0, 1, 2, 4, 8]))
    p = T.params(V, int(rng.integers(0, 10000)))
    if rng.integers(0, 3) == 0: p['hertz'][0, 5] = 2.0      # a voice outside the closed form's range
    M = torch.tensor(rng.standard_normal((64, 64)) * 0.3, dtype=torch.float32, device='cuda')
    T.geometry(1, span, steady=steady)
    out = torch.full((K * N, V), float('nan'), device='cuda')
    _native.fused_osc_biquad_mix(kind, btype, T.RATE, pos, N, K, T.CTX, T.dev(p['hertz']), T.dev(p['phase']), T.dev(p['cutoff']),
                                 T.dev(p['gain']) if gain else None, M, out)
    got = out.cpu().numpy()
    pp = dict(p)
    if not gain: pp['gain'] = np.ones_like(p['gain'])
    ref = T.oracle_chain(kind, btype, pp, pos, N, K)
    ref = (ref.reshape(K * N, V // 64, 64) @ M.cpu().numpy().astype(np.float64)).reshape(K * N, V)
    scale = max(1e-3, float(np.abs(ref).max()))
    err = T.maxerr(got, T.f32(ref)) if np.isfinite(got).all() else float('inf')
    ok = err < 2e-6 * max(1.0, scale)
    bad += not ok
    print('OK ' if ok else 'BAD', case, kind, btype, 'V', V, 'N', N, 'K', K, 'pos', pos, 'gain', gain, 'steady', steady, 'span', span, 'err %.2e' % err, 'scale %.2f' % scale)
T.geometry(0, 0, steady=-1)
print('bad', bad)
