import sys, torch, numpy as np
sys.path.insert(0, '.')
from signals_amd import _native, runtime
runtime.set_device('cuda:0')
x = torch.rand((16384, 4096), device='cuda') * 2 - 1
M = torch.tensor(np.linalg.qr(np.random.default_rng(0).standard_normal((64, 64)))[0], dtype=torch.float32, device='cuda')
out = torch.empty_like(x)
for _ in range(3): _native.mix_matrix(x, M, out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): _native.mix_matrix(x, M, out)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
ref = (x.double().reshape(16384, 64, 64) @ M.double()).reshape(16384, 4096)
print(f'mix_matrix 67M samples: {ms*1e3:.1f} us  {8*x.numel()/ms/1e6:.0f} GB/s  {128*x.numel()/ms/1e9:.1f} TFLOP/s  max err {float((out.double()-ref).abs().max()):.2e}')
