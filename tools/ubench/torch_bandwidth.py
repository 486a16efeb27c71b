import torch
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for mib in (256, 1024):
    n = mib * 2**20 // 4
    x = torch.rand(n, device='cuda'); y = torch.empty_like(x)
    ms = t(lambda: y.copy_(x)); print(f'{mib} MiB copy   : {ms*1e3:7.1f} us  {2*n*4/ms/1e6:6.0f} GB/s')
    ms = t(lambda: y.fill_(1.0)); print(f'{mib} MiB fill   : {ms*1e3:7.1f} us  {n*4/ms/1e6:6.0f} GB/s')
    ms = t(lambda: x.sum()); print(f'{mib} MiB sum    : {ms*1e3:7.1f} us  {n*4/ms/1e6:6.0f} GB/s')
    ms = t(lambda: torch.mul(x, 0.5, out=y)); print(f'{mib} MiB scale  : {ms*1e3:7.1f} us  {2*n*4/ms/1e6:6.0f} GB/s')
