import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for mb in (25, 100, 400, 1600):
    n = mb*1024*1024//2
    a = torch.randn(n, device='cuda', dtype=torch.bfloat16); b = torch.empty_like(a)
    us = t(lambda: b.copy_(a)); print(f"copy {mb}MB: {us:.1f} us  {2*mb*1.048576/us*1e3:.0f} GB/s")
    us = t(lambda: a.mul_(1.0001)); print(f"inplace mul {mb}MB: {us:.1f} us  {2*mb*1.048576/us*1e3:.0f} GB/s")
    us = t(lambda: torch.nn.functional.gelu(a, out=b) if False else b.copy_(a).mul_(2)); 
    us = t(lambda: b.fill_(1.0)); print(f"fill {mb}MB: {us:.1f} us  {mb*1.048576/us*1e3:.0f} GB/s")
    us = t(lambda: a.sum()); print(f"sum {mb}MB: {us:.1f} us  {mb*1.048576/us*1e3:.0f} GB/s")
