"""HBM ceilings seen from torch: pure write (fill), copy (1R:1W), pure read (sum) - context for the write-heavy kernels."""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (100, 512, 2048):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
    a.normal_()
    us = t(lambda: b.fill_(1.0)); print(f"{mb:5d} MB fill   {us:8.1f} us  {mb * 1.048576 / us * 1e3:7.0f} GB/s written")
    us = t(lambda: b.copy_(a)); print(f"{mb:5d} MB copy   {us:8.1f} us  {2 * mb * 1.048576 / us * 1e3:7.0f} GB/s moved")
    us = t(lambda: a.sum()); print(f"{mb:5d} MB sum    {us:8.1f} us  {mb * 1.048576 / us * 1e3:7.0f} GB/s read")
