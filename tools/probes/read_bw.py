"""What a pure-read pass reaches on this part (developer probe): torch reductions / copies at the sizes of the reduce kernels."""
import torch
dev = torch.device("cuda:0")


def t(name, fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:44s} {us:8.1f} us {nbytes / us / 1e3:8.1f} GB/s", flush=True)


for mb in (49, 99, 198, 396, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev).to(torch.bfloat16)
    xf = x.view(torch.int32)
    y = torch.empty_like(x)
    t(f"sum bf16 {mb} MB", lambda: x.sum(dtype=torch.float32), 2 * n)
    t(f"sum as int32 {mb} MB", lambda: xf.sum(), 2 * n)
    t(f"amax bf16 {mb} MB", lambda: x.amax(), 2 * n)
    t(f"colsum [rows][672] {mb} MB", lambda: x[: n // 672 * 672].view(-1, 672).sum(0, dtype=torch.float32), 2 * (n // 672 * 672))
    t(f"copy {mb} MB (read + write)", lambda: y.copy_(x), 4 * n)
    t(f"fill {mb} MB (write)", lambda: y.zero_(), 2 * n)
