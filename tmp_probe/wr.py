import torch,time
x=torch.empty(99*1024*1024//2,dtype=torch.bfloat16,device='cuda')
for f in (lambda: x.zero_(), lambda: x.fill_(1.0)):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3; print(f"write 99MB: {us:.1f} us  {x.numel()*2/us/1e3:.0f} GB/s")
