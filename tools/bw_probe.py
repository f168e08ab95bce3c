#!/usr/bin/env python3
"""What do plain streaming kernels reach on this GPU?  (fill = write only, copy = read + write)"""
import torch
dev = torch.device("cuda:0")
def t(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (78, 315, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev); y = torch.empty_like(x)
    tf = t(lambda: y.fill_(1.0)); tc = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB: fill {mb/1024/tf/1e3*1.048576:6.2f} TB/s ({tf*1e6:.0f} us)   copy {2*mb/1024/tc/1e3*1.048576:6.2f} TB/s ({tc*1e6:.0f} us)", flush=True)
