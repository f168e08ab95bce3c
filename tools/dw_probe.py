#!/usr/bin/env python3
"""Ablation of the depthwise 7x7 + LayerNorm kernel on the four ConvNeXt stage shapes (MTBT_DW_DEBUG bits:
1 = no FMA loop, 2 = no staging DMA, 4 = no LayerNorm / stores)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd.engine import Act, Plan
DEV = torch.device("cuda:0")
def bench(N, H, C, iters=30):
    p = Plan(DEV)
    x = Act.of(torch.randn(N, H, H, C, device=DEV).bfloat16())
    y = Act.of(torch.empty(N, H, H, C, device=DEV, dtype=torch.bfloat16))
    w = torch.randn(49, C, device=DEV).bfloat16()
    b, lw, lb = (torch.randn(C, device=DEV) for _ in range(3))
    p.dwconv(x, w, y, 7, bias=b, lnw=lw, lnb=lb, eps=1e-6)
    for _ in range(3): p.run(stream=torch.cuda.current_stream().cuda_stream)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): p.run(stream=torch.cuda.current_stream().cuda_stream)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3
for (H, C) in [(160, 96), (80, 192), (40, 384), (20, 768)]:
    r = []
    for dbg in (0, 1, 2, 4, 7):
        os.environ["MTBT_DW_DEBUG"] = str(dbg)
        r.append(f"{dbg}:{bench(16, H, C):6.1f}")
    print(f"H={H:3d} C={C:3d}  " + "  ".join(r), flush=True)
