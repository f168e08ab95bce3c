#!/usr/bin/env python3
"""ConvNeXt conv_dw + LayerNorm in one launch (the product form) against conv_dw + bias alone (scale / shift form) at the four stage shapes of
the 640x640 batch-16 forward: what moving the LayerNorm into the consumer (the fused MLP's prologue) would leave in the depthwise kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import _lib as L
from multitask_bonetumor_yolo_amd.engine import Act, Plan
dev = torch.device("cuda:0")
for (H, C) in [(160, 96), (80, 192), (40, 384), (20, 768)]:
    N = 16
    x = torch.randn(N, H, H, C, device=dev).bfloat16()
    w = (torch.randn(49, C, device=dev) * 0.1).bfloat16()
    bias, lnw, lnb = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    one = torch.ones(C, device=dev)
    for tag in ("dw+LN", "dw+bias"):
        p = Plan(dev)
        y = p.new(N, H, H, C, L.BF16)
        if tag == "dw+LN":
            p.dwconv(Act.of(x), w, y, 7, bias=bias, lnw=lnw, lnb=lnb, eps=1e-6)
        else:
            p.dwconv(Act.of(x), w, y, 7, scale=one, shift=bias, act=L.ACT_NONE)
        for _ in range(3):
            p.run(stream=torch.cuda.current_stream().cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            p.run(stream=torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
        print(f"{H}x{H}x{C} {tag:8s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
