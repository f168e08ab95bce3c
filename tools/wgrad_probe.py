#!/usr/bin/env python3
"""Time the weight-gradient kernel on the network's stride-1 conv shapes (batch 16, 640 x 640 input)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import backward as B
from multitask_bonetumor_yolo_amd.engine import Act
dev = "cuda:0"
for name, (N, H, C, K, k) in {"proto.cv2 3x3 256->256 @160": (16, 160, 256, 256, 3), "c2f_p3.m 3x3 128->128 @80": (16, 80, 128, 128, 3),
                              "fc1.s2 1x1 384->1536 @40": (16, 40, 384, 1536, 1), "fc2.s2 1x1 1536->384 @40": (16, 40, 1536, 384, 1),
                              "head 3x3 256->64 @80": (16, 80, 256, 64, 3)}.items():
    x = Act.of(torch.randn(N, H, H, C, device=dev).bfloat16())
    dy = Act.of(torch.randn(N, H, H, K, device=dev).bfloat16())
    out = B.conv_wgrad(x, dy, R=k, S=k, pad=k // 2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5):
        B.conv_wgrad(x, dy, R=k, S=k, pad=k // 2, out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 5 * 1e3
    fl = 2.0 * N * H * H * C * K * k * k
    print(f"{name:32s} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
