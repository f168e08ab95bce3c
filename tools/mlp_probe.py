#!/usr/bin/env python3
"""Ablation of the fused ConvNeXt MLP kernel (MTBT_MLP_DEBUG bits: 1 no GELU, 2 no GEMM2, 4 no GEMM1, 8 no weight DMA)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
for D, M in [(96, 16 * 160 * 160), (192, 16 * 80 * 80)]:
    t = torch.randn(M, D, device=dev).bfloat16(); res = torch.randn(M, D, device=dev).bfloat16()
    w1 = torch.randn(4 * D, D, device=dev).bfloat16() * 0.1; w2 = torch.randn(D, 4 * D, device=dev).bfloat16() * 0.05
    b1 = torch.zeros(4 * D, device=dev); b2 = torch.zeros(D, device=dev); y = torch.empty_like(t)
    out = []
    for dbg in (0, 256, 512):
        os.environ["MTBT_MLP_DEBUG"] = str(dbg)
        s = torch.cuda.current_stream().cuda_stream
        f = lambda: lib.mtbt_convnext_mlp_fused(t.data_ptr(), res.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), M, D, s)
        for _ in range(3): f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        out.append(f"{dbg}:{a.elapsed_time(b) / 20 * 1e3:6.1f}")
    print(f"D={D}  " + "  ".join(out), flush=True)
