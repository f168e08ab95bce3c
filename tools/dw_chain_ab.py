#!/usr/bin/env python3
"""Depthwise 7x7 + LayerNorm on the multi-chunk ConvNeXt stages: a hash of the output (to compare two library builds bit for bit under
tools/lib_ab_cmd.sh) and the time per launch inside a captured 32-launch chain."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MTBT_LANES"] = "1"
from multitask_bonetumor_yolo_amd.engine import Act, Plan
DEV = torch.device("cuda:0")
for (N, H, W, C) in [(16, 40, 40, 384), (16, 20, 20, 768), (16, 80, 80, 192), (3, 24, 20, 384), (2, 12, 28, 768), (1, 9, 11, 320)]:
    torch.manual_seed(C + H)
    a = Act.of(torch.randn(N, H, W, C, device=DEV).bfloat16())
    b = Act.of(torch.empty(N, H, W, C, device=DEV, dtype=torch.bfloat16))
    w = (torch.randn(49, C, device=DEV) / 7).bfloat16()
    bias, lw, lb = torch.randn(C, device=DEV) * 0.1, torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    p1 = Plan(DEV)
    p1.dwconv(a, w, b, 7, bias=bias, lnw=lw, lnb=lb, eps=1e-6)
    p1.run(stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = hashlib.sha1(b.buf.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    p = Plan(DEV)
    for i in range(32):
        x, y = (a, b) if i % 2 == 0 else (b, a)
        p.dwconv(x, w, y, 7, bias=bias, lnw=lw, lnb=lb, eps=1e-6)
    s = torch.cuda.Stream(DEV)
    with torch.cuda.stream(s):
        p.run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            p.run()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    print(f"N={N} {H}x{W} C={C}: sha1 {h}   {e0.elapsed_time(e1) / 10 / 32 * 1e3:6.2f} us per launch in a chain", flush=True)
