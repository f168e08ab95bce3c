"""Fused vs unfused ConvNeXt MLP at d = 384 (stage 2 of the 640x640 batch-16 forward: M = 25600 pixels) on one box in one process."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import _lib as L
from multitask_bonetumor_yolo_amd.engine import Act, Plan
from multitask_bonetumor_yolo_amd.model import _permute_hidden
dev = torch.device("cuda:0")
for (M, D, H) in [(25600, 384, 40), (102400, 192, 80), (409600, 96, 160)]:
    N = M // (H * H)
    t = torch.randn(N, H, H, D, device=dev).bfloat16()
    res = torch.randn(N, H, H, D, device=dev).bfloat16()
    w1 = (torch.randn(4 * D, D, device=dev) / D ** 0.5).bfloat16()
    w2 = (torch.randn(D, 4 * D, device=dev) / (4 * D) ** 0.5 * 0.1).bfloat16()
    b1, b2 = torch.randn(4 * D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
    p = Plan(dev)
    h = Act.of(torch.empty(N, H, H, 4 * D, device=dev, dtype=torch.bfloat16))
    y1, y2 = Act.of(torch.empty_like(t)), Act.of(torch.empty_like(t))
    p.conv(Act.of(t), w1, h, shift=b1, act=L.ACT_GELU_POLY, name="fc1")
    p.conv(h, w2, y1, shift=b2, res=Act.of(res), name="fc2")
    p.mlp_fused(Act.of(t), Act.of(res), w1, b1, _permute_hidden(w2).contiguous(), b2, y2, name="fused")
    for _ in range(3):
        p.run(stream=torch.cuda.current_stream().cuda_stream)
    acc = None
    for _ in range(5):
        ms = p.run_timed()
        acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
    torch.cuda.synchronize()
    us = [a / 5 * 1e3 for a in acc]
    fl = 2.0 * M * D * 4 * D * 2
    print(f"M={M} D={D}: fc1 {us[0]:.1f} + fc2 {us[1]:.1f} = {us[0]+us[1]:.1f} us ({fl/(us[0]+us[1])/1e6:.0f} TF/s) | fused {us[2]:.1f} us ({fl/us[2]/1e6:.0f} TF/s) | "
          f"max|diff| {(y1.buf.float()-y2.buf.float()).abs().max().item():.3g}", flush=True)
