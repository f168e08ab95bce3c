#!/usr/bin/env python3
"""Time the backward pieces of one ConvNeXt block (batch 16) per stage shape, beside the block's forward launches.
usage: backward_probe.py > profiles/rNN_backward_block.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import _lib as L, backward as B
from multitask_bonetumor_yolo_amd.engine import Act, Plan
dev = torch.device("cuda:0")
bf = torch.bfloat16


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def plan_of(build):
    p = Plan(dev)
    build(p)
    s = torch.cuda.current_stream().cuda_stream
    return lambda: p.run(stream=s)


for H, d in [(160, 96), (80, 192), (40, 384), (20, 768)]:
    N = 16
    t = lambda c: Act.of(torch.randn(N, H, H, c, device=dev).to(bf))
    x, a, tt, h, hg, o, dy = t(d), t(d), t(d), t(4 * d), t(4 * d), t(d), t(d)
    taps = torch.randn(49, d, device=dev).to(bf)
    W1, W2 = (torch.randn(4 * d, d, device=dev) / d ** 0.5).to(bf), (torch.randn(d, 4 * d, device=dev) / (4 * d) ** 0.5).to(bf)
    one, zero, vec = torch.ones(d, device=dev), torch.zeros(d, device=dev), torch.randn(d, device=dev)
    b4 = torch.zeros(4 * d, device=dev)
    rows = []
    rows.append(("fwd dw7x7 + LayerNorm", timed(plan_of(lambda p: p.dwconv(x, taps, a, 7, bias=vec, lnw=one, lnb=zero, eps=1e-6)))))
    rows.append(("fwd fc1 + GELU", timed(plan_of(lambda p: p.conv(tt, W1, hg, shift=b4, act=L.ACT_GELU)))))
    rows.append(("fwd fc2 + layer-scale + residual", timed(plan_of(lambda p: p.conv(hg, W2, o, scale=one, shift=zero, res=x)))))
    nf = len(rows)
    rows.append(("bwd channel sums (d gamma, d b2, d b1, d dw-bias)", timed(lambda: (B.channel_sum(dy, times=o), B.channel_sum(dy), B.channel_sum(h), B.channel_sum(a)))))
    rows.append(("bwd fc2 wgrad", timed(lambda: B.conv_wgrad(hg, dy, R=1, S=1, pad=0))))
    w2d, w1d = B.dgrad_weight(W2, 1, 1), B.dgrad_weight(W1, 1, 1)
    rows.append(("bwd fc2 dgrad", timed(plan_of(lambda p: B.conv_dgrad(p, dy, w2d, h, R=1, S=1, pad=0)))))
    rows.append(("bwd GELU'", timed(lambda: B.act_backward(h, hg, L.ACT_GELU, out=h))))
    rows.append(("bwd fc1 wgrad", timed(lambda: B.conv_wgrad(tt, h, R=1, S=1, pad=0))))
    rows.append(("bwd fc1 dgrad", timed(plan_of(lambda p: B.conv_dgrad(p, h, w1d, tt, R=1, S=1, pad=0)))))
    rows.append(("bwd LayerNorm (+ d gamma, d beta)", timed(lambda: B.layernorm_backward(a, tt, one, 1e-6))))
    rows.append(("bwd dw7x7 wgrad (one pass per filter row)", timed(lambda: B.dwconv_wgrad(x, a, 7))))
    tf = B.dwconv_dgrad_weight(taps, 7)
    rows.append(("bwd dw7x7 dgrad", timed(plan_of(lambda p: B.dwconv_dgrad(p, a, tf, o, 7, one, zero)))))
    fwd, bwd = sum(v for _, v in rows[:nf]), sum(v for _, v in rows[nf:])
    print(f"ConvNeXt block, batch {N}, {H}x{H}x{d} (bf16):  forward {fwd:.0f} us (unfused launches), backward pieces {bwd:.0f} us ({bwd / fwd:.1f}x)")
    for n, v in rows:
        print(f"    {v:9.1f} us  {n}")
    sys.stdout.flush()
