#!/usr/bin/env python3
"""Tile sweep of EVERY implicit-GEMM conv shape of the batch-16 inference plan, timed the way the step runs them: 32 launches on one stream
in a captured HIP graph, operands rotating through a ring of buffers larger than the L2s (so every launch fetches out of the Infinity
Cache / HBM like a launch that follows its producer), no host in the loop.  tools/conv_tune.py's eager same-buffer loop is host-bound
below ~15 us per launch and L2-warm: its verdicts on the small shapes (64-byte K-steps for short rows, 128 x 64 tiles) do not hold here.
    python tools/chain_tune.py [--img 640] [--batch 16] [filter]     -> per shape: default choice, best hint, launches per step, saving"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MTBT_LANES"] = "1"
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_  # noqa: E402
from multitask_bonetumor_yolo_amd import _lib as L  # noqa: E402
from multitask_bonetumor_yolo_amd.engine import Act, Plan  # noqa: E402

DEV = torch.device("cuda:0")
CHAIN, RING = 32, 6
DT = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16}


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def plan_shapes(img, batch):
    model = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval()
    model.set_compute_dtype(torch.bfloat16)
    x = torch.rand(batch, 3, img, img, device=DEV)
    with torch.no_grad():
        model(x, "infer")
    torch.cuda.synchronize()
    plans = list(model.__dict__["_plans"].values())
    shapes = {}
    for pl in plans:
        plan = pl if isinstance(pl, Plan) else getattr(pl, "plan", None) or next(v for v in (pl if isinstance(pl, (tuple, list)) else vars(pl).values()) if isinstance(v, Plan))
        for l in plan.launches:
            a = l.keep[0] if l.keep else None
            if not isinstance(a, L.ConvArgs):
                continue
            key = (a.N, a.H, a.W, a.C, a.K, a.R, a.stride, a.pad, a.act, bool(a.res), a.dtype, a.out_dtype, a.out_mode, bool(a.scale))
            shapes.setdefault(key, []).append(l.name)
    del model
    torch.cuda.empty_cache()
    return shapes


def chain_time(key, hint):
    N, H, W, C, K, R, stride, pad, act, has_res, dt, odt, om, has_scale = key
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    p = Plan(DEV)
    Kq = K // 4 if om == L.OUT_CONVT2X2 else K
    Hy, Wy = (2 * Ho, 2 * Wo) if om == L.OUT_CONVT2X2 else (Ho, Wo)
    xs = [Act.of(torch.randn(N, H, W, C, device=DEV).to(DT[dt])) for _ in range(RING)]
    ys = [Act.of(torch.empty(N, Hy, Wy, Kq, device=DEV, dtype=DT[odt])) for _ in range(RING)]
    rs = [Act.of(torch.randn(N, Hy, Wy, Kq, device=DEV).to(DT[dt])) for _ in range(RING)] if has_res else None
    w = (torch.randn(K, R * R * C, device=DEV) / (R * R * C) ** 0.5).to(DT[dt])
    sc, sh = (torch.ones(K, device=DEV) if has_scale else None), torch.zeros(K, device=DEV)    # (a head's output conv has a bias only: the streaming kernel's case)
    try:
        for i in range(CHAIN):
            p.conv(xs[i % RING], w, ys[i % RING], R=R, S=R, stride=stride, pad=pad, scale=sc, shift=sh, act=act, res=rs[i % RING] if rs else None,
                   out_mode=om, tile_hint=hint)
        s = torch.cuda.Stream(DEV)
        with torch.cuda.stream(s):
            p.run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                p.run()
            for _ in range(2):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(8):
                g.replay()
            e1.record(s)
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 8 / CHAIN * 1e3
    except RuntimeError:
        return None


def main():
    img, batch = arg("--img", 640), arg("--batch", 16)
    flt = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
    shapes = plan_shapes(img, batch)
    hints = [(tc, tp, nar) for tc, tp in [(128, 128), (128, 64), (64, 128), (64, 64), (96, 128), (96, 64), (32, 128), (32, 64)] for nar in (0, 1)]
    total_default = total_best = 0.0
    print(f"# {len(shapes)} distinct conv shapes, batch {batch} x {img}^2; per launch in a {CHAIN}-launch graph chain", flush=True)
    for key, names in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
        N, H, W, C, K, R, stride, pad, act, has_res, dt, odt, om, has_scale = key
        label = f"{R}x{R}/{stride} {C}->{K} @{H}x{W} act{act}{' +res' if has_res else ''}{' f32out' if odt == L.F32 else ''}{' convT' if om else ''}"
        if flt and not any(f in label or any(f in n for n in names) for f in flt):
            continue
        t0 = chain_time(key, 0)
        res = []
        for tc, tp, nar in hints:
            if (tc == 96 and K % 96) or (tc == 128 and K < 96) or (tc == 64 and K % 64 and K > 64) or (tc == 32 and K > 32):
                continue
            if nar == 0 and C % (64 if dt != L.F32 else 32):
                continue
            t = chain_time(key, (2 << 28) | (nar << 27) | (1 << 26) | (tc << 16) | tp)
            if t is not None:
                res.append((t, f"{tc}x{tp}{'n' if nar else 'w'}"))
        res.sort()
        best = res[0] if res else (t0, "-")
        n = len(names)
        total_default += n * t0
        total_best += n * min(t0, best[0])
        top = "  ".join(f"{h} {t:.1f}" for t, h in res[:4])
        print(f"{label:44s} x{n:2d}  default {t0:6.1f} us | {top} | saves {n * max(0.0, t0 - best[0]):6.1f} us   [{names[0]}]", flush=True)
    print(f"# sum over the plan: default {total_default:.0f} us, best-per-shape {total_best:.0f} us", flush=True)


if __name__ == "__main__":
    main()
