#!/usr/bin/env python3
"""Per-launch time of a conv INSIDE A DEPENDENT CHAIN, as the step runs it: L launches ping-ponging between two buffers (each reads what
the previous one wrote -- on another XCD, so out of the Infinity Cache, not a warm L2), captured in one HIP graph and replayed.  The eager
same-buffer loop of conv_tune.py is host-bound below ~15 us and keeps the operands L2-warm: it hid what bounds the small pyramid levels.
    python tools/chain_probe.py [filter]
tile hints: (nbuf << 28) | (narrow << 27) | (TC << 16) | TP.  (Round 3 ran this with a deep-pipeline unit -- 4 / 6 / 8 LDS stages on 64x64, 128x64,
64x32, 128x32 tiles -- built in: profiles/r03_chain_probe_deep_pipelines.txt; no gain, the unit is gone and nbuf >= 3 runs two stages.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MTBT_LANES"] = "1"
from multitask_bonetumor_yolo_amd import _lib as L  # noqa: E402
from multitask_bonetumor_yolo_amd.engine import Act, Plan  # noqa: E402

DEV = torch.device("cuda:0")
CHAIN = int(os.environ.get("CHAIN", "32"))
SHAPES = {
    # name: (N, H, W, C(=K), k)
    "neck 3x3 128->128 @20": (16, 20, 20, 128, 3),
    "c2f_p5 3x3 256->256 @20": (16, 20, 20, 256, 3),
    "neck 3x3 128->128 @40": (16, 40, 40, 128, 3),
    "c2f_p4 3x3 192->192 @40": (16, 40, 40, 192, 3),
    "neck 1x1 256->256 @20": (16, 20, 20, 256, 1),
    "neck 1x1 256->256 @40": (16, 40, 40, 256, 1),
    "c2f_p3 3x3 128->128 @80": (16, 80, 80, 128, 3),
}
HINTS = [("default", 0)] + [(f"nbuf={nb} {tc}x{tp}", (nb << 28) | (tc << 16) | tp | (1 << 26))
                            for tc, tp in [(64, 64), (128, 64), (128, 128), (64, 128)] for nb in (2,)]


def chain_time(shape, hint, dtype=torch.bfloat16):
    N, H, W, C, k = shape
    p = Plan(DEV)
    a = Act.of(torch.randn(N, H, W, C, device=DEV).to(dtype))
    b = Act.of(torch.empty(N, H, W, C, device=DEV, dtype=dtype))
    w = (torch.randn(C, k * k * C, device=DEV) / (k * k * C) ** 0.5).to(dtype)
    sc, sh = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    try:
        for i in range(CHAIN):
            x, y = (a, b) if i % 2 == 0 else (b, a)
            p.conv(x, w, y, R=k, S=k, pad=k // 2, scale=sc, shift=sh, act=L.ACT_SILU, tile_hint=hint)
        s = torch.cuda.Stream(DEV)
        with torch.cuda.stream(s):
            p.run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                p.run()
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                g.replay()
            e1.record(s)
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 / CHAIN * 1e3
    except RuntimeError as e:
        return str(e)[:60]


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, shape in SHAPES.items():
        if flt and flt not in name:
            continue
        print(f"== {name}", flush=True)
        for hn, h in HINTS:
            t = chain_time(shape, h)
            print(f"   {hn:18s}: {t if isinstance(t, str) else f'{t:6.2f} us per launch'}", flush=True)


if __name__ == "__main__":
    main()
