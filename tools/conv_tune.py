#!/usr/bin/env python3
"""Tile / pipeline-depth sweep of the implicit-GEMM conv on the shapes the 640x640 batch-16 forward runs.
Development tool (GPU box): `python tools/conv_tune.py [filter]` prints TF/s and GB/s per (shape, hint)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import _lib as L  # noqa: E402
from multitask_bonetumor_yolo_amd.engine import Act, Plan  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = {
    # name: (N, H, W, C, K, k, act, residual)
    "proto.cv2 3x3 256->256 @160": (16, 160, 160, 256, 256, 3, 1, False),
    "c2f_p3.m 3x3 128->128 @80": (16, 80, 80, 128, 128, 3, 1, False),
    "c2f_p4.m 3x3 192->192 @40": (16, 40, 40, 192, 192, 3, 1, False),
    "c2f_p5.m 3x3 256->256 @20": (16, 20, 20, 256, 256, 3, 1, False),
    "bifpn.m 3x3 128->128 @40": (16, 40, 40, 128, 128, 3, 1, False),
    "head 3x3 256->64 @80": (16, 80, 80, 256, 64, 3, 1, False),
    "bifpn.m 3x3 128->128 @20": (16, 20, 20, 128, 128, 3, 1, False),
    "head 3x3 256->64 @20": (16, 20, 20, 256, 64, 3, 1, False),
    "head 3x3 64->64 @20": (16, 20, 20, 64, 64, 3, 1, False),
    "bifpn pw 1x1 256->256 @20": (16, 20, 20, 256, 256, 1, 2, False),
    "fc1.s0 1x1 96->384 @160": (16, 160, 160, 96, 384, 1, 3, False),
    "fc2.s0 1x1 384->96 @160": (16, 160, 160, 384, 96, 1, 0, True),
    "fc1.s1 1x1 192->768 @80": (16, 80, 80, 192, 768, 1, 3, False),
    "fc2.s1 1x1 768->192 @80": (16, 80, 80, 768, 192, 1, 0, True),
    "fc1.s2 1x1 384->1536 @40": (16, 40, 40, 384, 1536, 1, 3, False),
    "fc2.s2 1x1 1536->384 @40": (16, 40, 40, 1536, 384, 1, 0, True),
    "fc1.s3 1x1 768->3072 @20": (16, 20, 20, 768, 3072, 1, 3, False),
    "fc2.s3 1x1 3072->768 @20": (16, 20, 20, 3072, 768, 1, 0, True),
    "c2f.cv2 1x1 512->256 @80": (16, 80, 80, 512, 256, 1, 1, False),
    "c2f.cv1 1x1 256->256 @80": (16, 80, 80, 256, 256, 1, 1, False),
}


def bench(shape, hint, dtype=torch.bfloat16, iters=20):
    N, H, W, C, K, k, act, use_res = shape
    N = int(os.environ.get("MTBT_TUNE_BATCH", N))      # (the training step runs the same layers at batch 32)
    sc_ = int(os.environ.get("MTBT_TUNE_SCALE", "1"))   # configs[4]: the same layers on 1280^2 inputs (maps x 2), batch 64, fp16
    H, W = H * sc_, W * sc_
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("MTBT_TUNE_DTYPE", "bf16")] if dtype == torch.bfloat16 else dtype
    p = Plan(DEV)
    x = Act.of(torch.randn(N, H, W, C, device=DEV).to(dtype))
    w = (torch.randn(K, k * k * C, device=DEV) / (k * k * C) ** 0.5).to(dtype)
    y = Act.of(torch.empty(N, H, W, K, device=DEV, dtype=dtype))
    res = Act.of(torch.randn(N, H, W, K, device=DEV).to(dtype)) if use_res else None
    sc, sh = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    try:
        p.conv(x, w, y, R=k, S=k, pad=k // 2, scale=sc, shift=sh, act=act, res=res, tile_hint=hint)
        for _ in range(3):
            p.run()
    except RuntimeError as e:
        return None, None, str(e)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        p.run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    l = p.launches[0]
    return l.flops / ms / 1e9, l.bytes / ms / 1e6, f"{ms*1e3:.1f} us"


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    hints = [(0, 0, 0, 0)] + [(2, tc, tp, nar) for tc, tp in [(128, 128), (128, 64), (64, 128), (96, 128), (96, 64), (64, 64), (32, 128)]
                              for nar in (0, 1)]   # (two LDS stages: the deeper pipelines and the 256-pixel tiles are no longer built)
    for name, shape in SHAPES.items():
        if flt and flt not in name:
            continue
        print(f"== {name}")
        for nb, tc, tp, nar in hints:
            K = shape[4]
            if tc and ((tc == 96 and K % 96) or (tc == 128 and K < 96) or (tc == 64 and K % 64 and K > 64)):
                continue
            tf, gb, t = bench(shape, (nb << 28) | (nar << 27) | (tc << 16) | tp)
            if tf is None:
                continue
            print(f"   nbuf={nb} bk={'64B ' if nar else '128B'} tile={tc:3d}x{tp:3d}: {tf:7.1f} TF/s {gb:7.0f} GB/s  {t}")


if __name__ == "__main__":
    main()
