#!/usr/bin/env python3
"""Ablation of the implicit-GEMM conv on latency-bound (20x20) shapes: MTBT_CONV_DEBUG bits x pipeline depth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_tune import SHAPES, bench
for name in ["bifpn.m 3x3 128->128 @20", "c2f_p5.m 3x3 256->256 @20", "bifpn pw 1x1 256->256 @20"]:
    for nb in (2, 3):
        for dbg in (0, 1, 2, 3, 16, 19, 8):
            os.environ["MTBT_CONV_DEBUG"] = str(dbg)
            r = bench(SHAPES[name], (nb << 28) | (64 << 16) | 64, iters=int(os.environ.get('PROBE_ITERS','200')))
            print(f"{name:32s} nbuf={nb} debug={dbg:2d}: {r[2]}", flush=True)
