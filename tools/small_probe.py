#!/usr/bin/env python3
"""Ablation of the implicit-GEMM conv (needs a -DMTBT_CONV_ABLATION build: `MTBT_CONV_ABLATION=1 python -m multitask_bonetumor_yolo_amd.build --force`):
MTBT_CONV_DEBUG bits  1 = no DMA after the prologue stages, 2 = no fragment reads / MFMAs, 4 = fragment reads without MFMAs, 8 = MFMAs on
constants (no fragment reads), 16 = no epilogue.  Latency-bound small maps and the short-K 1x1 GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_tune import SHAPES, bench
NAMES = {0: "full", 16: "no epilogue", 1: "no DMA", 2: "no reads/MFMA", 3: "no DMA, no reads/MFMA", 19: "skeleton (none of the three)", 8: "MFMA on constants", 4: "reads only"}
for name in sys.argv[1:] or ["bifpn.m 3x3 128->128 @20", "bifpn pw 1x1 256->256 @20", "fc1.s1 1x1 192->768 @80", "fc2.s2 1x1 1536->384 @40", "c2f.cv1 1x1 256->256 @80"]:
    for dbg, what in NAMES.items():
        os.environ["MTBT_CONV_DEBUG"] = str(dbg)
        r = bench(SHAPES[name], 0, iters=int(os.environ.get('PROBE_ITERS', '100')))
        print(f"{name:32s} {what:30s}: {r[2]}", flush=True)
