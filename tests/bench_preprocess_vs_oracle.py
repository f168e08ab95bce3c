#!/usr/bin/env python3
"""Time the device input pipeline (mtbt_letterbox_batch) on a batch of BTXRD-sized radiographs and the numpy oracle
beside it.  usage: python tests/bench_preprocess_vs_oracle.py [B] [H0] [W0] [S]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import preprocess as P
from oracle import preprocess as O
B, H0, W0, S = (int(v) for v in (sys.argv[1:5] + ["16", "2048", "1536", "640"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(H0 - 8 * i, W0 + 4 * i, 3), dtype=np.uint8) for i in range(B)]
masks = [rng.integers(0, 256, size=a.shape[:2], dtype=np.uint8) for a in imgs]
di, dm = [torch.from_numpy(a).cuda() for a in imgs], [torch.from_numpy(a).cuda() for a in masks]
for _ in range(3):
    P.letterbox_batch(di, dm, S)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    P.letterbox_batch(di, dm, S)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
out_bytes = B * 4 * S * S * 4
# a source pixel is touched only if a destination tap lands on it: at most 4 taps x 3 B + 1 mask byte per output pixel
src_bytes = sum(min(x.size + m.size, 13 * S * S) for x, m in zip(imgs, masks))
print(f"device: {us:.1f} us / batch of {B} ({B / us * 1e6:.0f} images/s); output {out_bytes / 1e6:.1f} MB + source <= {src_bytes / 1e6:.1f} MB "
      f"-> {(out_bytes + src_bytes) / us / 1e3:.0f} GB/s")
t0 = time.perf_counter()
for x, m in zip(imgs[:4], masks[:4]):
    O.letterbox(x, m, S)
dt = (time.perf_counter() - t0) / 4
print(f"numpy oracle (1 core): {dt * 1e3:.1f} ms / image ({1 / dt:.0f} images/s)")
