#!/usr/bin/env python3
"""Time the device multitask loss (batch 32, 640 x 640: BASELINE config 2's shape) and the CPU oracle (the reference's
per-image loop, restated) on the same inputs."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import multitask_loss
from oracle.loss import multitask_loss as oracle_loss
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, S, NC = 32, 640, 2
det = [torch.randn(B, 64 + NC, h, h, generator=g) for h in (80, 40, 20)]
n_gt = 3
gt = torch.cat([torch.cat([torch.full((n_gt, 1), float(b)), torch.randint(0, NC, (n_gt, 1), generator=g).float(),
                           torch.rand(n_gt, 2, generator=g) * 0.6 + 0.2, torch.rand(n_gt, 2, generator=g) * 0.35 + 0.05], 1) for b in range(B)])
protos, logits = torch.randn(B, 32, 160, 160, generator=g), torch.randn(B, 2, generator=g)
masks, gcls = (torch.rand(B, 1, S, S, generator=g) > 0.7).float(), torch.randint(0, 2, (B,), generator=g)
pw, pb = torch.randn(1, 32, 1, 1, generator=g) * 0.2, torch.tensor([0.1])
kw = dict(img_size=S, nc_det=NC, label_smoothing=0.1, training=True)
d = [t.to(dev) for t in det]
args = (d, protos.to(dev), logits.to(dev), gt.to(dev), masks.to(dev), gcls.to(dev), pw.to(dev), pb.to(dev))
for _ in range(3): out = multitask_loss(*args, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): out = multitask_loss(*args, **kw)
torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t0) / 20 * 1e3
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): out = multitask_loss(*args, **kw)
b.record(); torch.cuda.synchronize()
torch.set_num_threads(min(32, os.cpu_count()))
t0 = time.perf_counter(); ref = oracle_loss(det, protos, logits, gt, masks, gcls, pw, pb, **kw); cpu_ms = (time.perf_counter() - t0) * 1e3
print(f"batch {B}: device loss {gpu_ms:.3f} ms wall per call ({a.elapsed_time(b)/20:.3f} ms GPU time), CPU oracle {cpu_ms:.1f} ms; "
      f"total {float(out[0]):.5f} vs {float(ref[0]):.5f}, positives {float(out[6]):.0f} vs {float(ref[6]):.0f}")
