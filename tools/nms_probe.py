#!/usr/bin/env python3
"""Time the NMS kernel phases on the bench's own inputs (dev tool): top_k=1 ~ compaction + sort, top_k=100 = full."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_, postprocess as pp
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval().set_compute_dtype(torch.bfloat16)
x = torch.rand(16, 3, 640, 640, generator=torch.Generator().manual_seed(0)).to(dev)
with torch.no_grad():
    out = m(x, "infer")
d = pp.decode_boxes(out["detect_features"], 640)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for k in (1, 10, 100):
    r = pp.nms_batched(d["boxes"], d["best_score"], d["best_label"], 640.0, 0.05, 0.6, k)
    print(f"top_k={k}: {t(lambda: pp.nms_batched(d['boxes'], d['best_score'], d['best_label'], 640.0, 0.05, 0.6, k)):.1f} us  n_cand={r['n_cand'].tolist()[:4]} counts={r['counts'].tolist()[:4]}")
print(f"decode: {t(lambda: pp.decode_boxes(out['detect_features'], 640)):.1f} us")
feats, mc, protos = out["segment_protos"]
r = pp.nms_batched(d["boxes"], d["best_score"], d["best_label"], 640.0, 0.05, 0.6, 100)
print(f"masks: {t(lambda: pp.assemble_masks(protos, mc, r['keep_anchor'], r['counts'], (640, 640))):.1f} us")
