#!/usr/bin/env python3
"""Print the lane schedule of the batch-16 640x640 plan (built on CPU tensors, nothing is launched).
usage: MTBT_LANES=3 MTBT_LANE_WIDE_US=60 python tools/dump_schedule.py [B S]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_
from multitask_bonetumor_yolo_amd.model import _Lowering
from multitask_bonetumor_yolo_amd.engine import code_of

B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 640)
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).eval()
xs = torch.empty(B, 3, S, S)
with torch.no_grad():
    lo = _Lowering(m, xs, code_of(torch.bfloat16))
    c3, c4, c5 = lo.backbone()
    feats = list(lo.neck(c3, c4, c5))
    lo.p.pool.reuse = os.environ.get("MTBT_HEAD_REUSE", "0") == "1"
    lo.det_branch(feats, m.detect, "detect")
    lo.det_branch(feats, m.segment, "segment")
    lo.seg_extras(feats, m.segment)
    lo.cls_head(feats[2])
s = lo.p.schedule()
for i, l in enumerate(lo.p.launches):
    est = max(l.flops / 4e14, l.bytes / 2e12) * 1e6
    print(f"{i:3d} lane {s.lane[i]} est {est:6.1f}us deps {s.deps[i]} waits {s.waits[i]} rec {s.records[i]:3d}  {l.name}")
print("lanes used", sorted(set(s.lane)), "events", s.n_events, "est makespan %.2f ms" % (s.est_makespan * 1e3))
