#!/usr/bin/env python3
"""Stress the multi-lane execution: N eager steps and N graph replays with MTBT_LANES=4 against the single-stream result."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_
from multitask_bonetumor_yolo_amd.graphed import GraphedInference
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval().set_compute_dtype(torch.bfloat16)
x = torch.rand(16, 3, 640, 640, device=dev)
def snap(fwd, det):
    return [t.clone() for t in (fwd["segment_preds_cat"], fwd["detect_preds_cat"], fwd["segment_protos"][2], fwd["img_cls_logits"],
                                det["keep_idx"], det["masks"])]
names = ["segment_preds_cat", "detect_preds_cat", "protos", "cls", "keep_idx", "masks"]
os.environ["MTBT_LANES"] = "1"
ref = snap(*m.infer_and_detect(x, 640)); torch.cuda.synchronize()
os.environ["MTBT_LANES"] = sys.argv[2] if len(sys.argv) > 2 else "4"
m.__dict__.pop("_plans", None)
bad = {}
for i in range(N):
    got = snap(*m.infer_and_detect(x, 640)); torch.cuda.synchronize()
    for n_, a, b in zip(names, ref, got):
        if not torch.equal(a, b): bad[("eager", n_)] = bad.get(("eager", n_), 0) + 1
g = GraphedInference(m, x, 640) if os.environ.get("STRESS_GRAPH", "1") == "1" else None
for i in range(N if g is not None else 0):
    g.replay(); torch.cuda.synchronize()
    got = snap(g.fwd, g.out)
    for n_, a, b in zip(names, ref, got):
        if not torch.equal(a, b): bad[("graph", n_)] = bad.get(("graph", n_), 0) + 1
print("steps", N, "mismatches", bad, flush=True)
