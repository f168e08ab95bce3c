#!/usr/bin/env python3
"""Which part of the multi-lane step breaks HIP stream capture?  capture_probe.py MODE  (plan | step | nomask)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_
mode = sys.argv[1]
dev = torch.device("cuda:0")
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval().set_compute_dtype(torch.bfloat16)
x = torch.rand(16, 3, 640, 640, device=dev)
st = torch.cuda.Stream(device=dev)
side = torch.cuda.Stream(device=dev)
g = torch.cuda.CUDAGraph()
def step():
    if mode == "plan":
        c = m.compile(x); m._bind_input(c, x)
        lim = int(os.environ.get("PLAN_LIMIT", "0"))
        if lim and len(c.plan.launches) > lim:
            c.plan.launches = c.plan.launches[:lim]
        c.plan.run()
    else:
        m.infer_and_detect(x, 640, masks=(mode != "nomask"), side_stream=side)
with torch.cuda.stream(st), torch.no_grad():
    step(); step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("ok", mode, flush=True)
