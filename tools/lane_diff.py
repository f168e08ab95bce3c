#!/usr/bin/env python3
"""Find the first launch whose output differs between single-stream and multi-lane execution of the same plan."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MTBT_POOL_REUSE"] = "0"
os.environ["MTBT_LANES"] = "1"
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_
dev = torch.device("cuda:0")
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval().set_compute_dtype(torch.bfloat16)
B, S = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16, 640)
x = torch.rand(B, 3, S, S, device=dev)
c = m.compile(x); m._bind_input(c, x)
p = c.plan
L = p.launches
bufs, last_writer = {}, {}
for i, l in enumerate(L):
    for t in l.keep:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            bufs[t.untyped_storage().data_ptr()] = t
    for w in l.writes:
        last_writer[w[0]] = i
p.run(); torch.cuda.synchronize()
ref = {k: bufs[k].clone() for k in last_writer if k in bufs}
os.environ["MTBT_LANES"] = sys.argv[1] if len(sys.argv) > 1 else "4"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for it in range(N):
    p.run(); torch.cuda.synchronize()
    bad = sorted((last_writer[k], k) for k in ref if not torch.equal(bufs[k], ref[k]))
    if bad:
        sch = p.schedule()
        print(f"step {it}: {len(bad)} buffers differ; earliest writers:", flush=True)
        for i, k in bad[:6]:
            d = (bufs[k].float() - ref[k].float())
            nbad = int((d != 0).sum())
            print(f"   launch {i} lane {sch.lane[i]} {L[i].name}: {nbad} of {d.numel()} elements, max |d| {d.abs().max().item():.4g}, "
                  f"deps {sch.deps[i]} waits {sch.waits[i]}", flush=True)
        p.run(stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()   # restore sequentially
print("done", flush=True)
