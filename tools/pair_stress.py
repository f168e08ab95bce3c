#!/usr/bin/env python3
"""Which two launches of the plan disturb each other when they run at the same time?  Runs pairs of independent launches
concurrently on two streams and compares every buffer they write against the sequential result."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MTBT_LANES"] = "1"
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_
from multitask_bonetumor_yolo_amd.engine import _overlap
dev = torch.device("cuda:0")
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval().set_compute_dtype(torch.bfloat16)
os.environ["MTBT_POOL_REUSE"] = "0"
BS = os.environ.get("PAIR_SHAPE", "16x640").split("x")
x = torch.rand(int(BS[0]), 3, int(BS[1]), int(BS[1]), device=dev)
c = m.compile(x); m._bind_input(c, x)
p = c.plan
p.run(); torch.cuda.synchronize()
L = p.launches
bufs = {}
for l in L:
    for t in l.keep:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            bufs[t.untyped_storage().data_ptr()] = t
def written(l):
    return [bufs[w[0]] for w in l.writes if w[0] in bufs]
ref = {i: [t.clone() for t in written(l)] for i, l in enumerate(L)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def pick(spec):   # comma list of launch indices or name substrings
    out = []
    for v in spec.split(","):
        out += [int(v)] if v.isdigit() else [i for i, l in enumerate(L) if v in l.name]
    return out
A, B = pick(sys.argv[1]), pick(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
def indep(a, b):
    ra, wa, rb, wb = L[a].reads, L[a].writes, L[b].reads, L[b].writes
    return not any(_overlap(u, v) for u in wa for v in rb + wb) and not any(_overlap(u, v) for u in wb for v in ra)
for a in A:
    for b in B:
        if a == b or not indep(a, b):
            continue
        bad_a = bad_b = 0
        for r in range(reps):
            torch.cuda.synchronize()
            pa, pb = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)
            for _ in range(3):
                L[a].fn(*L[a].args, pa); L[b].fn(*L[b].args, pb)
            torch.cuda.synchronize()
            for t, q in zip(written(L[a]), ref[a]):
                if not torch.equal(t, q) and os.environ.get("PAIR_VERBOSE"):
                    idx = (t != q).nonzero()
                    i0 = idx[0].tolist()
                    sl = (i0[0], i0[1], i0[2], slice(max(i0[3] - 4, 0), i0[3] + 8))
                    print("   got", [round(v, 4) for v in t[sl].float().tolist()], "\n   ref", [round(v, 4) for v in q[sl].float().tolist()], flush=True)
                    print("   differing elements", idx.shape[0], "first", idx[:6].tolist(), "last", idx[-3:].tolist(),
                          "max |d|", (t.float() - q.float()).abs().max().item(), flush=True)
            bad_a += any(not torch.equal(t, q) for t, q in zip(written(L[a]), ref[a]))
            bad_b += any(not torch.equal(t, q) for t, q in zip(written(L[b]), ref[b]))
            if bad_a or bad_b:   # restore
                L[a].fn(*L[a].args, pa); torch.cuda.synchronize(); L[b].fn(*L[b].args, pb); torch.cuda.synchronize()
        if bad_a or bad_b:
            print(f"PAIR {a} {L[a].name}  x  {b} {L[b].name}: wrong {bad_a}/{bad_b} of {reps}", flush=True)
print("done", flush=True)
