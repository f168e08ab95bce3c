#!/usr/bin/env python3
"""Do HIP stream priorities reach the hardware scheduler -- eagerly, and for kernel nodes captured into a graph?  A chain of N small dependent
kernels (stream H) beside machine-filling GEMMs (stream L): wall time of each alone, of both at equal priority, and with H at high priority;
then the same fork / join captured into ONE graph from streams of those priorities and replayed.  If priority works, "both" ~ max(alone);
if not, the small chain stretches behind the big kernels' workgroups."""
import torch
dev = torch.device("cuda:0")
lo_p, hi_p = 0, -1
try:
    r = torch.cuda.Stream.priority_range()
    print("priority range (least, greatest):", r, flush=True)
    lo_p, hi_p = r[0], r[1]
except Exception as e:  # noqa: BLE001
    print("priority_range unavailable:", e)
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
B = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
Cb = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
a = torch.randn(64, 256, device=dev, dtype=torch.bfloat16)
b = torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
c = torch.empty(64, 256, device=dev, dtype=torch.bfloat16)
NBIG, NSMALL = 8, 1500


def big(): 
    for _ in range(NBIG):
        torch.mm(A, B, out=Cb)


def small():
    for _ in range(NSMALL):
        torch.mm(a, b, out=c)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def forked(sH, sL):
    def run():
        cur = torch.cuda.current_stream()
        sH.wait_stream(cur); sL.wait_stream(cur)
        with torch.cuda.stream(sL):
            big()
        with torch.cuda.stream(sH):
            small()
        cur.wait_stream(sH); cur.wait_stream(sL)
    return run


def graphed(fn, cap_stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap_stream):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap_stream):
            fn()
    return lambda: g.replay()


s0 = torch.cuda.Stream(dev)
with torch.cuda.stream(s0):
    print(f"eager   big alone {timed(big):7.2f} ms | small chain alone {timed(small):7.2f} ms (host-bound if >> graph figure)", flush=True)
    gb, gs = graphed(big, s0), graphed(small, s0)
    print(f"graph   big alone {timed(gb):7.2f} ms | small chain alone {timed(gs):7.2f} ms", flush=True)
    for name, ph, pl in (("equal priority", lo_p, lo_p), ("small chain HIGH", hi_p, lo_p), ("small chain LOW (control)", lo_p, hi_p)):
        sH, sL = torch.cuda.Stream(dev, priority=ph), torch.cuda.Stream(dev, priority=pl)
        f = forked(sH, sL)
        te = timed(f)
        tg = timed(graphed(f, s0))
        print(f"{name:28s}: eager fork/join {te:7.2f} ms | ONE graph {tg:7.2f} ms   (stream priorities H={sH.priority} L={sL.priority})", flush=True)
