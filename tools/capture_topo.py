#!/usr/bin/env python3
"""Minimal stream-capture topologies with plain torch kernels (is a capture crash ours or the runtime's?)."""
import sys, torch
t = sys.argv[1]
dev = torch.device("cuda:0")
a, b, c = (torch.zeros(1 << 20, device=dev) for _ in range(3))
main, l1, l2 = (torch.cuda.Stream(device=dev) for _ in range(3))
E = lambda: torch.cuda.Event()
def body():
    fork, e6, e7, j1, j2 = E(), E(), E(), E(), E()
    fork.record(main)
    l1.wait_event(fork)
    if t != "nofork2":
        l2.wait_event(fork)
    with torch.cuda.stream(main): a.add_(1)
    with torch.cuda.stream(l1): b.add_(1)
    e6.record(l1)
    l2.wait_event(e6)
    with torch.cuda.stream(l2): c.add_(1)
    e7.record(l2)
    if t != "noback":
        l1.wait_event(e7)
    with torch.cuda.stream(l1): b.add_(1)
    j1.record(l1); main.wait_event(j1)
    j2.record(l2); main.wait_event(j2)
if t == "viamain":
    def body():
        fork, e6, e, e7, e8, j1, j2 = (E() for _ in range(7))
        fork.record(main); l1.wait_event(fork); l2.wait_event(fork)
        with torch.cuda.stream(l1): b.add_(1)
        e6.record(l1); main.wait_event(e6)
        with torch.cuda.stream(main): a.add_(1)
        e.record(main); l2.wait_event(e)
        with torch.cuda.stream(l2): c.add_(1)
        e7.record(l2); main.wait_event(e7)
        with torch.cuda.stream(main): a.add_(1)
        e8.record(main); l1.wait_event(e8)
        with torch.cuda.stream(l1): b.add_(1)
        j1.record(l1); main.wait_event(j1)
        j2.record(l2); main.wait_event(j2)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    body(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=main):
        body()
g.replay(); torch.cuda.synchronize()
print("ok", t, a[0].item(), b[0].item(), c[0].item())
