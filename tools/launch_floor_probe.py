#!/usr/bin/env python3
"""Floor of a dependent launch inside a replayed HIP graph: 256 trivial kernels (one 64-element add each) on one captured stream, time per node.
Run under different HIP runtime settings (DEBUG_CLR_GRAPH_PACKET_CAPTURE, AMD_OPT_FLUSH) to see what the runtime itself puts between two nodes."""
import torch
dev = torch.device("cuda:0")
t = torch.zeros(64, device=dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(3):
        t.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(256):
            t.add_(1.0)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20):
        g.replay()
    e1.record(s)
    torch.cuda.synchronize()
print(f"trivial node: {e0.elapsed_time(e1) / 20 / 256 * 1e3:.2f} us per launch", flush=True)
