"""Does a single-round 3x3 launch (400 tiles on 512 workgroup slots: every workgroup in the same phase at the same time) run faster as TWO
half-batch launches on two streams, whose phases are offset by the dispatch delay?  A chain of `depth` dependent convs (a C2f's four 3x3s),
captured into a HIP graph as the product replays it: (a) one launch per layer, (b) two half-batch chains on two streams.  One box, one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd.engine import Act, Plan

dev = torch.device("cuda:0")
N, H, C, depth = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 80, 128, 4
torch.manual_seed(0)
ws = [(torch.randn(C, 9 * C, device=dev) / (9 * C) ** 0.5).bfloat16() for _ in range(depth)]
sh = torch.zeros(C, device=dev)


def chain(n):
    p = Plan(dev)
    bufs = [Act.of(torch.randn(n, H, H, C, device=dev).bfloat16())] + [Act.of(torch.empty(n, H, H, C, device=dev, dtype=torch.bfloat16)) for _ in range(depth)]
    for i in range(depth):
        p.conv(bufs[i], ws[i], bufs[i + 1], R=3, S=3, pad=1, shift=sh, act=1)
    return p


full, halves = chain(N), [chain(N // 2), chain(N // 2)]
side = torch.cuda.Stream(dev)


def run_full():
    full.run(stream=torch.cuda.current_stream(dev).cuda_stream)


def run_halves():
    cur = torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    halves[0].run(stream=cur.cuda_stream)
    with torch.cuda.stream(side):
        halves[1].run(stream=side.cuda_stream)
    cur.wait_stream(side)


def graph_time(fn, reps=30):
    s = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):          # four chains per replay: amortises the replay's own launch cost
                fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        g.replay()
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / 4 * 1e3


for rep in range(3):
    print(f"{H}x{H} x{depth} layers: one launch per layer {graph_time(run_full):7.1f} us | two half-batch chains on two streams {graph_time(run_halves):7.1f} us", flush=True)
