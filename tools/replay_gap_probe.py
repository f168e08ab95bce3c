"""What sits BETWEEN two graph replays: K replays of the one-step graph against K/2 replays of a graph that holds the step twice (the second
copy overwrites the first's static outputs: timing only), and against the sum of the step's span inside one replay.  Also the two-in-flight
variant of tools/two_inflight_probe.py with the current kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, GraphedInference, init_synthetic_
from multitask_bonetumor_yolo_amd.model import calibrate_synthetic_heads_, synthetic_images
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval()
m.set_compute_dtype(torch.bfloat16)
B, IMG = 16, 640
x = synthetic_images(B, IMG, 0).to(dev)
calibrate_synthetic_heads_(m, x[:4].contiguous())
g1 = GraphedInference(m, x, IMG, autotune=True)


class Twice(GraphedInference):
    def _step(self):
        GraphedInference._step(self)
        return GraphedInference._step(self)


g2 = Twice(m, x, IMG)


def run(g, n):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    a, b = run(g1, 40), run(g2, 20)
    print(f"one step per graph {a:.3f} ms/step | two steps per graph {b / 2:.3f} ms/step  ({(a - b / 2) * 1e3:+.0f} us per step between replays)", flush=True)
