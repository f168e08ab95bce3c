"""Throughput with TWO batches in flight: two model replicas (same weights), each with its own HIP-graph, replayed alternately on two streams --
the tail of step i (small pyramid-level launches that do not fill the chip) overlaps the backbone of step i + 1.  Against one graph."""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, GraphedInference, init_synthetic_
from multitask_bonetumor_yolo_amd.model import calibrate_synthetic_heads_, synthetic_images
dev = torch.device("cuda:0")
torch.manual_seed(0)
mA = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval()
mA.set_compute_dtype(torch.bfloat16)
B, IMG = 16, 640
xA = synthetic_images(B, IMG, 0).to(dev)
xB = synthetic_images(B, IMG, 1).to(dev)
calibrate_synthetic_heads_(mA, xA)
mB = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval()
mB.load_state_dict(mA.state_dict())
mB.set_compute_dtype(torch.bfloat16)
gA, gB = GraphedInference(mA, xA, IMG), GraphedInference(mB, xB, IMG)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def run(n, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if two and (i & 1):
            with torch.cuda.stream(sB): gB.replay()
        else:
            with torch.cuda.stream(sA): gA.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    run(6, True); a = run(40, False); b = run(40, True)
    print(f"one graph {a:.3f} ms/step ({B / a * 1e3:.0f} img/s) | two in flight {b:.3f} ms/step ({B / b * 1e3:.0f} img/s)", flush=True)
