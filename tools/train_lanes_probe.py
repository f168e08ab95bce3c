"""How the lane scheduler spreads the training plans: launches per lane, estimated makespan vs serial, and measured step time with
MTBT_TRAIN_LANES=0/1 in ONE process (batch 32, 640x640, bf16)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_
from multitask_bonetumor_yolo_amd.trainstep import TrainStep
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
from bench import synthetic_targets
dev = torch.device("cuda:0")
B, S = int(os.environ.get("B", "32")), 640
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).set_compute_dtype(torch.bfloat16)
x = torch.rand(B, 3, S, S, device=dev)
boxes, masks, cls = synthetic_targets(B, S, 0, dev)
ts = TrainStep(m, (B, 3, S, S), optimizer="sgd", lr=1e-4)
for name, plan in (("fwd", ts.tp.fwd), ("bwd", ts.bwd)):
    sch = plan.schedule()
    lanes = [sch.lane.count(k) for k in range(sch.n_lanes)]
    serial = sum(max(l.flops / 4e14, l.bytes / 2e12) + 6e-6 for l in plan.launches)
    print(f"{name}: {len(plan.launches)} launches, per lane {lanes}, {sch.n_events} cross-lane events, est makespan {sch.est_makespan*1e3:.2f} ms vs serial {serial*1e3:.2f} ms", flush=True)
for rnd in range(2):
    for lanes in ("0", "1"):
        os.environ["MTBT_TRAIN_LANES"] = lanes
        ts.tp.reload_env()
        for _ in range(2):
            ts.step(x, boxes, masks, cls)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ts.step(x, boxes, masks, cls)
        torch.cuda.synchronize()
        print(f"round {rnd} MTBT_TRAIN_LANES={lanes}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/step", flush=True)
