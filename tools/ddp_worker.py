"""One rank of the data-parallel training-step rehearsal (tests/test_gpu_train.py::test_two_rank_step_equals_averaged_gradients):
launched by `python -m torch.distributed.run --nproc-per-node 2 tools/ddp_worker.py OUT`.  Every rank builds the same seeded model,
takes its half of a seeded batch, runs `TrainStep.step` twice and rank 0 saves the parameter buckets.  Backend: MTBT_DIST_BACKEND
(default "gloo": the ranks share one GPU here; on an 8-GPU node the same code runs with "nccl" = RCCL over xGMI)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_batch(S, B, seed=21):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, S, S, generator=g)
    boxes = torch.tensor([[b, b % 2, 0.3 + 0.1 * (b % 3), 0.5, 0.3, 0.35] for b in range(B)], dtype=torch.float32)
    masks = torch.zeros(B, 1, S, S)
    for b in range(B):
        masks[b, 0, S // 4: S // 4 + 20 + 4 * b, S // 3: S // 3 + 30] = 1
    cls = torch.tensor([b % 2 for b in range(B)])
    return x, boxes, masks, cls


def shard(batch, rank, per):
    x, boxes, masks, cls = batch
    sel = (boxes[:, 0] >= rank * per) & (boxes[:, 0] < (rank + 1) * per)
    bx = boxes[sel].clone()
    bx[:, 0] -= rank * per
    return x[rank * per:(rank + 1) * per], bx, masks[rank * per:(rank + 1) * per], cls[rank * per:(rank + 1) * per]


def build_model(dev, seed=8):
    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_
    torch.manual_seed(seed)
    return init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False), seed).to(dev)


STEP_KW = dict(optimizer="sgd", lr=0.05, weight_decay=5e-4, momentum=0.9, clip_norm=10.0, iou_match_thresh=0.05)

if __name__ == "__main__":
    out = sys.argv[1]
    dist.init_process_group(os.environ.get("MTBT_DIST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    from multitask_bonetumor_yolo_amd.trainstep import TrainStep
    S, per = 128, 2
    model = build_model(dev)
    torch.manual_seed(3)
    proj = torch.nn.Conv2d(32, 1, 1)
    ts = TrainStep(model, (per, 3, S, S), projector=proj, overlap=os.environ.get("MTBT_DDP_OVERLAP", "1") == "1", **STEP_KW)
    batch = make_batch(S, per * world)
    losses = []
    for step in range(2):
        x, bx, mk, cl = (t.to(dev) for t in shard(batch, rank, per))
        losses.append(ts.step(x, bx, mk, cl).cpu())
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"buckets": [b.cpu() for b in ts.params.buckets], "proj": ts.pj.cpu(), "losses": losses, "host_staged": ts._host_staged}, out)
    dist.barrier()
    dist.destroy_process_group()
