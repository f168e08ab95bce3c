#!/usr/bin/env python3
"""Post-process load of the synthetic model (bench configuration): candidates / kept boxes per image, score range, and how well the kept
boxes of the bf16 path agree with the fp32 path, for raw and calibrated heads, noise and noise + blob inputs.
usage: python tools/head_stats.py [B S]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, calibrate_synthetic_heads_, init_synthetic_, synthetic_images
from multitask_bonetumor_yolo_amd.metrics import box_iou_xyxy, MeanAveragePrecision
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 640)
dev = torch.device("cuda:0")


def run(m, x, dt):
    m.set_compute_dtype(dt)
    _, det = m.infer_and_detect(x, S)
    torch.cuda.synchronize()
    return {k: det[k].cpu() for k in ("boxes", "scores", "labels", "counts", "n_cand")} | {"mask_px": det["masks"].float().sum().item()}


for inp in ("noise", "blobs"):
    for frac, tq in (((0.12, 0.999), (0.12, 1.0), (0.03, 1.0), (0.03, 0.999)) if inp == "blobs" else ((0.12, 1.0), (0.03, 1.0))):
        torch.manual_seed(0)
        m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval()
        x = (torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(0)) if inp == "noise" else synthetic_images(B, S, 0)).to(dev)
        if frac is not None:
            m.set_compute_dtype(torch.float32)
            calibrate_synthetic_heads_(m, x[:4].contiguous(), cand_frac=frac, top_quantile=tq)
        r32, r16 = run(m, x, torch.float32), run(m, x, torch.bfloat16)
        hit = tot = hit7 = 0
        preds, targets = [], []
        for b in range(B):
            n32, n16 = int(r32["counts"][b]), int(r16["counts"][b])
            if n32 and n16:
                iou = box_iou_xyxy(r16["boxes"][b, :n16].numpy(), r32["boxes"][b, :n32].numpy())
                hit += int((iou.max(1) >= 0.9).sum()) + int((iou.max(0) >= 0.9).sum())
                hit7 += int((iou.max(1) >= 0.7).sum()) + int((iou.max(0) >= 0.7).sum())
            tot += n16 + n32
            preds.append(dict(boxes=r16["boxes"][b, :n16], scores=r16["scores"][b, :n16], labels=r16["labels"][b, :n16]))
            targets.append(dict(boxes=r32["boxes"][b, :n32], labels=r32["labels"][b, :n32]))
        mp = MeanAveragePrecision([0.5], [1, 10, 100], dist_sync=False)
        mp.update(preds, targets)
        cnt = r16["counts"]
        valid = torch.arange(r16["scores"].shape[1])[None, :] < cnt[:, None]
        sc = r16["scores"][valid]
        wh = (r16["boxes"][valid][:, 2:] - r16["boxes"][valid][:, :2])
        print(f"{inp} frac={frac} tq={tq}: n_cand {r16['n_cand'].tolist()} kept {cnt.tolist()} scores [{sc.min():.3f}, {sc.max():.3f}] med {sc.median():.3f} "
              f"side med {wh.median():.0f}px mask px/box {r16['mask_px'] / max(int(cnt.sum()), 1):.0f} | bf16~fp32 kept agreement IoU.9 {hit / max(tot, 1):.3f} IoU.7 {hit7 / max(tot, 1):.3f} "
              f"mAP50(bf16 | fp32 as GT) {mp.compute()['map_50']:.3f}", flush=True)
