"""Oracle restatement of the post-process the reference's trainer/eval code runs on the model
outputs.  TEST INFRASTRUCTURE (see oracle/__init__.py).

  batch_bbox_iou, dist2bbox     /root/reference/src/running_main_v3.py:71-110   (pinned by fixtures)
  decode_levels                 running_main_v3.py:510-533 (== :264-290 in the loss)
  nms                           torchvision.ops.nms as called at running_main_v3.py:549 -- third-party,
                                absent here, **parity unpinned**; restates the published CPU kernel
                                `torchvision/csrc/ops/cpu/nms_kernel.cpp` (greedy, stable descending
                                sort, strict `>` on fp32 IoU, area = (x2-x1)*(y2-y1))
  filter_and_nms                running_main_v3.py:535-552 (max over classes, >CONF_TH, clamp, nms, [:TOP_K])
  proto_projector_logits        running_main_v3.py:186,251-255 ; evaluate_model.py:160-171
  assemble_masks                test_model.py:80-85 in its intended form
"""
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

CONF_TH = 0.05   # running_main_v3.py:54
NMS_IOU = 0.6    # running_main_v3.py:55
TOP_K = 100      # running_main_v3.py:56 (300 in running_main_v2.py:53)


def batch_bbox_iou(boxes1, boxes2, eps: float = 1e-7):
    """running_main_v3.py:71-97."""
    if boxes1.numel() == 0 or boxes2.numel() == 0:
        return torch.zeros((boxes1.shape[0], boxes2.shape[0]))
    ix1 = torch.max(boxes1[:, 0].unsqueeze(1), boxes2[:, 0].unsqueeze(0))
    iy1 = torch.max(boxes1[:, 1].unsqueeze(1), boxes2[:, 1].unsqueeze(0))
    ix2 = torch.min(boxes1[:, 2].unsqueeze(1), boxes2[:, 2].unsqueeze(0))
    iy2 = torch.min(boxes1[:, 3].unsqueeze(1), boxes2[:, 3].unsqueeze(0))
    inter = (ix2 - ix1).clamp(min=0) * (iy2 - iy1).clamp(min=0)
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    return inter / (a1.unsqueeze(1) + a2.unsqueeze(0) - inter + eps)


def dist2bbox(distance, anchor_points, box_format="xyxy"):
    """running_main_v3.py:100-110."""
    lt, rb = torch.split(distance, 2, dim=-1)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    if box_format == "xyxy":
        return torch.cat((x1y1, x2y2), dim=-1)
    if box_format == "xywh":
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim=-1)
    raise NotImplementedError(f"Box format '{box_format}' not implemented.")


def decode_levels(det_maps: List[torch.Tensor], img_size: int, reg_max: int = 16):
    """running_main_v3.py:510-533.  det_maps: per level [B, 4*reg_max+nc, h, w].
    Returns boxes [B,A,4] xyxy in pixels, class scores (sigmoid) [B,A,nc], raw logits [B,A,nc]."""
    proj = torch.arange(reg_max, dtype=torch.float32)
    boxes, scores, logits = [], [], []
    for fm in det_maps:
        b, ch, h, w = fm.shape
        stride = img_size / w
        flat = fm.permute(0, 2, 3, 1).reshape(b, h * w, ch)
        dist = flat[..., : reg_max * 4].reshape(b, h * w, 4, reg_max)
        cls = flat[..., reg_max * 4:]
        ltrb = torch.einsum("ijkl,l->ijk", F.softmax(dist, dim=-1), proj)
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                                indexing="ij")
        anchors = torch.stack((gx + 0.5, gy + 0.5), dim=-1).view(1, h * w, 2).repeat(b, 1, 1)
        boxes.append(dist2bbox(ltrb * stride, anchors * stride))
        scores.append(cls.sigmoid())
        logits.append(cls)
    return torch.cat(boxes, 1), torch.cat(scores, 1), torch.cat(logits, 1)


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS with torchvision's CPU-kernel arithmetic, all in float32.
    Returns int64 indices of kept boxes in descending-score order (stable for ties)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().to(torch.float32).numpy()
    s = scores.detach().to(torch.float32).numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-s, kind="stable")  # stable descending (ties keep ascending index)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    zero = np.float32(0)
    for _i in range(n):
        if suppressed[_i]:
            continue
        i = order[_i]
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(zero, xx2 - xx1)
        h = np.maximum(zero, yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[_i + 1:] |= ovr > thr
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def filter_and_nms(boxes: torch.Tensor, cls_scores: torch.Tensor, img_size: float,
                   conf_th: float = CONF_TH, iou_th: float = NMS_IOU, top_k: int = TOP_K):
    """running_main_v3.py:535-552 for ONE image.  boxes [A,4], cls_scores [A,nc].
    Returns (kept indices into the conf-filtered list [int64], kept anchor indices, boxes, scores, labels)."""
    top_scores, top_labels = cls_scores.max(dim=1)
    keep_conf = top_scores > conf_th
    anchor_idx = torch.nonzero(keep_conf).flatten()
    if anchor_idx.numel() == 0:
        e = torch.empty((0,), dtype=torch.int64)
        return e, e, torch.empty((0, 4)), torch.empty((0,)), e
    b = boxes[keep_conf].clamp(0, img_size)
    s = top_scores[keep_conf]
    lab = top_labels[keep_conf]
    k = nms(b, s, iou_th)[:top_k]
    return k, anchor_idx[k], b[k], s[k], lab[k]


def proto_projector_logits(protos: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, img_size: int):
    """running_main_v3.py:251-255: Conv2d(nm,1,1)(protos) -> bilinear to img_size (align_corners=False)."""
    low = F.conv2d(protos, weight.view(1, -1, 1, 1), bias.view(1))
    return F.interpolate(low, size=(img_size, img_size), mode="bilinear", align_corners=False)


def assemble_masks(coeffs: torch.Tensor, protos: torch.Tensor, out_size: Tuple[int, int]):
    """test_model.py:80-85 (intended dims): coeffs [K,nm] for one image's kept boxes, protos [nm,h,w]
    -> (mask logits [K,H,W] fp32, masks bool) with masks = sigmoid(logits) > 0.5."""
    low = torch.einsum("kc,chw->khw", coeffs, protos)
    up = F.interpolate(low.unsqueeze(0), size=out_size, mode="bilinear", align_corners=False)[0]
    return up, torch.sigmoid(up) > 0.5
