"""Oracle restatement of `MultiTaskLitModel._multitask_loss` (`/root/reference/src/running_main_v3.py:232-387`).

TEST INFRASTRUCTURE (CPU, fp32).  Pinned: `tests/golden/ref_blocks.pt` holds inputs and the outputs of the REAL reference
method (called unbound on a stand-in `self`, see tests/golden/make_ref_fixtures.py); tests/test_oracle_golden.py checks
this restatement against them.

Terms (reference lines):
  image classification  CrossEntropy(logits, gt)                                              :237, :189
  segmentation          BCEWithLogits(bilinear(Conv1x1(protos)) -> S x S, gt masks), mean     :251-257, :190
  detection, per image with GT boxes (:297-368):
      decode every anchor (softmax over 16 bins . arange, anchors (x+.5, y+.5), stride S/w)   :268-290
      IoU [A, G] against the image's GT boxes (cxcywh normalised -> xyxy pixels)              :301-316
      positives: max_g IoU > iou_match_thresh, matched GT = argmax                            :319-321
      box    sum (1 - IoU)                                                                    :329-331
      class  BCEWithLogits(sum) against one-hot / label-smoothed targets (train only)         :334-346
      DFL    two-bin cross-entropy of the 4 side distributions, target (anchor*stride -/+ GT)/stride
             clamped to [0, reg_max - 1.01]                                                   :351-367
  normalisation: every detection sum / #positives of the BATCH (or / batch size if none)     :369-375
  total = w_seg*seg + w_box*box + w_dfl*dfl + w_cls*cls + w_img*img                           :377-383
"""
import torch
import torch.nn.functional as F

from .postprocess import batch_bbox_iou


def multitask_loss(det_maps, protos, img_logits, gt_boxes, gt_masks, gt_cls, proj_w, proj_b, *, img_size, nc_det, reg_max=16,
                   iou_match_thresh=0.5, label_smoothing=0.0, training=True,
                   weights=(1.0, 2.0, 1.5, 0.5, 1.0)):
    """det_maps: 3 x [B, 4*reg_max+nc, h, w]; protos [B, nm, hp, wp]; gt_boxes [G, 6] = (batch_idx, cls, cx, cy, w, h).
    Returns the reference's tuple: (total, seg, box, dfl, cls_det, img_cls[, n_pos, mean matched IoU])."""
    w_seg, w_box, w_dfl, w_cls, w_img = weights
    loss_img = F.cross_entropy(img_logits, gt_cls)
    seg_logits = F.interpolate(F.conv2d(protos, proj_w, proj_b), size=(img_size, img_size), mode="bilinear", align_corners=False)
    loss_seg = F.binary_cross_entropy_with_logits(seg_logits, gt_masks)

    B = det_maps[0].shape[0]
    project = torch.arange(reg_max, dtype=torch.float32)
    boxes, cls_logits, dists, anchors, strides = [], [], [], [], []
    for fm in det_maps:
        b, ch, h, w = fm.shape
        stride = img_size / w
        flat = fm.permute(0, 2, 3, 1).reshape(b, h * w, ch)
        raw = flat[..., : 4 * reg_max].view(b, h * w, 4, reg_max)
        ltrb = torch.einsum("ijkl,l->ijk", F.softmax(raw, dim=-1), project)
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        anc = torch.stack((gx + 0.5, gy + 0.5), dim=-1).view(1, h * w, 2).repeat(b, 1, 1)
        d, a = ltrb * stride, anc * stride
        boxes.append(torch.cat((a - d[..., :2], a + d[..., 2:]), dim=-1))
        cls_logits.append(flat[..., 4 * reg_max:])
        dists.append(raw)
        anchors.append(anc)
        strides.append(torch.full((b, h * w, 1), stride, dtype=torch.float32))
    boxes, cls_logits, dists = torch.cat(boxes, 1), torch.cat(cls_logits, 1), torch.cat(dists, 1)
    anchors, strides = torch.cat(anchors, 1), torch.cat(strides, 1)

    box_sum, cls_sum, dfl_sum, iou_sum, n_pos = 0.0, 0.0, 0.0, 0.0, 0
    for b in range(B):
        gt = gt_boxes[gt_boxes[:, 0] == b]
        if gt.numel() == 0:
            continue
        gcls, c = gt[:, 1].long(), gt[:, 2:6]
        gxyxy = torch.cat([(c[:, 0] - c[:, 2] / 2) * img_size, (c[:, 1] - c[:, 3] / 2) * img_size,
                           (c[:, 0] + c[:, 2] / 2) * img_size, (c[:, 1] + c[:, 3] / 2) * img_size], dim=-1).view(-1, 4)
        # (sic: the reference concatenates the four COLUMNS end to end and views as [-1, 4]; with G > 1 boxes this
        #  scrambles coordinates across boxes -- reproduced as is, running_main_v3.py:303-308)
        iou = batch_bbox_iou(boxes[b], gxyxy)
        best, gi = iou.max(dim=1)
        pos = best > iou_match_thresh
        k = int(pos.sum())
        if k == 0:
            continue
        n_pos += k
        mgt = gxyxy[gi[pos]]
        miou = batch_bbox_iou(boxes[b][pos], mgt).diag()
        box_sum = box_sum + (1.0 - miou).sum()
        iou_sum += float(miou.detach().sum())
        mlog, mcls = cls_logits[b][pos], gcls[gi[pos]]
        if label_smoothing > 0.0 and training:
            tgt = torch.full_like(mlog, label_smoothing / (nc_det - 1))
            tgt.scatter_(-1, mcls.unsqueeze(1), 1.0 - label_smoothing)
        else:
            tgt = F.one_hot(mcls, num_classes=nc_det).float()
        cls_sum = cls_sum + F.binary_cross_entropy_with_logits(mlog, tgt, reduction="sum")
        ap, st = anchors[b][pos] * strides[b][pos], strides[b][pos]
        t = (torch.cat([ap - mgt[:, :2], mgt[:, 2:] - ap], dim=-1) / st).clamp(min=0, max=reg_max - 1.01)
        tl = t.floor().long().clamp(min=0, max=reg_max - 1)
        tr = (tl + 1).clamp(min=0, max=reg_max - 1)
        wl, wr = tr.float() - t, t - tl.float()
        pd = dists[b][pos]
        for side in range(4):
            dfl_sum = dfl_sum + (F.cross_entropy(pd[:, side, :], tl[:, side], reduction="none") * wl[:, side]).sum() \
                              + (F.cross_entropy(pd[:, side, :], tr[:, side], reduction="none") * wr[:, side]).sum()
    norm = float(n_pos) if n_pos > 0 else float(B)
    to_t = lambda v: v if isinstance(v, torch.Tensor) else torch.tensor(float(v))
    box, cls_, dfl = to_t(box_sum) / norm, to_t(cls_sum) / norm, to_t(dfl_sum) / norm
    total = w_seg * loss_seg + w_box * box + w_dfl * dfl + w_cls * cls_ + w_img * loss_img
    out = (total, loss_seg, box, dfl, cls_, loss_img)
    if training:
        out = out + (torch.tensor(float(n_pos)), torch.tensor(iou_sum / n_pos if n_pos > 0 else 0.0))
    return out
