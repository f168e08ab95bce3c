"""Multitask loss value on the GPU: the reference's `MultiTaskLitModel._multitask_loss`
(`/root/reference/src/running_main_v3.py:232-387`) as three kernel launches (csrc/loss.hip) instead of a per-image Python
loop with `.item()` synchronisations.  The returned 0-d tensors carry no autograd history; `with_grads=True` additionally
returns the gradient of the total with respect to the head outputs -- what `trainstep.TrainStep` feeds the backward launch plan
(`grad_out` writes it straight into that plan's input buffers).  (Under the drop-in autograd route the trainer's own torch loss is
used instead, on the tensors `forward(x, "train")` returns.)

No host synchronisation: the ground-truth boxes are grouped by image with device-side tensor ops, the positive count and
the mean matched IoU come back as tensors (the reference returns Python floats, `:385`)."""
import ctypes as C
from typing import Sequence

import torch

from . import _lib as L
from .postprocess import _mask_call, _need_cuda, _nhwc_rows, _stream


def group_gt_boxes(gt_boxes: torch.Tensor, n_images: int, img_size: float):
    """[G,6] = (batch_idx, cls, cx, cy, w, h) normalised  ->  (xyxy [G,4] pixels grouped by image, cls [G] int32, off [N+1] int32).

    The pixel boxes are laid out exactly as the reference builds them (`:303-308`): per image it concatenates the four
    coordinate COLUMNS end to end and views the result as [-1, 4], which for G > 1 boxes mixes coordinates of different
    boxes -- reproduced, not repaired (drop-in parity)."""
    dev = gt_boxes.device
    G = gt_boxes.shape[0]
    off = torch.zeros(n_images + 1, dtype=torch.int32, device=dev)
    if G == 0:
        return torch.zeros((1, 4), dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), off
    bidx = gt_boxes[:, 0].long()
    order = torch.argsort(bidx, stable=True)
    g, bidx = gt_boxes[order].float(), bidx[order]
    counts = torch.zeros(n_images, dtype=torch.long, device=dev).scatter_add_(0, bidx.clamp(0, n_images - 1), torch.ones_like(bidx))
    starts = torch.cumsum(counts, 0) - counts
    off[1:] = torch.cumsum(counts, 0).int()
    cols = torch.stack([(g[:, 2] - g[:, 4] / 2) * img_size, (g[:, 3] - g[:, 5] / 2) * img_size,
                        (g[:, 2] + g[:, 4] / 2) * img_size, (g[:, 3] + g[:, 5] / 2) * img_size], 0)      # [4, G]
    gi, o = counts[bidx], starts[bidx]
    local = torch.arange(G, device=dev) - o
    q = 4 * local[:, None] + torch.arange(4, device=dev)[None, :]                                       # position in cat(...)
    xyxy = cols[q // gi[:, None], o[:, None] + q % gi[:, None]]
    return xyxy.contiguous(), g[:, 1].int().contiguous(), off


def multitask_loss(det_maps: Sequence[torch.Tensor], protos: torch.Tensor, img_logits: torch.Tensor, gt_boxes: torch.Tensor,
                   gt_masks: torch.Tensor, gt_cls: torch.Tensor, proj_weight: torch.Tensor, proj_bias: torch.Tensor, *, img_size: int,
                   nc_det: int, reg_max: int = 16, iou_match_thresh: float = 0.5, label_smoothing: float = 0.0, training: bool = True,
                   weights=(1.0, 2.0, 1.5, 0.5, 1.0), with_grads: bool = False, grad_out=None):
    """det_maps: the raw Detect maps of `forward(x, "train")` (3 x [B, 4*reg_max+nc, h, w]); protos [B, nm, hp, wp];
    gt_masks [B,1,S,S] float; gt_cls [B] int64; proj_*: the trainer's `seg_proto_projector` (`:186`).
    Returns the reference's tuple as 0-d fp32 tensors: (total, seg, box, dfl, cls_det, img_cls[, n_pos, mean matched IoU]).
    `with_grads=True` returns `(that tuple, grads)` where grads = d total / d {det_maps (list, like det_maps), seg_logits [B,1,S,S]
    (the projector output after the bilinear resize, `:251-255`), img_logits}: the first operator of the backward pass."""
    lib = L.load()
    _need_cuda(det_maps[0], "multitask_loss")
    dev = det_maps[0].device
    B = det_maps[0].shape[0]
    a = L.LossArgs()
    keep = []
    A = 0
    for i, m in enumerate(det_maps):
        t, ld = _nhwc_rows(m)
        keep.append(t)
        a.map[i], a.h[i], a.w[i], a.map_pixel_stride[i] = t.data_ptr(), m.shape[2], m.shape[3], ld
        A += m.shape[2] * m.shape[3]
    a.n_levels, a.N, a.nc, a.reg_max, a.img_size = len(det_maps), B, nc_det, reg_max, float(img_size)
    xyxy, gcls, off = group_gt_boxes(gt_boxes.to(dev), B, float(img_size))
    a.gt_xyxy, a.gt_cls, a.gt_off = xyxy.data_ptr(), gcls.data_ptr(), off.data_ptr()
    a.iou_thresh, a.label_smoothing, a.training = float(iou_match_thresh), float(label_smoothing), int(training)
    # segmentation logits: Conv1x1(protos) -> bilinear S x S (mtbt_mask_assemble's projector path), bias added in the kernel
    w = proj_weight.detach().reshape(-1).float().contiguous()
    seg_logits, _ = _mask_call(protos, w, 0, 0, 1, None, None, 0.0, 1, (img_size, img_size), True, False)
    tgt = gt_masks.to(dev).float().contiguous()
    bias = proj_bias.detach().reshape(-1).float().contiguous()
    a.seg_logits, a.seg_targets, a.seg_bias, a.seg_n = seg_logits.data_ptr(), tgt.data_ptr(), bias.data_ptr(), seg_logits.numel()
    il = img_logits.float().contiguous()
    ig = gt_cls.to(dev).long().contiguous()
    a.img_logits, a.img_gt, a.n_img_classes = il.data_ptr(), ig.data_ptr(), il.shape[1]
    a.w_seg, a.w_box, a.w_dfl, a.w_cls, a.w_img = (float(v) for v in weights)
    nbytes = lib.mtbt_loss_workspace_bytes(B, A, seg_logits.numel())
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out = torch.empty(8, dtype=torch.float32, device=dev)
    a.workspace, a.workspace_bytes, a.out = ws.data_ptr(), nbytes, out.data_ptr()
    L.check(lib.mtbt_multitask_loss(C.byref(a), _stream(dev)), "mtbt_multitask_loss")
    res = tuple(out[i] for i in range(8 if training else 6))
    if not with_grads:
        del keep
        return res
    # gradient of the total w.r.t. the head outputs (csrc/loss.hip: det_loss_grad_kernel, seg_img_grad_kernel)
    no = 4 * reg_max + nc_det
    # grad_out = {"det_maps": [NHWC fp32 buffers], "img_logits": [B, n] fp32}: write straight into a training plan's input buffers
    d_maps = (list(grad_out["det_maps"]) if grad_out is not None else
              [torch.empty(m.shape[0], m.shape[2], m.shape[3], no, dtype=torch.float32, device=dev) for m in det_maps])
    ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in d_maps], *([None] * (3 - len(d_maps))))
    lds = (C.c_int32 * 3)(*([no] * len(d_maps)), *([0] * (3 - len(d_maps))))
    d_seg = torch.empty_like(seg_logits)
    d_img = grad_out["img_logits"] if grad_out is not None else torch.empty_like(il)
    L.check(lib.mtbt_multitask_loss_grad(C.byref(a), ptrs, lds, d_seg.data_ptr(), d_img.data_ptr(), _stream(dev)), "mtbt_multitask_loss_grad")
    del keep
    grads = {"det_maps": [t.permute(0, 3, 1, 2) for t in d_maps],      # [B, no, h, w] views of channels-last memory
             "seg_logits": d_seg.view(B, 1, img_size, img_size), "img_logits": d_img}
    return res, grads
