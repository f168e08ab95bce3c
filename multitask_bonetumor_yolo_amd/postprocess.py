"""Device post-process over the HIP kernels: box decode, per-image NMS, mask assembly, proto projector.

Mirrors what the reference's trainer does on the model outputs
(`/root/reference/src/running_main_v3.py:510-552`, `:251-257`; `/root/reference/src/test_model.py:80-85`),
batched on the GPU with no per-image Python loop and no host synchronisation.  No CPU path.
"""
import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib as L

CONF_TH, NMS_IOU, TOP_K = 0.05, 0.6, 100  # running_main_v3.py:54-56


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a CUDA/HIP tensor on an MI355X (no CPU path)")


def _nhwc_rows(t: torch.Tensor):
    """[N,C,H,W] logical tensor -> (fp32 tensor whose memory is NHWC rows, pixel stride).  Zero-copy for the
    channels-last tensors the model returns; NCHW-contiguous input is re-laid out once (a torch copy)."""
    assert t.dim() == 4
    t = t.float() if t.dtype != torch.float32 else t
    n, c, h, w = t.shape
    if not (t.stride(1) == 1 and t.stride(2) == w * t.stride(3) and t.stride(0) == h * w * t.stride(3)):
        t = t.contiguous(memory_format=torch.channels_last)
    return t, t.stride(3)


def decode_boxes(det_maps: Sequence[torch.Tensor], img_size: Optional[float] = None, reg_max: int = 16, xywh: bool = False,
                 strides: Optional[Sequence[float]] = None, preds_cat: Optional[torch.Tensor] = None, want_scores: bool = True):
    """Decode raw Detect maps (per level [B, 4*reg_max+nc, h, w]).

    Trainer semantics (running_main_v3.py:510-533) by default: xyxy pixels with stride = img_size / w.
    `xywh=True` + explicit `strides` gives ultralytics `Detect._inference` (stride may be 0).
    Returns dict(boxes [B,A,4], scores [B,A,nc] sigmoid, best_score [B,A], best_label [B,A] int32)."""
    lib = L.load()
    maps = []
    for m in det_maps:
        _need_cuda(m, "decode_boxes")
        maps.append(_nhwc_rows(m))
    B, no = det_maps[0].shape[0], det_maps[0].shape[1]
    nc = no - 4 * reg_max
    dev = det_maps[0].device
    A = sum(m.shape[2] * m.shape[3] for m in det_maps)
    a = L.DecodeArgs()
    for i, ((t, ld), m) in enumerate(zip(maps, det_maps)):
        a.map[i] = t.data_ptr()
        a.h[i], a.w[i], a.map_pixel_stride[i] = m.shape[2], m.shape[3], ld
        a.stride[i] = float(strides[i]) if strides is not None else float(img_size) / m.shape[3]
    a.n_levels, a.N, a.nc, a.reg_max, a.xywh = len(maps), B, nc, reg_max, int(xywh)
    out = {}
    if preds_cat is None:
        out["boxes"] = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        out["best_score"] = torch.empty((B, A), dtype=torch.float32, device=dev)
        out["best_label"] = torch.empty((B, A), dtype=torch.int32, device=dev)
        a.boxes, a.best_score, a.best_label = out["boxes"].data_ptr(), out["best_score"].data_ptr(), out["best_label"].data_ptr()
        if want_scores:
            out["scores"] = torch.empty((B, A, nc), dtype=torch.float32, device=dev)
            a.scores = out["scores"].data_ptr()
    else:
        assert preds_cat.is_contiguous() and preds_cat.shape[:2] == (B, A)
        a.preds_cat, a.cat_stride = preds_cat.data_ptr(), preds_cat.shape[2]
    L.check(lib.mtbt_decode_boxes(C.byref(a), _stream(dev)), "mtbt_decode_boxes")
    out["_keep"] = maps
    return out


def detect_inference(maps, head, mc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ultralytics `Detect._inference` / `Segment.forward` eval output: [B, 4+nc(+nm), A] =
    cat(dist2bbox(dfl(box), anchors, xywh) * head.stride, sigmoid(cls)(, mask coefficients)).
    `maps`: engine.Act list (plan-owned raw maps); `mc`: [B,A,nm] buffer.  Returned as a [B,4+nc+nm,A] view of
    a fresh [B,A,4+nc+nm] buffer."""
    B = maps[0].N
    A = sum(m.H * m.W for m in maps)
    width = 4 + head.nc + (mc.shape[2] if mc is not None else 0)
    cat = torch.empty((B, A, width), dtype=torch.float32, device=maps[0].buf.device)
    decode_boxes([m.nchw() for m in maps], reg_max=head.reg_max, xywh=True, strides=[float(s) for s in head.stride], preds_cat=cat)
    if mc is not None:
        cat[:, :, 4 + head.nc:].copy_(mc)
    return cat.permute(0, 2, 1)


def batch_bbox_iou(boxes1: torch.Tensor, boxes2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """running_main_v3.py:71-97: pairwise IoU [N,M] of xyxy boxes; an empty side gives the reference's zero matrix."""
    lib = L.load()
    _need_cuda(boxes1, "batch_bbox_iou")
    n, m = boxes1.shape[0], boxes2.shape[0]
    out = torch.zeros((n, m), dtype=torch.float32, device=boxes1.device)
    if n and m:
        b1, b2 = boxes1.contiguous().float(), boxes2.to(boxes1.device).contiguous().float()
        L.check(lib.mtbt_bbox_iou_pairwise(b1.data_ptr(), n, b2.data_ptr(), m, C.c_float(eps), out.data_ptr(), _stream(boxes1.device)),
                "mtbt_bbox_iou_pairwise")
    return out


def nms_batched(boxes: torch.Tensor, best_score: torch.Tensor, best_label: Optional[torch.Tensor], clamp_max: float,
                conf_th: float = CONF_TH, iou_th: float = NMS_IOU, top_k: int = TOP_K):
    """running_main_v3.py:535-552 for the whole batch: score > conf_th, clamp to [0, clamp_max], greedy NMS
    (torchvision.ops.nms arithmetic and ordering), first top_k.  boxes [B,A,4] xyxy, best_score [B,A].
    Returns dict(keep_idx int64 [B,top_k] (index into the confidence-filtered list, -1 padded), keep_anchor int32,
    boxes [B,top_k,4], scores, labels int64, counts int32 [B], n_cand int32 [B])."""
    lib = L.load()
    _need_cuda(boxes, "nms_batched")
    boxes = boxes.contiguous().float()
    best_score = best_score.contiguous().float()
    B, A = best_score.shape
    dev = boxes.device
    if best_label is not None:
        best_label = best_label.contiguous().to(torch.int32)
    o = {
        "keep_idx": torch.empty((B, top_k), dtype=torch.int64, device=dev),
        "keep_anchor": torch.empty((B, top_k), dtype=torch.int32, device=dev),
        "boxes": torch.empty((B, top_k, 4), dtype=torch.float32, device=dev),
        "scores": torch.empty((B, top_k), dtype=torch.float32, device=dev),
        "labels": torch.empty((B, top_k), dtype=torch.int64, device=dev),
        "counts": torch.empty((B,), dtype=torch.int32, device=dev),
        "n_cand": torch.empty((B,), dtype=torch.int32, device=dev),
    }
    wsb = lib.mtbt_nms_workspace_bytes(B, A)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    rc = lib.mtbt_nms_batched(boxes.data_ptr(), best_score.data_ptr(), best_label.data_ptr() if best_label is not None else None,
                              B, A, conf_th, iou_th, clamp_max, top_k, o["keep_idx"].data_ptr(), o["keep_anchor"].data_ptr(),
                              o["boxes"].data_ptr(), o["scores"].data_ptr(), o["labels"].data_ptr(), o["counts"].data_ptr(),
                              o["n_cand"].data_ptr(), ws.data_ptr(), wsb, _stream(dev))
    L.check(rc, "mtbt_nms_batched")
    o["_keep"] = (boxes, best_score, best_label, ws)
    return o


def _mask_call(protos, coeff, cbs, cks, ccs, gather, counts, bias, K, out_hw, want_logits, want_masks):
    lib = L.load()
    _need_cuda(protos, "mask assembly")
    pr, ld = _nhwc_rows(protos)
    B, nm, hp, wp = protos.shape
    if ld != nm:
        pr = pr.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    dev = protos.device
    H, W = out_hw
    a = L.MaskArgs()
    a.protos, a.coeff = pr.data_ptr(), coeff.data_ptr()
    a.coeff_batch_stride, a.coeff_k_stride, a.coeff_c_stride = cbs, cks, ccs
    a.gather_idx = gather.data_ptr() if gather is not None else None
    a.counts = counts.data_ptr() if counts is not None else None
    a.bias = float(bias)
    a.N, a.K, a.nm, a.hp, a.wp, a.Hout, a.Wout = B, K, nm, hp, wp, H, W
    logits = torch.empty((B, K, H, W), dtype=torch.float32, device=dev) if want_logits else None
    masks = torch.empty((B, K, H, W), dtype=torch.bool, device=dev) if want_masks else None
    a.logits = logits.data_ptr() if logits is not None else None
    a.masks = masks.data_ptr() if masks is not None else None
    L.check(lib.mtbt_mask_assemble(C.byref(a), _stream(dev)), "mtbt_mask_assemble")
    return logits, masks


def assemble_masks(protos: torch.Tensor, mc: torch.Tensor, keep_anchor: torch.Tensor, counts: Optional[torch.Tensor],
                   out_size, want_logits: bool = False):
    """Instance masks of the kept boxes (test_model.py:80-85 intended form): masks[b,k] =
    sigmoid(bilinear(sum_c mc[b,c,anchor(b,k)] * protos[b,c])) > 0.5.
    protos [B,nm,hp,wp]; mc [B,nm,A] (any strides, fp32); keep_anchor int32 [B,K]; counts int32 [B] or None.
    Rows k >= counts[b] are zero.  Returns (masks bool [B,K,H,W], logits or None)."""
    assert mc.dtype == torch.float32
    keep_anchor = keep_anchor.contiguous().to(torch.int32)
    if counts is not None:
        counts = counts.contiguous().to(torch.int32)
    logits, masks = _mask_call(protos, mc, mc.stride(0), mc.stride(2), mc.stride(1), keep_anchor, counts, 0.0,
                               keep_anchor.shape[1], tuple(out_size), want_logits, True)
    return masks, logits


def proto_projector_logits(protos: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, img_size: int) -> torch.Tensor:
    """Trainer's seg_proto_projector path (running_main_v3.py:186, :251-255): Conv2d(nm,1,1)(protos) then bilinear
    to img_size.  Returns logits [B,1,S,S] fp32."""
    w = weight.detach().reshape(-1).float().contiguous()
    logits, _ = _mask_call(protos, w, 0, 0, 1, None, None, 0.0, 1, (img_size, img_size), True, False)
    return logits.add_(bias.detach().reshape(1, 1, 1, 1).to(logits.dtype))      # device-side add: no host synchronisation (bilinear weights sum to 1)


def detect_and_segment(det_maps: List[torch.Tensor], mc: torch.Tensor, protos: torch.Tensor, img_size: int,
                       conf_th: float = CONF_TH, iou_th: float = NMS_IOU, top_k: int = TOP_K, masks: bool = True):
    """The whole validation post-process for a batch: decode -> filter/NMS/top-k -> instance masks."""
    d = decode_boxes(det_maps, img_size, want_scores=False)
    k = nms_batched(d["boxes"], d["best_score"], d["best_label"], float(img_size), conf_th, iou_th, top_k)
    out = {"boxes": k["boxes"], "scores": k["scores"], "labels": k["labels"], "counts": k["counts"],
           "keep_idx": k["keep_idx"], "keep_anchor": k["keep_anchor"], "n_cand": k["n_cand"]}
    if masks:
        out["masks"], _ = assemble_masks(protos, mc, k["keep_anchor"], k["counts"], (img_size, img_size))
    return out
