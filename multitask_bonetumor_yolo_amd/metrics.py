"""Validation metric accumulators with the reference's definitions.

The reference trainer builds torchmetrics objects (`/root/reference/src/running_main_v3.py:198-217`) and feeds them in
`validation_step` (`:466-498` segmentation, `:535-575` boxes).  torchmetrics (and its pycocotools / faster-coco-eval
backend) is a third-party dependency that is absent here and unversioned in the reference: PARITY UNPINNED -- what is
restated is the published COCO evaluation (pycocotools `COCOeval.evaluateImg` / `accumulate` / `summarize`) and
torchmetrics' binary stat-score definitions.

  SegmentationMetrics     pixel counts on the device (`mtbt_seg_confusion`: one pass over logits + gt, no host sync in
                          `update`); F1 / precision / recall / accuracy / Dice / IoU and the single-instance mask mAP of
                          :480-497 from those counts in `compute`.
  MeanAveragePrecision    COCO box mAP over IoU thresholds with `max_detection_thresholds` (mAP@0.5 and @0.5:0.95 of :206-214):
                          host-side numpy -- at most 100 kept boxes per image after the device NMS, a few GT boxes.

Data-parallel validation (configs[3]: one process per GPU, each rank sees its shard of the validation set): the reference's metric
objects are built with `dist_sync_on_step=True` (`running_main_v3.py:193-218`), i.e. torchmetrics gathers every rank's state before it
computes.  Here `compute()` does the same when a `torch.distributed` process group with more than one rank is alive (`dist_sync=True`,
the default): the per-image records (detections x ground-truth IoU matrices; pixel counts) of all ranks are all-gathered in rank order,
every rank computes the SAME global value, and the local state is left as it was (torchmetrics' sync / unsync around compute).
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L


def _world(group=None) -> int:
    import torch.distributed as dist
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _all_gather_records(local, group=None):
    """Every rank's list of per-image records, concatenated in rank order (a few KB per image: host objects; RCCL moves the pickled bytes
    through device tensors, gloo through the host).  All ranks must call it."""
    import torch.distributed as dist
    world = _world(group)
    if world == 1:
        return list(local)
    parts = [None] * world
    dist.all_gather_object(parts, list(local), group=group)
    return [rec for part in parts for rec in part]


def box_iou_xyxy(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """COCO box IoU in float64 (pycocotools `bbIou` without crowd boxes): [D,4] x [G,4] -> [D,G]."""
    a, b = np.asarray(a, np.float64).reshape(-1, 4), np.asarray(b, np.float64).reshape(-1, 4)
    iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    union = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union > 0, inter / union, 0.0)


class MeanAveragePrecision:
    """COCO mAP / mAR for one area range ("all"), any IoU thresholds, any max-detection thresholds.

    `update(preds, targets)` takes the torchmetrics layout the reference builds (`running_main_v3.py:553-575`): per image
    `dict(boxes [D,4] xyxy, scores [D], labels [D])` and `dict(boxes [G,4], labels [G])`."""

    def __init__(self, iou_thresholds: Optional[Sequence[float]] = None, max_detection_thresholds: Sequence[int] = (1, 10, 100),
                 dist_sync: bool = True, process_group=None):
        self.dist_sync, self.group = dist_sync, process_group
        self.iou_thresholds = np.asarray(iou_thresholds if iou_thresholds is not None else np.linspace(0.5, 0.95, 10), np.float64)
        self.max_dets = sorted(int(m) for m in max_detection_thresholds)
        self.rec_thresholds = np.linspace(0.0, 1.0, 101)
        self._images: List[tuple] = []

    def reset(self):
        self._images = []

    @staticmethod
    def _np(t, dtype):
        return (t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)).astype(dtype)

    def update(self, preds: Sequence[Dict], targets: Sequence[Dict]):
        if len(preds) != len(targets):
            raise ValueError("MeanAveragePrecision.update: preds and targets differ in length")
        for p, t in zip(preds, targets):
            db, gb = self._np(p["boxes"], np.float64).reshape(-1, 4), self._np(t["boxes"], np.float64).reshape(-1, 4)
            self.add_image(self._np(p["scores"], np.float64), self._np(p["labels"], np.int64), self._np(t["labels"], np.int64), box_iou_xyxy(db, gb))

    def add_image(self, scores, labels, gt_labels, iou):
        """One image with a precomputed IoU matrix [D, G] (boxes above; mask IoU for the segmentation mAP)."""
        scores, labels, gt_labels = np.asarray(scores, np.float64).ravel(), np.asarray(labels, np.int64).ravel(), np.asarray(gt_labels, np.int64).ravel()
        iou = np.asarray(iou, np.float64).reshape(len(scores), len(gt_labels))
        self._images.append((scores, labels, gt_labels, iou))

    def _match(self, iou: np.ndarray) -> np.ndarray:
        """pycocotools evaluateImg without crowd / ignore flags: detections in score order, each takes the still unmatched GT
        of highest IoU >= threshold (of equal IoUs the later GT).  -> matched [T, D] bool."""
        D, G = iou.shape
        out = np.zeros((len(self.iou_thresholds), D), bool)
        if G == 0:
            return out
        for ti, t in enumerate(self.iou_thresholds):
            free = np.ones(G, bool)
            lim = min(t, 1 - 1e-10)
            for d in range(D):
                v = np.where(free, iou[d], -1.0)
                m = G - 1 - int(np.argmax(v[::-1]))
                if v[m] >= lim:
                    free[m] = False
                    out[ti, d] = True
        return out

    def compute(self) -> Dict[str, float]:
        """With a live process group (and dist_sync): over the images of ALL ranks -- a collective, every rank must call it."""
        images = _all_gather_records(self._images, self.group) if self.dist_sync else self._images
        T, R, M = len(self.iou_thresholds), len(self.rec_thresholds), len(self.max_dets)
        classes = sorted(set(int(c) for im in images for c in np.concatenate([im[1], im[2]])))
        precision, recall = -np.ones((T, R, len(classes), M)), -np.ones((T, len(classes), M))
        for k, c in enumerate(classes):
            per_image, npig = [], 0
            for scores, labels, gt_labels, iou in images:
                di, gi = np.nonzero(labels == c)[0], np.nonzero(gt_labels == c)[0]
                npig += len(gi)
                if len(di):
                    di = di[np.argsort(-scores[di], kind="mergesort")][: self.max_dets[-1]]
                    per_image.append((scores[di], iou[np.ix_(di, gi)]))
            if npig == 0:
                continue
            for m, maxdet in enumerate(self.max_dets):
                sc = np.concatenate([s[:maxdet] for s, _ in per_image]) if per_image else np.zeros(0)
                tp = np.concatenate([self._match(i[:maxdet]) for _, i in per_image], axis=1) if per_image else np.zeros((T, 0), bool)
                order = np.argsort(-sc, kind="mergesort")
                tp = tp[:, order]
                tps, fps = np.cumsum(tp, axis=1).astype(np.float64), np.cumsum(~tp, axis=1).astype(np.float64)
                for t in range(T):
                    nd = tps.shape[1]
                    rc = tps[t] / npig
                    pr = tps[t] / (fps[t] + tps[t] + np.spacing(1))
                    recall[t, k, m] = rc[-1] if nd else 0.0
                    pr = np.maximum.accumulate(pr[::-1])[::-1]                       # precision envelope
                    inds = np.searchsorted(rc, self.rec_thresholds, side="left")
                    q = np.zeros(R)
                    ok = inds < nd
                    q[ok] = pr[inds[ok]]
                    precision[t, :, k, m] = q

        def mean(a):
            a = a[a > -1]
            return float(a.mean()) if a.size else -1.0

        out = {"map": mean(precision[:, :, :, -1])}
        for name, thr in (("map_50", 0.5), ("map_75", 0.75)):
            hit = np.nonzero(np.isclose(self.iou_thresholds, thr))[0]
            out[name] = mean(precision[hit[0], :, :, -1]) if len(hit) else -1.0
        for m, maxdet in enumerate(self.max_dets):
            out[f"mar_{maxdet}"] = mean(recall[:, :, m])
        return out


class SegmentationMetrics:
    """Binary segmentation metrics of `validation_step` (`running_main_v3.py:466-498`) from device-side pixel counts."""

    def __init__(self, dist_sync: bool = True, process_group=None):
        self.dist_sync, self.group = dist_sync, process_group
        self._counts: List[torch.Tensor] = []
        self._psum: List[torch.Tensor] = []

    def reset(self):
        self._counts, self._psum = [], []

    def update(self, seg_logits: torch.Tensor, masks_gt: torch.Tensor):
        """seg_logits, masks_gt: [B,1,S,S] (or [B,S,S]) fp32 CUDA tensors.  Asynchronous: no host synchronisation."""
        if not (seg_logits.is_cuda and masks_gt.is_cuda):
            raise RuntimeError("SegmentationMetrics.update: expected CUDA/HIP tensors on an MI355X (no CPU path)")
        if seg_logits.shape != masks_gt.shape:
            raise ValueError("SegmentationMetrics.update: logits and masks differ in shape")
        lib = L.load()
        x, t = seg_logits.float().contiguous(), masks_gt.float().contiguous()
        B, n = x.shape[0], x[0].numel()
        dev = x.device
        counts = torch.empty(B, 4, dtype=torch.int64, device=dev)
        psum = torch.empty(B, dtype=torch.float32, device=dev)
        ws_bytes = lib.mtbt_seg_confusion_workspace_bytes(B)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        L.check(lib.mtbt_seg_confusion(x.data_ptr(), t.data_ptr(), B, n, counts.data_ptr(), psum.data_ptr(), ws.data_ptr(), ws_bytes,
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtbt_seg_confusion")
        self._counts.append(counts)
        self._psum.append(psum)

    def per_image(self, sync: bool = False):
        """-> (counts [N,4] int64 = TP, FP, FN, TN; mask scores [N] = sum(prob * mask) / (sum(mask) + 1e-6), :483).
        `sync`: the images of all ranks in rank order (a collective)."""
        if self._counts:
            c = torch.cat(self._counts).cpu().numpy()
            p = torch.cat(self._psum).cpu().numpy()
        else:
            c, p = np.zeros((0, 4), np.int64), np.zeros(0, np.float32)
        if sync and _world(self.group) > 1:
            recs = _all_gather_records([(c, p)], self.group)
            c, p = np.concatenate([r[0] for r in recs]), np.concatenate([r[1] for r in recs])
        return c, (p / ((c[:, 0] + c[:, 1]).astype(np.float32) + np.float32(1e-6))).astype(np.float32)

    def compute(self) -> Dict[str, float]:
        """With a live process group (and dist_sync): from the pixel counts of ALL ranks -- a collective, every rank must call it."""
        c, score = self.per_image(sync=self.dist_sync)
        tp, fp, fn, tn = (float(v) for v in c.sum(axis=0)) if len(c) else (0.0, 0.0, 0.0, 0.0)
        div = lambda a, b: a / b if b else 0.0                                    # torchmetrics _safe_divide
        out = {"f1": div(2 * tp, 2 * tp + fp + fn), "precision": div(tp, tp + fp), "recall": div(tp, tp + fn),
               "accuracy": div(tp + tn, tp + tn + fp + fn), "iou": div(tp, tp + fp + fn), "dice_global": div(2 * tp, 2 * tp + fp + fn)}
        den = (2 * c[:, 0] + c[:, 1] + c[:, 2]).astype(np.float64)
        dice = np.where(den > 0, 2 * c[:, 0] / np.maximum(den, 1), np.nan)         # per sample; empty-vs-empty samples are skipped
        out["dice"] = float(np.nanmean(dice)) if np.any(den > 0) else 0.0
        # segmentation mAP (:478-497): one predicted instance (class 0) and one GT instance per image
        m = MeanAveragePrecision(dist_sync=False)                              # (c, score already hold every rank's images)
        union = (c[:, 0] + c[:, 1] + c[:, 2]).astype(np.float64)
        iou = np.where(union > 0, c[:, 0] / np.maximum(union, 1), 0.0)
        for i in range(len(c)):
            m.add_image([score[i]], [0], [0], [[iou[i]]])
        seg = m.compute() if len(c) else {"map": -1.0, "map_50": -1.0}
        out["seg_map"], out["seg_map_50"] = seg["map"], seg["map_50"]
        return out
