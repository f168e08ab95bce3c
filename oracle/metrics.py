"""Oracle restatement of the metric definitions behind the reference's validation loop
(`/root/reference/src/running_main_v3.py:198-217`, `:466-498`, `:535-575`).

TEST INFRASTRUCTURE.  PARITY UNPINNED: the reference delegates to torchmetrics (`MeanAveragePrecision`, binary F1 /
precision / recall / accuracy, `DiceScore`), which is absent and unversioned.  This file follows the text of pycocotools'
`COCOeval` (evaluateImg / accumulate) with plain Python loops -- no vectorisation shared with the product's
implementation -- and is itself anchored by hand-computed cases in tests/test_cpu_metrics.py."""
import numpy as np
import torch


def box_iou(d, g):
    iw = min(d[2], g[2]) - max(d[0], g[0])
    ih = min(d[3], g[3]) - max(d[1], g[1])
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    return inter / ((d[2] - d[0]) * (d[3] - d[1]) + (g[2] - g[0]) * (g[3] - g[1]) - inter)


def coco_map(preds, targets, iou_thresholds, max_dets=(1, 10, 100)):
    """preds: per image dict(boxes, scores, labels) of lists; targets: dict(boxes, labels).  -> dict like torchmetrics'."""
    rec_thrs = list(np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True))   # COCOeval.Params.recThrs (not i / 100)
    cats = sorted({int(c) for p in preds for c in p["labels"]} | {int(c) for t in targets for c in t["labels"]})
    prec, rec = {}, {}
    for c in cats:
        for md in max_dets:
            entries, npig = [], 0                                   # (score, [matched at threshold t])
            for p, t in zip(preds, targets):
                gts = [b for b, l in zip(t["boxes"], t["labels"]) if int(l) == c]
                dts = sorted([(float(s), i) for i, (s, l) in enumerate(zip(p["scores"], p["labels"])) if int(l) == c], key=lambda v: -v[0])[:md]
                npig += len(gts)
                matched = {thr: [-1] * len(gts) for thr in iou_thresholds}
                for s, i in dts:
                    flags = []
                    for thr in iou_thresholds:
                        best, m = min(thr, 1 - 1e-10), -1
                        for gi, g in enumerate(gts):
                            if matched[thr][gi] >= 0:
                                continue
                            v = box_iou([float(x) for x in p["boxes"][i]], [float(x) for x in g])
                            if v < best:
                                continue
                            best, m = v, gi
                        if m >= 0:
                            matched[thr][m] = i
                        flags.append(m >= 0)
                    entries.append((s, flags))
            if npig == 0:
                continue
            entries.sort(key=lambda e: -e[0])                       # Python's sort is stable, like mergesort
            for ti, thr in enumerate(iou_thresholds):
                tp = fp = 0
                rc, pr = [], []
                for s, flags in entries:
                    tp, fp = tp + (1 if flags[ti] else 0), fp + (0 if flags[ti] else 1)
                    rc.append(tp / npig)
                    pr.append(tp / (fp + tp + np.spacing(1)))
                rec[(ti, c, md)] = rc[-1] if rc else 0.0
                for i in range(len(pr) - 1, 0, -1):
                    if pr[i] > pr[i - 1]:
                        pr[i - 1] = pr[i]
                q = []
                for r in rec_thrs:
                    j = int(np.searchsorted(rc, r, side="left"))
                    q.append(pr[j] if j < len(pr) else 0.0)
                prec[(ti, c, md)] = q
    last = max_dets[-1]
    mean = lambda v: float(np.mean(v)) if len(v) else -1.0
    out = {"map": mean([x for (ti, c, md), q in prec.items() if md == last for x in q])}
    for name, thr in (("map_50", 0.5), ("map_75", 0.75)):
        hit = [i for i, t in enumerate(iou_thresholds) if abs(t - thr) < 1e-9]
        out[name] = mean([x for (ti, c, md), q in prec.items() if md == last and hit and ti == hit[0] for x in q]) if hit else -1.0
    for md in max_dets:
        out[f"mar_{md}"] = mean([v for (ti, c, m), v in rec.items() if m == md])
    return out


def seg_counts(seg_logits: torch.Tensor, masks_gt: torch.Tensor):
    """Per image TP, FP, FN, TN and the mask score of running_main_v3.py:470-483, with torch ops as the reference writes them."""
    probs = seg_logits.sigmoid()
    gt = masks_gt.int()
    out, scores = [], []
    for i in range(probs.shape[0]):
        pred = probs[i] > 0.5
        tgt = gt[i] > 0.5
        out.append([int((pred & tgt).sum()), int((pred & ~tgt).sum()), int((~pred & tgt).sum()), int((~pred & ~tgt).sum())])
        scores.append(float((probs[i] * pred.float()).sum() / (pred.float().sum() + 1e-6)))
    return np.array(out, np.int64), np.array(scores, np.float32)
