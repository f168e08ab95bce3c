"""COCO mAP host logic (multitask_bonetumor_yolo_amd.metrics) against hand-computed cases and the loop-form oracle.
torchmetrics / pycocotools are absent: PARITY UNPINNED against them; the algorithm is pycocotools' published one."""
import numpy as np
import torch

from oracle import metrics as O
from multitask_bonetumor_yolo_amd.metrics import MeanAveragePrecision, box_iou_xyxy


def _box(x, y, w=10.0, h=10.0):
    return [x, y, x + w, y + h]


def test_hand_computed_ap():
    # two GT boxes; detections by score: hit, miss, hit  -> recall .5 .5 1, precision 1 .5 2/3, envelope 1 2/3 2/3
    preds = [dict(boxes=[_box(0, 0), _box(200, 200), _box(50, 50)], scores=[0.9, 0.8, 0.7], labels=[0, 0, 0])]
    targets = [dict(boxes=[_box(0, 0), _box(50, 50)], labels=[0, 0])]
    m = MeanAveragePrecision(iou_thresholds=[0.5])
    m.update(preds, targets)
    r = m.compute()
    want = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    assert abs(r["map"] - want) < 1e-12 and abs(r["map_50"] - want) < 1e-12 and r["map_75"] == -1.0
    assert r["mar_1"] == 0.5 and r["mar_10"] == 1.0 and r["mar_100"] == 1.0


def test_perfect_none_and_unlabelled_class():
    t = [dict(boxes=[_box(0, 0), _box(30, 30)], labels=[0, 1])]
    m = MeanAveragePrecision()
    m.update([dict(boxes=[_box(0, 0), _box(30, 30)], scores=[0.5, 0.6], labels=[0, 1])], t)
    assert abs(m.compute()["map"] - 1.0) < 1e-12
    m.reset()
    m.update([dict(boxes=np.zeros((0, 4)), scores=[], labels=[])], t)
    assert m.compute()["map"] == 0.0 and m.compute()["mar_100"] == 0.0
    m.reset()   # detections of a class with no GT anywhere are left out of the mean (precision stays -1 for it)
    m.update([dict(boxes=[_box(0, 0), _box(99, 99)], scores=[0.5, 0.9], labels=[0, 7])], [dict(boxes=[_box(0, 0)], labels=[0])])
    assert abs(m.compute()["map"] - 1.0) < 1e-12
    assert MeanAveragePrecision().compute()["map"] == -1.0


def test_greedy_matching_takes_best_free_gt_and_iou_values():
    iou = box_iou_xyxy(np.array([_box(0, 0), _box(5, 0)]), np.array([_box(0, 0), _box(4, 0)]))
    assert abs(iou[0, 0] - 1.0) < 1e-12 and abs(iou[1, 1] - 90 / 110) < 1e-12 and abs(iou[0, 1] - 60 / 140) < 1e-12
    m = MeanAveragePrecision(iou_thresholds=[0.4])
    # the higher-scored detection overlaps both GTs and takes its best one; the second still finds the other
    m.update([dict(boxes=[_box(5, 0), _box(0, 0)], scores=[0.9, 0.8], labels=[0, 0])], [dict(boxes=[_box(0, 0), _box(4, 0)], labels=[0, 0])])
    assert abs(m.compute()["map"] - 1.0) < 1e-12


def test_matches_loop_oracle_on_random_sets():
    rng = np.random.default_rng(0)
    thr = np.linspace(0.5, 0.95, 10).tolist()
    for trial in range(4):
        preds, targets = [], []
        for _ in range(12):
            G, D = int(rng.integers(0, 4)), int(rng.integers(0, 14))
            gb = np.concatenate([rng.uniform(0, 80, (G, 2)), rng.uniform(8, 30, (G, 2))], 1)
            gb[:, 2:] += gb[:, :2]
            src = gb[rng.integers(0, G, D)] if G else np.zeros((D, 4))
            db = src + rng.normal(0, 2.5, (D, 4)) if G else np.concatenate([rng.uniform(0, 80, (D, 2)), rng.uniform(90, 120, (D, 2))], 1)
            scores = np.round(rng.uniform(0, 1, D), 1)                            # ties on purpose: stable ordering matters
            preds.append(dict(boxes=db, scores=scores, labels=rng.integers(0, 2, D)))
            targets.append(dict(boxes=gb, labels=rng.integers(0, 2, G)))
        m = MeanAveragePrecision(iou_thresholds=thr, max_detection_thresholds=[1, 3, 10])
        m.update([{k: torch.as_tensor(v) for k, v in p.items()} for p in preds], [{k: torch.as_tensor(v) for k, v in t.items()} for t in targets])
        got, want = m.compute(), O.coco_map(preds, targets, thr, (1, 3, 10))
        assert set(got) == set(want)
        for k in want:
            assert abs(got[k] - want[k]) < 1e-12, (trial, k, got[k], want[k])
