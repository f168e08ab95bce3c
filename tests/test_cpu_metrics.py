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


def test_gloo_world_size_2_metric_sync(tmp_path):
    """Data-parallel validation (configs[3]): the reference's metrics are `dist_sync_on_step=True` (running_main_v3.py:193-218) -- every
    rank's state is gathered before compute.  Two gloo ranks hold DIFFERENT images; `compute()` on each returns the value one process
    holding all images (in rank order) computes, for the box mAP and for the segmentation counts; the local state is untouched."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, json, numpy as np, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "from multitask_bonetumor_yolo_amd.metrics import MeanAveragePrecision, SegmentationMetrics\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "def images(rank):\n"
        "    rng = np.random.default_rng(100 + rank)\n"
        "    P, T = [], []\n"
        "    for _ in range(5 + rank):\n"
        "        G, D = int(rng.integers(1, 4)), int(rng.integers(0, 9))\n"
        "        gb = np.concatenate([rng.uniform(0, 80, (G, 2)), rng.uniform(8, 30, (G, 2))], 1); gb[:, 2:] += gb[:, :2]\n"
        "        db = gb[rng.integers(0, G, D)] + rng.normal(0, 3.0, (D, 4))\n"
        "        P.append(dict(boxes=db, scores=np.round(rng.uniform(0, 1, D), 2), labels=rng.integers(0, 2, D)))\n"
        "        T.append(dict(boxes=gb, labels=rng.integers(0, 2, G)))\n"
        "    return P, T\n"
        "def counts(rank):\n"
        "    g = torch.Generator().manual_seed(7 + rank)\n"
        "    return torch.randint(0, 500, (3 + rank, 4), generator=g), torch.rand(3 + rank, generator=g) * 100\n"
        "m = MeanAveragePrecision(); m.update(*images(r)); n_local = len(m._images)\n"
        "got = m.compute()\n"
        "ref = MeanAveragePrecision(dist_sync=False)\n"
        "for k in range(w): ref.update(*images(k))\n"
        "want = ref.compute()\n"
        "assert got == want and len(m._images) == n_local, (got, want)\n"
        "local = MeanAveragePrecision(dist_sync=False); local.update(*images(r))\n"
        "assert local.compute() != want          # the rank-local value (what round 2 reported) is a different number\n"
        "s = SegmentationMetrics(); c, p = counts(r); s._counts.append(c); s._psum.append(p)\n"
        "gs = s.compute()\n"
        "rs = SegmentationMetrics(dist_sync=False)\n"
        "for k in range(w):\n"
        "    c, p = counts(k); rs._counts.append(c); rs._psum.append(p)\n"
        "assert gs == rs.compute() and len(s._counts) == 1\n"
        "print(f'RANK{r} ok {got[\"map\"]:.6f} {gs[\"iou\"]:.6f}', flush=True)\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    rows = sorted(l.split() for o in outs for l in o.splitlines() if l.startswith("RANK"))
    assert len(rows) == 2 and rows[0][1:] == rows[1][1:]          # both ranks: the same global numbers
