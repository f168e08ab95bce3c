"""Device segmentation accumulators (mtbt_seg_confusion) vs the torch restatement of validation_step's own operations."""
import numpy as np
import pytest
import torch

from oracle import metrics as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(B, S, seed, empty=()):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, 1, S, S, generator=g) * 3
    logits[0, 0, 0, :4] = torch.tensor([0.0, 2.0 ** -24, 2.0 ** -23, -0.0])         # the sigmoid(x) > 0.5 boundary in fp32
    gt = (torch.rand(B, 1, S, S, generator=g) > 0.7).float()
    for i in empty:
        logits[i] = -5.0
        gt[i] = 0.0
    return logits, gt


@pytest.mark.parametrize("B,S", [(3, 64), (2, 640), (1, 36)])
def test_counts_match_reference_ops(B, S):
    from multitask_bonetumor_yolo_amd.metrics import SegmentationMetrics
    logits, gt = _batch(B, S, B * S)
    m = SegmentationMetrics()
    m.update(logits.to(DEV), gt.to(DEV))
    counts, score = m.per_image()
    rc, rs = O.seg_counts(logits, gt)
    assert np.array_equal(counts, rc)                                               # integer counts: exact
    assert np.allclose(score, rs, rtol=2e-5, atol=0)                                # fp32 sum of ~S*S/2 probabilities, different order


def test_metric_values_and_empty_images():
    from multitask_bonetumor_yolo_amd.metrics import SegmentationMetrics
    m = SegmentationMetrics()
    tot = np.zeros(4)
    for seed in (1, 2):
        logits, gt = _batch(4, 32, seed, empty=(1,))
        m.update(logits.to(DEV), gt.to(DEV))
        tot += O.seg_counts(logits, gt)[0].sum(0)
    r = m.compute()
    tp, fp, fn, tn = tot
    assert abs(r["f1"] - 2 * tp / (2 * tp + fp + fn)) < 1e-12 and abs(r["precision"] - tp / (tp + fp)) < 1e-12
    assert abs(r["recall"] - tp / (tp + fn)) < 1e-12 and abs(r["accuracy"] - (tp + tn) / tot.sum()) < 1e-12
    assert 0.0 < r["dice"] < 1.0 and 0.0 <= r["seg_map"] <= 1.0
    # perfect prediction -> every metric 1, including the mask mAP
    m.reset()
    gt = (torch.rand(2, 1, 32, 32) > 0.5).float()
    m.update(((gt * 2 - 1) * 4).to(DEV), gt.to(DEV))
    r = m.compute()
    assert r["f1"] == 1.0 and r["iou"] == 1.0 and r["dice"] == 1.0 and abs(r["seg_map"] - 1.0) < 1e-12


def test_rejects_cpu_tensors():
    from multitask_bonetumor_yolo_amd.metrics import SegmentationMetrics
    with pytest.raises(RuntimeError):
        SegmentationMetrics().update(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
