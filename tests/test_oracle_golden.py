"""Pin the oracle to the reference: every reference-owned block must reproduce, bit for bit, the
outputs the REAL reference classes gave for the same state_dict and inputs
(fixtures: tests/golden/ref_blocks.pt, made by tests/golden/make_ref_fixtures.py)."""
import os

import pytest
import torch

from oracle import blocks, postprocess


@pytest.fixture(scope="module")
def cases(golden_dir):
    return torch.load(os.path.join(golden_dir, "ref_blocks.pt"), weights_only=True)


BUILDERS = {
    "ConvBlock_3x3": lambda: blocks.ConvBlock(16, 32, 3, 1),
    "ConvBlock_1x1": lambda: blocks.ConvBlock(24, 16, 1),
    "ConvBlock_3x3_train": lambda: blocks.ConvBlock(16, 32, 3, 1),
    "Bottleneck": lambda: blocks.Bottleneck(16, 16, False, kernel=(3, 3), e=1.0),
    "Bottleneck_add": lambda: blocks.Bottleneck(16, 16, True),
    "C2f": lambda: blocks.C2f(24, 32),
    "DepthwiseConvBlock": lambda: blocks.DepthwiseConvBlock(32, 32),
    "BiFPNUnit": lambda: blocks.BiFPNUnit(32),
    "BiFPN": lambda: blocks.BiFPN([16, 24, 32], 32, 2),
}


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_block_bit_exact(cases, name):
    case = cases[name]
    m = BUILDERS[name]()
    missing, unexpected = m.load_state_dict(case["state_dict"], strict=True)
    assert not missing and not unexpected  # identical parameter names
    m.train(case["train"])
    with torch.no_grad():
        out = m(*case["inputs"])
    ref = case["output"]
    if isinstance(ref, (list, tuple)):
        assert len(out) == len(ref)
        for o, r in zip(out, ref):
            assert torch.equal(o, r)
    else:
        assert torch.equal(out, ref)
    if case["train"]:  # BN running statistics after one train-mode step (momentum .9997)
        for k, v in case["state_dict_after"].items():
            assert torch.equal(m.state_dict()[k], v), k


def test_autopad(cases):
    for args, want in zip(cases["autopad"]["inputs"], cases["autopad"]["output"]):
        assert blocks.same_pad(*args) == want


def test_box_utils_bit_exact(cases):
    for k in ("batch_bbox_iou", "batch_bbox_iou_empty"):
        assert torch.equal(postprocess.batch_bbox_iou(*cases[k]["inputs"]), cases[k]["output"])
    assert torch.equal(postprocess.dist2bbox(*cases["dist2bbox_xyxy"]["inputs"], "xyxy"), cases["dist2bbox_xyxy"]["output"])
    assert torch.equal(postprocess.dist2bbox(*cases["dist2bbox_xywh"]["inputs"], "xywh"), cases["dist2bbox_xywh"]["output"])
    with pytest.raises(NotImplementedError):
        postprocess.dist2bbox(*cases["dist2bbox_xyxy"]["inputs"], "cxcywh")


def test_constants(cases):
    c = cases["constants"]
    assert (postprocess.CONF_TH, postprocess.NMS_IOU, postprocess.TOP_K) == (c["CONF_TH"], c["NMS_IOU"], c["TOP_K"])


def test_weighted_add_of_oldest_variant_bit_exact(cases):
    """src/model.py:27-36: `sum(w_i + f)` -- the weights are added, not multiplied."""
    from oracle.model import WeightedAdd
    for n in (2, 3):
        c = cases[f"WeightedAdd_{n}"]
        m = WeightedAdd(n)
        with torch.no_grad():
            m.w.copy_(c["w"])
            assert torch.equal(m(c["inputs"]), c["output"])


@pytest.mark.parametrize("name", ["loss_train", "loss_train_nosmooth", "loss_eval"])
def test_multitask_loss_matches_reference_method(cases, name):
    """oracle.loss.multitask_loss against MultiTaskLitModel._multitask_loss itself (running_main_v3.py:232-387), incl. the
    reference's column-concatenated GT boxes (two boxes in image 0), an image without boxes, label smoothing on / off and
    the eval-mode tuple."""
    from oracle.loss import multitask_loss
    c = cases[name]
    out = multitask_loss(c["det"], c["protos"], c["logits"], c["gt_boxes"], c["gt_masks"], c["gt_cls"], c["proj_w"], c["proj_b"],
                         img_size=c["img_size"], nc_det=c["nc_det"], label_smoothing=c["smoothing"], training=c["training"])
    assert len(out) == len(c["output"]) and len(out) == (8 if c["training"] else 6)
    for a, b in zip(out, c["output"]):
        assert torch.allclose(a.float(), b.float(), rtol=1e-6, atol=1e-6), (name, float(a), float(b))
