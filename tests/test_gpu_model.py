"""End-to-end parity of the drop-in module against the CPU oracle with the SAME state_dict and inputs.
fp32 mode (exact-fp32 MFMA): logits / raw maps / protos within 1e-3 (north_star).  bf16 mode: relative error
bound stated below (bf16 storage of ~100 layers of activations)."""
import pytest
import torch

from oracle import postprocess as opp
from oracle.model import ConvNeXtBiFPNYOLO as OracleModel
from oracle.model import ConvNeXtBiFPNYOLOv2 as OracleModelV2
from oracle.model import randomize_

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, ConvNeXtBiFPNYOLOv2, postprocess as pp

DEV = "cuda:0"


@pytest.fixture(scope="module")
def pair():
    torch.manual_seed(0)
    ora = randomize_(OracleModel(2, 2, pretrained_backbone=False)).eval()
    hip = ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)
    missing, unexpected = hip.load_state_dict(ora.state_dict(), strict=True)
    assert not missing and not unexpected
    return ora, hip.to(DEV).eval()


def maxdiff(a, b):
    return (a.float().cpu() - b).abs().max().item()


def relerr(a, b):
    return ((a.float().cpu() - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("B,S", [(2, 64), (1, 160), (1, 256)])   # 256: the /8 and /16 maps are 16-aligned -> direct 3x3 kernel (fp32)
def test_infer_fp32_parity(pair, B, S):
    ora, hip = pair
    hip.set_compute_dtype(torch.float32)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(S))
    with torch.no_grad():
        ref = ora(x, "infer")
        out = hip(x.to(DEV), "infer")
    torch.cuda.synchronize()
    assert set(out) == set(ref)
    for o, r in zip(out["detect_features"], ref["detect_features"]):
        assert o.shape == r.shape and maxdiff(o, r) < 1e-3
    seg_feats, mc, protos = out["segment_protos"]
    rseg_feats, rmc, rprotos = ref["segment_protos"]
    for o, r in zip(seg_feats, rseg_feats):
        assert o.shape == r.shape and maxdiff(o, r) < 1e-3
    assert mc.shape == rmc.shape and maxdiff(mc, rmc) < 1e-3
    assert protos.shape == rprotos.shape and maxdiff(protos, rprotos) < 1e-3
    assert maxdiff(out["img_cls_logits"], ref["img_cls_logits"]) < 1e-3
    assert maxdiff(out["img_cls_probs"], ref["img_cls_probs"]) < 1e-3
    # Detect eval output with the reference's never-set stride (zeros): boxes are exactly 0
    for key in ("detect_preds_cat", "segment_preds_cat"):
        assert out[key].shape == ref[key].shape
        assert maxdiff(out[key], ref[key]) < 1e-3
    assert torch.all(out["detect_preds_cat"][:, :4] == 0)


def test_infer_with_strides_set(pair):
    ora, hip = pair
    hip.set_compute_dtype(torch.float32)
    x = torch.rand(1, 3, 96, 96, generator=torch.Generator().manual_seed(5))
    st = torch.tensor([8.0, 16.0, 32.0])
    old = (ora.detect.stride, ora.segment.stride, hip.detect.stride, hip.segment.stride)
    try:
        ora.detect.stride = ora.segment.stride = st
        hip.detect.stride = hip.segment.stride = st
        with torch.no_grad():
            ref = ora(x, "infer")
            out = hip(x.to(DEV), "infer")
        torch.cuda.synchronize()
        for key in ("detect_preds_cat", "segment_preds_cat"):
            assert torch.allclose(out[key].cpu(), ref[key], rtol=1e-4, atol=2e-3)
    finally:
        ora.detect.stride, ora.segment.stride, hip.detect.stride, hip.segment.stride = old


def test_infer_bf16_close(pair):
    ora, hip = pair
    hip.set_compute_dtype(torch.bfloat16)
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = ora(x, "infer")
        out = hip(x.to(DEV), "infer")
    torch.cuda.synchronize()
    hip.set_compute_dtype(torch.float32)
    # bf16 activations/weights (8 mantissa bits) through ~100 layers: relative L2 error bound 5e-2
    for o, r in zip(out["detect_features"], ref["detect_features"]):
        assert relerr(o, r) < 5e-2
    _, mc, protos = out["segment_protos"]
    assert relerr(mc, ref["segment_protos"][1]) < 5e-2
    assert relerr(protos, ref["segment_protos"][2]) < 5e-2
    assert maxdiff(out["img_cls_probs"], ref["img_cls_probs"]) < 3e-2


def test_postprocess_pipeline_on_model_outputs(pair):
    """decode -> NMS -> masks on the HIP model's outputs equals the oracle post-process run on the same outputs
    (kept indices bit-exact), and equals the all-CPU pipeline up to boxes that sit on a decision boundary."""
    ora, hip = pair
    hip.set_compute_dtype(torch.float32)
    S = 160
    x = torch.rand(2, 3, S, S, generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        out = hip(x.to(DEV), "infer")
    seg_feats, mc, protos = out["segment_protos"]
    res = pp.detect_and_segment(out["detect_features"], mc, protos, S)
    torch.cuda.synchronize()
    maps_cpu = [m.float().cpu() for m in out["detect_features"]]
    boxes, scores, _ = opp.decode_levels(maps_cpu, S)
    d = pp.decode_boxes(out["detect_features"], S)
    for b in range(2):
        # oracle NMS on the GPU-decoded boxes: identical kept set
        bs, bl = d["scores"][b].cpu().max(dim=1)
        k, anchors, kb, ks, kl = opp.filter_and_nms(d["boxes"][b].cpu(), d["scores"][b].cpu(), S)
        n = int(res["counts"][b])
        assert n == len(k)
        assert torch.equal(res["keep_idx"][b, :n].cpu(), k)
        assert torch.equal(res["boxes"][b, :n].cpu(), kb)
        # masks of the kept boxes
        coeffs = mc[b].cpu()[:, anchors].t()
        ref_logits, ref_masks = opp.assemble_masks(coeffs, protos[b].cpu(), (S, S))
        diff = res["masks"][b, :n].cpu() != ref_masks
        assert torch.all(ref_logits[diff].abs() < 1e-4)
        # decode parity against the CPU decode of the same maps
        assert torch.allclose(d["boxes"][b].cpu(), boxes[b], rtol=1e-5, atol=1e-3)


def test_validation_metrics_end_to_end(pair):
    """SURVEY §8c acceptance metric: box mAP@0.5 / @0.5:0.95 and the projector-mask IoU / F1 of the whole HIP path (fp32 forward
    -> decode -> NMS -> device metric accumulators) within 1e-3 of the all-CPU oracle path on a fixed synthetic det / GT set."""
    from multitask_bonetumor_yolo_amd import MeanAveragePrecision, SegmentationMetrics
    from oracle import metrics as om
    ora, hip = pair
    hip.set_compute_dtype(torch.float32)
    B, S = 3, 160
    g = torch.Generator().manual_seed(33)
    x = torch.rand(B, 3, S, S, generator=g)
    with torch.no_grad():
        out = hip(x.to(DEV), "infer")
        ref = ora(x, "infer")
    _, mc, protos = out["segment_protos"]
    res = pp.detect_and_segment(out["detect_features"], mc, protos, S, masks=False)
    rboxes, rscores, _ = opp.decode_levels(ref["detect_features"], S)
    preds_hip, preds_ora, targets = [], [], []
    for b in range(B):
        n = int(res["counts"][b])
        preds_hip.append(dict(boxes=res["boxes"][b, :n], scores=res["scores"][b, :n], labels=res["labels"][b, :n]))
        _, _, kb, ks, kl = opp.filter_and_nms(rboxes[b], rscores[b], S)
        preds_ora.append(dict(boxes=kb, scores=ks, labels=kl))
        pick = torch.arange(0, min(len(kb), 12), 3)                     # GT = every third of the oracle's best boxes, jittered
        targets.append(dict(boxes=kb[pick] + torch.randn(len(pick), 4, generator=g) * 1.5, labels=kl[pick]))
    assert sum(len(t["labels"]) for t in targets) > 0
    for thr in ([0.5], torch.linspace(0.5, 0.95, 10).tolist()):          # running_main_v3.py:206-214
        a, b_ = MeanAveragePrecision(thr, [1, 10, 100]), MeanAveragePrecision(thr, [1, 10, 100])
        a.update(preds_hip, targets)
        b_.update(preds_ora, targets)
        ra, rb = a.compute(), b_.compute()
        assert 0.0 < rb["map"] <= 1.0
        for k in rb:
            assert abs(ra[k] - rb[k]) <= 1e-3, (k, ra[k], rb[k])
    # segmentation: projector logits (:251-255) -> pixel counts
    w, bias = torch.randn(1, protos.shape[1], 1, 1, generator=g) * 0.5, torch.randn(1, generator=g) * 0.1
    gt = (torch.rand(B, 1, S, S, generator=g) > 0.6).float()
    sm = SegmentationMetrics()
    sm.update(pp.proto_projector_logits(protos, w.to(DEV), bias.to(DEV), S), gt.to(DEV))
    got = sm.compute()
    rc, _ = om.seg_counts(opp.proto_projector_logits(ref["segment_protos"][2], w, bias, S), gt)
    tp, fp, fn, tn = (float(v) for v in rc.sum(0))
    assert abs(got["iou"] - tp / (tp + fp + fn)) <= 1e-3 and abs(got["f1"] - 2 * tp / (2 * tp + fp + fn)) <= 1e-3


def test_train_mode_forward_batch_stat_heads():
    """forward(x, "train") under model.eval(): the reference flips the heads to train mode (main_model.py:358-359), so
    their BatchNorms use batch statistics and update the running statistics.  Fresh pair: state mutates."""
    torch.manual_seed(3)
    ora = randomize_(OracleModel(2, 2, pretrained_backbone=False)).eval()
    hip = ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)
    hip.load_state_dict(ora.state_dict(), strict=True)
    hip = hip.to(DEV).eval().set_compute_dtype(torch.float32)
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        rdet, (rseg, rmc, rprotos), rlog = ora(x, "train")
        det, (seg, mc, protos), log = hip(x.to(DEV), "train")
    torch.cuda.synchronize()
    assert len(det) == 3 and len(seg) == 3
    for o, r in zip(list(det) + list(seg), list(rdet) + list(rseg)):
        assert o.shape == r.shape and maxdiff(o, r) < 1e-3
    assert mc.shape == rmc.shape and maxdiff(mc, rmc) < 1e-3
    assert protos.shape == rprotos.shape and maxdiff(protos, rprotos) < 1e-3
    assert maxdiff(log, rlog) < 1e-3
    # flag handling (F14): top-level flags restored, children left in train mode; running statistics advanced identically
    assert hip.detect.training is False and hip.segment.training is False
    assert hip.detect.cv2[0][0].bn.training and ora.detect.cv2[0][0].bn.training
    hsd, osd = hip.state_dict(), ora.state_dict()
    for k in osd:
        if "running_" in k or "num_batches_tracked" in k:
            assert maxdiff(hsd[k], osd[k].float()) < 1e-4, k
    assert int(hsd["segment.proto.cv2.bn.num_batches_tracked"]) == 1
    # a following infer call must use the UPDATED running statistics (plans that folded the old ones are invalidated)
    with torch.no_grad():
        ri = ora(x, "infer")
        hi = hip(x.to(DEV), "infer")
    torch.cuda.synchronize()
    for o, r in zip(hi["detect_features"], ri["detect_features"]):
        assert maxdiff(o, r) < 1e-3
    assert maxdiff(hi["segment_protos"][2], ri["segment_protos"][2]) < 1e-3


def test_train_mode_forward_bf16_and_v2():
    torch.manual_seed(5)
    ora = randomize_(OracleModelV2(2, 2, pretrained_backbone=False)).eval()
    hip = ConvNeXtBiFPNYOLOv2(2, 2, pretrained_backbone=False)
    hip.load_state_dict(ora.state_dict(), strict=True)
    hip = hip.to(DEV).eval().set_compute_dtype(torch.bfloat16)
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        (rseg, rmc, rprotos), rlog = ora(x, "train")
        (seg, mc, protos), log = hip(x.to(DEV), "train")
    torch.cuda.synchronize()
    for o, r in zip(seg, rseg):
        assert o.shape == r.shape and relerr(o, r) < 6e-2
    assert relerr(mc, rmc) < 6e-2 and relerr(protos, rprotos) < 6e-2


def test_v0_variant_src_model_py():
    """BASELINE config 0: the `src/model.py` variant (nearest / max-pool BiFPN, DWConv nodes, weight-adding WeightedAdd)."""
    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLOv0
    from oracle.model import ConvNeXtBiFPNYOLOv0 as OracleModelV0
    torch.manual_seed(7)
    ora = randomize_(OracleModelV0(3, 3)).eval()
    with torch.no_grad():
        for u in ora.neck.units:  # non-trivial fusion weights
            for a in (u.add_p4_td, u.add_p3_td, u.add_p4_out, u.add_p5_out):
                a.w.copy_(torch.rand_like(a.w) + 0.2)
    hip = ConvNeXtBiFPNYOLOv0(3, 3)
    hip.load_state_dict(ora.state_dict(), strict=True)
    hip = hip.to(DEV).eval().set_compute_dtype(torch.float32)
    x = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        ref = ora(x)
        out = hip(x.to(DEV))
    torch.cuda.synchronize()
    assert set(out) == {"detect", "segment", "img_cls"}
    assert maxdiff(out["detect"][0], ref["detect"][0]) < 1e-3
    for o, r in zip(out["detect"][1], ref["detect"][1]):
        assert o.shape == r.shape and maxdiff(o, r) < 1e-3
    assert maxdiff(out["segment"][0], ref["segment"][0]) < 1e-3
    feats, mc, protos = out["segment"][1]
    assert maxdiff(mc, ref["segment"][1][1]) < 1e-3 and maxdiff(protos, ref["segment"][1][2]) < 1e-3
    assert maxdiff(out["img_cls"], ref["img_cls"]) < 1e-3
    raw = hip(x.to(DEV), mode="raw")
    assert isinstance(raw, tuple) and len(raw) == 3


def test_v2_variant_layout():
    torch.manual_seed(1)
    ora = randomize_(OracleModelV2(2, 3, pretrained_backbone=False)).eval()
    hip = ConvNeXtBiFPNYOLOv2(2, 3, pretrained_backbone=False)
    hip.load_state_dict(ora.state_dict(), strict=True)
    hip = hip.to(DEV).eval()
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = ora(x, "infer")
        out = hip(x.to(DEV), "infer")
    torch.cuda.synchronize()
    assert set(out) == set(ref)
    assert out["detect_preds_cat"].shape == ref["detect_preds_cat"].shape
    assert maxdiff(out["segment_preds_cat"], ref["segment_preds_cat"]) < 1e-3
    assert maxdiff(out["img_cls_logits"], ref["img_cls_logits"]) < 1e-3


def test_errors(pair):
    _, hip = pair
    with pytest.raises(ValueError):
        hip(torch.rand(1, 3, 64, 64, device=DEV), "eval")
    with pytest.raises(ValueError):
        hip(torch.rand(1, 3, 65, 64, device=DEV), "infer")
    with pytest.raises(RuntimeError):
        hip(torch.rand(1, 3, 64, 64), "infer")  # CPU tensor: no CPU path


def test_lane_schedule_matches_sequential(monkeypatch):
    """engine.Plan.run over 4 HIP streams (dependency-driven lanes) gives bit-identical results to one stream."""
    torch.manual_seed(3)
    from multitask_bonetumor_yolo_amd import init_synthetic_
    hip = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    x = torch.rand(2, 3, 128, 128, device=DEV)
    outs = []
    for lanes in ("1", "4", "6"):
        monkeypatch.setenv("MTBT_LANES", lanes)
        hip.__dict__.pop("_plans", None)
        for _ in range(3):  # repeated runs: lanes of run k+1 must not overtake readers of run k
            fwd, det = hip.infer_and_detect(x, 128)
        torch.cuda.synchronize()
        outs.append((fwd["segment_preds_cat"].clone(), fwd["detect_preds_cat"].clone(), fwd["img_cls_logits"].clone(),
                     fwd["segment_protos"][2].clone(), det["keep_idx"].clone(), det["masks"].clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_high_resolution_1280_pipeline():
    """BASELINE config 4 shape (1280x1280, 33600 anchors per image; the large-A NMS path): the fused step equals the two
    plain calls bit for bit, and the oracle's filter + NMS on the GPU-decoded boxes keeps the same indices."""
    from multitask_bonetumor_yolo_amd import init_synthetic_
    torch.manual_seed(11)
    hip = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    S = 1280
    x = torch.rand(1, 3, S, S, device=DEV)
    fwd, det = hip.infer_and_detect(x, S)
    with torch.no_grad():
        out = hip(x, "infer")
    feats, mc, protos = out["segment_protos"]
    assert out["detect_preds_cat"].shape == (1, 4 + 2, 33600) and protos.shape == (1, 32, S // 4, S // 4)
    res = pp.detect_and_segment(out["detect_features"], mc, protos, S)
    torch.cuda.synchronize()
    assert torch.equal(det["keep_idx"], res["keep_idx"]) and torch.equal(det["masks"], res["masks"])
    d = pp.decode_boxes(out["detect_features"], S)
    k, anchors, kb, ks, kl = opp.filter_and_nms(d["boxes"][0].cpu(), d["scores"][0].cpu(), S)
    n = int(res["counts"][0])
    assert n == len(k) and n > 0
    assert torch.equal(res["keep_idx"][0, :n].cpu(), k) and torch.equal(res["boxes"][0, :n].cpu(), kb)


def test_concurrent_lanes_are_deterministic_at_full_size(monkeypatch):
    """Batch 16 x 640^2 on four lanes, eager and graph replay, 40 steps each: bit-identical to the single-stream result.
    (Guards the LDS-DMA pipelines' barrier discipline: a missing lgkmcnt(0) before the restaging barrier produced rare wrong
    tiles only when a CU was shared with another kernel.)"""
    from multitask_bonetumor_yolo_amd import init_synthetic_
    from multitask_bonetumor_yolo_amd.graphed import GraphedInference
    torch.manual_seed(5)
    hip = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    x = torch.rand(16, 3, 640, 640, device=DEV)

    def snap(fwd, det):
        return [t.clone() for t in (fwd["segment_preds_cat"], fwd["detect_preds_cat"], fwd["segment_protos"][2],
                                    fwd["img_cls_logits"], det["keep_idx"], det["masks"])]
    monkeypatch.setenv("MTBT_LANES", "1")
    ref = snap(*hip.infer_and_detect(x, 640))
    torch.cuda.synchronize()
    monkeypatch.setenv("MTBT_LANES", "4")
    hip.__dict__.pop("_plans", None)
    for _ in range(40):
        got = snap(*hip.infer_and_detect(x, 640))
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, got))
    g = GraphedInference(hip, x, 640)
    for _ in range(40):
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, snap(g.fwd, g.out)))


def test_autotuned_schedule_keeps_the_results():
    """GraphedInference(autotune=True) times the launch-schedule options on the live machine and keeps the fastest in model.plan_options.  Whatever it
    picks, the step computes the same thing: with the lowering option NODE_FUSED pinned off (the only one that changes arithmetic, by one bf16 ulp)
    every output of the tuned graph is BIT-IDENTICAL to the default schedule's; and the candidates are freed as it goes (plans cache empty but for
    the winner)."""
    from multitask_bonetumor_yolo_amd import init_synthetic_, graphed
    from multitask_bonetumor_yolo_amd.graphed import GraphedInference
    torch.manual_seed(7)
    hip = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    x = torch.rand(4, 3, 256, 256, device=DEV)

    def snap(g):
        g.replay()
        torch.cuda.synchronize()
        return [t.clone() for t in (g.fwd["segment_preds_cat"], g.fwd["detect_preds_cat"], g.fwd["segment_protos"][2], g.fwd["img_cls_logits"],
                                    g.out["keep_idx"], g.out["masks"])]
    ref = snap(GraphedInference(hip, x, 256))
    lines = []
    knobs = tuple(k for k in graphed.AUTOTUNE_KNOBS if k[0] != "NODE_FUSED")
    old = graphed.AUTOTUNE_KNOBS
    graphed.AUTOTUNE_KNOBS = knobs
    try:
        g = GraphedInference(hip, x, 256, autotune=True, log=lines.append)
    finally:
        graphed.AUTOTUNE_KNOBS = old
    assert len(lines) == 2 + sum(len(v) for _, v in knobs) and lines[-1].startswith("autotune: kept")
    assert isinstance(hip.plan_options, dict) and set(hip.plan_options) <= {k for k, _ in knobs}
    assert len(hip.__dict__.get("_plans", {})) == 1
    got = snap(g)
    assert all(torch.equal(a, b) for a, b in zip(ref, got))


def test_reserved_streams_are_distinct_hip_streams():
    """torch.cuda.Stream() cycles through a pool of 32: the 33rd object IS the first.  The plan's lane streams, the graph's capture / side streams
    and the gradient-exchange stream come from engine.reserved_stream, which keeps them pairwise distinct however many streams the process created
    before -- an aliased capture stream segfaulted hipStreamEndCapture (a whole test session was enough to get there)."""
    from multitask_bonetumor_yolo_amd.engine import reserved_stream
    junk = [torch.cuda.Stream(device=DEV) for _ in range(40)]            # wrap torch's pool
    roles = ["lane1", "lane2", "lane3", "graph_capture", "graph_side", "eager_side", "grad_exchange", "test_extra"]
    got = [reserved_stream(DEV, r) for r in roles]
    assert len({s.cuda_stream for s in got}) == len(roles) and all(s.cuda_stream != 0 for s in got)
    assert all(reserved_stream(DEV, r) is s for r, s in zip(roles, got))   # one per role, process-wide
    del junk


def test_graph_outlives_plan_eviction_and_refuses_stale_weights():
    """a captured graph points into its launch plan's buffers and folded weights.  The GraphedInference object keeps that plan
    alive when the model's plan cache drops it, refuses to replay after the weights changed, and a graph can be destroyed and a new one
    captured in the same process."""
    from multitask_bonetumor_yolo_amd import init_synthetic_
    from multitask_bonetumor_yolo_amd.graphed import GraphedInference
    torch.manual_seed(6)
    hip = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    x = torch.rand(2, 3, 128, 128, device=DEV)
    g = GraphedInference(hip, x, 128)
    ref = [t.clone() for t in (g.replay()["keep_idx"], g.fwd["segment_protos"][2])]
    torch.cuda.synchronize()
    hip.__dict__.pop("_plans")                       # what a signature change does to the cache entry
    junk = [torch.randn(1 << 20, device=DEV) for _ in range(64)]      # would land in the freed pool if the plan had been released
    out = g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["keep_idx"], ref[0]) and torch.equal(g.fwd["segment_protos"][2], ref[1])
    del junk
    with torch.no_grad():
        hip.cls_fc.weight.mul_(1.5)                   # an in-place update torch's version counter sees
    with pytest.raises(RuntimeError, match="changed since the capture"):
        g.replay()
    del g                                            # destroy the captured (multi-stream) graph ...
    torch.cuda.synchronize()
    g2 = GraphedInference(hip, x, 128)               # ... and capture again in the same process
    out2 = g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out2["keep_idx"], ref[0])      # (cls_fc does not feed the boxes)
    hip.mark_weights_updated()                        # what raw-pointer updates (fused optimiser) must call
    with pytest.raises(RuntimeError, match="changed since the capture"):
        g2.replay()


# ---------------------------------------------------------------------------------------------------------------------------
# Parity where the benchmark actually runs: 640 x 640 (configs[1]) and 1280 x 1280 (configs[4]).  At these sizes the 80^2 / 160^2 maps
# take kernel instances the small cases above never reach end to end (direct 3x3 with multi-tile halos, fused MLP over 409 600
# pixels, the 4-lane schedule).  The oracle forward costs ~0.4 s (640) / ~1.6 s (1280) per image on the box's host cores.
# ---------------------------------------------------------------------------------------------------------------------------
def _out_list(out):
    seg_feats, mc, protos = out["segment_protos"]
    return {"det0": out["detect_features"][0], "det1": out["detect_features"][1], "det2": out["detect_features"][2],
            "seg0": seg_feats[0], "seg2": seg_feats[2], "mc": mc, "protos": protos, "logits": out["img_cls_logits"]}


def _kept_agreement(a_boxes, a_cnt, b_boxes, b_cnt, tol):
    """fraction of a's kept boxes that have a box of b within `tol` pixels (all four coordinates)"""
    hit = tot = 0
    for n in range(a_boxes.shape[0]):
        A, Bx = a_boxes[n, :int(a_cnt[n])].cpu(), b_boxes[n, :int(b_cnt[n])].cpu()
        tot += len(A)
        if len(A) and len(Bx):
            hit += int(((A[:, None, :] - Bx[None, :, :]).abs().amax(dim=2).amin(dim=1) <= tol).sum())
    return hit / max(tot, 1)


def test_infer_fp32_parity_at_640(pair):
    ora, hip = pair
    hip.set_compute_dtype(torch.float32)
    x = torch.rand(1, 3, 640, 640, generator=torch.Generator().manual_seed(640))
    with torch.no_grad():
        ref, out = _out_list(ora(x, "infer")), _out_list(hip(x.to(DEV), "infer"))
    torch.cuda.synchronize()
    for k in ref:
        assert out[k].shape == ref[k].shape and maxdiff(out[k], ref[k]) < 1e-3, (k, maxdiff(out[k], ref[k]))


def test_infer_bf16_at_640_per_output_bounds(pair):
    """bf16 (the benchmarked mode) at the benchmarked size, per output: absolute error against the oracle bounded relative to that output's
    own scale; and what the post-process consumes agrees with the fp32 HIP path anchor by anchor: decoded boxes within 2 px for >= 99 % of
    the 2 x 8400 anchors, best class scores within 0.03 everywhere.  (With random-init weights ~all anchors score ~0.5, so the SET of 100
    boxes NMS keeps is decided by differences far below any arithmetic's resolution -- it is compared bit for bit on identical inputs in
    test_postprocess_pipeline_on_model_outputs, not across precisions.)"""
    ora, hip = pair
    S = 640
    x = torch.rand(2, 3, S, S, generator=torch.Generator().manual_seed(641))
    with torch.no_grad():
        ref = _out_list(ora(x, "infer"))
        hip.set_compute_dtype(torch.float32)
        d32 = pp.decode_boxes(hip(x.to(DEV), "infer")["detect_features"], S)
        hip.set_compute_dtype(torch.bfloat16)
        b16 = hip(x.to(DEV), "infer")
        d16 = pp.decode_boxes(b16["detect_features"], S)
    torch.cuda.synchronize()
    hip.set_compute_dtype(torch.float32)
    out = _out_list(b16)
    # bf16 keeps 8 mantissa bits; through ~100 layers the error stays a few percent of each output's dynamic range
    for k, r in ref.items():
        scale = r.abs().max().item()
        assert maxdiff(out[k], r) <= 0.06 * scale + 1e-3, (k, maxdiff(out[k], r), scale)
        assert relerr(out[k], r) < 3e-2, (k, relerr(out[k], r))
    box_err = (d16["boxes"] - d32["boxes"]).abs().amax(dim=2)
    assert (box_err <= 2.0).float().mean().item() >= 0.99, (box_err <= 2.0).float().mean().item()
    assert (d16["best_score"] - d32["best_score"]).abs().max().item() <= 0.03


def test_bf16_post_process_agrees_with_fp32_on_calibrated_heads():
    """The cross-precision check of what the benchmark's post-process produces (running_main_v3.py:535-552 on the model's own outputs), on
    the benchmark's model: synthetic weights whose BatchNorm statistics and heads are calibrated on the batch (`calibrate_synthetic_heads_`:
    ~10^3 of the 8400 anchors per image pass conf 0.05, scores spread over (0.05, 0.99)).
    What holds, and is asserted: the confidence filter is real (300 .. 3000 candidates per image, not "all 8400"), the kept scores are
    spread, >= 75 % of the boxes NMS keeps in bf16 have an fp32-kept box of IoU >= 0.7 (and vice versa; measured 0.84) and the COCO
    mAP@0.5 of the bf16 detections against the fp32 detections taken as ground truth is >= 0.70 (measured 0.79).
    What does NOT hold with random weights, and why (measured with tools/head_stats.py, several calibrations and inputs): agreement at
    IoU >= 0.9 is 0.70 - 0.78 and that mAP 0.76 - 0.80, not the 0.90 / 0.95 a trained detector would give.  bf16 storage through ~100
    layers leaves ~3e-2 relative L2 error on the head maps (test_infer_bf16_at_640_per_output_bounds); a random network's class logits
    are a CONTINUOUS field with ~10^3 candidates a few hundredths of a logit apart, so that error reorders neighbours, greedy NMS then
    keeps a different representative of a cluster (one anchor over = IoU 0.83 .. 0.91 at these box sizes) and boxes around rank 100
    swap in and out of the top-k.  A trained head separates objects from background by margins far above that noise.  The decisions
    themselves are compared bit for bit on IDENTICAL inputs (test_postprocess_pipeline_on_model_outputs, test_nms_*)."""
    from multitask_bonetumor_yolo_amd import MeanAveragePrecision, calibrate_synthetic_heads_, init_synthetic_
    from multitask_bonetumor_yolo_amd.metrics import box_iou_xyxy
    torch.manual_seed(77)
    hip = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval()
    S, B = 640, 4
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(78)).to(DEV)
    hip.set_compute_dtype(torch.float32)
    calibrate_synthetic_heads_(hip, x)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        hip.set_compute_dtype(dt)
        _, det = hip.infer_and_detect(x, S, masks=False)
        torch.cuda.synchronize()
        res[dt] = {k: det[k].cpu() for k in ("boxes", "scores", "labels", "counts", "n_cand")}
    hip.set_compute_dtype(torch.float32)
    r32, r16 = res[torch.float32], res[torch.bfloat16]
    nc = r32["n_cand"].float()
    assert 300 <= nc.min().item() and nc.max().item() <= 3000, nc           # a real confidence filter: neither "all 8400" nor "none"
    sc = torch.cat([r32["scores"][b, :int(r32["counts"][b])] for b in range(B)])
    assert sc.max().item() > 0.8 and sc.min().item() < 0.4                  # spread scores, not a band around 0.5
    hit = tot = 0
    preds, targets = [], []
    for b in range(B):
        n32, n16 = int(r32["counts"][b]), int(r16["counts"][b])
        assert n32 > 0 and n16 > 0
        iou = box_iou_xyxy(r16["boxes"][b, :n16].numpy(), r32["boxes"][b, :n32].numpy())
        hit += int((iou.max(axis=1) >= 0.7).sum()) + int((iou.max(axis=0) >= 0.7).sum())
        tot += n16 + n32
        preds.append(dict(boxes=r16["boxes"][b, :n16], scores=r16["scores"][b, :n16], labels=r16["labels"][b, :n16]))
        targets.append(dict(boxes=r32["boxes"][b, :n32], labels=r32["labels"][b, :n32]))
    assert hit / tot >= 0.75, hit / tot
    m = MeanAveragePrecision([0.5], [1, 10, 100], dist_sync=False)
    m.update(preds, targets)
    assert m.compute()["map_50"] >= 0.70, m.compute()


def test_infer_fp16_at_1280_vs_oracle(pair):
    """BASELINE configs[4]'s arithmetic and shape: fp16 storage, v_mfma_f32_16x16x32_f16, 1280 x 1280.  fp16 keeps 11 mantissa bits (8x
    bf16's resolution) inside +-65504: outputs within 1 % of each output's range / 5e-3 relative L2 of the fp32 oracle; the NMS kept
    indices equal the oracle NMS run on the SAME (GPU-decoded) boxes bit for bit."""
    ora, hip = pair
    S = 1280
    x = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(1280))
    with torch.no_grad():
        ref = _out_list(ora(x, "infer"))
        hip.set_compute_dtype(torch.float16)
        raw = hip(x.to(DEV), "infer")
    out = _out_list(raw)
    torch.cuda.synchronize()
    hip.set_compute_dtype(torch.float32)
    for k, r in ref.items():
        assert torch.isfinite(out[k]).all(), k
        scale = r.abs().max().item()
        assert maxdiff(out[k], r) <= 1e-2 * scale + 1e-3, (k, maxdiff(out[k], r), scale)
        assert relerr(out[k], r) < 5e-3, (k, relerr(out[k], r))
    res = pp.detect_and_segment(raw["detect_features"], raw["segment_protos"][1], raw["segment_protos"][2], S)
    d = pp.decode_boxes(raw["detect_features"], S)
    torch.cuda.synchronize()
    k, *_ = opp.filter_and_nms(d["boxes"][0].cpu(), d["scores"][0].cpu(), S)
    n = int(res["counts"][0])
    assert n == len(k) and torch.equal(res["keep_idx"][0, :n].cpu(), k)


def test_infer_fp16_batch64_1280_properties():
    """configs[4] at full size (batch 64, 1280 x 1280, fp16) through size-independent properties: every output finite, every image's result
    identical to the same image run in a batch of 4 (images are independent: no cross-image arithmetic, tile order does not change a
    pixel's reduction order), kept boxes sorted by score, inside the image, at most top-k; the instance masks (64 x 100 x 1280^2 booleans,
    10.5 GB) of the full batch equal those of the 4-image runs bit for bit and are empty beyond each image's kept count."""
    from multitask_bonetumor_yolo_amd import calibrate_synthetic_heads_, init_synthetic_
    torch.manual_seed(64)
    hip = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(DEV).eval().set_compute_dtype(torch.float16)
    S, B = 1280, 64
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(65)).to(DEV)
    calibrate_synthetic_heads_(hip, x[:2].contiguous())
    fwd, det = hip.infer_and_detect(x, S, masks=True)
    big = {k: v.clone() for k, v in _out_list(fwd).items()}
    kept = {k: det[k].clone() for k in ("keep_idx", "counts", "boxes", "scores", "n_cand")}
    masks = det["masks"]
    torch.cuda.synchronize()
    assert masks.shape == (B, 100, S, S) and masks.dtype == torch.bool
    for k, v in big.items():
        assert torch.isfinite(v).all(), k
    for lo in (0, 60):
        fs, ds = hip.infer_and_detect(x[lo:lo + 4].contiguous(), S, masks=True)
        small = _out_list(fs)
        torch.cuda.synchronize()
        for k, v in small.items():
            assert torch.equal(v, big[k][lo:lo + 4]), (k, lo)
        assert torch.equal(ds["keep_idx"], kept["keep_idx"][lo:lo + 4]) and torch.equal(ds["counts"], kept["counts"][lo:lo + 4])
        assert torch.equal(ds["masks"], masks[lo:lo + 4]), ("masks", lo)
    cnt = kept["counts"].cpu()
    assert int(cnt.max()) <= 100 and int(cnt.sum()) > 0
    ncand = kept["n_cand"].cpu()
    assert int(ncand.min()) > 0 and int(ncand.max()) < 33600          # calibrated heads: a real confidence filter, not "every anchor"
    for n in range(B):
        c = int(cnt[n])
        sc, bx = kept["scores"][n, :c].cpu(), kept["boxes"][n, :c].cpu()
        assert torch.all(sc[:-1] >= sc[1:]) and bx.min().item() >= 0 and bx.max().item() <= S
        if c < 100:
            assert not masks[n, c:].any()
    assert masks[:, 0].any()
