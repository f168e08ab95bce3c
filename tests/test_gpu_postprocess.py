"""GPU parity of the post-process (decode, NMS, mask assembly, proto projector) against the oracle.
NMS kept indices are compared BIT-EXACT on identical inputs."""
import pytest
import torch

from oracle import postprocess as opp

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from multitask_bonetumor_yolo_amd import postprocess as pp

DEV = "cuda:0"


def synth_maps(B, sizes, nc, seed, spread=3.0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, 64 + nc, h, w, generator=g) * spread for h, w in sizes]


def test_decode_matches_trainer_decode():
    maps = synth_maps(2, [(8, 8), (4, 4), (2, 2)], 2, 0)
    boxes, scores, _ = opp.decode_levels(maps, 64)
    d = pp.decode_boxes([m.to(DEV) for m in maps], 64)
    torch.cuda.synchronize()
    assert torch.allclose(d["boxes"].cpu(), boxes, rtol=1e-5, atol=1e-3)
    assert torch.allclose(d["scores"].cpu(), scores, rtol=1e-5, atol=1e-6)
    bs, bl = scores.max(dim=2)
    assert torch.allclose(d["best_score"].cpu(), bs, rtol=1e-5, atol=1e-6)
    agree = (d["best_label"].cpu().long() == bl)
    near_tie = (scores[..., 0] - scores[..., 1]).abs() < 1e-6
    assert torch.all(agree | near_tie)


def test_decode_channels_last_and_many_classes():
    maps = [m.contiguous(memory_format=torch.channels_last) for m in synth_maps(1, [(6, 5), (3, 3)], 7, 1)]
    boxes, scores, _ = opp.decode_levels(maps, 48)
    d = pp.decode_boxes([m.to(DEV) for m in maps], 48)
    torch.cuda.synchronize()
    assert torch.allclose(d["boxes"].cpu(), boxes, rtol=1e-5, atol=1e-3)
    assert torch.allclose(d["scores"].cpu(), scores, rtol=1e-5, atol=1e-6)


def clustered_boxes(A, seed, img=640.0, n_clusters=5):
    """SURVEY 8(d) NMS stress set: boxes jittered around a few centres, ~all above the conf threshold."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(n_clusters, 2, generator=g) * img * 0.6 + img * 0.2
    which = torch.randint(0, n_clusters, (A,), generator=g)
    c = centres[which] + torch.randn(A, 2, generator=g) * 12
    wh = torch.rand(A, 2, generator=g) * 60 + 30
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = torch.rand(A, generator=g)
    labels = torch.randint(0, 2, (A,), generator=g, dtype=torch.int32)
    return boxes, scores, labels


def oracle_nms_image(boxes, scores, labels, img, conf, iou, top_k):
    keep_conf = scores > conf
    b = boxes[keep_conf].clamp(0, img)
    s = scores[keep_conf]
    k = opp.nms(b, s, iou)[:top_k]
    return k, torch.nonzero(keep_conf).flatten()[k], b[k], s[k], labels[keep_conf][k]


@pytest.mark.parametrize("A,top_k", [(1, 10), (63, 100), (64, 100), (65, 5), (1000, 100), (8400, 100), (8400, 300), (3000, 3000)])
def test_nms_bit_exact(A, top_k):
    B = 3
    data = [clustered_boxes(A, 100 + i) for i in range(B)]
    boxes = torch.stack([d[0] for d in data])
    scores = torch.stack([d[1] for d in data])
    labels = torch.stack([d[2] for d in data])
    out = pp.nms_batched(boxes.to(DEV), scores.to(DEV), labels.to(DEV), 640.0, 0.05, 0.6, top_k)
    torch.cuda.synchronize()
    for i in range(B):
        k, anchors, kb, ks, kl = oracle_nms_image(boxes[i], scores[i], labels[i], 640.0, 0.05, 0.6, top_k)
        n = int(out["counts"][i])
        assert n == len(k)
        assert int(out["n_cand"][i]) == int((scores[i] > 0.05).sum())
        assert torch.equal(out["keep_idx"][i, :n].cpu(), k)                       # bit-exact kept indices
        assert torch.equal(out["keep_anchor"][i, :n].cpu().long(), anchors)
        assert torch.equal(out["boxes"][i, :n].cpu(), kb)
        assert torch.equal(out["scores"][i, :n].cpu(), ks)
        assert torch.equal(out["labels"][i, :n].cpu(), kl.long())
        assert torch.all(out["keep_idx"][i, n:] == -1)


def test_nms_edge_cases():
    # score ties (stable order), IoU exactly at the threshold, zero-area boxes, nothing above conf, duplicates
    boxes = torch.tensor([
        [0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 6],      # duplicate; IoU(0,2) = 0.6 exactly -> NOT suppressed (strict >)
        [5, 5, 5, 5], [5, 5, 5, 5],                          # zero area: 0/0 = nan -> not suppressed
        [100, 100, 120, 120], [101, 101, 121, 121], [300, 300, 310, 310],
    ], dtype=torch.float32)
    scores = torch.tensor([0.9, 0.9, 0.8, 0.7, 0.7, 0.6, 0.6, 0.01])
    labels = torch.zeros(8, dtype=torch.int32)
    out = pp.nms_batched(boxes[None].to(DEV), scores[None].to(DEV), labels[None].to(DEV), 640.0, 0.05, 0.6, 100)
    torch.cuda.synchronize()
    k, *_ = oracle_nms_image(boxes, scores, labels, 640.0, 0.05, 0.6, 100)
    n = int(out["counts"][0])
    assert torch.equal(out["keep_idx"][0, :n].cpu(), k)
    assert k.tolist() == [0, 2, 3, 4, 5]
    # nothing passes the confidence filter
    out = pp.nms_batched(boxes[None].to(DEV), (scores * 0)[None].to(DEV), None, 640.0, 0.05, 0.6, 10)
    torch.cuda.synchronize()
    assert int(out["counts"][0]) == 0 and int(out["n_cand"][0]) == 0


@pytest.mark.parametrize("kind", ["scattered", "narrow", "ties", "few_values"])
def test_nms_preselection_bit_exact(kind):
    """The kernel sorts only the candidates of the top score-histogram bins first and falls back to the full sort when
    they run out: scattered boxes end inside the preselection, tied / quantised scores straddle and overflow its bins."""
    A, top_k = 8400, 100
    g = torch.Generator().manual_seed(11)
    xy = torch.rand(A, 2, generator=g) * 600
    wh = torch.rand(A, 2, generator=g) * 30 + 4
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(A, generator=g)
    if kind == "narrow":
        scores = 0.5 + (scores - 0.5) * 0.02                 # random-init logits: everything near 0.5
    elif kind == "ties":
        scores = torch.round(scores * 2000) / 2000          # many exact ties (stable order by candidate index)
    elif kind == "few_values":
        scores = torch.round(scores * 3) / 4 + 0.1          # four distinct values: the top bin alone overflows the selection
    labels = torch.zeros(A, dtype=torch.int32)
    out = pp.nms_batched(boxes[None].to(DEV), scores[None].to(DEV), labels[None].to(DEV), 640.0, 0.05, 0.6, top_k)
    torch.cuda.synchronize()
    k, anchors, kb, ks, _ = oracle_nms_image(boxes, scores, labels, 640.0, 0.05, 0.6, top_k)
    n = int(out["counts"][0])
    assert n == len(k) and torch.equal(out["keep_idx"][0, :n].cpu(), k)
    assert torch.equal(out["keep_anchor"][0, :n].cpu().long(), anchors)
    assert torch.equal(out["boxes"][0, :n].cpu(), kb) and torch.equal(out["scores"][0, :n].cpu(), ks)


def test_nms_large_anchor_count_global_sort_path():
    """A = 33600 (1280x1280): the key array no longer fits LDS and is sorted in the workspace."""
    A = 33600
    boxes, scores, labels = clustered_boxes(A, 7, img=1280.0, n_clusters=40)
    scores = torch.where(torch.arange(A) % 3 == 0, scores, torch.zeros(()))  # ~1/3 pass conf
    out = pp.nms_batched(boxes[None].to(DEV), scores[None].to(DEV), labels[None].to(DEV), 1280.0, 0.05, 0.6, 100)
    torch.cuda.synchronize()
    k, *_ = oracle_nms_image(boxes, scores, labels, 1280.0, 0.05, 0.6, 100)
    n = int(out["counts"][0])
    assert n == len(k) and torch.equal(out["keep_idx"][0, :n].cpu(), k)


@pytest.mark.parametrize("hp,wp", [(40, 40), (64, 64), (40, 64), (16, 16)])  # general path / MFMA x4 fast path (W % 64 == 0)
def test_mask_assembly_and_projector(hp, wp):
    g = torch.Generator().manual_seed(11)
    B, nm, A, K = 2, 32, 300, 21
    SH, SW = 4 * hp, 4 * wp
    protos = torch.randn(B, nm, hp, wp, generator=g)
    mc = torch.randn(B, A, nm, generator=g).permute(0, 2, 1)  # logical [B,nm,A], strided like the model's
    keep_anchor = torch.randint(0, A, (B, K), generator=g, dtype=torch.int32)
    counts = torch.tensor([K, 3], dtype=torch.int32)
    masks, logits = pp.assemble_masks(protos.to(DEV).contiguous(memory_format=torch.channels_last), mc.to(DEV), keep_anchor.to(DEV),
                                      counts.to(DEV), (SH, SW), want_logits=True)
    torch.cuda.synchronize()
    for b in range(B):
        n = int(counts[b])
        coeffs = mc[b, :, keep_anchor[b, :n].long()].t()
        ref_logits, ref_masks = opp.assemble_masks(coeffs, protos[b], (SH, SW))
        assert (logits[b, :n].cpu() - ref_logits).abs().max().item() < 1e-3
        diff = masks[b, :n].cpu() != ref_masks
        assert torch.all(ref_logits[diff].abs() < 1e-4)  # only sign-ambiguous pixels may differ
        assert not masks[b, n:].any()
    if hp == wp:
        w, bias = torch.randn(nm, generator=g) / 6, torch.tensor([0.3])
        out = pp.proto_projector_logits(protos.to(DEV), w.to(DEV), bias.to(DEV), SH)
        torch.cuda.synchronize()
        ref = opp.proto_projector_logits(protos, w, bias, SH)
        assert out.shape == ref.shape and (out.cpu() - ref).abs().max().item() < 1e-3


def test_batch_bbox_iou_bit_exact():
    """pairwise IoU against the reference's own vectors (tests/golden) and the oracle, bit for bit."""
    import os
    cases = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ref_blocks.pt"), weights_only=True)
    b1, b2 = cases["batch_bbox_iou"]["inputs"]
    got = pp.batch_bbox_iou(b1.to(DEV), b2.to(DEV))
    assert torch.equal(got.cpu(), cases["batch_bbox_iou"]["output"])
    e1, e2 = cases["batch_bbox_iou_empty"]["inputs"]
    assert torch.equal(pp.batch_bbox_iou(e1.to(DEV), e2.to(DEV)).cpu(), cases["batch_bbox_iou_empty"]["output"])
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(8400, 2, generator=g) * 600
    a = torch.cat([xy, xy + torch.rand(8400, 2, generator=g) * 120], 1)
    xy2 = torch.rand(7, 2, generator=g) * 600
    b = torch.cat([xy2, xy2 + torch.rand(7, 2, generator=g) * 200], 1)
    b[3] = a[17]  # identical box: IoU ~ 1
    assert torch.equal(pp.batch_bbox_iou(a.to(DEV), b.to(DEV)).cpu(), opp.batch_bbox_iou(a, b))
