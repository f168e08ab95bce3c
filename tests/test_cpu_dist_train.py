"""Flat gradient buckets and the data-parallel exchange step (multitask_bonetumor_yolo_amd/dist_train.py) on CPU: layout, and
the world_size-2 gloo path of the all-reduce / parameter broadcast."""
import os
import subprocess
import sys

import torch

from multitask_bonetumor_yolo_amd.dist_train import FlatBuckets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bucket_layout_reverse_order_aligned_views():
    shapes = [("a.weight", (64, 27)), ("a.bias", (64,)), ("b.weight", (7, 5, 3)), ("c.weight", (1000, 100)), ("d.bias", (3,))]
    fb = FlatBuckets(shapes, "cpu", bucket_bytes=8000)
    names = [n for lay in fb.layout for n, _, _ in lay]
    assert names == [n for n, _ in reversed(shapes)]                                   # reverse registration order
    assert len(fb.buckets) == 2 and [len(l) for l in fb.layout] == [2, 3]              # a bucket closes once it reaches bucket_bytes
    for lay, flat in zip(fb.layout, fb.buckets):
        for name, off, n in lay:
            assert off % 4 == 0 and off + n <= flat.numel()
            v = fb.views[name]
            assert tuple(v.shape) == dict(shapes)[name] and v.data_ptr() == flat.data_ptr() + 4 * off
    fb.views["a.bias"].fill_(2.0)
    assert float(sum(b.sum() for b in fb.buckets)) == 128.0
    fb.zero_()
    assert all(float(b.abs().sum()) == 0 for b in fb.buckets)
    # the model's own parameter list packs into 25 MB buckets
    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO
    m = ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)
    big = FlatBuckets([(n, p.shape) for n, p in m.named_parameters()], "cpu")
    total = sum(p.numel() for p in m.parameters())
    assert 7 <= len(big.buckets) <= 8 and sum(b.numel() for b in big.buckets) >= total
    assert big.layout[0][0][0] == list(m.named_parameters())[-1][0]                    # last registered parameter first


def test_gloo_world_size_2_gradient_exchange(tmp_path):
    """Two CPU ranks: broadcast of rank 0's parameters, then the bucketed all-reduce leaves the MEAN gradient on both."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from multitask_bonetumor_yolo_amd.dist_train import FlatBuckets, broadcast_parameters\n"
        "dist.init_process_group('gloo')\n"
        "r = dist.get_rank()\n"
        "shapes = [('w1', (300, 7)), ('b1', (300,)), ('w2', (5, 300)), ('b2', (5,))]\n"
        "params = FlatBuckets(shapes, 'cpu', bucket_bytes=4096)\n"
        "grads = FlatBuckets(shapes, 'cpu', bucket_bytes=4096)\n"
        "g = torch.Generator().manual_seed(100 + r)\n"
        "for b in params.buckets: b.copy_(torch.randn(b.shape, generator=g))\n"
        "broadcast_parameters(params.buckets, src=0)\n"
        "g0 = torch.Generator().manual_seed(100)\n"
        "assert all(torch.equal(b, torch.randn(b.shape, generator=g0)) for b in params.buckets)\n"
        "for n, _ in shapes: grads.views[n].fill_(float(r + 1))\n"
        "for w in grads.all_reduce_mean(): w.wait()\n"
        "assert all(torch.allclose(grads.views[n], torch.full_like(grads.views[n], 1.5)) for n, _ in shapes)\n"
        "assert len(grads.buckets) == 2\n"
        "print(f'RANK{r} ok', flush=True)\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert sorted(l for o in outs for l in o.splitlines() if l.startswith("RANK")) == ["RANK0 ok", "RANK1 ok"]


def test_gradient_arena_layouts_and_unused_parameter_bucket():
    """train.make_arena (no GPU needed): every trainable parameter has a slot whose kernel-layout view maps back to the parameter's own shape
    (4-D conv weights channels-last, depthwise taps tap-major, the ConvTranspose, the zero-padded class conv), and the parameters the loss never
    reaches (Segment cv2 / cv3 / cv4) form leading bucket(s) of their own that the data-parallel step neither reduces nor steps."""
    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO
    from multitask_bonetumor_yolo_amd.train import make_arena
    from multitask_bonetumor_yolo_amd.trainstep import UNUSED_BY_THE_LOSS
    m = ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)
    arena, gview, n_tail = make_arena(m, "cpu", UNUSED_BY_THE_LOSS)
    names = {n for n, p in m.named_parameters() if p.requires_grad}
    assert set(arena.views) == names
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        v = arena.views[n]
        pv = gview[n](v) if gview[n] is not None else v
        assert tuple(pv.shape) == tuple(p.shape), n
        assert pv.untyped_storage().data_ptr() == v.untyped_storage().data_ptr()          # a VIEW of the bucket, not a copy
        assert v.data_ptr() % 16 == 0
    w = m.backbone.c2f_p3.m[0].cv1.conv.weight                                      # [128,128,3,3] -> slot [K,R,S,C]
    assert tuple(arena.views["backbone.c2f_p3.m.0.cv1.conv.weight"].shape) == (128, 3, 3, 128)
    assert tuple(arena.views["detect.cv3.0.2.weight"].shape) == (32, 1, 1, 256)       # nc = 2 rows padded to 32
    assert tuple(arena.views["backbone.body.stages_0.blocks.0.conv_dw.weight"].shape) == (49, 96)
    assert n_tail >= 1
    tail = {n for lay in arena.layout[:n_tail] for n, _, _ in lay}
    assert tail and all(n.startswith(UNUSED_BY_THE_LOSS) for n in tail)
    rest = {n for lay in arena.layout[n_tail:] for n, _, _ in lay}
    assert not any(n.startswith(UNUSED_BY_THE_LOSS) for n in rest)
    # the same layout twice (the parameter buckets and the gradient buckets of TrainStep must coincide)
    arena2, _, n_tail2 = make_arena(m, "cpu", UNUSED_BY_THE_LOSS)
    assert [b.numel() for b in arena.buckets] == [b.numel() for b in arena2.buckets] and n_tail == n_tail2
    # permuted parameter views round-trip values (what re-homing relies on)
    v = arena.views["backbone.c2f_p3.m.0.cv1.conv.weight"]
    gview["backbone.c2f_p3.m.0.cv1.conv.weight"](v).copy_(w.detach())
    assert torch.equal(v.permute(0, 3, 1, 2), w.detach())


def test_every_gradient_bucket_waits_for_all_its_writer_lanes(monkeypatch):
    """The overlapped all-reduce of configs[3] (trainstep.TrainStep._backward_and_exchange; running_main_v3.py:824-828 = implicit DDP):
    the REAL backward launch plan, lowered here on CPU tensors (nothing is issued), scheduled onto 4 lanes.  For every gradient bucket
    that is reduced: every launch that writes into the bucket's storage is marked, the lanes of those writers all carry one of the
    bucket's events, each writer is at or before its lane's event, and the buckets are reduced in the order of their latest writer.
    (Round 2 marked only the program-order-last writer: its event covered ONE of the lanes.)"""
    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, _lib as L, train as T
    from multitask_bonetumor_yolo_amd.engine import _overlap, _region
    from multitask_bonetumor_yolo_amd.trainstep import UNUSED_BY_THE_LOSS, bucket_writers, exchange_marks
    monkeypatch.setenv("MTBT_LANES", "4")
    monkeypatch.setattr(T, "DRY_LOWERING", True)
    torch.manual_seed(0)
    m = ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False).train()
    tp = T.TrainPlan(m, (2, 3, 64, 64), torch.device("cpu"), L.BF16, tail_prefixes=UNUSED_BY_THE_LOSS)
    bwd = tp.backward_plan(("det", "logits", "protos"))
    assert bwd.lane_any and bwd.lanes() == 4
    buckets = tp.arena.buckets
    live = list(range(tp.n_tail_buckets, len(buckets)))
    writers = bucket_writers(bwd.launches, buckets)
    # (1) region-based detection finds a writer for every parameter the plan produces a gradient for -- also where the launch was handed
    #     a derived view of the slot (fc1.bias goes in as `.view(-1)`, which an id()-based match misses)
    for name in bwd.written:
        slot = _region(tp.arena.views[name])
        hit = [i for i, l in enumerate(bwd.launches) if any(_overlap(slot, w) for w in l.writes)]
        if not hit:
            # the one legitimate case: a conv bias in front of a batch-statistic BatchNorm -- its gradient is EXACTLY zero and the slot
            # keeps the arena's zero (train.py conv_bn_act)
            owner = dict(m.named_modules())[name.rsplit(".", 2)[0]]
            assert name.endswith(".conv.bias") and owner.bn.training, f"no launch records a write to the gradient slot of {name}"
            continue
        b = next(k for k, lay in enumerate(tp.arena.layout) if any(n == name for n, _, _ in lay))
        assert set(hit) <= set(writers[b])
    assert not any(writers[b] for b in range(tp.n_tail_buckets))          # Segment cv2 / cv3 / cv4: never written, never reduced
    assert all(writers[b] for b in live)
    # (2) lanes of the writers == lanes that carry one of the bucket's events; every writer at or before its lane's event
    marks, order = exchange_marks(writers, live)
    sch = bwd.schedule()
    points = bwd.mark_points(sch, marks)
    multi = 0
    for b in live:
        w_lanes = {sch.lane[i] for i in writers[b]}
        e_at = {sch.lane[i]: i for i in points[str(b)]}
        assert w_lanes == set(e_at), (b, w_lanes, set(e_at))
        assert all(i <= e_at[sch.lane[i]] for i in writers[b])
        multi += len(w_lanes) > 1
        # what round 2 did -- one event after the last writer -- leaves the other lanes uncovered
        if len(w_lanes) > 1:
            old = bwd.mark_points(sch, {"b": [max(writers[b])]})["b"]
            assert {sch.lane[i] for i in old} != w_lanes
    assert multi > 0, "expected buckets whose writers are spread over several lanes"
    # (3) collectives in the order the buckets complete; a bucket without a writer would wait for the whole plan (last)
    assert sorted(order) == live and [max(writers[b]) for b in order] == sorted(max(writers[b]) for b in live)
    marks2, order2 = exchange_marks([[], [5, 2], [3]], [0, 1, 2])
    assert marks2 == {"1": [5, 2], "2": [3]} and order2 == [2, 1, 0]
    # (4) single-stream issue (MTBT_TRAIN_LANES=0): one event after the highest marked index covers everything (stream order)
    monkeypatch.setenv("MTBT_LANES", "1")
    bwd.reload_env()
    s1 = bwd.schedule()
    assert set(s1.lane) == {0} and all(bwd.mark_points(s1, marks)[str(b)] == [max(writers[b])] for b in live)


def test_snapshot_views_are_fresh_copies():
    """FlatBuckets.snapshot_views: what the autograd node of forward(x, "train") hands out -- views into ONE fresh copy per bucket."""
    shapes = [("a", (4, 3)), ("b", (5,)), ("c", (2, 2, 2)), ("d", (7,))]
    fb = FlatBuckets(shapes, "cpu", bucket_bytes=64)
    for i, (n, _) in enumerate(shapes):
        fb.views[n].fill_(float(i + 1))
    snap = fb.snapshot_views(["a", "c", "d"])
    own = {b.untyped_storage().data_ptr() for b in fb.buckets}
    assert set(snap) == {"a", "c", "d"}
    for n, v in snap.items():
        assert tuple(v.shape) == dict(shapes)[n] and torch.equal(v, fb.views[n]) and v.untyped_storage().data_ptr() not in own
    fb.zero_()
    assert float(snap["c"].sum()) == 8 * 3.0                       # the snapshot does not follow the arena
    same_bucket = [n for n in snap if fb.where[n][0] == fb.where["a"][0]]
    assert len({snap[n].untyped_storage().data_ptr() for n in same_bucket}) == 1   # one copy per bucket, not per tensor


def test_src_model_py_variant_training_plan_lowers(monkeypatch):
    """BASELINE config 0's graph under module.train() (reference src/model.py:27-123): the training lowering of its neck -- lateral Convs, the
    weight-adding WeightedAdd over identity / nearest-x2 / max-pooled inputs, DWConv nodes -- built on CPU tensors (nothing is issued): every
    trainable parameter gets a writer in the backward plan, each WeightedAdd input a resample-backward launch that reads the node's gradient,
    and the max-pooled inputs hand their FORWARD activation to it (the argmax is recomputed, not stored)."""
    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLOv0, _lib as L, train as T
    from multitask_bonetumor_yolo_amd.engine import _overlap, _region
    monkeypatch.setattr(T, "DRY_LOWERING", True)
    torch.manual_seed(0)
    m = ConvNeXtBiFPNYOLOv0(2, 2).train()
    tp = T.TrainPlan(m, (2, 3, 64, 64), torch.device("cpu"), L.F32)
    bwd = tp.backward_plan(("det", "seg", "mc", "protos", "logits"))
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    assert sorted(bwd.written) == sorted(trainable)
    for name in trainable:
        if name.endswith(".conv.bias"):
            continue
        slot = _region(tp.arena.views[name])
        assert any(_overlap(slot, w) for l in bwd.launches for w in l.writes), f"no launch writes the gradient slot of {name}"
    names = [l.name for l in bwd.launches]
    for ui in range(2):
        for key, n_in in (("p4_td", 2), ("p3_td", 2), ("p4_out", 3), ("p5_out", 2)):
            base = f"neck.units.{ui}.add_{key}"
            assert names.count(base + ".norm.bwd") == 1 and names.count(base + ".dysum") == 1
            assert [n for n in names if n.startswith(base + ".bwd")] == [f"{base}.bwd{i}" for i in range(n_in)]
    # mode 4 (max pooling) launches carry the forward input pointer
    pooled = [l for l in bwd.launches if l.name.endswith("add_p5_out.bwd1") or l.name.endswith("add_p4_out.bwd2")]
    assert len(pooled) == 4 and all(l.args[2] == L.RES_MAXPOOL and l.args[1] for l in pooled)
    fwd_names = [l.name for l in tp.fwd.launches]
    assert sum(n.endswith(".norm") and ".add_" in n for n in fwd_names) == 8
