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
