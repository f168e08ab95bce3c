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
