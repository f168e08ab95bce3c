"""Fused AdamW over flat buckets (csrc/optim.hip) vs torch.optim.AdamW with the reference trainer's settings
(running_main_v3.py:732-734) on the CPU, several steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_adamw_matches_torch_over_steps():
    from multitask_bonetumor_yolo_amd.dist_train import FlatAdamW, FlatBuckets
    shapes = [("conv.weight", (64, 3, 3, 3)), ("conv.bias", (64,)), ("fc.weight", (10, 1731)), ("fc.bias", (7,)), ("odd", (5,))]
    g = torch.Generator().manual_seed(0)
    ref_p = {n: torch.randn(s, generator=g).requires_grad_() for n, s in shapes}
    opt = torch.optim.AdamW(list(ref_p.values()), lr=1e-3, weight_decay=0.0005, foreach=False)
    params, grads = FlatBuckets(shapes, DEV, bucket_bytes=40000), FlatBuckets(shapes, DEV, bucket_bytes=40000)
    assert len(params.buckets) == 2
    for n, _ in shapes:
        params.views[n].copy_(ref_p[n].detach())
    ours = FlatAdamW(params, grads, lr=1e-3)
    for step in range(6):
        for n, s in shapes:
            gr = torch.randn(s, generator=g) * (10.0 if step == 3 else 1.0)
            ref_p[n].grad = gr.clone()
            grads.views[n].copy_(gr)
        if step == 4:                                   # cosine schedule moves the rate between epochs
            lr = ours.cosine_lr(1e-3, epoch=3, t_max=10)
            for gp in opt.param_groups:
                gp["lr"] = lr
        opt.step()
        ours.step()
        torch.cuda.synchronize()
        for n, _ in shapes:
            a, b = params.views[n].cpu(), ref_p[n].detach()
            assert ((a - b).abs() / (b.abs() + 1e-3)).max().item() < 2e-6, (step, n)
    # the closed-form cosine rate equals torch's scheduler
    p = torch.nn.Parameter(torch.zeros(1))
    o = torch.optim.AdamW([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=10, eta_min=1e-5)
    for e in range(1, 8):
        o.step(); sch.step()
        assert abs(ours.cosine_lr(1e-3, e, 10) - o.param_groups[0]["lr"]) < 1e-12


def test_adamw_rejects_cpu_buckets():
    from multitask_bonetumor_yolo_amd.dist_train import FlatAdamW, FlatBuckets
    b = FlatBuckets([("w", (8,))], "cpu")
    with pytest.raises(RuntimeError):
        FlatAdamW(b, FlatBuckets([("w", (8,))], "cpu"), lr=1e-3).step()
