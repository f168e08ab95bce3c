"""Input gradients of the convolutions through the forward kernels (multitask_bonetumor_yolo_amd/backward.py) vs torch autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

if torch.cuda.is_available():
    from multitask_bonetumor_yolo_amd import backward as B
    from multitask_bonetumor_yolo_amd.engine import Act, Plan


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def run(p):
    p.run(stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


def check(got_nhwc, want_nchw, dtype):
    got = got_nhwc.float().cpu().permute(0, 3, 1, 2)
    if dtype == torch.float32:
        assert (got - want_nchw).abs().max().item() < 2e-4
    else:
        assert ((got - want_nchw).abs() / (want_nchw.abs() + 1.0)).max().item() < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,W,C,K,k", [(2, 16, 32, 64, 128, 3), (1, 20, 20, 128, 64, 3), (2, 9, 7, 96, 384, 1), (1, 12, 12, 384, 96, 1)])
def test_conv_dgrad(dtype, N, H, W, C, K, k):
    g = torch.Generator().manual_seed(N * H + K)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(K, C, k, k, generator=g) / (C * k * k) ** 0.5
    dy = torch.randn(N, K, H, W, generator=g)
    if dtype == torch.bfloat16:
        w, dy = w.bfloat16().float(), dy.bfloat16().float()
    (want,) = torch.autograd.grad(F.conv2d(x, w, None, 1, k // 2), x, dy)
    wp = w.permute(0, 2, 3, 1).reshape(K, -1).contiguous().to(DEV, dtype)              # the forward layout [K, R*S*C]
    p = Plan(torch.device(DEV))
    dx = Act.of(torch.zeros(N, H, W, C, dtype=dtype, device=DEV))
    B.conv_dgrad(p, Act.of(nhwc(dy).to(dtype)), B.dgrad_weight(wp, k, k), dx, R=k, S=k, pad=k // 2)
    run(p)
    check(dx.buf, want, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_downsample_dgrad(dtype):
    g = torch.Generator().manual_seed(5)
    N, H, W, C, K = 2, 12, 8, 96, 192
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(K, C, 2, 2, generator=g) / (4 * C) ** 0.5
    dy = torch.randn(N, K, H // 2, W // 2, generator=g)
    if dtype == torch.bfloat16:
        w, dy = w.bfloat16().float(), dy.bfloat16().float()
    (want,) = torch.autograd.grad(F.conv2d(x, w, None, 2, 0), x, dy)
    wp = w.permute(0, 2, 3, 1).reshape(K, -1).contiguous().to(DEV, dtype)
    p = Plan(torch.device(DEV))
    dx = Act.of(torch.zeros(N, H, W, C, dtype=dtype, device=DEV))
    B.downsample2x2_dgrad(p, Act.of(nhwc(dy).to(dtype)), B.downsample2x2_dgrad_weight(wp), dx)
    run(p)
    check(dx.buf, want, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,k,H", [(96, 7, 24), (256, 3, 20), (384, 7, 12)])
def test_dwconv_dgrad(dtype, C, k, H):
    g = torch.Generator().manual_seed(C + k)
    N = 2
    x = torch.randn(N, C, H, H, generator=g, requires_grad=True)
    w = torch.randn(C, 1, k, k, generator=g) / k
    dy = torch.randn(N, C, H, H, generator=g)
    if dtype == torch.bfloat16:
        w, dy = w.bfloat16().float(), dy.bfloat16().float()
    (want,) = torch.autograd.grad(F.conv2d(x, w, None, 1, k // 2, groups=C), x, dy)
    taps = w.view(C, k * k).t().contiguous().to(DEV, dtype)                              # the forward layout [k*k, C]
    p = Plan(torch.device(DEV))
    dx = Act.of(torch.zeros(N, H, H, C, dtype=dtype, device=DEV))
    B.dwconv_dgrad(p, Act.of(nhwc(dy).to(dtype)), B.dwconv_dgrad_weight(taps, k), dx, k, torch.ones(C, device=DEV), torch.zeros(C, device=DEV))
    run(p)
    check(dx.buf, want, dtype)


@pytest.mark.parametrize("N,H,W,C,K,k", [(2, 16, 16, 64, 64, 3), (1, 20, 20, 128, 96, 3), (2, 9, 7, 96, 384, 1), (3, 40, 40, 256, 64, 3), (1, 5, 3, 8, 8, 3),
                                         (2, 9, 7, 384, 512, 1),      # wide tile on the output channels (256 x 128: K % 256 == 0, K >= C), ragged last pixel step
                                         (2, 9, 7, 512, 384, 1),      # wide tile on the input channels (128 x 256)
                                         (1, 12, 12, 256, 256, 1),    # square 256: one wide tile per row
                                         (1, 6, 6, 256, 192, 3),      # k x k taps with a wide input side, ragged output-channel tile
                                         (1, 12, 11, 768, 512, 1),    # 256 x 256 tile, 512 threads (the stage-3 MLP shapes)
                                         (2, 16, 24, 96, 160, 3)])    # 3x3 halo kernel (eight waves, taps split 5 + 4) with ragged channel tiles on both sides
def test_conv_wgrad(N, H, W, C, K, k):
    """Weight gradient (csrc/wgrad.hip) vs autograd, bf16-rounded operands, ragged channel tiles and pixel slices."""
    g = torch.Generator().manual_seed(N * H + K + k)
    x = (torch.randn(N, C, H, W, generator=g)).bfloat16().float()
    w = (torch.randn(K, C, k, k, generator=g) / (C * k * k) ** 0.5).requires_grad_()
    dy = torch.randn(N, K, H, W, generator=g).bfloat16().float()
    (want,) = torch.autograd.grad(F.conv2d(x, w, None, 1, k // 2), w, dy)
    want = want.permute(0, 2, 3, 1).reshape(K, -1)                                   # packed [K, R*S*C]
    xa, dya = Act.of(nhwc(x).bfloat16()), Act.of(nhwc(dy).bfloat16())
    got = B.conv_wgrad(xa, dya, R=k, S=k, pad=k // 2)
    torch.cuda.synchronize()
    scale = want.abs().max().item()
    assert (got.cpu() - want).abs().max().item() <= 2e-4 * scale + 1e-5               # fp32 accumulation of exact bf16 products
    again = B.conv_wgrad(xa, dya, R=k, S=k, pad=k // 2)
    assert torch.equal(got, again)                                                    # deterministic
    acc = B.conv_wgrad(xa, dya, R=k, S=k, pad=k // 2, out=got.clone(), accumulate=True)
    assert torch.allclose(acc, 2 * got, rtol=1e-5, atol=1e-5 * scale)                # (dw + partials) rounds differently from 2 * dw


@pytest.mark.parametrize("P,C,K", [(2 * 9 * 7, 384, 96), (3 * 8 * 8, 768, 192), (130, 256, 256)])
def test_conv_wgrad_with_gelu_applied_to_the_staged_operand(P, C, K):
    """mtbt_conv_wgrad_xact: dW[k][c] = sum_p dy[p][k] * gelu(x[p][c]) with x the kept fc1 PRE-activation (the fc2 weight gradient behind the fused
    training forward) equals the plain kernel on the activated tensor rounded to bf16 -- both tile orientations."""
    from multitask_bonetumor_yolo_amd import _lib as L
    g = torch.Generator().manual_seed(P + C)
    x = (torch.randn(1, P, 1, C, generator=g) * 2).bfloat16()
    dy = torch.randn(1, P, 1, K, generator=g).bfloat16()
    h = torch.nn.functional.gelu(x.float()).bfloat16()
    want = dy.float().view(P, K).t() @ h.float().view(P, C)
    lib = L.load()
    xd, dyd = x.to(DEV), dy.to(DEV)
    out = torch.zeros(K, C, dtype=torch.float32, device=DEV)
    nbytes = lib.mtbt_conv_wgrad_workspace_bytes(1, P, 1, C, K, 1, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    rc = lib.mtbt_conv_wgrad_xact(xd.data_ptr(), dyd.data_ptr(), out.data_ptr(), 1, P, 1, C, K, 1, 1, 0, 1, P * C, C, P * K, K, L.BF16, L.ACT_GELU_POLY, 0,
                                  ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    scale = want.abs().max().item()
    assert (out.cpu() - want).abs().max().item() <= 6e-3 * scale          # the polynomial GELU (2.3e-4 absolute) and one bf16 rounding of h


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["silu", "elu", "gelu", "none"])
def test_act_backward(dtype, act):
    from multitask_bonetumor_yolo_amd import _lib as L
    code = {"silu": L.ACT_SILU, "elu": L.ACT_ELU, "gelu": L.ACT_GELU, "none": L.ACT_NONE}[act]
    fn = {"silu": F.silu, "elu": F.elu, "gelu": F.gelu, "none": lambda t: t}[act]
    g = torch.Generator().manual_seed(3)
    z = (torch.randn(2, 24, 9, 11, generator=g) * 3).to(dtype).float().requires_grad_()
    dy = torch.randn(2, 24, 9, 11, generator=g).to(dtype).float()
    (want,) = torch.autograd.grad(fn(z), z, dy)
    got = B.act_backward(Act.of(nhwc(dy).to(dtype)), Act.of(nhwc(z.detach()).to(dtype)), code)
    torch.cuda.synchronize()
    check(got.buf, want, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_channel_sum(dtype):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 40, 31, 23, generator=g).to(dtype).float()          # 2139 pixels: ragged last workgroup
    a = Act.of(nhwc(x).to(dtype))
    got = B.channel_sum(a)
    torch.cuda.synchronize()
    want = x.sum(dim=(0, 2, 3))
    assert torch.allclose(got.cpu(), want, rtol=1e-5, atol=1e-3)
    assert torch.equal(got, B.channel_sum(a))
    twice = B.channel_sum(a, out=got.clone(), accumulate=True)
    assert torch.allclose(twice.cpu(), 2 * want, rtol=1e-5, atol=2e-3)
    y = torch.randn(3, 40, 31, 23, generator=g).to(dtype).float()
    dot = B.channel_sum(a, times=Act.of(nhwc(y).to(dtype)))
    assert torch.allclose(dot.cpu(), (x * y).sum(dim=(0, 2, 3)), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("k", [3, 1])
def test_convblock_backward_composes(k):
    """Backward of one reference-owned block -- ConvBlock = Conv2d(bias) -> BatchNorm2d (running statistics) -> SiLU,
    main_model.py:113-141 -- assembled from the pieces (activation derivative, channel sums, dgrad through the forward kernel,
    wgrad) against torch autograd through the oracle block: dx, dW, d bias, d gamma, d beta.  bf16 activations, fp32 gradients."""
    from multitask_bonetumor_yolo_amd import _lib as L
    from oracle.blocks import ConvBlock
    torch.manual_seed(11)
    N, H, W, C, K = 2, 20, 20, 64, 128
    blk = ConvBlock(C, K, k).eval()
    with torch.no_grad():
        blk.bn.running_mean.normal_(0, 0.1); blk.bn.running_var.uniform_(0.5, 1.5)
        blk.bn.weight.uniform_(0.5, 1.5); blk.bn.bias.normal_(0, 0.1)
        blk.conv.weight.copy_(blk.conv.weight.bfloat16().float())
    x = torch.randn(N, C, H, W).bfloat16().float().requires_grad_()
    dy = torch.randn(N, K, H, W).bfloat16().float()
    blk(x).backward(dy)
    # ---- the same with the HIP pieces ----
    s = (blk.bn.weight / torch.sqrt(blk.bn.running_var + blk.bn.eps)).detach()           # u = s * (W*x + b) + t
    t = (blk.bn.bias - blk.bn.running_mean * s).detach()
    wp = blk.conv.weight.detach().permute(0, 2, 3, 1).reshape(K, -1).contiguous()          # [K, R*S*C]
    p = Plan(torch.device(DEV))
    xa = Act.of(nhwc(x.detach()).bfloat16())
    u = Act.of(torch.empty(N, H, W, K, dtype=torch.bfloat16, device=DEV))                 # pre-activation kept by a training forward
    p.conv(xa, wp.to(DEV, torch.bfloat16), u, R=k, S=k, pad=k // 2, scale=s.to(DEV), shift=(blk.conv.bias.detach() * s + t).to(DEV))
    run(p)
    du = B.act_backward(Act.of(nhwc(dy).bfloat16()), u, L.ACT_SILU)                        # dy * SiLU'(u)
    d_beta = B.channel_sum(du)
    d_udotu = B.channel_sum(du, times=u)
    sd, td = s.to(DEV), t.to(DEV)
    gamma, beta = blk.bn.weight.detach().to(DEV), blk.bn.bias.detach().to(DEV)
    d_gamma = (d_udotu - beta * d_beta) / gamma                                            # sum du * (u - beta) / gamma
    d_bias = d_beta * sd                                                                   # d(W*x + b) = s * du
    dW = B.conv_wgrad(xa, du, R=k, S=k, pad=k // 2) * sd[:, None]
    p2 = Plan(torch.device(DEV))
    dx = Act.of(torch.empty(N, H, W, C, dtype=torch.bfloat16, device=DEV))
    B.conv_dgrad(p2, du, B.dgrad_weight((wp.to(DEV) * sd[:, None]).to(torch.bfloat16), k, k), dx, R=k, S=k, pad=k // 2)
    run(p2)

    def close(got, want, tol):
        return (got.float().cpu() - want).abs().max().item() <= tol * want.abs().max().item()
    assert close(dx.buf.permute(0, 3, 1, 2), x.grad, 3e-2)                                 # bf16 du, bf16 folded weights, bf16 dx
    assert close(dW, blk.conv.weight.grad.permute(0, 2, 3, 1).reshape(K, -1), 2e-2)
    assert close(d_bias, blk.conv.bias.grad, 2e-2)
    assert close(d_beta, blk.bn.bias.grad, 2e-2)
    assert close(d_gamma, blk.bn.weight.grad, 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_train_backward(dtype):
    """BatchNorm2d on batch statistics (ultralytics Conv inside the heads in train mode) followed by SiLU: dx, d gamma, d beta."""
    from multitask_bonetumor_yolo_amd import _lib as L
    torch.manual_seed(5)
    N, C, H, W = 4, 64, 12, 10
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    x = (torch.randn(N, C, H, W) * 2 + 0.3).requires_grad_()
    dy = torch.randn(N, C, H, W).to(dtype).float()
    u_ref = bn(x)
    F.silu(u_ref).backward(dy)
    var = x.detach().var(dim=(0, 2, 3), unbiased=False)
    u = Act.of(nhwc(u_ref.detach()).to(dtype))
    dz = B.act_backward(Act.of(nhwc(dy).to(dtype)), u, L.ACT_SILU)
    dx, dg, db = B.batchnorm_train_backward(dz, u, bn.weight.detach().to(DEV), bn.bias.detach().to(DEV), var.to(DEV), bn.eps)
    torch.cuda.synchronize()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    rel = lambda got, want: (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
    assert rel(dx.buf.permute(0, 3, 1, 2), x.grad) < tol
    assert rel(dg, bn.weight.grad) < tol and rel(db, bn.bias.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [96, 384, 768])
def test_layernorm_backward(dtype, C):
    """timm ConvNeXt block norm: LayerNorm over channels (eps 1e-6) in NHWC."""
    g = torch.Generator().manual_seed(C)
    N, H, W = 2, 9, 7
    x = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.2).to(dtype).float().requires_grad_()
    ln = torch.nn.LayerNorm(C, eps=1e-6)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.1)
    dy = torch.randn(N, H, W, C, generator=g).to(dtype).float()
    ln(x).backward(dy)
    dx, dg, db = B.layernorm_backward(Act.of(x.detach().to(DEV, dtype)), Act.of(dy.to(DEV, dtype)), ln.weight.detach().to(DEV), 1e-6)
    torch.cuda.synchronize()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    rel = lambda got, want: (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
    assert rel(dx.buf, x.grad) < tol and rel(dg, ln.weight.grad) < tol and rel(db, ln.bias.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,k,H", [(96, 7, 18), (256, 3, 11), (192, 7, 40), (384, 3, 16)])   # (two chunks x many tiles per slot; three chunks)
def test_dwconv_wgrad(dtype, C, k, H):
    g = torch.Generator().manual_seed(C * k)
    N = 2
    x = torch.randn(N, C, H, H, generator=g).to(dtype).float()
    w = (torch.randn(C, 1, k, k, generator=g) / k).requires_grad_()
    dy = torch.randn(N, C, H, H, generator=g).to(dtype).float()
    (want,) = torch.autograd.grad(F.conv2d(x, w, None, 1, k // 2, groups=C), w, dy)
    got = B.dwconv_wgrad(Act.of(nhwc(x).to(dtype)), Act.of(nhwc(dy).to(dtype)), k)
    torch.cuda.synchronize()
    want = want.view(C, k * k).t()
    assert (got.cpu() - want).abs().max().item() <= 2e-4 * want.abs().max().item() + 1e-5


def test_convnext_block_backward_composes():
    """Backward of a whole ConvNeXt block (timm: dw 7x7 + bias -> LayerNorm -> Linear -> GELU -> Linear -> layer-scale -> + x) assembled
    from the pieces, against torch autograd through the oracle block: dx and the gradient of every parameter.  bf16 activations."""
    from multitask_bonetumor_yolo_amd import _lib as L
    from oracle.convnext import ConvNeXtBlock
    torch.manual_seed(21)
    N, H, W, d = 2, 12, 12, 96
    blk = ConvNeXtBlock(d)
    with torch.no_grad():
        blk.gamma.uniform_(0.2, 1.0)
        blk.norm.weight.uniform_(0.5, 1.5); blk.norm.bias.normal_(0, 0.1)
        for prm in (blk.conv_dw.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight):
            prm.copy_(prm.bfloat16().float())
    x = torch.randn(N, d, H, W).bfloat16().float().requires_grad_()
    dy = torch.randn(N, d, H, W).bfloat16().float()
    blk(x).backward(dy)

    bf = torch.bfloat16
    dev = torch.device(DEV)
    ones = lambda c: torch.ones(c, device=DEV)
    zeros = lambda c: torch.zeros(c, device=DEV)
    new = lambda c: Act.of(torch.empty(N, H, W, c, dtype=bf, device=DEV))
    taps = blk.conv_dw.weight.detach().view(d, 49).t().contiguous().to(DEV, bf)
    W1 = blk.mlp.fc1.weight.detach().to(DEV, bf).contiguous()            # [4d, d] = packed [K, C]
    W2 = blk.mlp.fc2.weight.detach().to(DEV, bf).contiguous()            # [d, 4d]
    gam = blk.gamma.detach().to(DEV)
    # ---- a training forward that keeps what the backward needs ----
    xa = Act.of(nhwc(x.detach()).to(bf))
    a, t, h, hg, o = new(d), new(d), new(4 * d), new(4 * d), new(d)
    p = Plan(dev)
    p.dwconv(xa, taps, a, 7, scale=ones(d), shift=blk.conv_dw.bias.detach().to(DEV))
    p.layernorm(a, blk.norm.weight.detach().to(DEV), blk.norm.bias.detach().to(DEV), 1e-6, t)
    p.conv(t, W1, h, shift=blk.mlp.fc1.bias.detach().to(DEV))
    p.conv(t, W1, hg, shift=blk.mlp.fc1.bias.detach().to(DEV), act=L.ACT_GELU)
    p.conv(hg, W2, o, shift=blk.mlp.fc2.bias.detach().to(DEV))
    run(p)
    # ---- backward ----
    dya = Act.of(nhwc(dy).to(bf))
    d_gamma = B.channel_sum(dya, times=o)

    def affine2(x1, x2, c1, c2):
        import ctypes as C
        out = Act.of(torch.empty_like(x1.buf))
        z = zeros(x1.C)
        L.check(L.load().mtbt_channel_affine2(x1.ptr, x2.ptr, c1.data_ptr(), c2.data_ptr(), z.data_ptr(), out.ptr, N * H * W, x1.C, x1.code,
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "affine2")
        torch.cuda.synchronize()
        return out
    do = affine2(dya, dya, gam, zeros(d))                                 # d o = dy * gamma
    dW2, db2 = B.conv_wgrad(hg, do, R=1, S=1, pad=0), B.channel_sum(do)
    dhg, dt, dxdw = new(4 * d), new(d), new(d)
    p2 = Plan(dev)
    B.conv_dgrad(p2, do, B.dgrad_weight(W2, 1, 1), dhg, R=1, S=1, pad=0)
    run(p2)
    dh = B.act_backward(dhg, h, L.ACT_GELU)
    dW1, db1 = B.conv_wgrad(t, dh, R=1, S=1, pad=0), B.channel_sum(dh)
    p3 = Plan(dev)
    B.conv_dgrad(p3, dh, B.dgrad_weight(W1, 1, 1), dt, R=1, S=1, pad=0)
    run(p3)
    da, d_lnw, d_lnb = B.layernorm_backward(a, dt, blk.norm.weight.detach().to(DEV), 1e-6)
    d_taps, d_dwb = B.dwconv_wgrad(xa, da, 7), B.channel_sum(da)
    p4 = Plan(dev)
    B.dwconv_dgrad(p4, da, B.dwconv_dgrad_weight(taps, 7), dxdw, 7, ones(d), zeros(d))
    run(p4)
    dx = affine2(dxdw, dya, ones(d), ones(d))                             # + the residual path

    def close(got, want, tol=4e-2):
        return (got.float().cpu() - want).abs().max().item() <= tol * want.abs().max().item()
    assert close(dx.buf.permute(0, 3, 1, 2), x.grad)
    assert close(d_gamma, blk.gamma.grad)
    assert close(dW2, blk.mlp.fc2.weight.grad) and close(db2, blk.mlp.fc2.bias.grad)
    assert close(dW1, blk.mlp.fc1.weight.grad) and close(db1, blk.mlp.fc1.bias.grad)
    assert close(d_lnw, blk.norm.weight.grad) and close(d_lnb, blk.norm.bias.grad)
    assert close(d_taps, blk.conv_dw.weight.grad.view(d, 49).t()) and close(d_dwb, blk.conv_dw.bias.grad)


def test_conv_wgrad_strided_downsample():
    """ConvNeXt downsample conv: 2x2, stride 2, no padding."""
    g = torch.Generator().manual_seed(8)
    N, H, W, C, K = 2, 12, 20, 96, 192
    x = torch.randn(N, C, H, W, generator=g).bfloat16().float()
    w = (torch.randn(K, C, 2, 2, generator=g) / (4 * C) ** 0.5).requires_grad_()
    dy = torch.randn(N, K, H // 2, W // 2, generator=g).bfloat16().float()
    (want,) = torch.autograd.grad(F.conv2d(x, w, None, 2, 0), w, dy)
    want = want.permute(0, 2, 3, 1).reshape(K, -1)
    got = B.conv_wgrad(Act.of(nhwc(x).bfloat16()), Act.of(nhwc(dy).bfloat16()), R=2, S=2, pad=0, stride=2)
    torch.cuda.synchronize()
    assert (got.cpu() - want).abs().max().item() <= 2e-4 * want.abs().max().item() + 1e-5


def test_head_conv_train_mode_backward_composes():
    """ultralytics Conv as the heads run it in `forward(x, "train")`: Conv2d(3x3, no bias) -> BatchNorm2d on BATCH statistics
    (mtbt_bn_train_nhwc) -> SiLU.  Backward from the pieces vs autograd: dx, dW, d gamma, d beta."""
    from multitask_bonetumor_yolo_amd import _lib as L
    torch.manual_seed(31)
    N, H, W, C, K = 4, 20, 20, 64, 64
    conv = torch.nn.Conv2d(C, K, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(K, eps=1e-3, momentum=0.03).train()
    with torch.no_grad():
        conv.weight.copy_(conv.weight.bfloat16().float())
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    x = torch.randn(N, C, H, W).bfloat16().float().requires_grad_()
    dy = torch.randn(N, K, H, W).bfloat16().float()
    F.silu(bn(conv(x))).backward(dy)

    import copy
    bf, dev = torch.bfloat16, torch.device(DEV)
    wp = conv.weight.detach().permute(0, 2, 3, 1).reshape(K, -1).contiguous().to(DEV, bf)
    xa = Act.of(nhwc(x.detach()).to(bf))
    z, u = Act.of(torch.empty(N, H, W, K, dtype=bf, device=DEV)), Act.of(torch.empty(N, H, W, K, dtype=bf, device=DEV))
    bn_dev = copy.deepcopy(bn).to(DEV)
    bn_dev.running_mean.zero_(); bn_dev.running_var.fill_(1.0)
    p = Plan(dev)
    p.conv(xa, wp, z, R=3, S=3, pad=1)
    p.bn_train(z, u, bn_dev, L.ACT_NONE)
    run(p)
    var = z.buf.float().var(dim=(0, 1, 2), unbiased=False)                      # the batch variance the forward kernel used
    du = B.act_backward(Act.of(nhwc(dy).to(bf)), u, L.ACT_SILU)
    dz, dg, db = B.batchnorm_train_backward(du, u, bn.weight.detach().to(DEV), bn.bias.detach().to(DEV), var, bn.eps)
    dW = B.conv_wgrad(xa, dz, R=3, S=3, pad=1)
    dx = Act.of(torch.empty(N, H, W, C, dtype=bf, device=DEV))
    p2 = Plan(dev)
    B.conv_dgrad(p2, dz, B.dgrad_weight(wp, 3, 3), dx, R=3, S=3, pad=1)
    run(p2)
    close = lambda got, want, tol=4e-2: (got.float().cpu() - want).abs().max().item() <= tol * want.abs().max().item()
    assert close(dx.buf.permute(0, 3, 1, 2), x.grad)
    assert close(dW, conv.weight.grad.permute(0, 2, 3, 1).reshape(K, -1))
    assert close(dg, bn.weight.grad) and close(db, bn.bias.grad)
