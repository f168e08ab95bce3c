"""The training-step kernels (csrc/train_ops.hip, resample_bwd.hip, wgrad.hip fp32 / stem, optim.hip, the conv epilogue's y2 /
derivative modes) one by one against torch autograd on the same inputs, through the C ABI."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

if torch.cuda.is_available():
    from multitask_bonetumor_yolo_amd import _lib as L
    from multitask_bonetumor_yolo_amd.engine import Act, Plan


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def nhwc(t, dtype=torch.float32):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)


def back(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def close(got, want, rtol, what=""):
    err = (got.float().cpu() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= rtol * scale + 1e-7, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
CODE = {torch.float32: 0, torch.bfloat16: 1}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("running", [False, True])
@pytest.mark.parametrize("act", ["silu", "elu"])
def test_bn_forward_backward(dtype, running, act):
    """mtbt_bn_forward_nhwc writing a channel SLICE + mtbt_bn_backward_nhwc reading its gradient from a slice vs autograd."""
    lib = L.load()
    torch.manual_seed(3)
    N, Cc, H, W, LD, OFF = 3, 64, 9, 7, 160, 32
    bn = torch.nn.BatchNorm2d(Cc, eps=4e-5, momentum=0.9997)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    bn.train(not running)
    fn = {"silu": F.silu, "elu": F.elu}[act]
    code = {"silu": L.ACT_SILU, "elu": L.ACT_ELU}[act]
    x = (torch.randn(N, Cc, H, W) * 1.5 + 0.2).to(dtype).float().requires_grad_()
    dy = torch.randn(N, Cc, H, W).to(dtype).float()
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    y_ref = fn(bn(x))
    y_ref.backward(dy)
    g, b = bn.weight.detach().to(DEV), bn.bias.detach().to(DEV)
    rm, rv = rm0.to(DEV), rv0.to(DEV)
    xd = nhwc(x.detach(), dtype)
    P = N * H * W
    ycat = torch.zeros(N, H, W, LD, dtype=dtype, device=DEV)
    stats = torch.zeros(2 * Cc, device=DEV)
    nb = lib.mtbt_bn_train_workspace_bytes(P, Cc)
    ws = torch.empty(nb // 4 + 16, device=DEV)
    yv = ycat.view(-1)[OFF:]
    L.check(lib.mtbt_bn_forward_nhwc(xd.data_ptr(), yv.data_ptr(), LD, g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.9997, 4e-5, code,
                                     P, Cc, CODE[dtype], int(running), stats.data_ptr(), ws.data_ptr(), nb, S()), "bn_forward")
    torch.cuda.synchronize()
    close(back(ycat[..., OFF:OFF + Cc]), y_ref.detach(), TOL[dtype] * 5, "y")
    assert ycat[..., :OFF].abs().max().item() == 0 and ycat[..., OFF + Cc:].abs().max().item() == 0
    if not running:
        close(rm, bn.running_mean, 1e-5, "running_mean")
        close(rv, bn.running_var, 1e-4 if dtype == torch.float32 else 1e-2, "running_var")
    dcat = torch.zeros(N, H, W, LD, dtype=dtype, device=DEV)
    dcat[..., OFF:OFF + Cc] = nhwc(dy, dtype)
    dx = torch.empty(N, H, W, Cc, dtype=dtype, device=DEV)
    dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    nb2 = lib.mtbt_bn_backward_workspace_bytes(P, Cc)
    ws2 = torch.empty(nb2 // 4, device=DEV)
    L.check(lib.mtbt_bn_backward_nhwc(dcat.view(-1)[OFF:].data_ptr(), LD, xd.data_ptr(), stats.data_ptr(), g.data_ptr(), b.data_ptr(), 4e-5, code,
                                      int(running), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, P, Cc, CODE[dtype], ws2.data_ptr(), nb2, S()), "bn_backward")
    torch.cuda.synchronize()
    close(back(dx), x.grad, TOL[dtype] * 5, "dx")
    close(dg, bn.weight.grad, TOL[dtype] * 5, "dgamma")
    close(db, bn.bias.grad, TOL[dtype] * 5, "dbeta")


def test_weight_prep_layouts():
    lib = L.load()
    torch.manual_seed(0)
    w = torch.randn(24, 16, 3, 3, device=DEV)
    wl = torch.randn(40, 24, device=DEV)                                       # Linear
    wcl = torch.randn(24, 16, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    g = torch.rand(40, device=DEV) + 0.5
    small = torch.randn(2, 24, 1, 1, device=DEV)                                # class conv: K padded to 8
    specs = []

    def add(src, dims, ss, flips=(0, 0, 0, 0), s0=None, s1=None, dtype=torch.float32, src_dim3=0):
        dst = torch.full(tuple(dims), 7.0, dtype=dtype, device=DEV)
        specs.append((src, dst, dims, ss, flips, s0, s1, src_dim3))
        return dst
    fwd = add(w, (24, 3, 3, 16), (w.stride(0), w.stride(2), w.stride(3), w.stride(1)), dtype=torch.bfloat16)
    dgr = add(w, (16, 3, 3, 24), (w.stride(1), w.stride(2), w.stride(3), w.stride(0)), flips=(0, 1, 1, 0))
    fcl = add(wcl, (24, 3, 3, 16), (wcl.stride(0), wcl.stride(2), wcl.stride(3), wcl.stride(1)))
    lin_t = add(wl, (24, 1, 1, 40), (wl.stride(1), 0, 0, wl.stride(0)), s0=(g, 3))
    lin_c = add(wl, (40, 1, 1, 24), (wl.stride(0), 0, 0, wl.stride(1)), s0=(g, 0))
    padded = add(small, (24, 1, 1, 8), (small.stride(1), 0, 0, small.stride(0)), src_dim3=2)
    n = len(specs)
    table = (L.PrepDesc * n)()
    starts, total = [], 0
    for i, (src, dst, dims, ss, flips, s0, s1, sd3) in enumerate(specs):
        e = table[i]
        e.src, e.dst = src.data_ptr(), dst.data_ptr()
        e.scale0, e.scale0_dim = (s0[0].data_ptr(), s0[1]) if s0 else (None, 0)
        e.scale1, e.scale1_dim = (s1[0].data_ptr(), s1[1]) if s1 else (None, 0)
        for q in range(4):
            e.sstride[q], e.dim[q], e.flip[q] = ss[q], dims[q], flips[q]
        e.dst_dtype = 0 if dst.dtype == torch.float32 else 1
        e.src_dim3 = sd3
        starts.append(total)
        total += lib.mtbt_weight_prep_blocks(dst.numel())
    tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV)
    sdev = torch.tensor(starts, dtype=torch.int32, device=DEV)
    L.check(lib.mtbt_weight_prep(tdev.data_ptr(), sdev.data_ptr(), n, total, S()), "weight_prep")
    torch.cuda.synchronize()
    assert torch.equal(fwd, w.permute(0, 2, 3, 1).bfloat16())
    assert torch.equal(dgr, w.flip(2, 3).permute(1, 2, 3, 0))
    assert torch.equal(fcl, wcl.permute(0, 2, 3, 1))
    assert torch.equal(lin_t.view(24, 40), wl.t() * g[None, :])
    assert torch.equal(lin_c.view(40, 24), wl * g[:, None])
    want = torch.zeros(24, 8, device=DEV)
    want[:, :2] = small.view(2, 24).t()
    assert torch.equal(padded.view(24, 8), want)


def test_bifpn_norm_weights():
    lib = L.load()
    for n in (2, 3):
        w = torch.tensor([[0.7, -0.4], [1.3, 0.2], [-1.1, 2.0]][:n], requires_grad=True)
        e = F.elu(w)
        a = e / (e.sum(dim=0, keepdim=True) + 1e-4)
        da = torch.randn(n, 2)
        a.backward(da)
        wd = w.detach().to(DEV).contiguous()
        out = torch.empty(2 * n, device=DEV)
        L.check(lib.mtbt_bifpn_norm_weights(wd.data_ptr(), n, 1e-4, out.data_ptr(), S()), "norm")
        dout = da.t().contiguous().to(DEV)                                       # transposed [2][n]
        dw = torch.empty(n, 2, device=DEV)
        L.check(lib.mtbt_bifpn_norm_weights_backward(wd.data_ptr(), n, 1e-4, dout.data_ptr(), dw.data_ptr(), 0, S()), "norm bwd")
        torch.cuda.synchronize()
        close(out.view(2, n).t(), a.detach(), 1e-6, "norm")
        close(dw, w.grad, 1e-5, "norm bwd")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fuse_backward(dtype, mode):
    lib = L.load()
    torch.manual_seed(mode)
    N, Cc, H, W = 2, 16, 6, 10
    hi, wi = {0: (H, W), 1: (H // 2, W // 2), 2: (2 * H, 2 * W)}[mode]
    x = torch.randn(N, Cc, hi, wi).to(dtype).float().requires_grad_()
    wgt = torch.tensor(0.37, requires_grad=True)
    dy = torch.randn(N, Cc, H, W).to(dtype).float()
    r = x if mode == 0 else (F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) if mode == 1
                             else F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False))
    (wgt * r).backward(dy)
    prev = torch.randn(N, hi, wi, Cc).to(dtype)                                  # dx accumulates onto this
    dx = prev.clone().to(DEV)
    dwg = torch.zeros(1, device=DEV)
    wdev = torch.tensor([0.37], device=DEV)
    nb = lib.mtbt_bifpn_fuse_backward_workspace_bytes()
    ws = torch.empty(nb // 4, device=DEV)
    dyd, xd = nhwc(dy, dtype), nhwc(x.detach(), dtype)
    L.check(lib.mtbt_bifpn_fuse_backward(dyd.data_ptr(), xd.data_ptr(), mode, wdev.data_ptr(), dx.data_ptr(), 1, dwg.data_ptr(), 0, N, H, W, Cc,
                                         CODE[dtype], ws.data_ptr(), nb, S()), "fuse bwd")
    torch.cuda.synchronize()
    close(back(dx) - prev.float().permute(0, 3, 1, 2), x.grad, TOL[dtype] * 8, "dx")
    close(dwg, wgt.grad.view(1), TOL[dtype] * 4, "dwgt")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 3, 4])
def test_resample_backward_of_the_src_model_py_neck(dtype, mode):
    """reference src/model.py:60-74: the WeightedAdd inputs are identity, F.interpolate(scale_factor=2, mode="nearest") and F.max_pool2d(., 2);
    mtbt_resample_backward = their transposes.  The pooled input carries exact TIES (constant windows, equal pairs): torch routes the
    gradient to the first maximum in row-major order, so must the kernel."""
    lib = L.load()
    torch.manual_seed(10 + mode)
    N, Cc, H, W = 2, 16, 6, 10
    hi, wi = {0: (H, W), 3: (H // 2, W // 2), 4: (2 * H, 2 * W)}[mode]
    x = torch.randn(N, Cc, hi, wi).to(dtype).float()
    if mode == 4:
        x[:, :, 0:2, 0:2] = 0.5                       # a constant window
        x[:, :, 2, 2] = x[:, :, 3, 3] = 9.0           # first and last position tie
        x[:, :, 4, 5] = x[:, :, 5, 4] = 9.0           # second and third position tie
    x.requires_grad_()
    dy = torch.randn(N, Cc, H, W).to(dtype).float()
    r = x if mode == 0 else (F.interpolate(x, scale_factor=2, mode="nearest") if mode == 3 else F.max_pool2d(x, 2))
    r.backward(dy)
    dyd, xd = nhwc(dy, dtype), nhwc(x.detach(), dtype)
    for accumulate in (0, 1):
        prev = torch.randn(N, hi, wi, Cc).to(dtype)
        dx = prev.clone().to(DEV)
        L.check(lib.mtbt_resample_backward(dyd.data_ptr(), xd.data_ptr(), mode, dx.data_ptr(), accumulate, N, H, W, Cc, CODE[dtype], S()), "resample bwd")
        torch.cuda.synchronize()
        got = back(dx) - (prev.float().permute(0, 3, 1, 2) if accumulate else 0.0)
        close(got, x.grad, TOL[dtype] * 4, f"dx accumulate={accumulate}")
    assert lib.mtbt_resample_backward(dx.data_ptr(), None, 4, dx.data_ptr(), 0, N, H, W, Cc, CODE[dtype], S()) != 0     # max pooling needs the forward input
    assert lib.mtbt_resample_backward(dx.data_ptr(), dx.data_ptr(), 1, dx.data_ptr(), 0, N, H, W, Cc, CODE[dtype], S()) != 0  # bilinear: mtbt_bifpn_fuse_backward


def test_wadd_norm_weights_of_the_src_model_py_neck():
    """reference src/model.py:27-37: w = relu(w); w = w / (w.sum() + eps); out = sum(w_i + f_i).  The output gets s / (s + eps) added to every
    element, so d w_j = [w_j > 0] eps / (s + eps)^2 sum(dy)."""
    lib = L.load()
    for vals in ([0.7, 1.3], [1.2, -0.3, 0.8], [0.0, 2.0]):
        n, Cc = len(vals), 24
        # float64 reference: autograd differentiates the quotient as 1 / (s + eps) - s / (s + eps)^2, two terms that agree to ~5 digits -- in fp32
        # torch's own gradient is only good to ~1e-3 relative, the kernel evaluates eps / (s + eps)^2 directly
        w = torch.tensor(vals, dtype=torch.float64, requires_grad=True)
        f = [torch.randn(2, Cc, 4, 5, dtype=torch.float64) for _ in range(n)]
        wn = F.relu(w) / (F.relu(w).sum() + 1e-4)
        y = sum(w_i + f_i for w_i, f_i in zip(wn, f))
        dy = torch.randn_like(y).float().double()
        y.backward(dy)
        wd = w.detach().float().to(DEV)
        out = torch.empty(n, device=DEV)
        L.check(lib.mtbt_wadd_norm_weights(wd.data_ptr(), n, 1e-4, out.data_ptr(), S()), "wadd norm")
        colsum = dy.sum(dim=(0, 2, 3)).float().to(DEV).contiguous()
        dw = torch.full((n,), 123.0, device=DEV)                                  # overwritten, then accumulated onto
        L.check(lib.mtbt_wadd_norm_weights_backward(wd.data_ptr(), n, 1e-4, colsum.data_ptr(), Cc, dw.data_ptr(), 0, S()), "wadd norm bwd")
        L.check(lib.mtbt_wadd_norm_weights_backward(wd.data_ptr(), n, 1e-4, colsum.data_ptr(), Cc, dw.data_ptr(), 1, S()), "wadd norm bwd")
        torch.cuda.synchronize()
        close(out, wn.detach().float(), 1e-6, "wadd norm")
        close(0.5 * dw, w.grad.float(), 1e-5, "wadd norm bwd")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_projector_backward(dtype):
    lib = L.load()
    torch.manual_seed(1)
    N, nm, hp, wp, So = 2, 32, 12, 12, 48
    protos = torch.randn(N, nm, hp, wp, requires_grad=True)
    conv = torch.nn.Conv2d(nm, 1, 1)
    dseg = torch.randn(N, 1, So, So)
    F.interpolate(conv(protos), size=(So, So), mode="bilinear", align_corners=False).backward(dseg)
    nb = lib.mtbt_projector_backward_workspace_bytes(N, hp, wp, nm)
    ws = torch.empty(nb // 4, device=DEV)
    dp = torch.empty(N, hp, wp, nm, dtype=dtype, device=DEV)
    dw, db = torch.empty(nm, device=DEV), torch.empty(1, device=DEV)
    pd = nhwc(protos.detach())
    w = conv.weight.detach().view(-1).to(DEV).contiguous()
    dsd = dseg.view(N, So, So).to(DEV).contiguous()
    L.check(lib.mtbt_projector_backward(dsd.data_ptr(), pd.data_ptr(), w.data_ptr(), dp.data_ptr(), CODE[dtype], 0, dw.data_ptr(), db.data_ptr(), 0,
                                        N, hp, wp, nm, So, So, ws.data_ptr(), nb, S()), "projector bwd")
    torch.cuda.synchronize()
    close(back(dp), protos.grad, TOL[dtype], "d protos")
    close(dw, conv.weight.grad.view(-1), 2e-5, "dw")
    close(db, conv.bias.grad, 2e-5, "db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gap_fc_backward(dtype):
    lib = L.load()
    torch.manual_seed(2)
    N, Cc, H, W, no = 3, 256, 4, 5, 2
    x = torch.randn(N, Cc, H, W).to(dtype).float().requires_grad_()
    fc = torch.nn.Linear(Cc, no)
    dl = torch.randn(N, no)
    fc(x.mean(dim=(2, 3))).backward(dl)
    prev = (torch.randn(N, H, W, Cc) * 0.01).to(dtype)                            # (small: a bf16 sum keeps 8 bits of the LARGER addend)
    dx = prev.clone().to(DEV)
    dw, db = torch.empty(no, Cc, device=DEV), torch.empty(no, device=DEV)
    pool = torch.empty(N, Cc, device=DEV)
    xd, dld, wd = nhwc(x.detach(), dtype), dl.to(DEV), fc.weight.detach().to(DEV).contiguous()
    L.check(lib.mtbt_gap_fc_backward(xd.data_ptr(), dld.data_ptr(), wd.data_ptr(), dx.data_ptr(), 1, dw.data_ptr(), db.data_ptr(), 0, pool.data_ptr(),
                                     N, H * W, Cc, no, CODE[dtype], S()), "gap_fc bwd")
    torch.cuda.synchronize()
    close(back(dx) - prev.float().permute(0, 3, 1, 2), x.grad, TOL[dtype] * 4, "dx")
    close(dw, fc.weight.grad, TOL[dtype] * 4, "dw")
    close(db, fc.bias.grad, 1e-5, "db")


def test_copy_strided_and_scale_grad():
    lib = L.load()
    torch.manual_seed(4)
    src = torch.randn(2, 5, 7, 66, device=DEV)
    dst = torch.full((2, 5, 7, 32), 9.0, dtype=torch.bfloat16, device=DEV)
    L.check(lib.mtbt_copy_strided(src.data_ptr() + 64 * 4, 0, 5 * 7 * 66, 66, dst.data_ptr(), 1, 5 * 7 * 32, 32, 2, 35, 2, 32, S()), "copy")
    torch.cuda.synchronize()
    assert torch.equal(dst[..., :2], src[..., 64:66].bfloat16()) and dst[..., 2:].abs().max().item() == 0
    # mode 0: y = x + gamma * (W h + b)
    K, Cc, P = 24, 96, 50
    W_ = torch.randn(K, Cc, requires_grad=True); b = torch.randn(K, requires_grad=True); gm = torch.rand(K, requires_grad=True)
    h, dy = torch.randn(P, Cc), torch.randn(P, K)
    (gm * (h @ W_.t() + b)).backward(dy)
    G = (dy.t() @ h).to(DEV).contiguous(); s = dy.sum(0).to(DEV)
    dW, dg, db = (torch.empty(K, Cc, device=DEV), torch.empty(K, device=DEV), torch.empty(K, device=DEV))
    Wd, gd, bd = W_.detach().to(DEV), gm.detach().to(DEV), b.detach().to(DEV)
    L.check(lib.mtbt_scale_grad(0, G.data_ptr(), Wd.data_ptr(), gd.data_ptr(), bd.data_ptr(), s.data_ptr(), dW.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                K, Cc, 0, S()), "scale_grad rows")
    torch.cuda.synchronize()
    close(dW, W_.grad, 1e-5, "dW"); close(dg, gm.grad, 1e-5, "dgamma"); close(db, b.grad, 1e-5, "db")
    # mode 1: y = W (v * x)
    W2 = torch.randn(K, Cc, requires_grad=True); v = torch.rand(Cc, requires_grad=True)
    x = torch.randn(P, Cc)
    ((x * v) @ W2.t()).backward(dy)
    G2 = (dy.t() @ x).to(DEV).contiguous()
    dW2, dv = torch.empty(K, Cc, device=DEV), torch.empty(Cc, device=DEV)
    W2d, vd = W2.detach().to(DEV), v.detach().to(DEV)
    L.check(lib.mtbt_scale_grad(1, G2.data_ptr(), W2d.data_ptr(), vd.data_ptr(), None, None, dW2.data_ptr(), dv.data_ptr(), None, K, Cc, 0, S()),
            "scale_grad cols")
    torch.cuda.synchronize()
    close(dW2, W2.grad, 1e-5, "dW2"); close(dv, v.grad, 1e-5, "dv")


@pytest.mark.parametrize("N,H,W,Cc,K,k,stride", [(2, 16, 16, 64, 64, 3, 1), (1, 10, 12, 32, 96, 1, 1), (2, 12, 8, 96, 192, 2, 2), (1, 5, 3, 16, 8, 3, 1)])
def test_conv_wgrad_fp32(N, H, W, Cc, K, k, stride):
    from multitask_bonetumor_yolo_amd import backward as B
    lib = L.load()
    g = torch.Generator().manual_seed(N + K + k)
    pad = k // 2 if stride == 1 else 0
    x = torch.randn(N, Cc, H, W, generator=g)
    w = (torch.randn(K, Cc, k, k, generator=g) / (Cc * k * k) ** 0.5).requires_grad_()
    y = F.conv2d(x, w, None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    (want,) = torch.autograd.grad(y, w, dy)
    xa, dya = Act.of(nhwc(x)), Act.of(nhwc(dy))
    out = torch.empty(K, k * k * Cc, device=DEV)
    nb = lib.mtbt_conv_wgrad_workspace_bytes(N, H, W, Cc, K, k, k)
    ws = torch.empty(nb // 4, device=DEV)
    L.check(lib.mtbt_conv_wgrad(xa.ptr, dya.ptr, out.data_ptr(), N, H, W, Cc, K, k, k, pad, stride, xa.bs, xa.ld, dya.bs, dya.ld, 0, 0, ws.data_ptr(), nb, S()),
            "wgrad f32")
    torch.cuda.synchronize()
    close(out, want.permute(0, 2, 3, 1).reshape(K, -1), 2e-5, "dW")


def test_conv_wgrad_workspace_is_checked_against_the_launched_slices():
    """a padding > (R-1)/2 makes Ho*Wo exceed H*W; the entry point must size the check from the slices it launches."""
    lib = L.load()
    N, H, W, Cc, K = 1, 38, 38, 8, 8
    nb = lib.mtbt_conv_wgrad_workspace_bytes(N, H, W, Cc, K, 1, 1)
    x = torch.zeros(N, H, W, Cc, dtype=torch.bfloat16, device=DEV)
    dy = torch.zeros(N, H + 2, W + 2, K, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(K, Cc, device=DEV)
    ws = torch.empty(nb // 4, device=DEV)
    rc = lib.mtbt_conv_wgrad(x.data_ptr(), dy.data_ptr(), out.data_ptr(), N, H, W, Cc, K, 1, 1, 1, 1, H * W * Cc, Cc, (H + 2) * (W + 2) * K, K, 1, 0,
                             ws.data_ptr(), nb, S())
    assert rc in (0, -4)          # launched with enough room, or refused -- never a silent overrun
    big = torch.empty(nb, device=DEV)
    assert lib.mtbt_conv_wgrad(x.data_ptr(), dy.data_ptr(), out.data_ptr(), N, H, W, Cc, K, 1, 1, 1, 1, H * W * Cc, Cc, (H + 2) * (W + 2) * K, K, 1, 0,
                               big.data_ptr(), nb * 4, S()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_train_and_wgrad(dtype):
    lib = L.load()
    torch.manual_seed(6)
    N, Hh, Ww, K = 2, 64, 128, 96
    img = torch.rand(N, 3, Hh, Ww)
    conv = torch.nn.Conv2d(3, K, 4, 4)
    ln_w, ln_b = torch.rand(K) + 0.5, torch.randn(K) * 0.1
    raw_ref = conv(img)
    y_ref = F.layer_norm(raw_ref.permute(0, 2, 3, 1), (K,), ln_w, ln_b, 1e-6)
    d = torch.randn(N, K, Hh // 4, Ww // 4).to(dtype).float()
    (gw,) = torch.autograd.grad(raw_ref, conv.weight, d)
    xd = img.to(DEV)
    wd, bd = conv.weight.detach().reshape(K, 48).to(DEV).contiguous(), conv.bias.detach().to(DEV)
    y = torch.empty(N, Hh // 4, Ww // 4, K, dtype=dtype, device=DEV)
    raw = torch.empty_like(y)
    lwd, lbd = ln_w.to(DEV), ln_b.to(DEV)
    L.check(lib.mtbt_stem_conv4x4_ln_train(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), 1e-6, y.data_ptr(),
                                           raw.data_ptr(), N, Hh, Ww, K, CODE[dtype], S()), "stem train")
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    close(back(raw), raw_ref.detach(), tol, "raw")
    close(y.float().cpu(), y_ref.detach(), tol, "y")
    nb = lib.mtbt_stem_wgrad_workspace_bytes(K)
    ws = torch.empty(nb // 4, device=DEV)
    dW = torch.empty(K, 48, device=DEV)
    dd = nhwc(d, dtype)
    L.check(lib.mtbt_stem_wgrad(xd.data_ptr(), dd.data_ptr(), dW.data_ptr(), N, Hh, Ww, K, CODE[dtype], 0, ws.data_ptr(), nb, S()), "stem wgrad")
    torch.cuda.synchronize()
    close(dW, gw.reshape(K, 48), 2e-5, "stem dW")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_epilogue_second_output_and_derivative_mode(dtype):
    """y2 = the pre-activation next to y = GELU(.) (training fc1), and the backward epilogue y = conv * GELU'(res)."""
    torch.manual_seed(8)
    N, H, W, Cc, K = 2, 8, 8, 64, 128
    x = torch.randn(N, Cc, H, W).to(dtype).float()
    w = (torch.randn(K, Cc) / 8).to(dtype).float()
    b = torch.randn(K) * 0.1
    z_ref = F.conv2d(x, w.view(K, Cc, 1, 1), b)
    p = Plan(torch.device(DEV))
    xa = Act.of(nhwc(x, dtype))
    y, y2 = Act.of(torch.empty(N, H, W, K, dtype=dtype, device=DEV)), Act.of(torch.empty(N, H, W, K, dtype=dtype, device=DEV))
    a = p.conv(xa, w.to(DEV, dtype), y, shift=b.to(DEV), act=L.ACT_GELU)
    a.y2 = y2.ptr
    # derivative mode: dz = (dy W^T) * GELU'(z)
    dy = torch.randn(N, Cc, H, W).to(dtype).float()
    wt = torch.randn(K, Cc) / 8
    wt = wt.to(dtype).float()
    zz = torch.randn(N, K, H, W).to(dtype).float().requires_grad_()
    lin = F.conv2d(dy, wt.view(K, Cc, 1, 1))
    (gelu_grad,) = torch.autograd.grad(F.gelu(zz), zz, torch.ones_like(zz))
    dz = Act.of(torch.empty(N, H, W, K, dtype=dtype, device=DEV))
    p.conv(Act.of(nhwc(dy, dtype)), wt.to(DEV, dtype), dz, act=L.ACT_DGELU, res=Act.of(nhwc(zz.detach(), dtype)))
    p.run(stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    close(back(y2.buf), z_ref, tol, "pre-activation")
    close(back(y.buf), F.gelu(z_ref), tol, "activated")
    close(back(dz.buf), lin * gelu_grad, tol, "derivative epilogue")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_train_outputs(dtype):
    lib = L.load()
    torch.manual_seed(9)
    N, Cc, H, W = 2, 96, 12, 20
    x = torch.randn(N, Cc, H, W).to(dtype).float()
    w = (torch.randn(Cc, 1, 7, 7) / 7).to(dtype).float()
    bias, lw, lb = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.1
    raw_ref = F.conv2d(x, w, bias, 1, 3, groups=Cc)
    t_ref = F.layer_norm(raw_ref.permute(0, 2, 3, 1), (Cc,), lw, lb, 1e-6)
    taps = w.view(Cc, 49).t().contiguous().to(DEV, dtype)
    xd = nhwc(x, dtype)
    t, raw = torch.empty_like(xd), torch.empty_like(xd)
    bd, lwd, lbd = bias.to(DEV), lw.to(DEV), lb.to(DEV)
    L.check(lib.mtbt_dwconv_nhwc_train(xd.data_ptr(), taps.data_ptr(), bd.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), 1e-6, None, None,
                                       0, t.data_ptr(), raw.data_ptr(), None, N, H, W, Cc, 7, CODE[dtype], S()), "dwconv train LN")
    torch.cuda.synchronize()
    tol = 5e-5 if dtype == torch.float32 else 3e-2
    close(back(raw), raw_ref, tol, "raw")
    close(t.float().cpu(), t_ref, tol, "ln out")
    # scale/shift form with an aliased residual: y = conv(x) + y
    acc0 = (torch.randn(N, H, W, Cc) * 0.1).to(dtype)
    acc = acc0.clone().to(DEV)
    one, zero = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    L.check(lib.mtbt_dwconv_nhwc_train(xd.data_ptr(), taps.data_ptr(), None, None, None, 0.0, one.data_ptr(), zero.data_ptr(), 0, acc.data_ptr(), None,
                                       acc.data_ptr(), N, H, W, Cc, 7, CODE[dtype], S()), "dwconv train res")
    torch.cuda.synchronize()
    close(back(acc), F.conv2d(x, w, None, 1, 3, groups=Cc) + acc0.float().permute(0, 3, 1, 2), tol, "conv + aliased residual")


def test_layernorm_backward_accumulates():
    lib = L.load()
    torch.manual_seed(10)
    P, Cc = 77, 192
    x = torch.randn(P, Cc, requires_grad=True)
    g = torch.rand(Cc) + 0.5
    dy = torch.randn(P, Cc)
    F.layer_norm(x, (Cc,), g, None, 1e-6).backward(dy)
    prev = torch.randn(P, Cc)
    dx = prev.clone().to(DEV)
    xhat = torch.empty(P, Cc, device=DEV)
    xd, dyd, gd = x.detach().to(DEV), dy.to(DEV), g.to(DEV)
    L.check(lib.mtbt_layernorm_backward_nhwc(xd.data_ptr(), dyd.data_ptr(), gd.data_ptr(), 1e-6, dx.data_ptr(), xhat.data_ptr(),
                                             P, Cc, 0, 1, S()), "ln bwd acc")
    torch.cuda.synchronize()
    close(dx.cpu() - prev, x.grad, 2e-5, "dx (accumulated)")


def test_sgd_sumsq_clip_and_scaled_adamw():
    lib = L.load()
    torch.manual_seed(11)
    n = 10007
    p0, grads = torch.randn(n), [torch.randn(n) * 3 for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9, weight_decay=5e-4, nesterov=True)
    p = p0.clone().to(DEV)
    buf = torch.zeros(n, device=DEV)
    coef, sq, norm = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    ws = torch.empty(lib.mtbt_sumsq_workspace_bytes() // 4, device=DEV)
    for step, g in enumerate(grads, 1):
        ref.grad = g.clone()
        total = torch.nn.utils.clip_grad_norm_([ref], 10.0)
        opt.step()
        gd = g.to(DEV)
        L.check(lib.mtbt_sumsq(gd.data_ptr(), n // 2, sq.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4, S()), "sumsq a")       # two "buckets"
        L.check(lib.mtbt_sumsq(gd[n // 2:].data_ptr(), n - n // 2, sq.data_ptr(), 1, ws.data_ptr(), ws.numel() * 4, S()), "sumsq b")
        L.check(lib.mtbt_clip_coef(sq.data_ptr(), 10.0, coef.data_ptr(), norm.data_ptr(), S()), "clip")
        L.check(lib.mtbt_sgd_step(p.data_ptr(), gd.data_ptr(), buf.data_ptr(), n, 0.01, 0.9, 0.0, 5e-4, 1, step, coef.data_ptr(), S()), "sgd")
        torch.cuda.synchronize()
        assert abs(norm.item() - total.item()) <= 1e-4 * total.item()
        close(p, ref.detach(), 2e-6, f"sgd step {step}")
    # AdamW with the clip coefficient
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=5e-4)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step, g in enumerate(grads, 1):
        ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref], 10.0)
        opt.step()
        gd = g.to(DEV)
        L.check(lib.mtbt_sumsq(gd.data_ptr(), n, sq.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4, S()), "sumsq")
        L.check(lib.mtbt_clip_coef(sq.data_ptr(), 10.0, coef.data_ptr(), None, S()), "clip")
        L.check(lib.mtbt_adamw_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 5e-4, step, coef.data_ptr(), S()), "adamw")
        torch.cuda.synchronize()
        close(p, ref.detach(), 5e-6, f"adamw step {step}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgrad_with_bias_gradient(dtype):
    """mtbt_conv_wgrad_bias: dW and sum_p dy (the bias gradient) from one launch -- ragged channel tiles, several slices."""
    lib = L.load()
    g = torch.Generator().manual_seed(31)
    N, H, W, Cc, K, k = 3, 24, 20, 96, 200, 3
    x = torch.randn(N, Cc, H, W, generator=g).to(dtype).float()
    dy = torch.randn(N, K, H, W, generator=g).to(dtype).float()
    w = torch.zeros(K, Cc, k, k, requires_grad=True)
    b = torch.zeros(K, requires_grad=True)
    F.conv2d(x, w, b, 1, 1).backward(dy)
    xa, dya = Act.of(nhwc(x, dtype)), Act.of(nhwc(dy, dtype))
    out, db = torch.empty(K, k * k * Cc, device=DEV), torch.empty(K, device=DEV)
    nb = lib.mtbt_conv_wgrad_workspace_bytes(N, H, W, Cc, K, k, k)
    ws = torch.empty(nb // 4, device=DEV)
    L.check(lib.mtbt_conv_wgrad_bias(xa.ptr, dya.ptr, out.data_ptr(), db.data_ptr(), N, H, W, Cc, K, k, k, 1, 1, xa.bs, xa.ld, dya.bs, dya.ld, CODE[dtype], 0,
                                     ws.data_ptr(), nb, S()), "wgrad+bias")
    torch.cuda.synchronize()
    close(out, w.grad.permute(0, 2, 3, 1).reshape(K, -1), 3e-4, "dW")
    close(db, b.grad, 3e-4, "dbias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cc,k", [(96, 7), (256, 3), (384, 7)])
def test_dwconv_wgrad_with_bias_gradient(dtype, Cc, k):
    lib = L.load()
    g = torch.Generator().manual_seed(Cc + k)
    N, H, W = 2, 19, 13
    x = torch.randn(N, Cc, H, W, generator=g).to(dtype).float()
    dy = torch.randn(N, Cc, H, W, generator=g).to(dtype).float()
    w = torch.zeros(Cc, 1, k, k, requires_grad=True)
    b = torch.zeros(Cc, requires_grad=True)
    F.conv2d(x, w, b, 1, k // 2, groups=Cc).backward(dy)
    xd, dyd = nhwc(x, dtype), nhwc(dy, dtype)
    dw, db = torch.empty(k * k, Cc, device=DEV), torch.empty(Cc, device=DEV)
    nb = lib.mtbt_dwconv_wgrad_workspace_bytes(N, H, W, Cc, k)
    ws = torch.empty(nb // 4, device=DEV)
    L.check(lib.mtbt_dwconv_wgrad_bias(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, Cc, k, CODE[dtype], 0, ws.data_ptr(), nb, S()), "dw wgrad")
    torch.cuda.synchronize()
    close(dw, w.grad.view(Cc, k * k).t(), 3e-5, "dW")
    close(db, b.grad, 3e-5, "dbias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cc", [96, 384, 768])
def test_layernorm_backward_with_parameter_gradients(dtype, Cc):
    lib = L.load()
    torch.manual_seed(Cc)
    P = 1000                                         # not a multiple of a wave's pixel run: dead pixels inside live waves
    x = (torch.randn(P, Cc) * 1.3 + 0.2).to(dtype).float().requires_grad_()
    gam = (torch.rand(Cc) + 0.5).requires_grad_()
    bet = torch.zeros(Cc, requires_grad=True)
    dy = torch.randn(P, Cc).to(dtype).float()
    F.layer_norm(x, (Cc,), gam, bet, 1e-6).backward(dy)
    prev = (torch.randn(P, Cc) * 0.01).to(dtype)
    dx = prev.clone().to(DEV)
    dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    nb = lib.mtbt_layernorm_backward_params_workspace_bytes(P, Cc)
    ws = torch.empty(nb // 4, device=DEV)
    xd, dyd, gd = x.detach().to(DEV, dtype), dy.to(DEV, dtype), gam.detach().to(DEV)
    L.check(lib.mtbt_layernorm_backward_params_nhwc(xd.data_ptr(), dyd.data_ptr(), gd.data_ptr(), 1e-6, dx.data_ptr(), P, Cc, CODE[dtype], 1, dg.data_ptr(),
                                                    db.data_ptr(), 0, ws.data_ptr(), nb, S()), "ln bwd params")
    torch.cuda.synchronize()
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    close(dx.float().cpu() - prev.float(), x.grad, tol, "dx")
    close(dg, gam.grad, tol, "dgamma")
    close(db, bet.grad, tol, "dbeta")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    # (N, H, W, C, K, k, act, residual-as-GELU'-input)
    (2, 9, 7, 64, 128, 1, 0, False),      # implicit GEMM 1x1, 128-channel tile, ragged pixel tile
    (2, 16, 16, 64, 64, 3, 0, False),     # direct 3x3 (row-reuse kernel for 64-channel tiles in 16-bit modes)
    (1, 32, 32, 128, 256, 3, 0, False),   # direct 3x3, two channel tiles
    (2, 10, 10, 64, 192, 3, 0, False),    # 3x3 on the implicit-GEMM kernel (map not 16-aligned); K = 192 must avoid the 96-channel tile
    (3, 8, 8, 96, 384, 1, 7, True),       # fc2-dgrad * GELU'(pre-activation): d fc1.bias from the same launch
    (1, 5, 5, 64, 32, 1, 0, False),       # 32-channel tile
])
def test_conv_column_sums(dtype, cfg):
    """mtbt_conv_args.colsum: per-channel sums (and sums of squares about a shift) of the STORED conv output, from the epilogue --
    against the same sums taken over the output tensor the call wrote; then BatchNorm statistics from them against torch."""
    N, H, W, Cc, K, k, act, deriv = cfg
    lib = L.load()
    torch.manual_seed(21)
    x = torch.randn(N, Cc, H, W).to(dtype).float()
    w = (torch.randn(K, Cc, k, k) / (Cc * k * k) ** 0.5).to(dtype).float()
    b = torch.randn(K) * 0.3 + 0.5
    shift = torch.randn(K) * 0.1 + 0.4
    p = Plan(torch.device(DEV))
    xa = Act.of(nhwc(x, dtype))
    y = Act.of(torch.empty(N, H, W, K, dtype=dtype, device=DEV))
    wp = w.permute(0, 2, 3, 1).reshape(K, -1).contiguous().to(DEV, dtype)
    res = Act.of(nhwc(torch.randn(N, K, H, W), dtype)) if deriv else None
    a = p.conv(xa, wp, y, R=k, S=k, pad=k // 2, shift=b.to(DEV), act=act, res=res)
    sums = torch.full((2 * K,), 7.0, device=DEV)
    sh = shift.to(DEV)
    nb = lib.mtbt_conv_colsum_workspace_bytes(N * H * W, K, 1)
    ws = torch.empty(nb // 4 + 4, device=DEV)
    a.colsum, a.colsum_shift, a.colsum_sq, a.colsum_accumulate = sums.data_ptr(), sh.data_ptr(), 1, 0
    a.colsum_ws, a.colsum_ws_bytes = ws.data_ptr(), nb
    p.run()
    torch.cuda.synchronize()
    stored = y.buf.float().cpu().reshape(-1, K).double()
    want1 = (stored - shift.double()).sum(0)
    want2 = ((stored - shift.double()) ** 2).sum(0)
    got = sums.cpu().double()
    tol = 2e-5 if dtype == torch.float32 else 2e-4        # (fp32 accumulation order only: the summed values are the stored ones)
    assert (got[:K] - want1).abs().max() <= tol * (stored - shift.double()).abs().sum(0).max() + 1e-6
    assert (got[K:] - want2).abs().max() <= tol * want2.max() + 1e-6
    # accumulate form, sums only
    a.colsum_sq, a.colsum_accumulate = 0, 1
    p.run()
    torch.cuda.synchronize()
    assert (sums.cpu().double()[:K] - 2 * want1).abs().max() <= 2 * tol * (stored - shift.double()).abs().sum(0).max() + 1e-6
    if act != 0:
        return
    # BatchNorm forward from the sums == BatchNorm forward with its own statistics pass
    a.colsum_sq, a.colsum_accumulate = 1, 0
    p.run()
    bn = torch.nn.BatchNorm2d(K, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.copy_(shift); bn.running_var.uniform_(0.5, 1.5)
    rm, rv = bn.running_mean.clone().to(DEV), bn.running_var.clone().to(DEV)
    g, be = bn.weight.detach().to(DEV), bn.bias.detach().to(DEV)
    out = torch.empty_like(y.buf)
    stats = torch.empty(2 * K, device=DEV)
    pixels = N * H * W
    L.check(lib.mtbt_bn_forward_sums_nhwc(y.ptr, out.data_ptr(), K, g.data_ptr(), be.data_ptr(), rm.data_ptr(), rv.data_ptr(), C.c_float(0.03), C.c_float(1e-3),
                                          L.ACT_SILU, pixels, K, CODE[dtype], sums.data_ptr(), rm.data_ptr(), stats.data_ptr(), S()), "bn from sums")
    torch.cuda.synchronize()
    ref_in = y.buf.float().cpu().permute(0, 3, 1, 2)
    bn.train()
    ref = F.silu(bn(ref_in))
    close(back(out), ref.detach(), 2e-5 if dtype == torch.float32 else 2e-2, "bn(y) from column sums")
    close(stats[:K], ref_in.mean((0, 2, 3)), 2e-5, "batch mean")
    close(stats[K:], ref_in.var((0, 2, 3), unbiased=False), 1e-4, "batch variance")
    close(rm, bn.running_mean, 2e-5, "running mean"); close(rv, bn.running_var, 1e-4, "running var")
    # the consumer form: only the partial rows are written (layout from mtbt_conv_colsum_layout), BatchNorm reduces them itself
    rows, pitch = C.c_int64(0), C.c_int32(0)
    a.colsum = None
    L.check(lib.mtbt_conv_colsum_layout(C.byref(a), C.byref(rows), C.byref(pitch)), "layout")
    assert pitch.value == 2 * K and 0 < rows.value * pitch.value * 4 <= nb
    part = torch.full((rows.value * pitch.value,), float("nan"), device=DEV)
    a.colsum_ws, a.colsum_ws_bytes = part.data_ptr(), part.numel() * 4
    p.run()
    torch.cuda.synchronize()
    assert torch.isfinite(part).all(), "every partial row the layout announces is written"
    rm2, rv2 = shift.clone().to(DEV), bn.running_var.clone().to(DEV)
    out2, stats2 = torch.empty_like(out), torch.empty_like(stats)
    L.check(lib.mtbt_bn_forward_partials_nhwc(y.ptr, out2.data_ptr(), K, g.data_ptr(), be.data_ptr(), rm2.data_ptr(), rv2.data_ptr(), C.c_float(0.03),
                                              C.c_float(1e-3), L.ACT_SILU, pixels, K, CODE[dtype], part.data_ptr(), rows.value, pitch.value, rm2.data_ptr(),
                                              stats2.data_ptr(), S()), "bn from partial rows")
    torch.cuda.synchronize()
    close(stats2, stats.cpu(), 1e-5, "statistics from the partial rows == from the finished sums")
    assert torch.equal(out2, out) or (out2.float() - out.float()).abs().max().item() <= 2e-2
    a.colsum_ws_bytes = part.numel() * 4 - 4
    assert lib.mtbt_conv2d_nhwc(C.byref(a), S()) == -4, "a partial buffer smaller than the layout is refused (MTBT_EWORKSPACE)"
