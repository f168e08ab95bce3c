"""GPU parity of each HIP kernel against the CPU oracle / plain torch fp32 on the same seeded inputs.
Everything goes through the C ABI (ctypes -> libmtbt_hip.so).  fp32 kernels: <= 1e-3 absolute
(north_star tolerance; observed ~1e-5); bf16 kernels: relative-to-max tolerance stated per test."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from multitask_bonetumor_yolo_amd import _lib as L
    from multitask_bonetumor_yolo_amd.engine import Act, Plan

DEV = "cuda:0"
TOL32 = 1e-3


def nhwc(t):  # [N,C,H,W] cpu -> dense NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def back(t):  # NHWC cuda -> NCHW cpu fp32
    return t.float().cpu().permute(0, 3, 1, 2)


def run(plan):
    plan.run()
    torch.cuda.synchronize()


ACTS = {0: lambda v: v, 1: F.silu, 2: F.elu, 3: F.gelu, 4: F.gelu}  # 4 = polynomial GELU (|err| <= 2.3e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    # (N, H, W, C, K, k, stride, pad, act, residual, tile_hint)
    (2, 12, 12, 64, 64, 3, 1, 1, 1, False, 0),
    (1, 20, 20, 128, 128, 3, 1, 1, 1, False, 0),
    (2, 9, 7, 96, 384, 1, 1, 0, 3, False, 0),        # ConvNeXt fc1 (+GELU), ragged pixel tile, TC=128
    (2, 9, 7, 384, 96, 1, 1, 0, 0, True, 0),         # ConvNeXt fc2 + layer-scale residual, TC=96
    (2, 9, 7, 96, 384, 1, 1, 0, 4, False, 0),        # fc1 with the polynomial GELU of the bf16 mode
    (1, 16, 16, 96, 192, 2, 2, 0, 0, False, 0),      # downsample 2x2/2
    (1, 8, 8, 256, 2, 1, 1, 0, 0, False, 0),         # cls conv, K=2 (scalar epilogue)
    (1, 10, 10, 64, 32, 1, 1, 0, 1, False, 0),
    (3, 16, 16, 256, 256, 3, 1, 1, 1, False, (128 << 16) | 128),
    (3, 16, 16, 256, 256, 3, 1, 1, 2, False, (128 << 16) | 64),
    (1, 16, 16, 192, 192, 3, 1, 1, 1, False, (96 << 16) | 128),
    (1, 16, 16, 192, 192, 3, 1, 1, 1, False, (96 << 16) | 64),
    (1, 16, 16, 64, 64, 3, 1, 1, 1, False, (64 << 16) | 128),
    (1, 16, 16, 64, 64, 3, 1, 1, 1, False, (64 << 16) | 64),
    (1, 16, 16, 64, 48, 3, 1, 1, 1, False, (32 << 16) | 128),
    (1, 16, 16, 64, 48, 3, 1, 1, 1, False, (32 << 16) | 64),
    (1, 1, 1, 32, 16, 1, 1, 0, 0, False, 0),         # single pixel
    (2, 32, 48, 128, 256, 3, 1, 1, 1, True, 0),      # direct 3x3 (LDS halo tile): 2 channel tiles, residual, 2 images
    (1, 16, 32, 64, 64, 3, 1, 1, 2, False, 0),       # direct 3x3, TC=64, single channel chunk
    (1, 48, 16, 192, 96, 3, 1, 1, 1, False, 0),      # direct 3x3, K=96 (ragged channel tile), 3 chunks
    (2, 32, 48, 128, 256, 3, 1, 1, 1, True, 1 << 26),  # same shape forced onto the implicit-GEMM kernel
    (2, 32, 48, 128, 256, 3, 1, 1, 1, True, 1 << 25),  # ... and onto the row-reuse direct kernel (TC=128, 4 slabs)
    (1, 16, 32, 64, 64, 3, 1, 1, 2, False, 1 << 25),   # row-reuse, TC=64, 2 slabs
    (1, 48, 16, 96, 96, 3, 1, 1, 1, False, 1 << 25),   # row-reuse, ragged channel tile, odd slab count (3)
])
def test_conv_igemm(dtype, cfg):
    N, H, W, Cin, K, k, st, pad, act, use_res, hint = cfg
    if dtype == torch.float32 and Cin % 16:
        pytest.skip("C % 16")
    if dtype == torch.bfloat16 and Cin % 32:
        pytest.skip("C % 32")
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(K, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(K, generator=g) + 0.5
    shift = torch.randn(K, generator=g) * 0.1
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    res = torch.randn(N, K, Ho, Wo, generator=g) if use_res else None
    if dtype == torch.bfloat16:  # the kernel sees bf16-rounded operands; give the reference the same
        x, w = x.bfloat16().float(), w.bfloat16().float()
        res = res.bfloat16().float() if use_res else None
    ref = ACTS[act](F.conv2d(x, w, None, st, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    if use_res:
        ref = ref + res
    p = Plan(torch.device(DEV))
    xa = Act.of(nhwc(x).to(dtype))
    wp = w.permute(0, 2, 3, 1).reshape(K, -1).contiguous().to(DEV, dtype)
    ya = Act.of(torch.zeros(N, Ho, Wo, K, dtype=dtype, device=DEV))
    ra = Act.of(nhwc(res).to(dtype)) if use_res else None
    p.conv(xa, wp, ya, R=k, S=k, stride=st, pad=pad, scale=scale.to(DEV), shift=shift.to(DEV), act=act, res=ra, tile_hint=hint)
    run(p)
    out = back(ya.buf)
    if dtype == torch.float32:
        assert (out - ref).abs().max().item() < TOL32
    else:  # bf16 output rounding: 2^-8 relative
        assert ((out - ref).abs() / (ref.abs() + 1.0)).max().item() < 1.5e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    (2, 32, 48, 128, 256, 3, 1, 1, 1, True, 0),      # two channel tiles, residual, 2 images
    (1, 16, 32, 64, 64, 3, 1, 1, 2, False, 0),       # TC = 64, single channel chunk
    (1, 48, 16, 192, 96, 3, 1, 1, 1, False, 0),      # K = 96 (ragged channel tile), 3 chunks
])
def test_conv3x3_direct_first_formulation(dtype, cfg, monkeypatch):
    """The row-reuse kernel is the default direct 3x3 kernel; policy bit 4 selects the first formulation (conv3x3_direct_kernel): same shapes."""
    monkeypatch.setenv("MTBT_CONV_POLICY", str(7 | 16))
    test_conv_igemm(dtype, cfg)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_channel_slices_and_f32_out(dtype):
    """Concat-free C2f addressing: read a channel slice, write a channel slice of a wider buffer; fp32 output."""
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 10, 10
    xin = torch.randn(N, 128, H, W, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    if dtype == torch.bfloat16:
        xin, w = xin.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(xin[:, 64:128], w, None, 1, 1)
    p = Plan(torch.device(DEV))
    xa = Act.of(nhwc(xin).to(dtype)).slice(64, 64)
    ybuf = torch.full((N, H, W, 66), 7.0, dtype=torch.float32, device=DEV)
    ya = Act.of(ybuf).slice(2, 64)   # misaligned for vector stores -> scalar epilogue
    p.conv(xa, w.permute(0, 2, 3, 1).reshape(64, -1).contiguous().to(DEV, dtype), ya, R=3, S=3, pad=1)
    run(p)
    out = back(ybuf)
    assert torch.all(out[:, :2] == 7.0)
    tol = TOL32 if dtype == torch.float32 else 2e-2
    assert (out[:, 2:66] - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [
    # (N, H, W, C, K, pitch of the fp32 map, channel offset in it, per-image rows of the map (> H*W: a pyramid level inside a larger buffer))
    (2, 10, 10, 64, 64, 68, 0, 100),      # Detect cv2[i][2]: 4 * reg_max box channels of the [N,h,w,68] map
    (2, 10, 10, 64, 2, 68, 64, 100),      # Detect cv3[i][2]: nc = 2 class channels behind them (scalar stores)
    (3, 7, 9, 64, 32, 32, 0, 84),         # Segment cv4[i][2]: mask coefficients of one level inside [N, A, 32]; ragged last wave
    (1, 5, 5, 128, 48, 48, 0, 25),        # four reduction steps, three channel fragments
    (2, 6, 6, 256, 2, 68, 64, 36),        # the class conv behind a 256-channel branch
    (1, 1, 1, 32, 16, 16, 0, 1),          # one pixel, one step
])
def test_head_output_conv_streams(dtype, cfg):
    """main_model.py:300-340 (ultralytics Detect.cv2 / cv3 / Segment.cv4 `[i][2]`: Conv2d(c, k, 1) with bias): mtbt_conv2d_nhwc hands these
    to pw_stream_kernel (no LDS).  Equal BIT FOR BIT to the implicit-GEMM kernel on the same call (a tile hint keeps it there), within
    bf16 / fp16 operand rounding of torch's conv, and nothing outside the slice is written."""
    N, H, W, Cin, K, pitch, c0, rows = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = (torch.randn(N, Cin, H, W, generator=g)).to(dtype).float()
    w = (torch.randn(K, Cin, 1, 1, generator=g) / Cin ** 0.5).to(dtype).float()
    b = torch.randn(K, generator=g)
    ref = F.conv2d(x, w, b)
    outs = []
    for hint in (0, (64 << 16) | 64):
        ybuf = torch.full((N, rows, pitch), 7.0, dtype=torch.float32, device=DEV)
        ya = Act(ybuf, c0, N, H, W, K, pitch, rows * pitch)
        p = Plan(torch.device(DEV))
        p.conv_policy = 0x100 | 7 | 128      # policy bit 7: the streaming kernel takes every shape it can run (by default only the narrow ones: round 3)
        p.conv(Act.of(nhwc(x).to(dtype)), w.reshape(K, Cin).contiguous().to(DEV, dtype), ya, R=1, S=1, shift=b.to(DEV), tile_hint=hint)
        run(p)
        outs.append(ybuf.clone())
    assert torch.equal(outs[0], outs[1])
    y = outs[0][:, :H * W, c0:c0 + K].reshape(N, H, W, K).permute(0, 3, 1, 2).cpu()
    assert (y - ref).abs().max().item() < 1e-3          # fp32 accumulation of exactly representable products
    mask = torch.ones_like(outs[0], dtype=torch.bool)
    mask[:, :H * W, c0:c0 + K] = False
    assert torch.all(outs[0][mask] == 7.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose_2x2(dtype):
    g = torch.Generator().manual_seed(4)
    N, H, W, Cin, Cout = 2, 6, 5, 64, 32
    x = torch.randn(N, Cin, H, W, generator=g)
    wt = torch.randn(Cin, Cout, 2, 2, generator=g) / 8
    b = torch.randn(Cout, generator=g)
    if dtype == torch.bfloat16:
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
    ref = F.conv_transpose2d(x, wt, b, 2)
    p = Plan(torch.device(DEV))
    ya = Act.of(torch.zeros(N, 2 * H, 2 * W, Cout, dtype=dtype, device=DEV))
    p.conv(Act.of(nhwc(x).to(dtype)), wt.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).contiguous().to(DEV, dtype), ya,
           shift=b.repeat(4).to(DEV), out_mode=L.OUT_CONVT2X2)
    run(p)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert (back(ya.buf) - ref).abs().max().item() < tol


def test_conv_rejects_bad_args():
    lib = L.load()
    a = L.ConvArgs()
    assert lib.mtbt_conv2d_nhwc(C.byref(a), None) == -1
    assert lib.mtbt_conv2d_nhwc(None, None) == -1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem(dtype):
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 32, 64, generator=g)
    w = torch.randn(96, 3, 4, 4, generator=g) / 7
    b = torch.randn(96, generator=g) * 0.1
    lw, lb = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g) * 0.1
    y = F.conv2d(x, w, b, 4)
    ref = F.layer_norm(y.permute(0, 2, 3, 1), (96,), lw, lb, 1e-6).permute(0, 3, 1, 2)
    p = Plan(torch.device(DEV))
    ya = Act.of(torch.zeros(2, 8, 16, 96, dtype=dtype, device=DEV))
    p.stem(x.to(DEV), w.reshape(96, 48).contiguous().to(DEV), b.to(DEV), lw.to(DEV), lb.to(DEV), 1e-6, ya)
    run(p)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert (back(ya.buf) - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 96, 9, 11), (1, 192, 8, 8), (1, 768, 5, 6), (1, 384, 4, 4), (1, 96, 20, 37), (2, 384, 17, 16),
                                   (1, 384, 40, 40), (2, 384, 12, 20), (1, 768, 20, 24), (1, 192, 16, 32)])   # whole half-width / full tiles (no bounds branches)
def test_dwconv7_layernorm(dtype, shape):
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(6)
    x = torch.randn(N, Cc, H, W, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    w = torch.randn(Cc, 1, 7, 7, generator=g) / 7
    if dtype == torch.bfloat16:
        w = w.bfloat16().float()
    b = torch.randn(Cc, generator=g) * 0.1
    lw, lb = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    y = F.conv2d(x, w, b, 1, 3, groups=Cc)
    ref = F.layer_norm(y.permute(0, 2, 3, 1), (Cc,), lw, lb, 1e-6).permute(0, 3, 1, 2)
    p = Plan(torch.device(DEV))
    ya = Act.of(torch.zeros(N, H, W, Cc, dtype=dtype, device=DEV))
    p.dwconv(Act.of(nhwc(x).to(dtype)), w.reshape(Cc, 49).t().contiguous().to(DEV, dtype), ya, 7, bias=b.to(DEV), lnw=lw.to(DEV),
             lnb=lb.to(DEV), eps=1e-6)
    run(p)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert (back(ya.buf) - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv3_affine_silu(dtype):
    g = torch.Generator().manual_seed(7)
    N, Cc, H, W = 2, 256, 7, 10
    x = torch.randn(N, Cc, H, W, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    w = torch.randn(Cc, 1, 3, 3, generator=g) / 3
    if dtype == torch.bfloat16:
        w = w.bfloat16().float()
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    ref = F.silu(F.conv2d(x, w, None, 1, 1, groups=Cc) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    p = Plan(torch.device(DEV))
    ya = Act.of(torch.zeros(N, H, W, Cc, dtype=dtype, device=DEV))
    p.dwconv(Act.of(nhwc(x).to(dtype)), w.reshape(Cc, 9).t().contiguous().to(DEV, dtype), ya, 3, scale=sc.to(DEV), shift=sh.to(DEV), act=1)
    run(p)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert (back(ya.buf) - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 384, 12, 20), (1, 768, 8, 24), (1, 192, 16, 16), (2, 96, 9, 11), (1, 256, 40, 40)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_dwconv7_affine_wide(dtype, shape, act):
    """The scale / shift form of the 7x7 kernel (depthwise dgrad): wider tensors run the one-chunk kernel with the 128-channel chunks as
    grid rows; half-width tiles when the width is not a whole number of 16-pixel tiles; activation constants and the run-time form."""
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, Cc, H, W, generator=g)
    w = torch.randn(Cc, 1, 7, 7, generator=g) / 7
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    ref = ACTS[act](F.conv2d(x, w, None, 1, 3, groups=Cc) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    p = Plan(torch.device(DEV))
    ya = Act.of(torch.zeros(N, H, W, Cc, dtype=dtype, device=DEV))
    p.dwconv(Act.of(nhwc(x).to(dtype)), w.reshape(Cc, 49).t().contiguous().to(DEV, dtype), ya, 7, scale=sc.to(DEV), shift=sh.to(DEV), act=act)
    run(p)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert (back(ya.buf) - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cc", [96, 192, 768])
def test_layernorm(dtype, Cc):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, Cc, 5, 7, generator=g) * 2 + 0.5
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    lw, lb = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (Cc,), lw, lb, 1e-6).permute(0, 3, 1, 2)
    p = Plan(torch.device(DEV))
    ya = Act.of(torch.zeros(2, 5, 7, Cc, dtype=dtype, device=DEV))
    p.layernorm(Act.of(nhwc(x).to(dtype)), lw.to(DEV), lb.to(DEV), 1e-6, ya)
    run(p)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert (back(ya.buf) - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bifpn_fuse_modes(dtype):
    g = torch.Generator().manual_seed(9)
    Cc = 64
    mid = torch.randn(2, Cc, 8, 6, generator=g)
    small = torch.randn(2, Cc, 4, 3, generator=g)
    big = torch.randn(2, Cc, 16, 12, generator=g)
    if dtype == torch.bfloat16:
        mid, small, big = (t.bfloat16().float() for t in (mid, small, big))
    w = [0.3, 0.45, 0.25]
    cases = [
        ([mid, small], [0, 1], w[0] * mid + w[1] * F.interpolate(small, scale_factor=2, mode="bilinear")),
        ([mid, mid, big], [0, 0, 2], w[0] * mid + w[1] * mid + w[2] * F.interpolate(big, scale_factor=0.5, mode="bilinear")),
        ([mid, small], [0, 3], w[0] * mid + w[1] * F.interpolate(small, scale_factor=2, mode="nearest")),
        ([mid, big], [0, 4], w[0] * mid + w[1] * F.max_pool2d(big, 2)),
    ]
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    for ins, modes, ref in cases:
        p = Plan(torch.device(DEV))
        ya = Act.of(torch.zeros(2, 8, 6, Cc, dtype=dtype, device=DEV))
        p.fuse([Act.of(nhwc(t).to(dtype)) for t in ins], w[:len(ins)], modes, ya)
        run(p)
        assert (back(ya.buf) - ref).abs().max().item() < tol, modes
    # src/model.py:33-36 WeightedAdd bug: sum(w_i + f_i)
    p = Plan(torch.device(DEV))
    ya = Act.of(torch.zeros(2, 8, 6, Cc, dtype=dtype, device=DEV))
    p.fuse([Act.of(nhwc(mid).to(dtype)), Act.of(nhwc(small).to(dtype))], w[:2], [0, 3], ya, bug=True)
    run(p)
    ref = (w[0] + mid) + (w[1] + F.interpolate(small, scale_factor=2, mode="nearest"))
    assert (back(ya.buf) - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gap_fc(dtype):
    g = torch.Generator().manual_seed(10)
    x = torch.randn(3, 256, 5, 4, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    w, b = torch.randn(2, 256, generator=g) / 16, torch.randn(2, generator=g)
    ref = F.linear(x.mean((2, 3)), w, b)
    p = Plan(torch.device(DEV))
    y = torch.zeros(3, 2, device=DEV)
    p.gap_fc(Act.of(nhwc(x).to(dtype)), w.to(DEV), b.to(DEV), y)
    run(p)
    assert (y.cpu() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("D,M", [(96, 1000), (192, 517), (96, 256), (384, 128 * 3), (384, 128 * 7 + 37)])
def test_convnext_mlp_fused(D, M):
    """mlp_fused.hip against torch: res + fc2'(GELU(fc1(t))) with bf16 storage of t, the hidden activations and the weights
    (the unfused path rounds at the same points); ragged M exercises the tail masking."""
    from multitask_bonetumor_yolo_amd.model import _permute_hidden
    from multitask_bonetumor_yolo_amd import _lib as L
    g = torch.Generator().manual_seed(D + M)
    t = torch.randn(M, D, generator=g).bfloat16()
    res = torch.randn(M, D, generator=g).bfloat16()
    w1 = (torch.randn(4 * D, D, generator=g) / D ** 0.5).bfloat16()
    w2 = (torch.randn(D, 4 * D, generator=g) / (4 * D) ** 0.5 * 0.1).bfloat16()
    b1, b2 = torch.randn(4 * D, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    hid = torch.nn.functional.gelu(t.float() @ w1.float().t() + b1).bfloat16().float()
    ref = res.float() + hid @ w2.float().t() + b2
    y = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    dev = [v.to(DEV) for v in (t, res, w1, b1, _permute_hidden(w2), b2)]
    lib = L.load()
    rc = lib.mtbt_convnext_mlp_fused(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(),
                                     dev[5].data_ptr(), y.data_ptr(), M, D, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < 0.03 * max(1.0, ref.abs().max().item()), err   # bf16 output rounding (2^-8 relative) dominates
    assert ((y.float().cpu() - ref).norm() / ref.norm()).item() < 5e-3


def _perm_rows_train(n4d):
    """Row order of mtbt_convnext_mlp_fused_train's fc1 weight / bias (header): staged row 16 b + 4 g + e of a 32-row chunk = hidden 8 g + 4 b + e."""
    idx = torch.arange(n4d).view(-1, 4, 2, 4)            # [chunk][g][b][e] natural
    return idx.permute(0, 2, 1, 3).reshape(-1)           # [chunk][b][g][e] staged


@pytest.mark.parametrize("D,M", [(96, 200), (192, 333), (96, 4096)])
def test_convnext_mlp_fused_train_keeps_the_pre_activation(D, M):
    """Training forward of the ConvNeXt Mlp in one launch (timm Mlp under model.train(), main_model.py:21-26): y as the inference kernel's, plus
    hpre = fc1(t) + b1 in NATURAL hidden order (bf16) -- what the backward reads.  Ragged M: nothing is written past row M."""
    from multitask_bonetumor_yolo_amd import _lib as L
    g = torch.Generator().manual_seed(D + M)
    t = torch.randn(M, D, generator=g).bfloat16()
    res = torch.randn(M, D, generator=g).bfloat16()
    w1 = (torch.randn(4 * D, D, generator=g) / D ** 0.5).bfloat16()
    w2 = (torch.randn(D, 4 * D, generator=g) / (4 * D) ** 0.5 * 0.1).bfloat16()
    b1, b2 = torch.randn(4 * D, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    pre = t.float() @ w1.float().t() + b1
    ref = res.float() + torch.nn.functional.gelu(pre).bfloat16().float() @ w2.float().t() + b2
    perm = _perm_rows_train(4 * D)
    y = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    hp = torch.full((M + 3, 4 * D), 7.0, dtype=torch.bfloat16, device=DEV)
    dev = [v.to(DEV) for v in (t, res, w1[perm].contiguous(), b1[perm].contiguous(), w2, b2)]
    lib = L.load()
    rc = lib.mtbt_convnext_mlp_fused_train(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(),
                                           dev[5].data_ptr(), y.data_ptr(), hp.data_ptr(), M, D, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert ((y.float().cpu() - ref).norm() / ref.norm()).item() < 5e-3
    got = hp.float().cpu()
    assert torch.all(got[M:] == 7.0)
    assert (got[:M] - pre).abs().max().item() <= 2 ** -7 * max(1.0, pre.abs().max().item())          # bf16 rounding of the stored value
    assert lib.mtbt_convnext_mlp_fused_train(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), dev[5].data_ptr(),
                                             y.data_ptr(), None, M, D, torch.cuda.current_stream().cuda_stream) == -1      # MTBT_EINVAL: hpre is what this entry is for


def test_fp16_store_saturates_and_keeps_nan():
    """The fp16 arithmetic mode's stores (common.h f2h_bits, BASELINE configs[4]): finite values beyond the binary16 range saturate at
    +-65504, infinities too, and a NaN stays a NaN (v_med3_f32 alone would have returned -65504 for it and hidden a divergence)."""
    lib = L.load()
    src = torch.tensor([float("nan"), float("inf"), float("-inf"), 1.0e6, -1.0e6, 65504.0, 1.0, -2.5, 0.0, 70000.0, -float("nan"), 3.0e-8] * 16,
                       dtype=torch.float32, device=DEV)
    dst = torch.zeros(src.numel(), dtype=torch.float16, device=DEV)
    L.check(lib.mtbt_cast(src.data_ptr(), dst.data_ptr(), src.numel(), L.F32, L.F16, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cast")
    torch.cuda.synchronize()
    got = dst.float().cpu()[:12]
    want = torch.tensor([float("nan"), 65504.0, -65504.0, 65504.0, -65504.0, 65504.0, 1.0, -2.5, 0.0, 65504.0, float("nan"), 0.0])
    assert torch.isnan(got[0]) and torch.isnan(got[10])
    fin = [i for i in range(12) if i not in (0, 10)]
    assert torch.equal(got[fin][:-1], want[fin][:-1]) and abs(got[11].item() - 3.0e-8) < 6e-8   # (3e-8 is a binary16 subnormal: rounded, not flushed to garbage)
    # the conv epilogue's fp16 store takes the same path: a NaN input pixel stays visible in the output
    p = Plan(DEV)
    x = torch.zeros(1, 4, 4, 32, dtype=torch.float16, device=DEV)
    x[0, 1, 2, 3] = float("nan")
    w = torch.eye(32, dtype=torch.float16, device=DEV)
    y = p.new(1, 4, 4, 32, L.F16)
    p.conv(Act.of(x), w, y, name="nan probe")
    run(p)
    out = y.buf.float().cpu()
    assert torch.isnan(out[0, 1, 2]).all() and not torch.isnan(out[0, 0]).any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_upconv_fused_proto(dtype):
    """mtbt_convt2x2_conv3x3_nhwc = ConvTranspose2d(2, 2, bias) -> Conv 3x3 + folded BatchNorm + SiLU (ultralytics Proto.upsample -> Proto.cv2,
    main_model.py:326-328) against torch fp32 on the same operands: every output pixel incl. the border rows / columns, where the transposed
    conv's bias enters through fewer taps.  fp32 mode <= 1e-3 absolute; bf16 / fp16: relative L2 against the fp32 result of the ROUNDED
    composed operands <= 2e-2 / 3e-3 and border pixels no worse than interior ones."""
    from multitask_bonetumor_yolo_amd.model import compose_upconv
    g = torch.Generator().manual_seed(12)
    N, H, W, C, K = 2, 32, 48, 64, 128
    wt, bt = torch.randn(C, C, 2, 2, generator=g) * 0.08, torch.randn(C, generator=g) * 0.5
    w3 = torch.randn(K, C, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    x = torch.randn(N, C, H, W, generator=g)
    ref = F.silu(F.conv2d(F.conv_transpose2d(x, wt, bt, stride=2), w3, padding=1) * sc[None, :, None, None] + sh[None, :, None, None])
    wc, s9 = compose_upconv(wt, bt, w3, sc, sh)
    code = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}[dtype]
    p = Plan(DEV)
    xa = Act.of(nhwc(x).to(dtype))
    y = p.new(N, 2 * H, 2 * W, K, code)
    p.upconv(xa, wc.to(DEV).to(dtype), s9.to(DEV), y, act=L.ACT_SILU)
    run(p)
    out = back(y.buf)
    err = (out - ref).abs()
    border = torch.zeros(2 * H, 2 * W, dtype=torch.bool)
    border[0], border[-1], border[:, 0], border[:, -1] = True, True, True, True
    if dtype == torch.float32:
        assert err.max().item() < TOL32, err.max().item()
    else:
        rel = (out - ref).norm().item() / ref.norm().item()
        assert rel < (2e-2 if dtype == torch.bfloat16 else 3e-3), rel
        assert err[:, :, border].max().item() <= 1.5 * err[:, :, ~border].max().item() + 1e-3
    # the bias classes matter at this scale: the interior shift on a border row would be off by much more than the tolerance
    assert (s9[4] - s9[1]).abs().max().item() > 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["td_up", "out_down3", "ragged"])
def test_bifpn_node_fused_equals_fuse_then_pointwise(dtype, case):
    """mtbt_bifpn_node_nhwc (weighted sum + resample as the B-operand staging of the DepthwiseConvBlock GEMM, main_model.py:198-243 + :62-102)
    against the two launches it replaces (mtbt_bifpn_fuse, then the 1x1 conv + shift + ELU) on the same operands: equal up to a unit in the
    last place of a few elements (the fused map is rounded to the storage type exactly as the stand-alone kernel stores it; the GEMMs may
    accumulate their 32-channel steps in a different order), and within the bf16 / fp16 tolerance of torch fp32."""
    g = torch.Generator().manual_seed(3)
    code = {torch.bfloat16: L.BF16, torch.float16: L.F16}[dtype]
    N, H, W, Cc, K = (2, 16, 24, 256, 256) if case != "ragged" else (1, 10, 6, 128, 128)     # ragged: 60 pixels (< one 64-pixel tile), C = K = 128
    x0 = torch.randn(N, Cc, H, W, generator=g)
    if case == "out_down3":
        ins = [x0, torch.randn(N, Cc, H, W, generator=g), torch.randn(N, Cc, 2 * H, 2 * W, generator=g)]
        modes, wts = [L.RES_ID, L.RES_ID, L.RES_DOWN_MEAN], [0.31, 0.42, 0.27]
    else:
        ins = [x0, torch.randn(N, Cc, H // 2, W // 2, generator=g)]
        modes, wts = [L.RES_ID, L.RES_UP_BILINEAR], [0.55, 0.45]
    w = (torch.randn(K, Cc, generator=g) / Cc ** 0.5)
    shift = torch.randn(K, generator=g) * 0.3
    acts = [Act.of(nhwc(t).to(dtype)) for t in ins]
    wd, sd = w.to(DEV).to(dtype), shift.to(DEV)
    p = Plan(DEV)
    s = p.new(N, H, W, Cc, code)
    p.fuse(acts, wts, modes, s)
    y2 = p.new(N, H, W, K, code)
    p.conv(s, wd, y2, shift=sd, act=L.ACT_ELU)
    y1 = p.new(N, H, W, K, code)
    p.node(acts, wts, modes, wd, sd, y1, act=L.ACT_ELU)
    run(p)
    # the same operands through the same MFMAs: the fused map may differ from the stand-alone kernel's by a last-bit rounding in a few places
    # (hipcc contracts the weighted sum's multiply-adds differently in the two kernels) and the GEMMs may accumulate their 32-channel steps
    # in a different order -- a few units in the last place of a small fraction of the outputs
    d = (y1.buf.float() - y2.buf.float()).abs()
    scale = y2.buf.float().abs().max().item()
    assert d.max().item() <= (1e-2 if dtype == torch.bfloat16 else 2e-3) * scale and (d > 0).float().mean().item() < 0.05, \
        (d.max().item(), scale, (d > 0).float().mean().item())
    # torch fp32 reference of the node
    ref_in = [t.to(dtype).float() for t in ins]
    parts = []
    for t, m in zip(ref_in, modes):
        if m == L.RES_UP_BILINEAR:
            t = F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
        elif m == L.RES_DOWN_MEAN:
            t = F.avg_pool2d(t, 2)
        parts.append(t)
    fused = sum(wv * t for wv, t in zip(wts, parts))
    ref = F.elu(F.conv2d(fused, w.to(dtype).float()[:, :, None, None]) + shift[None, :, None, None])
    rel = (back(y1.buf) - ref).norm().item() / ref.norm().item()
    assert rel < (2e-2 if dtype == torch.bfloat16 else 3e-3), rel
