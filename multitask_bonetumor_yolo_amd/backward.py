"""Input gradients (dgrad) of the network's convolutions through the FORWARD kernels -- the first building block of the
backward pass.  No new GEMM kernel is needed for any of them:

  k x k, stride 1   dX = conv(dY, W'), W'[c][r][s][k] = W[k][R-1-r][S-1-s][c], padding R-1-pad     (mtbt_conv2d_nhwc)
  1 x 1 / Linear     the transposed weight (the case R = S = 1 of the line above)
  2 x 2, stride 2    dX = conv_transpose(dY, W): the OUT_CONVT2X2 scatter mode of the same kernel   (ConvNeXt downsample)
  depthwise k x k    the depthwise kernel on dY with the taps flipped                                (mtbt_dwconv_nhwc)

`conv_wgrad` is the weight gradient of the dense convolutions (any stride) (csrc/wgrad.hip: transposed LDS reads feed the MFMA, fp32 out,
deterministic split reduction; 3x3 / stride 1 from an LDS halo tile).  The helpers below are the STAND-ALONE forms of the backward operators
(one call = one launch, scratch allocated per call) that the kernel tests and probes use; the assembled training step -- the forward
that keeps what backward needs, the backward launch plan with plan-resident scratch, the autograd boundary -- is `train.py` /
`trainstep.py`.  The plan-building helpers (`conv_dgrad`, `dwconv_dgrad`, ...) append launches to an `engine.Plan`;
weights are re-laid out once by the `*_weight` helpers (device tensors, any dtype the kernels take)."""
import ctypes as C

import torch

from . import _lib as L
from .engine import Act, Plan


def dgrad_weight(w_packed: torch.Tensor, R: int, S: int) -> torch.Tensor:
    """Forward weight packed [K, R*S*C] (as `Plan.conv` takes it) -> dgrad weight packed [C, R*S*K]."""
    K = w_packed.shape[0]
    C = w_packed.shape[1] // (R * S)
    return w_packed.view(K, R, S, C).flip(1, 2).permute(3, 1, 2, 0).reshape(C, R * S * K).contiguous()


def conv_dgrad(plan: Plan, dy: Act, w_dgrad: torch.Tensor, dx: Act, *, R: int, S: int, pad: int, name="conv.dgrad"):
    """Stride-1 convolution: dx [N,H,W,C] from dy [N,H,W,K] (same spatial size, i.e. 2*pad == R-1 == S-1, or R = S = 1)."""
    assert R == S and 2 * pad == R - 1, "stride-1 'same' convolutions only"
    return plan.conv(dy, w_dgrad, dx, R=R, S=S, stride=1, pad=R - 1 - pad, name=name)


def downsample2x2_dgrad_weight(w_packed: torch.Tensor) -> torch.Tensor:
    """Forward weight of a 2x2 / stride-2 conv packed [K, 2*2*C] -> the [4*C, K] layout of the OUT_CONVT2X2 mode."""
    K = w_packed.shape[0]
    C = w_packed.shape[1] // 4
    return w_packed.view(K, 2, 2, C).permute(1, 2, 3, 0).reshape(4 * C, K).contiguous()


def downsample2x2_dgrad(plan: Plan, dy: Act, w_t: torch.Tensor, dx: Act, name="downsample.dgrad"):
    """dx [N,2H,2W,C] from dy [N,H,W,K]: every output pixel receives exactly one tap (non-overlapping patches)."""
    return plan.conv(dy, w_t, dx, out_mode=L.OUT_CONVT2X2, name=name)


def dwconv_dgrad_weight(w_taps: torch.Tensor, ksize: int) -> torch.Tensor:
    """Depthwise taps [k*k, C] -> flipped taps."""
    return w_taps.view(ksize, ksize, -1).flip(0, 1).reshape(ksize * ksize, -1).contiguous()


def dwconv_dgrad(plan: Plan, dy: Act, w_flipped: torch.Tensor, dx: Act, ksize: int, ones: torch.Tensor, zeros: torch.Tensor,
                 name="dwconv.dgrad"):
    """Depthwise stride-1 'same' convolution; `ones` / `zeros`: fp32 [C] (identity epilogue of the depthwise kernel)."""
    return plan.dwconv(dy, w_flipped, dx, ksize, scale=ones, shift=zeros, name=name)


def conv_wgrad(x: Act, dy: Act, *, R: int, S: int, pad: int, stride: int = 1, out: torch.Tensor = None, accumulate: bool = False) -> torch.Tensor:
    """dW of a convolution (any stride; the ConvNeXt 2x2 / stride-2 downsample included) as fp32 [K, R*S*C] (the packed forward layout) from x [N,H,W,C] and dy [N,H,W,K], both
    bf16.  `out` may be a view into a flat gradient bucket; `accumulate=True` adds to it.  Runs on the current stream."""
    lib = L.load()
    if x.code != L.BF16 or dy.code != L.BF16:
        raise NotImplementedError("conv_wgrad: bf16 operands only")
    assert x.N == dy.N and (dy.H, dy.W) == ((x.H + 2 * pad - R) // stride + 1, (x.W + 2 * pad - S) // stride + 1)
    K, Cc = dy.C, x.C
    dev = x.buf.device
    if out is None:
        out = torch.empty(K, R * S * Cc, dtype=torch.float32, device=dev)
    assert out.dtype == torch.float32 and out.numel() == K * R * S * Cc and out.is_contiguous()
    nbytes = lib.mtbt_conv_wgrad_workspace_bytes(x.N, x.H, x.W, Cc, K, R, S)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    L.check(lib.mtbt_conv_wgrad(x.ptr, dy.ptr, out.data_ptr(), x.N, x.H, x.W, Cc, K, R, S, pad, stride, x.batch_stride, x.ld, dy.batch_stride, dy.ld,
                                x.code, int(accumulate), ws.data_ptr(), nbytes, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
            "mtbt_conv_wgrad")
    return out


def act_backward(dy: Act, z: Act, act: int, out: Act = None) -> Act:
    """dz = dy * act'(z) (z = pre-activation), dense NHWC tensors of one dtype.  Runs on the current stream."""
    lib = L.load()
    assert dy.dense and z.dense and dy.code == z.code and dy.buf.shape == z.buf.shape
    if out is None:
        out = Act.of(torch.empty_like(dy.buf))
    dev = dy.buf.device
    L.check(lib.mtbt_act_backward(dy.ptr, z.ptr, out.ptr, dy.buf.numel(), act, dy.code, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
            "mtbt_act_backward")
    return out


def channel_sum(x: Act, out: torch.Tensor = None, accumulate: bool = False, times: Act = None) -> torch.Tensor:
    """fp32 [C] = sum over all pixels of x [N,H,W,C] (bias / shift gradient), or of x * times (scale gradient).  `out` may be a view
    into a gradient bucket."""
    lib = L.load()
    assert x.dense and (times is None or (times.dense and times.code == x.code and times.C == x.C))
    dev = x.buf.device
    P = x.N * x.H * x.W
    if out is None:
        out = torch.empty(x.C, dtype=torch.float32, device=dev)
    nbytes = lib.mtbt_channel_sum_workspace_bytes(P, x.C)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    L.check(lib.mtbt_channel_sum(x.ptr, times.ptr if times is not None else None, P, x.C, x.ld, times.ld if times is not None else 0, x.code,
                                 out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes,
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtbt_channel_sum")
    return out


def batchnorm_train_backward(dz: Act, u: Act, gamma: torch.Tensor, beta: torch.Tensor, batch_var: torch.Tensor, eps: float):
    """Backward of BatchNorm2d on BATCH statistics (the heads in `forward(x, "train")`, main_model.py:358-359) given dz = d loss /
    d(BN output) and u = the BN output itself (the kept pre-activation): returns (dx as an Act, d gamma, d beta).
        xhat = (u - beta) / gamma;  d beta = sum dz;  d gamma = sum dz * xhat;
        dx = gamma / sigma * (dz - d beta / M - xhat * d gamma / M),  sigma = sqrt(batch_var + eps),  M = pixels per channel
    Two deterministic channel reductions, a handful of [C]-sized tensor ops, one elementwise pass."""
    lib = L.load()
    dev = dz.buf.device
    M = float(dz.N * dz.H * dz.W)
    gamma, beta = gamma.float(), beta.float()
    s1 = channel_sum(dz)                                # sum dz
    s2 = channel_sum(dz, times=u)                       # sum dz * u
    d_beta = s1
    d_gamma = (s2 - beta * s1) / gamma
    inv_sigma = torch.rsqrt(batch_var.float() + eps)
    a = gamma * inv_sigma                               # coefficient of dz
    b = -(d_gamma / M) * inv_sigma                      # coefficient of u:  -(gamma/sigma) * (d_gamma/M) / gamma
    d = a * (-(s1 / M)) - b * beta                      # constant:          -(gamma/sigma) * d_beta/M + (d_gamma/M)/sigma * beta
    out = Act.of(torch.empty_like(dz.buf))
    a, b, d = a.contiguous(), b.contiguous(), d.contiguous()
    L.check(lib.mtbt_channel_affine2(dz.ptr, u.ptr, a.data_ptr(), b.data_ptr(), d.data_ptr(), out.ptr, dz.N * dz.H * dz.W, dz.C, dz.code,
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtbt_channel_affine2")
    out._keep = (a, b, d)                               # the launch is asynchronous
    return out, d_gamma, d_beta


def layernorm_backward(x: Act, dy: Act, gamma: torch.Tensor, eps: float):
    """Backward of a LayerNorm over the channels of each pixel (ConvNeXt `norm`, downsample LayerNorm2d): returns
    (dx as an Act, d gamma, d beta) from the LN input x and dy."""
    lib = L.load()
    assert x.dense and dy.dense and x.code == dy.code and x.buf.shape == dy.buf.shape
    dev = x.buf.device
    dx, xhat = Act.of(torch.empty_like(x.buf)), Act.of(torch.empty_like(x.buf))
    g = gamma.float().contiguous()
    L.check(lib.mtbt_layernorm_backward_nhwc(x.ptr, dy.ptr, g.data_ptr(), eps, dx.ptr, xhat.ptr, x.N * x.H * x.W, x.C, x.code, 0,
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtbt_layernorm_backward_nhwc")
    dx._keep = g
    return dx, channel_sum(dy, times=xhat), channel_sum(dy)


def dwconv_wgrad(x: Act, dy: Act, ksize: int, out: torch.Tensor = None, accumulate: bool = False) -> torch.Tensor:
    """Depthwise weight gradient as fp32 [k*k, C] (the forward tap layout) from x and dy [N,H,W,C]."""
    lib = L.load()
    assert x.dense and dy.dense and x.code == dy.code and x.buf.shape == dy.buf.shape
    dev = x.buf.device
    if out is None:
        out = torch.empty(ksize * ksize, x.C, dtype=torch.float32, device=dev)
    nbytes = lib.mtbt_dwconv_wgrad_workspace_bytes(x.N, x.H, x.W, x.C, ksize)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    L.check(lib.mtbt_dwconv_wgrad(x.ptr, dy.ptr, out.data_ptr(), x.N, x.H, x.W, x.C, ksize, x.code, int(accumulate), ws.data_ptr(), nbytes,
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtbt_dwconv_wgrad")
    return out
