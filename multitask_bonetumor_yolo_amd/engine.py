"""Host-side launch engine: NHWC activation views, a static buffer pool and pre-built C-ABI calls.

A forward pass is compiled once per (batch, size, dtype, weights version) into a flat list of
`Launch` records, each a C-ABI function plus its argument block (device pointers already resolved).
Running the plan is a tight loop of ctypes calls on the caller's current HIP stream: no allocation,
no host synchronisation, so it can be captured into a HIP graph.  PyTorch only owns the memory.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from . import _lib as L

TORCH_DTYPE = {L.F32: torch.float32, L.BF16: torch.bfloat16}
ESIZE = {L.F32: 4, L.BF16: 2}


def code_of(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return L.F32
    if dtype == torch.bfloat16:
        return L.BF16
    raise ValueError(f"unsupported compute dtype {dtype}: float32 (parity) or bfloat16 (throughput)")


@dataclass
class Act:
    """An NHWC activation view: element (n,y,x,c) lives at buf.data_ptr() + (off + n*bs + (y*W+x)*ld + c) * esize.
    ld > C makes it a channel slice of a wider buffer (concat-free C2f); bs lets a pyramid level live
    inside a larger per-image buffer (mask coefficients [N,A,nm])."""
    buf: torch.Tensor
    off: int
    N: int
    H: int
    W: int
    C: int
    ld: int
    bs: int

    @staticmethod
    def of(buf: torch.Tensor) -> "Act":
        N, H, W, Cc = buf.shape
        assert buf.is_contiguous()
        return Act(buf, 0, N, H, W, Cc, Cc, H * W * Cc)

    @property
    def code(self): return code_of(self.buf.dtype)
    @property
    def ptr(self): return self.buf.data_ptr() + self.off * self.buf.element_size()
    @property
    def batch_stride(self): return self.bs
    @property
    def dense(self): return self.ld == self.C and self.bs == self.H * self.W * self.C

    def slice(self, c0, Cc):
        assert 0 <= c0 and c0 + Cc <= self.C
        return Act(self.buf, self.off + c0, self.N, self.H, self.W, Cc, self.ld, self.bs)

    def nchw(self) -> torch.Tensor:
        """Logical [N,C,H,W] view (channels-last strides) of this activation; no copy."""
        return self.buf.as_strided((self.N, self.C, self.H, self.W), (self.bs, 1, self.W * self.ld, self.ld),
                                   self.buf.storage_offset() + self.off)


class Pool:
    """Plan-time buffer pool: buffers freed at plan-build time are handed to later ops of the same
    plan (stream order makes the reuse safe), keeping the working set small and cache-resident."""

    def __init__(self, device):
        self.device = device
        self.free_list = {}
        self.all = []
        self.bytes = 0

    def get(self, shape, dtype) -> torch.Tensor:
        key = (tuple(shape), dtype)
        lst = self.free_list.get(key)
        if lst:
            return lst.pop()
        t = torch.empty(shape, dtype=dtype, device=self.device)
        self.all.append(t)
        self.bytes += t.numel() * t.element_size()
        return t

    def put(self, t: torch.Tensor):
        self.free_list.setdefault((tuple(t.shape), t.dtype), []).append(t)


@dataclass
class Launch:
    fn: Callable
    args: tuple
    name: str
    keep: tuple = ()  # python objects that own the memory the argument block points to
    flops: float = 0.0
    bytes: float = 0.0


class Plan:
    def __init__(self, device):
        self.lib = L.load()
        self.device = device
        self.pool = Pool(device)
        self.launches: List[Launch] = []
        self.consts = []  # folded weights etc. (kept alive)

    # ---- execution ----
    def run(self, stream: Optional[int] = None, start: int = 0, end: Optional[int] = None):
        """Issue launches [start, end) on the current (or given) stream."""
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream)
        for l in self.launches[start:end]:
            rc = l.fn(*l.args, s)
            if rc != 0:
                L.check(rc, l.name)

    def run_timed(self):
        """Replay with a HIP event pair around every launch (all on torch's current stream, which is the stream
        the kernels are launched on).  Returns per-launch milliseconds; slower than run(), for roofline accounting."""
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        evs = []
        for l in self.launches:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = l.fn(*l.args, s)
            b.record()
            if rc != 0:
                L.check(rc, l.name)
            evs.append((a, b))
        torch.cuda.synchronize(self.device)
        return [a.elapsed_time(b) for a, b in evs]

    # ---- helpers ----
    def const(self, t: torch.Tensor, dtype=None) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=dtype or t.dtype).contiguous()
        self.consts.append(t)
        return t

    def new(self, N, H, W, Cc, code) -> Act:
        return Act.of(self.pool.get((N, H, W, Cc), TORCH_DTYPE[code]))

    def release(self, a: Act):
        self.pool.put(a.buf)

    # ---- op builders ----
    def conv(self, x: Act, w: torch.Tensor, y: Act, *, R=1, S=1, stride=1, pad=0, scale=None, shift=None,
             act=L.ACT_NONE, res: Optional[Act] = None, out_mode=L.OUT_NHWC, name="conv", tile_hint=0):
        """w: packed [K, R*S*C] in x's dtype.  y: output view (dtype may be f32)."""
        K = w.shape[0]
        assert w.shape[1] == R * S * x.C, (w.shape, R, S, x.C)
        Ho = (x.H + 2 * pad - R) // stride + 1
        Wo = (x.W + 2 * pad - S) // stride + 1
        a = L.ConvArgs()
        a.x, a.w, a.y = x.ptr, w.data_ptr(), y.ptr
        a.scale = scale.data_ptr() if scale is not None else None
        a.shift = shift.data_ptr() if shift is not None else None
        a.res = res.ptr if res is not None else None
        a.x_batch_stride, a.x_pixel_stride = x.batch_stride, x.ld
        a.y_batch_stride = y.batch_stride
        a.y_pixel_stride = y.ld
        a.res_batch_stride = res.batch_stride if res is not None else 0
        a.res_pixel_stride = res.ld if res is not None else 0
        a.N, a.H, a.W, a.C, a.K, a.R, a.S = x.N, x.H, x.W, x.C, K, R, S
        a.stride, a.pad, a.Ho, a.Wo = stride, pad, Ho, Wo
        a.dtype, a.out_dtype, a.act, a.out_mode, a.tile_hint = x.code, y.code, act, out_mode, tile_hint
        flops = 2.0 * x.N * Ho * Wo * K * R * S * x.C
        byts = (x.N * x.H * x.W * x.C + K * R * S * x.C) * ESIZE[x.code] + x.N * Ho * Wo * K * ESIZE[y.code]
        self.launches.append(Launch(self.lib.mtbt_conv2d_nhwc, (C.byref(a),), name, (a, x.buf, w, y.buf, scale, shift, res), flops, byts))
        return a

    def stem(self, x_nchw: torch.Tensor, w, b, lnw, lnb, eps, y: Act, name="stem"):
        N, _, H, W = x_nchw.shape
        args = (x_nchw.data_ptr(), w.data_ptr(), b.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), C.c_float(eps), y.ptr,
                N, H, W, y.C, y.code)
        self.launches.append(Launch(self.lib.mtbt_stem_conv4x4_ln, args, name, (x_nchw, w, b, lnw, lnb, y.buf),
                                    2.0 * N * (H // 4) * (W // 4) * y.C * 48,
                                    N * 3 * H * W * 4 + N * (H // 4) * (W // 4) * y.C * ESIZE[y.code]))

    def dwconv(self, x: Act, w, y: Act, ksize, *, bias=None, lnw=None, lnb=None, eps=0.0, scale=None, shift=None,
               act=L.ACT_NONE, name="dwconv"):
        assert x.dense and y.dense and x.C == y.C
        p = lambda t: t.data_ptr() if t is not None else None
        args = (x.ptr, w.data_ptr(), p(bias), p(lnw), p(lnb), C.c_float(eps), p(scale), p(shift), act, y.ptr,
                x.N, x.H, x.W, x.C, ksize, x.code)
        n = x.N * x.H * x.W * x.C
        self.launches.append(Launch(self.lib.mtbt_dwconv_nhwc, args, name, (x.buf, w, bias, lnw, lnb, scale, shift, y.buf),
                                    2.0 * n * ksize * ksize, 2.0 * n * ESIZE[x.code]))

    def layernorm(self, x: Act, w, b, eps, y: Act, name="layernorm"):
        assert x.dense and y.dense
        pixels = x.N * x.H * x.W
        args = (x.ptr, w.data_ptr(), b.data_ptr(), C.c_float(eps), y.ptr, pixels, x.C, x.code)
        self.launches.append(Launch(self.lib.mtbt_layernorm_nhwc, args, name, (x.buf, w, b, y.buf), 0.0,
                                    2.0 * pixels * x.C * ESIZE[x.code]))

    def fuse(self, inputs, weights, modes, y: Act, bug=False, name="bifpn_fuse"):
        a = L.FuseArgs()
        for i, (t, wv, m) in enumerate(zip(inputs, weights, modes)):
            assert t.dense
            a.x[i], a.wgt[i], a.resample[i] = t.ptr, float(wv), m
        a.n_in, a.y = len(inputs), y.ptr
        a.N, a.H, a.W, a.C, a.dtype, a.add_weight_bug = y.N, y.H, y.W, y.C, y.code, int(bug)
        n = y.N * y.H * y.W * y.C
        self.launches.append(Launch(self.lib.mtbt_bifpn_fuse, (C.byref(a),), name, (a, y.buf) + tuple(t.buf for t in inputs),
                                    0.0, n * ESIZE[y.code] * (1 + len(inputs))))
        return a

    def gap_fc(self, x: Act, w, b, y: torch.Tensor, name="gap_fc"):
        assert x.dense
        args = (x.ptr, w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), x.N, x.H * x.W, x.C,
                w.shape[0], x.code)
        self.launches.append(Launch(self.lib.mtbt_gap_fc, args, name, (x.buf, w, b, y), 0.0,
                                    x.N * x.H * x.W * x.C * ESIZE[x.code]))

    def bn_train(self, x: Act, y: Act, bn, act, name="bn_train"):
        """BatchNorm with batch statistics + activation over a dense NHWC tensor; updates bn.running_* in place."""
        assert x.dense and y.dense and x.C == y.C and x.code == y.code
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not supported")
        pixels = x.N * x.H * x.W
        nbytes = self.lib.mtbt_bn_train_workspace_bytes(pixels, x.C)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=self.device)
        rm = bn.running_mean.data_ptr() if bn.running_mean is not None else None
        rv = bn.running_var.data_ptr() if bn.running_var is not None else None
        g, b = self.const(bn.weight, torch.float32), self.const(bn.bias, torch.float32)
        args = (x.ptr, y.ptr, g.data_ptr(), b.data_ptr(), rm, rv, C.c_float(bn.momentum), C.c_float(bn.eps), act, pixels, x.C,
                x.code, ws.data_ptr(), nbytes)
        self.launches.append(Launch(self.lib.mtbt_bn_train_nhwc, args, name, (x.buf, y.buf, g, b, ws, bn), 0.0,
                                    3.0 * pixels * x.C * ESIZE[x.code]))

    def cast(self, x: Act, y: Act, name="cast"):
        assert x.dense and y.dense
        n = x.N * x.H * x.W * x.C
        self.launches.append(Launch(self.lib.mtbt_cast, (x.ptr, y.ptr, n, x.code, y.code), name, (x.buf, y.buf), 0.0,
                                    n * (ESIZE[x.code] + ESIZE[y.code])))

    def raw(self, fn, args, name, keep=()):
        self.launches.append(Launch(fn, args, name, keep))
