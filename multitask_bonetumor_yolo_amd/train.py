"""Training lowering of `ConvNeXtBiFPNYOLO`: forward in train mode that KEEPS what the backward pass needs, the backward
launch plan, and the autograd boundary that lets the reference trainer's `total_loss.backward()` drive it
(`/root/reference/src/running_main_v3.py:393-445` over `/root/reference/src/main_model.py:342-365`).

What the forward keeps (HBM is 288 GB: nothing is recomputed, nothing is checkpointed):
  * every convolution input (its weight gradient's operand) and, per BatchNorm, the conv output it normalised plus the (mean, var) used;
  * per ConvNeXt block the depthwise output before the LayerNorm, the LayerNorm output, fc1's pre-activation (second epilogue output
    of the GEMM) and its GELU;  per fusion node the inputs.
The backward is a static plan too: a tape of closures recorded while lowering the forward is replayed in reverse and emits C-ABI
launches (dgrad = the forward conv kernel on dY with re-laid-out weights, `mtbt_conv_wgrad`, `mtbt_bn_backward_nhwc`, ...).  Gradient
fan-in (C2f concat slices, ConvNeXt residuals, pyramid levels feeding six head branches) is accumulation INSIDE the producing
kernels (residual input of the conv epilogue, `accumulate` flags) -- a residual connection is a buffer alias, not a kernel.

Weights change every step, so the packed compute-dtype copies the kernels read (forward KRSC, dgrad CRSK with flipped taps, folded
layer scale / depthwise scale) are regenerated from the fp32 master parameters by ONE table-driven launch (`mtbt_weight_prep`) at
the head of the forward plan.  Parameter gradients land in flat fp32 buckets (`dist_train.FlatBuckets`, reverse registration
order) in the kernels' layouts; 4-D conv weights are channels-last there, so the tensors handed to autograd are permuted views.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch
import torch.nn as nn

from . import _lib as L
from . import model as M
from .dist_train import FlatBuckets
from .engine import Act, ESIZE, Launch, Plan, TORCH_DTYPE, _region

# column sums from the conv epilogue (BatchNorm batch statistics, d fc1.bias) instead of separate passes; MTBT_FUSED_COLSUM=0 turns them off (A/B)
FUSED_BN_STATS = os.environ.get("MTBT_FUSED_COLSUM", "1") != "0"
FUSED_TRAIN_MLP = os.environ.get("MTBT_TRAIN_FUSED_MLP", "1") != "0"      # ConvNeXt Mlp forward of stages 0-1 as one launch (bf16), see TrainPlan

WS_BYTES = 256 << 20           # shared scratch of the reduction kernels (sequential plan)
CLS_PAD = 32                   # the nc-channel class conv's gradient operand is zero-padded to this many channels


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


DRY_LOWERING = False           # tests/test_cpu_*: build the launch plans (regions, dependencies, schedule) on CPU tensors; nothing is ever issued


def _param_ptr(p: torch.Tensor, what: str) -> int:
    if p.dtype != torch.float32 or not (p.is_cuda or DRY_LOWERING):
        raise RuntimeError(f"{what}: the training plan reads parameters in place: fp32 CUDA/HIP tensors expected, got {p.dtype} on {p.device}")
    return p.data_ptr()


def _dense_vec(p: torch.Tensor, what: str) -> torch.Tensor:
    """A parameter the kernels read in place as a dense fp32 vector / matrix."""
    _param_ptr(p, what)
    if not p.is_contiguous():
        raise RuntimeError(f"{what}: expected a contiguous parameter")
    return p


class WsPool:
    """Scratch buffers of the reduction kernels, handed out round-robin: launches that got DIFFERENT buffers are independent for the lane
    scheduler (one shared buffer would serialise every weight gradient, channel sum and BatchNorm of the step); stream order plus the
    recorded write regions keep the reuse of one buffer safe."""
    SMALL, N_SMALL, N_BIG = 32 << 20, 6, 2

    def __init__(self, device):
        self.device = device
        self.small: List[torch.Tensor] = []
        self.big: List[torch.Tensor] = []
        self.i_small = self.i_big = 0

    def get(self, nbytes: int) -> torch.Tensor:
        if nbytes <= self.SMALL:
            if len(self.small) < self.N_SMALL:
                self.small.append(torch.empty(self.SMALL // 4, dtype=torch.float32, device=self.device))
            self.i_small = (self.i_small + 1) % len(self.small)
            return self.small[self.i_small]
        if nbytes > WS_BYTES:
            raise RuntimeError(f"training workspace too small: {nbytes} > {WS_BYTES} bytes (raise train.WS_BYTES)")
        if len(self.big) < self.N_BIG:
            self.big.append(torch.empty(WS_BYTES // 4, dtype=torch.float32, device=self.device))
        self.i_big = (self.i_big + 1) % len(self.big)
        return self.big[self.i_big]


class TPlan(Plan):
    """engine.Plan plus the training-step op builders.  Every builder records what its launch reads and writes, so the plan can be
    issued on one stream or spread over the engine's lanes."""

    def __init__(self, device, ws: WsPool):
        super().__init__(device)
        self.wsp = ws
        self.lane_any = os.environ.get("MTBT_TRAIN_LANE_ANY", "1") != "0"   # eager execution only: no capture-topology restriction on cross-lane waits
        self.cur_ws: Optional[torch.Tensor] = None
        self._late: List[torch.Tensor] = []
        self.pool.reuse = False

    def _ws(self, nbytes: int) -> int:
        """Scratch for the launch being built: returns its pointer; `self.cur_ws` (the tensor) goes into that launch's write set."""
        self.cur_ws = self.wsp.get(nbytes)
        return self.cur_ws.data_ptr()

    def release(self, a: Act):
        """Return a temporary to the pool -- a few releases LATER: handed out again at once, the buffer would tie its next writer (on the
        main chain) to its last reader (a weight gradient that could run beside the chain on a side lane)."""
        self.release_buf(a.buf)

    def release_buf(self, t: torch.Tensor):
        self._late.append(t)
        if len(self._late) > 8:
            self.pool.put(self._late.pop(0))

    def est(self, nbytes: float = 0.0, flops: float = 0.0):
        """Cost estimate of the launch just appended (the lane scheduler keeps machine-filling launches on lane 0)."""
        l = self.launches[-1]
        l.bytes, l.flops = max(l.bytes, nbytes), max(l.flops, flops)

    def conv2(self, x: Act, w, y: Act, *, y2: Optional[Act] = None, colsum: Optional[torch.Tensor] = None, colsum_sq=False,
              colsum_shift: Optional[torch.Tensor] = None, **kw):
        """`colsum` [K] / [2K] fp32: per-channel sums (and sums of squares) of the stored output minus `colsum_shift`, from the conv epilogue."""
        a = self.conv(x, w, y, **kw)
        l = self.launches[-1]
        if y2 is not None:
            assert y2.ld == y.ld and y2.bs == y.bs and y2.code == y.code
            a.y2 = y2.ptr
            l.keep = l.keep + (y2.buf,)
            l.writes = l.writes + (_region(y2),)
        if colsum is not None:
            K = w.shape[0]
            partial_only = colsum is True      # the consumer reduces the partial rows itself (BatchNorm): see colsum_partials()
            assert K % 8 == 0 and (partial_only or (colsum.dtype == torch.float32 and colsum.numel() == K * (2 if colsum_sq else 1)))
            nbytes = self.lib.mtbt_conv_colsum_workspace_bytes(a.N * a.Ho * a.Wo, K, int(colsum_sq))
            a.colsum, a.colsum_sq, a.colsum_accumulate = (None if partial_only else colsum.data_ptr()), int(colsum_sq), 0
            a.colsum_shift = colsum_shift.data_ptr() if colsum_shift is not None else None
            if partial_only:                   # the partial rows outlive the launch: a buffer of their own (exact size), not the rotating scratch
                a.colsum_ws, a.colsum_ws_bytes = 16, nbytes          # (placeholder for the layout query)
                rows, pitch = C.c_int64(0), C.c_int32(0)
                L.check(self.lib.mtbt_conv_colsum_layout(C.byref(a), C.byref(rows), C.byref(pitch)), "colsum layout")
                nbytes = rows.value * pitch.value * 4
                self.cs_partial = torch.empty(rows.value * pitch.value, dtype=torch.float32, device=self.device)
                a.colsum_ws = self.cs_partial.data_ptr()
                self.cs_layout = (self.cs_partial, rows.value, pitch.value)
            else:
                a.colsum_ws = self._ws(nbytes)
                self.cs_partial = self.cur_ws
            a.colsum_ws_bytes = nbytes
            l.keep = l.keep + (colsum if not partial_only else None, colsum_shift, self.cs_partial)
            l.writes = l.writes + ((_region(colsum),) if not partial_only else ()) + (_region(self.cs_partial),)
            if colsum_shift is not None:
                l.reads = l.reads + (_region(colsum_shift),)
        return a

    @staticmethod
    def colsum_ok(K: int) -> bool:
        """The conv epilogue can accumulate column sums for this output width (tile choices: conv_igemm.hip pick_tile with no96)."""
        return K % 8 == 0 and not (64 < K <= 96)

    def bn_forward(self, x: Act, y: Act, bn, act, stats: torch.Tensor, use_running: bool, name, sums: Optional[torch.Tensor] = None):
        """`sums` [2C]: column sums of x - running_mean and their squares, accumulated by the conv that produced x (one pass saved)."""
        assert x.dense and x.C == y.C and x.code == y.code and y.bs == y.H * y.W * y.ld
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not supported")
        pixels = x.N * x.H * x.W
        if sums is not None and not use_running:
            part, rows, pitch = sums          # (partial rows of the producing conv, mtbt_conv_colsum_layout)
            g, b = _dense_vec(bn.weight, name + ".weight"), _dense_vec(bn.bias, name + ".bias")
            args = (x.ptr, y.ptr, y.ld, g.data_ptr(), b.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), C.c_float(bn.momentum),
                    C.c_float(bn.eps), act, pixels, x.C, x.code, part.data_ptr(), rows, pitch, bn.running_mean.data_ptr(), stats.data_ptr())
            self.raw(self.lib.mtbt_bn_forward_partials_nhwc, args, name, keep=(x.buf, y.buf, stats, bn, part), reads=[x, part],
                     writes=[y, stats, bn.running_mean, bn.running_var])
            self.est(2.0 * pixels * x.C * ESIZE[x.code])
            return
        nbytes = self.lib.mtbt_bn_train_workspace_bytes(pixels, x.C)
        g, b = _dense_vec(bn.weight, name + ".weight"), _dense_vec(bn.bias, name + ".bias")
        args = (x.ptr, y.ptr, y.ld, g.data_ptr(), b.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), C.c_float(bn.momentum),
                C.c_float(bn.eps), act, pixels, x.C, x.code, int(use_running), stats.data_ptr(), self._ws(nbytes), nbytes)
        self.raw(self.lib.mtbt_bn_forward_nhwc, args, name, keep=(x.buf, y.buf, stats, bn), reads=[x], writes=[y, stats, self.cur_ws])
        self.est(3.0 * pixels * x.C * ESIZE[x.code])

    def bn_backward(self, dy: Act, x: Act, stats, bn, act, use_running: bool, dx: Act, dgamma, dbeta, name):
        assert x.dense and dx.dense and dy.C == x.C and dy.bs == dy.H * dy.W * dy.ld and dy.code == x.code == dx.code
        pixels = x.N * x.H * x.W
        nbytes = self.lib.mtbt_bn_backward_workspace_bytes(pixels, x.C)
        args = (dy.ptr, dy.ld, x.ptr, stats.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), C.c_float(bn.eps), act, int(use_running), dx.ptr,
                _ptr(dgamma), _ptr(dbeta), 0, pixels, x.C, x.code, self._ws(nbytes), nbytes)
        self.raw(self.lib.mtbt_bn_backward_nhwc, args, name, keep=(dy.buf, x.buf, stats, dx.buf, dgamma, dbeta, bn), reads=[dy, x, stats],
                 writes=[dx, dgamma, dbeta, self.cur_ws])
        self.est(5.0 * pixels * x.C * ESIZE[x.code])

    def wgrad(self, x: Act, dy: Act, out: torch.Tensor, *, R, S, pad, stride=1, dbias: Optional[torch.Tensor] = None, x_act: int = 0, name="wgrad"):
        """dW (and, with `dbias`, sum_p dy -- the bias gradient, from the same launch).  `x_act`: x is a pre-activation, the activation is
        applied while it is staged (mtbt_conv_wgrad_xact)."""
        assert x.code == dy.code and x.N == dy.N and out.dtype == torch.float32 and out.is_contiguous()
        assert out.numel() == dy.C * R * S * x.C, (out.shape, dy.C, R, S, x.C)
        assert dbias is None or (dbias.numel() == dy.C and dbias.dtype == torch.float32)
        nbytes = self.lib.mtbt_conv_wgrad_workspace_bytes(x.N, max(x.H, dy.H), max(x.W, dy.W), x.C, dy.C, R, S)
        tail = (x.N, x.H, x.W, x.C, dy.C, R, S, pad, stride, x.batch_stride, x.ld, dy.batch_stride, dy.ld, x.code, 0, self._ws(nbytes), nbytes)
        if x_act:
            assert dbias is None
            args = (x.ptr, dy.ptr, out.data_ptr()) + tail[:14] + (x_act,) + tail[14:]
            self.raw(self.lib.mtbt_conv_wgrad_xact, args, name, keep=(x.buf, dy.buf, out), reads=[x, dy], writes=[out, self.cur_ws])
        elif dbias is None:
            self.raw(self.lib.mtbt_conv_wgrad, (x.ptr, dy.ptr, out.data_ptr()) + tail, name, keep=(x.buf, dy.buf, out), reads=[x, dy], writes=[out, self.cur_ws])
        else:
            self.raw(self.lib.mtbt_conv_wgrad_bias, (x.ptr, dy.ptr, out.data_ptr(), dbias.data_ptr()) + tail, name, keep=(x.buf, dy.buf, out, dbias),
                     reads=[x, dy], writes=[out, dbias, self.cur_ws])
        self.est(1.0 * dy.N * dy.H * dy.W * (dy.C + x.C) * ESIZE[x.code], 2.0 * dy.N * dy.H * dy.W * dy.C * R * S * x.C)

    def mlp_fused_train(self, t: Act, res: Act, w1p, b1p, w2, b2, y: Act, hpre: Act, name):
        """ConvNeXt Mlp forward in one launch, keeping only the fc1 pre-activation (mtbt_convnext_mlp_fused_train)."""
        assert t.dense and res.dense and y.dense and hpre.dense and t.code == L.BF16 and hpre.C == 4 * t.C
        M, d = t.N * t.H * t.W, t.C
        self.raw(self.lib.mtbt_convnext_mlp_fused_train, (t.ptr, res.ptr, w1p.data_ptr(), b1p.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.ptr, hpre.ptr, M, d),
                 name, keep=(t.buf, res.buf, w1p, b1p, w2, b2, y.buf, hpre.buf), reads=[t, res, w1p, b1p, w2, b2], writes=[y, hpre])
        self.est(2.0 * M * d * (3 + 4), 16.0 * M * d * d)

    def channel_sum(self, x: Act, out: torch.Tensor, times: Optional[Act] = None, name="channel_sum"):
        assert x.bs == x.H * x.W * x.ld and (times is None or (times.bs == times.H * times.W * times.ld and times.code == x.code))
        P = x.N * x.H * x.W
        nbytes = self.lib.mtbt_channel_sum_workspace_bytes(P, x.C)
        args = (x.ptr, times.ptr if times is not None else None, P, x.C, x.ld, times.ld if times is not None else 0, x.code, out.data_ptr(), 0,
                self._ws(nbytes), nbytes)
        self.raw(self.lib.mtbt_channel_sum, args, name, keep=(x.buf, times.buf if times is not None else None, out), reads=[x, times],
                 writes=[out, self.cur_ws])
        self.est(1.0 * P * x.C * ESIZE[x.code] * (2 if times is not None else 1))

    def ln_backward_params(self, x: Act, dy: Act, gamma: torch.Tensor, eps, dx: Act, accumulate: bool, dgamma: torch.Tensor, dbeta: torch.Tensor, name):
        """LayerNorm backward with d gamma / d beta from the same pass (no xhat tensor, no channel-sum passes)."""
        assert x.dense and dy.dense and dx.dense
        P = x.N * x.H * x.W
        nbytes = self.lib.mtbt_layernorm_backward_params_workspace_bytes(P, x.C)
        args = (x.ptr, dy.ptr, gamma.data_ptr(), C.c_float(eps), dx.ptr, P, x.C, x.code, int(accumulate), dgamma.data_ptr(), dbeta.data_ptr(), 0,
                self._ws(nbytes), nbytes)
        self.raw(self.lib.mtbt_layernorm_backward_params_nhwc, args, name, keep=(x.buf, dy.buf, gamma, dx.buf, dgamma, dbeta),
                 reads=[x, dy] + ([dx] if accumulate else []), writes=[dx, dgamma, dbeta, self.cur_ws])
        self.est(3.0 * P * x.C * ESIZE[x.code])

    def dwconv_t(self, x: Act, w, y: Act, ksize, *, bias=None, lnw=None, lnb=None, eps=0.0, scale=None, shift=None, act=L.ACT_NONE,
                 raw: Optional[Act] = None, res: Optional[Act] = None, name="dwconv"):
        assert x.dense and y.dense and x.C == y.C and (raw is None or raw.dense) and (res is None or res.dense)
        args = (x.ptr, w.data_ptr(), _ptr(bias), _ptr(lnw), _ptr(lnb), C.c_float(eps), _ptr(scale), _ptr(shift), act, y.ptr,
                raw.ptr if raw is not None else None, res.ptr if res is not None else None, x.N, x.H, x.W, x.C, ksize, x.code)
        self.raw(self.lib.mtbt_dwconv_nhwc_train, args, name, keep=(x.buf, w, bias, lnw, lnb, scale, shift, y.buf, raw and raw.buf, res and res.buf),
                 reads=[x, res, w, bias, lnw, lnb, scale, shift], writes=[y, raw])
        n = x.N * x.H * x.W * x.C
        self.launches[-1].flops, self.launches[-1].bytes = 2.0 * n * ksize * ksize, 2.0 * n * ESIZE[x.code]

    def dw_wgrad(self, x: Act, dy: Act, out: torch.Tensor, ksize, name, dbias: Optional[torch.Tensor] = None):
        assert x.dense and dy.dense
        nbytes = self.lib.mtbt_dwconv_wgrad_workspace_bytes(x.N, x.H, x.W, x.C, ksize)
        tail = (x.N, x.H, x.W, x.C, ksize, x.code, 0, self._ws(nbytes), nbytes)
        if dbias is None:
            self.raw(self.lib.mtbt_dwconv_wgrad, (x.ptr, dy.ptr, out.data_ptr()) + tail, name, keep=(x.buf, dy.buf, out), reads=[x, dy], writes=[out, self.cur_ws])
        else:
            self.raw(self.lib.mtbt_dwconv_wgrad_bias, (x.ptr, dy.ptr, out.data_ptr(), dbias.data_ptr()) + tail, name, keep=(x.buf, dy.buf, out, dbias),
                     reads=[x, dy], writes=[out, dbias, self.cur_ws])
        self.est(6.0 * x.N * x.H * x.W * x.C * ESIZE[x.code])

    def copy_strided(self, src_ptr, src_code, sbs, sld, dst: Act, N, pixels, Cc, Cpad, keep, name):
        args = (src_ptr, src_code, sbs, sld, dst.ptr, dst.code, dst.bs, dst.ld, N, pixels, Cc, Cpad)
        self.raw(self.lib.mtbt_copy_strided, args, name, keep=keep + (dst.buf,), reads=list(k for k in keep if isinstance(k, torch.Tensor)), writes=[dst])


def arena_specs(model, tail_prefixes: Sequence[str] = ()):
    """Per trainable parameter (registration order): the shape its gradient has in the KERNEL's layout, and how to view a tensor of that
    layout as the parameter's own shape.  4-D conv weights are channels-last ([K,R,S,C]); depthwise taps [k*k, C]; the ConvTranspose
    [Cin,2,2,Cout]; the nc-channel class conv's rows are zero-padded to CLS_PAD.  Parameters whose name starts with one of
    `tail_prefixes` are moved to the END of the list (FlatBuckets fills in reverse order, so they form the first bucket(s)): the
    native training step keeps parameters that never receive a gradient out of the all-reduce and the optimiser."""
    specs, gview = [], {}
    mods = dict(model.named_modules())
    nc = model.nc_det
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        owner = mods[name.rsplit(".", 1)[0]] if "." in name else model
        leaf = name.rsplit(".", 1)[-1]
        kshape, back = tuple(p.shape), None
        if isinstance(owner, nn.ConvTranspose2d) and leaf == "weight":
            ci, co, r, s = p.shape
            kshape, back = (ci, r, s, co), (lambda v: v.permute(0, 3, 1, 2))
        elif isinstance(owner, nn.Conv2d) and leaf == "weight" and p.dim() == 4:
            k, c, r, s = p.shape
            if owner.groups > 1:                                    # depthwise: taps [k*k][C]
                if r == 1:
                    kshape, back = (k,), (lambda v, sh=tuple(p.shape): v.view(sh))
                else:
                    kshape, back = (r * s, k), (lambda v, r=r, s=s, k=k: v.view(r, s, k, 1).permute(2, 3, 0, 1))
            elif c == 3:                                            # the stem: torch layout flattened [K][48]
                kshape, back = (k, c * r * s), (lambda v, sh=tuple(p.shape): v.view(sh))
            elif k == nc and k % 8 != 0:                            # class conv: gradient rows zero-padded to CLS_PAD
                kshape, back = (CLS_PAD, r, s, c), (lambda v, k=k: v[:k].permute(0, 3, 1, 2))
            else:
                kshape, back = (k, r, s, c), (lambda v: v.permute(0, 3, 1, 2))
        elif isinstance(owner, nn.Conv2d) and leaf == "bias" and p.numel() == nc and nc % 8 != 0 and owner.out_channels == nc:
            kshape, back = (CLS_PAD,), (lambda v, k=nc: v[:k])
        specs.append((name, kshape))
        gview[name] = back
    if tail_prefixes:
        is_tail = lambda n: any(n.startswith(t) for t in tail_prefixes)
        specs = [s_ for s_ in specs if not is_tail(s_[0])] + [s_ for s_ in specs if is_tail(s_[0])]
    return specs, gview


def make_arena(model, device, tail_prefixes: Sequence[str] = ()):
    """(FlatBuckets in the kernels' layouts, name -> view-as-parameter function, number of leading buckets that hold only `tail` parameters)."""
    specs, gview = arena_specs(model, tail_prefixes)
    tail = [n for n, _ in specs if any(n.startswith(t) for t in tail_prefixes)]
    arena = FlatBuckets(specs, device, close_after=tail[:1])
    n_tail = 0
    for lay in arena.layout:
        if tail and all(n in tail for n, _, _ in lay):
            n_tail += 1
        else:
            break
    return arena, gview, n_tail


class _GradBuf:
    """Gradient of one forward buffer: same geometry; `init` = channel ranges already written, `left` = channels not yet consumed."""

    def __init__(self, t: torch.Tensor, channels: int):
        self.t, self.init, self.left = t, [], channels


class TrainPlan:
    """Forward + backward launch plans of the canonical model for one (batch shape, compute dtype, BatchNorm-mode tuple)."""

    OUT_NAMES = ("det", "seg", "mc", "protos", "logits")

    def __init__(self, model, shape, device, code: int, tail_prefixes: Sequence[str] = ()):
        self.m, self.code, self.dt, self.device = model, code, TORCH_DTYPE[code], device
        self.tail_prefixes = tuple(tail_prefixes)
        self.lib = L.load()
        self.shape = tuple(shape)
        self.ws = WsPool(device)
        self.fwd = TPlan(device, self.ws)
        self.tape: List = []
        self.prep: List[dict] = []
        self.train_bns: List[nn.BatchNorm2d] = []
        self.generation = 0
        self._stats_chunks: List[torch.Tensor] = []
        self.x = torch.empty(self.shape, dtype=torch.float32, device=device)
        self._ones: Dict[int, torch.Tensor] = {}
        self._zeros: Dict[int, torch.Tensor] = {}
        self._build_arena()
        with torch.no_grad():
            self._lower_forward()
        self._finish_prep()
        self._bwd_cache: Dict[Tuple[bool, ...], Tuple[TPlan, List[str]]] = {}
        self.param_ptrs = [p.data_ptr() for p in model.parameters()]
        self.use_lanes = os.environ.get("MTBT_TRAIN_LANES", "1") == "1"

    def reload_env(self):
        """Re-read MTBT_TRAIN_LANES / MTBT_LANES / ... (read once per plan, not per step) -- for tests and tools that flip them on a live plan."""
        self.use_lanes = os.environ.get("MTBT_TRAIN_LANES", "1") == "1"
        for plan in [self.fwd] + list(self._bwd_cache.values()):
            plan.reload_env()
        return self

    # ------------------------------------------------------------------------------------------------------------------
    # parameter-gradient arena (kernel layouts)
    # ------------------------------------------------------------------------------------------------------------------
    def _build_arena(self):
        self.arena, self._gview, self.n_tail_buckets = make_arena(self.m, self.device, self.tail_prefixes)
        self.pname = {id(p): n for n, p in self.m.named_parameters()}

    def pg(self, p: torch.Tensor) -> torch.Tensor:
        """Gradient slot of parameter p (kernel layout, fp32, inside a flat bucket); called while a backward plan is being emitted, it
        also records that this plan writes the slot."""
        name = self.pname[id(p)]
        self._touched.add(name)
        return self.arena.views[name]

    def param_grad(self, name: str) -> torch.Tensor:
        v = self.arena.views[name]
        back = self._gview[name]
        return back(v) if back is not None else v

    # ------------------------------------------------------------------------------------------------------------------
    # weight preparation table
    # ------------------------------------------------------------------------------------------------------------------
    def prep_w(self, src: torch.Tensor, dims, sstrides, *, flips=(0, 0, 0, 0), scale0=None, scale1=None, dtype=None, src_dim3=0, what="weight"):
        """Register a packed weight: returns the destination tensor (dense `dims`, compute dtype) that `mtbt_weight_prep` fills every step."""
        _param_ptr(src, what)
        dst = torch.zeros(tuple(dims), dtype=dtype or self.dt, device=self.device)
        self.prep.append(dict(src=src, dst=dst, dims=tuple(dims), ss=tuple(int(v) for v in sstrides), flips=tuple(flips), s0=scale0, s1=scale1,
                              src_dim3=src_dim3))
        return dst

    def w_fwd(self, w: torch.Tensor, **kw):
        """conv weight [K,C,R,S] (any strides) -> forward layout [K,R,S,C]; Linear [K,C] -> [K,1,1,C]."""
        if w.dim() == 2:
            K, Cc = w.shape
            return self.prep_w(w, (K, 1, 1, Cc), (w.stride(0), 0, 0, w.stride(1)), **kw).view(K, Cc)
        K, Cc, R, S = w.shape
        return self.prep_w(w, (K, R, S, Cc), (w.stride(0), w.stride(2), w.stride(3), w.stride(1)), **kw).view(K, R * S * Cc)

    def w_dgrad(self, w: torch.Tensor, k_pad: int = 0, **kw):
        """-> dgrad layout [C, R, S, K] with the taps flipped (K zero-padded to k_pad)."""
        if w.dim() == 2:
            K, Cc = w.shape
            return self.prep_w(w, (Cc, 1, 1, K), (w.stride(1), 0, 0, w.stride(0)), **kw).view(Cc, K)
        K, Cc, R, S = w.shape
        Kp = max(K, k_pad)
        return self.prep_w(w, (Cc, R, S, Kp), (w.stride(1), w.stride(2), w.stride(3), w.stride(0)), flips=(0, 1, 1, 0),
                           src_dim3=K if Kp != K else 0, **kw).view(Cc, R * S * Kp)

    def _finish_prep(self):
        n = len(self.prep)
        table = (L.PrepDesc * n)()
        starts, total = [], 0
        for i, d in enumerate(self.prep):
            e = table[i]
            e.src, e.dst = d["src"].data_ptr(), d["dst"].data_ptr()
            s0, s1 = d["s0"], d["s1"]
            e.scale0, e.scale0_dim = (s0[0].data_ptr(), s0[1]) if s0 else (None, 0)
            e.scale1, e.scale1_dim = (s1[0].data_ptr(), s1[1]) if s1 else (None, 0)
            for q in range(4):
                e.sstride[q], e.dim[q], e.flip[q] = d["ss"][q], d["dims"][q], d["flips"][q]
            e.dst_dtype = L.F32 if d["dst"].dtype == torch.float32 else L.BF16
            e.src_dim3 = d["src_dim3"]
            starts.append(total)
            nel = 1
            for v in d["dims"]:
                nel *= v
            total += self.lib.mtbt_weight_prep_blocks(nel)
        raw = bytes(table)
        self.prep_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.prep_starts = torch.tensor(starts, dtype=torch.int32).to(self.device)
        keep = (self.prep_table, self.prep_starts, tuple(d["src"] for d in self.prep), tuple(d["dst"] for d in self.prep))
        launch = Launch(self.lib.mtbt_weight_prep, (self.prep_table.data_ptr(), self.prep_starts.data_ptr(), n, total), "weight_prep", keep)
        launch.writes = tuple(_region(d["dst"]) for d in self.prep)
        self.fwd.launches.insert(0, launch)

    # ------------------------------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------------------------------
    def new(self, N, H, W, Cc, code=None) -> Act:
        return self.fwd.new(N, H, W, Cc, self.code if code is None else code)

    def ones(self, Cc):
        if Cc not in self._ones:
            self._ones[Cc] = torch.ones(Cc, dtype=torch.float32, device=self.device)
            self._zeros[Cc] = torch.zeros(Cc, dtype=torch.float32, device=self.device)
        return self._ones[Cc], self._zeros[Cc]

    def stats(self, Cc) -> torch.Tensor:
        t = torch.zeros(2 * Cc, dtype=torch.float32, device=self.device)
        self._stats_chunks.append(t)
        return t

    # ---- gradient bookkeeping (used while emitting the backward plan) ----
    def _gkey(self, a: Act):
        return a.buf.data_ptr()

    def _crange(self, a: Act):
        c0 = a.off % a.ld if a.ld != a.C else 0
        return c0, c0 + a.C

    def G(self, a: Act) -> Act:
        """The gradient view matching forward activation `a` (allocated on first use)."""
        gb = self.gmap.get(self._gkey(a))
        if gb is None:
            t = self.bwd.pool.get(tuple(a.buf.shape), self.dt)
            gb = self.gmap[self._gkey(a)] = _GradBuf(t, a.buf.shape[-1])
        return Act(gb.t, a.off, a.N, a.H, a.W, a.C, a.ld, a.bs)

    def has_grad(self, a: Act) -> bool:
        gb = self.gmap.get(self._gkey(a))
        if gb is None:
            return False
        c0, c1 = self._crange(a)
        return any(lo < c1 and c0 < hi for lo, hi in gb.init)

    def acc(self, a: Act) -> bool:
        """True if a's channels already hold a gradient (the writer must accumulate); marks them written."""
        self.G(a)
        gb = self.gmap[self._gkey(a)]
        c0, c1 = self._crange(a)
        covered = [r for r in gb.init if r[0] < c1 and c0 < r[1]]
        if covered:
            lo, hi = min(r[0] for r in covered), max(r[1] for r in covered)
            if not (lo <= c0 and c1 <= hi and sum(r[1] - r[0] for r in covered) >= c1 - c0):
                raise RuntimeError("partially initialised gradient region")
            return True
        gb.init.append((c0, c1))
        return False

    def done(self, a: Act):
        """a's producer has consumed its gradient: release the buffer once all channels are consumed."""
        gb = self.gmap.get(self._gkey(a))
        if gb is None:
            return
        gb.left -= a.C
        if gb.left <= 0:
            self.bwd.release_buf(gb.t)
            del self.gmap[self._gkey(a)]

    def alias_grad(self, dst_fwd: Act, src_fwd: Act):
        """grad(dst_fwd) := the buffer of grad(src_fwd) (a residual connection: d x = d y + ...); ownership moves."""
        assert dst_fwd.dense and src_fwd.dense and tuple(dst_fwd.buf.shape) == tuple(src_fwd.buf.shape)
        gb = self.gmap.pop(self._gkey(src_fwd))
        gb.left = dst_fwd.C
        gb.init = [(0, dst_fwd.C)]
        self.gmap[self._gkey(dst_fwd)] = gb

    # ------------------------------------------------------------------------------------------------------------------
    # building blocks: forward launches + a backward closure on the tape
    # ------------------------------------------------------------------------------------------------------------------
    def _dgrad(self, d_raw: Act, wd, x: Act, R, pad, name, stride=1, **kw):
        """dx (+)= conv(d_raw, wd) into grad(x) through the forward kernel (stride-1 'same' convs and 1x1)."""
        gx = self.G(x)
        acc = self.acc(x)
        self.bwd.conv2(d_raw, wd, gx, R=R, S=R, stride=1, pad=R - 1 - pad, res=gx if acc else None, name=name + ".dgrad", **kw)

    def conv_bn_act(self, x: Act, mod, y: Optional[Act], act, name: str) -> Act:
        """ConvBlock (main_model.py:113-141, conv bias) / ultralytics Conv (no bias): conv -> BatchNorm -> activation."""
        conv, bn = mod.conv, mod.bn
        k, K = conv.kernel_size[0], conv.out_channels
        wf, wd = self.w_fwd(conv.weight), self.w_dgrad(conv.weight)
        raw = self.new(x.N, x.H, x.W, K)
        bias = _dense_vec(conv.bias, name + ".bias") if conv.bias is not None else None
        running = not bn.training
        # batch statistics: the conv's own epilogue accumulates sum / sum of squares of what it stores (about the running mean)
        sums = None
        if not running and self.fwd.colsum_ok(K) and FUSED_BN_STATS:
            self.fwd.conv2(x, wf, raw, R=k, S=k, stride=1, pad=k // 2, shift=bias, name=name, colsum=True, colsum_sq=True, colsum_shift=bn.running_mean)
            sums = self.fwd.cs_layout
        else:
            self.fwd.conv(x, wf, raw, R=k, S=k, stride=1, pad=k // 2, shift=bias, name=name)
        if y is None:
            y = self.new(x.N, x.H, x.W, K)
        st = self.stats(K)
        if not running:
            self.train_bns.append(bn)
        self.fwd.bn_forward(raw, y, bn, act, st, running, name + ".bn", sums=sums)

        def bwd():
            if not self.has_grad(y):
                return
            d_raw = self.bwd.new(x.N, x.H, x.W, K, self.code)
            self.bwd.bn_backward(self.G(y), raw, st, bn, act, running, d_raw, self.pg(bn.weight), self.pg(bn.bias), name + ".bn.bwd")
            self.done(y)
            # bias gradient: running statistics -> sum_p d_raw, from the weight-gradient launch itself.  Batch statistics: sum_p d_raw = 0
            # EXACTLY (a bias in front of a batch-statistic BatchNorm cannot move the output), so the slot keeps the arena's zero --
            # autograd returns rounding noise of the order 1e-9 there
            slot = self.pg(conv.bias) if bias is not None else None
            self.bwd.wgrad(x, d_raw, self.pg(conv.weight), R=k, S=k, pad=k // 2, dbias=slot if running else None, name=name + ".wgrad")
            self._dgrad(d_raw, wd, x, k, k // 2, name)
            self.bwd.release(d_raw)
        self.tape.append(bwd)
        return y

    def conv_out(self, x: Act, conv: nn.Conv2d, y: Act, dy_src, name: str):
        """Head output conv (Conv2d 1x1 + bias, no BN) writing a channel slice of an fp32 output map.  `dy_src()` returns the dense
        compute-dtype gradient of that slice (channels zero-padded to a multiple of 8) or None when the output has no gradient."""
        K = conv.out_channels
        Kp = K if K % 8 == 0 else CLS_PAD
        wf, wd = self.w_fwd(conv.weight), self.w_dgrad(conv.weight, k_pad=Kp)
        self.fwd.conv(x, wf, y, shift=_dense_vec(conv.bias, name + ".bias"), name=name)

        def bwd():
            dy = dy_src()
            if dy is None:
                return
            self.bwd.wgrad(x, dy, self.pg(conv.weight), R=1, S=1, pad=0, dbias=self.pg(conv.bias), name=name + ".wgrad")
            self._dgrad(dy, wd, x, 1, 0, name)
        self.tape.append(bwd)

    def dw_bn_act(self, x: Act, mod, name: str) -> Act:
        """ultralytics DWConv: depthwise 3x3 -> BatchNorm -> SiLU (Detect.cv3, main_model.py:324 [ultralytics])."""
        conv, bn = mod.conv, mod.bn
        Cc = x.C
        w = conv.weight                                                     # [C,1,3,3]
        taps = self.prep_w(w, (1, 3, 3, Cc), (0, w.stride(2), w.stride(3), w.stride(0))).view(9, Cc)
        taps_f = self.prep_w(w, (1, 3, 3, Cc), (0, w.stride(2), w.stride(3), w.stride(0)), flips=(0, 1, 1, 0)).view(9, Cc)
        one, zero = self.ones(Cc)
        raw, y = self.new(x.N, x.H, x.W, Cc), self.new(x.N, x.H, x.W, Cc)
        self.fwd.dwconv_t(x, taps, raw, 3, scale=one, shift=zero, name=name)
        st = self.stats(Cc)
        running = not bn.training
        if not running:
            self.train_bns.append(bn)
        self.fwd.bn_forward(raw, y, bn, L.ACT_SILU, st, running, name + ".bn")

        def bwd():
            if not self.has_grad(y):
                return
            d_raw = self.bwd.new(x.N, x.H, x.W, Cc, self.code)
            self.bwd.bn_backward(self.G(y), raw, st, bn, L.ACT_SILU, running, d_raw, self.pg(bn.weight), self.pg(bn.bias), name + ".bn.bwd")
            self.done(y)
            self.bwd.dw_wgrad(x, d_raw, self.pg(conv.weight), 3, name + ".wgrad")
            gx = self.G(x)
            acc = self.acc(x)
            self.bwd.dwconv_t(d_raw, taps_f, gx, 3, scale=one, shift=zero, res=gx if acc else None, name=name + ".dgrad")
            self.bwd.release(d_raw)
        self.tape.append(bwd)
        return y

    def c2f(self, x: Act, mod, name: str) -> Act:
        """main_model.py:144-173, concat-free: every branch reads / writes channel slices of one [N,H,W,(2+n)c] buffer."""
        c, n = mod.c, len(mod.m)
        cat = self.new(x.N, x.H, x.W, (2 + n) * c)
        self.conv_bn_act(x, mod.cv1, cat.slice(0, 2 * c), L.ACT_SILU, name + ".cv1")
        prev = cat.slice(c, c)
        for i, b in enumerate(mod.m):
            if b.add:
                raise NotImplementedError("Bottleneck shortcut is never enabled by the reference model")
            t = self.conv_bn_act(prev, b.cv1, None, L.ACT_SILU, f"{name}.m.{i}.cv1")
            dst = cat.slice((2 + i) * c, c)
            self.conv_bn_act(t, b.cv2, dst, L.ACT_SILU, f"{name}.m.{i}.cv2")
            prev = dst
        return self.conv_bn_act(cat, mod.cv2, None, L.ACT_SILU, name + ".cv2")

    def dw_pointwise(self, x: Act, mod, name: str) -> Act:
        """DepthwiseConvBlock (main_model.py:62-102, k = 1): per-channel scale v -> pointwise W -> BatchNorm -> ELU, as ONE GEMM with
        W' = W * v (columns); the parameter gradients come back through mtbt_scale_grad."""
        dwv = mod.depthwise.weight                                          # [C,1,1,1]
        pw = mod.pointwise.weight                                           # [K,C,1,1]
        K, Cc = pw.shape[0], pw.shape[1]
        vec = dwv.view(-1)
        _dense_vec(dwv, name + ".depthwise.weight"), _dense_vec(pw, name + ".pointwise.weight")
        wf = self.prep_w(pw, (K, 1, 1, Cc), (pw.stride(0), 0, 0, pw.stride(1)), scale0=(vec, 3)).view(K, Cc)
        wd = self.prep_w(pw, (Cc, 1, 1, K), (pw.stride(1), 0, 0, pw.stride(0)), scale0=(vec, 0)).view(Cc, K)
        bn = mod.bn
        raw, y = self.new(x.N, x.H, x.W, K), self.new(x.N, x.H, x.W, K)
        running = not bn.training
        sums = None
        if not running and self.fwd.colsum_ok(K) and FUSED_BN_STATS:
            self.fwd.conv2(x, wf, raw, name=name, colsum=True, colsum_sq=True, colsum_shift=bn.running_mean)
            sums = self.fwd.cs_layout
        else:
            self.fwd.conv(x, wf, raw, name=name)
        st = self.stats(K)
        if not running:
            self.train_bns.append(bn)
        self.fwd.bn_forward(raw, y, bn, L.ACT_ELU, st, running, name + ".bn", sums=sums)
        gtmp = torch.empty(K, Cc, dtype=torch.float32, device=self.device)

        def bwd():
            if not self.has_grad(y):
                return
            d_raw = self.bwd.new(x.N, x.H, x.W, K, self.code)
            self.bwd.bn_backward(self.G(y), raw, st, bn, L.ACT_ELU, running, d_raw, self.pg(bn.weight), self.pg(bn.bias), name + ".bn.bwd")
            self.done(y)
            self.bwd.wgrad(x, d_raw, gtmp, R=1, S=1, pad=0, name=name + ".wgrad")
            dW, dv = self.pg(pw), self.pg(dwv)
            self.bwd.raw(self.lib.mtbt_scale_grad, (1, gtmp.data_ptr(), pw.data_ptr(), vec.data_ptr(), None, None, dW.data_ptr(), dv.data_ptr(), None,
                                                    K, Cc, 0), name + ".scale_grad", keep=(gtmp, pw, vec, dW, dv), reads=[gtmp], writes=[dW, dv])
            self._dgrad(d_raw, wd, x, 1, 0, name)
            self.bwd.release(d_raw)
        self.tape.append(bwd)
        return y

    # -- ConvNeXt-T feature extractor (timm, main_model.py:21-26,33-38) --
    def features(self, body):
        N, _, H, W = self.shape
        T = self.code
        st0 = body.stem_0
        a, raw0 = self.new(N, H // 4, W // 4, M.DIMS[0]), self.new(N, H // 4, W // 4, M.DIMS[0])
        w0 = _dense_vec(st0.weight, "stem_0.weight")
        args = (self.x.data_ptr(), w0.data_ptr(), st0.bias.data_ptr(), body.stem_1.weight.data_ptr(), body.stem_1.bias.data_ptr(),
                C.c_float(body.stem_1.eps), a.ptr, raw0.ptr, N, H, W, a.C, T)
        self.fwd.raw(self.lib.mtbt_stem_conv4x4_ln_train, args, "backbone.body.stem", keep=(self.x, w0, a.buf, raw0.buf), reads=[self.x], writes=[a, raw0])

        def stem_bwd(a=a, raw0=raw0):
            if not self.has_grad(a):
                return
            d_raw = self.bwd.new(a.N, a.H, a.W, a.C, T)
            self.bwd.ln_backward_params(raw0, self.G(a), body.stem_1.weight, body.stem_1.eps, d_raw, False, self.pg(body.stem_1.weight),
                                        self.pg(body.stem_1.bias), "stem.ln.bwd")
            self.done(a)
            self.bwd.channel_sum(d_raw, self.pg(st0.bias), name="stem.dbias")
            nbytes = self.lib.mtbt_stem_wgrad_workspace_bytes(a.C)
            dW = self.pg(st0.weight)
            self.bwd.raw(self.lib.mtbt_stem_wgrad, (self.x.data_ptr(), d_raw.ptr, dW.data_ptr(), N, H, W, a.C, T, 0, self.bwd._ws(nbytes), nbytes),
                         "stem.wgrad", keep=(self.x, d_raw.buf, dW), reads=[self.x, d_raw], writes=[dW, self.bwd.cur_ws])
            self.bwd.est(1.0 * N * 3 * H * W * 4)
            self.bwd.release(d_raw)
        self.tape.append(stem_bwd)

        feats = []
        for si in range(4):
            stg = getattr(body, f"stages_{si}")
            nm = f"backbone.body.stages_{si}"
            if si > 0:
                a = self.downsample(a, stg.downsample[0], stg.downsample[1], nm + ".downsample")
            for bi, blk in enumerate(stg.blocks):
                a = self.cn_block(a, blk, f"{nm}.blocks.{bi}")
            if si >= 1:
                feats.append(a)
        return feats

    def downsample(self, a: Act, ln, cv, name) -> Act:
        """timm stage transition: LayerNorm2d -> Conv2d(2, stride 2, bias)."""
        T = self.code
        t = self.new(a.N, a.H, a.W, a.C)
        self.fwd.layernorm(a, _dense_vec(ln.weight, name), _dense_vec(ln.bias, name), ln.eps, t, name=name + ".0")
        K = cv.out_channels
        nxt = self.new(a.N, a.H // 2, a.W // 2, K)
        w = cv.weight
        wf = self.w_fwd(w)
        wd = self.prep_w(w, (2, 2, a.C, K), (w.stride(2), w.stride(3), w.stride(1), w.stride(0))).view(4 * a.C, K)   # OUT_CONVT2X2 rows (r,s,c)
        self.fwd.conv(t, wf, nxt, R=2, S=2, stride=2, pad=0, shift=_dense_vec(cv.bias, name), name=name + ".1")

        def bwd():
            if not self.has_grad(nxt):
                return
            dy = self.G(nxt)
            self.bwd.wgrad(t, dy, self.pg(cv.weight), R=2, S=2, pad=0, stride=2, dbias=self.pg(cv.bias), name=name + ".1.wgrad")
            d_t = self.bwd.new(a.N, a.H, a.W, a.C, T)
            self.bwd.conv2(dy, wd, d_t, out_mode=L.OUT_CONVT2X2, name=name + ".1.dgrad")
            self.done(nxt)
            ga = self.G(a)
            acc = self.acc(a)
            self.bwd.ln_backward_params(a, d_t, ln.weight, ln.eps, ga, acc, self.pg(ln.weight), self.pg(ln.bias), name + ".0.bwd")
            self.bwd.release(d_t)
        self.tape.append(bwd)
        return nxt

    def cn_block(self, cur: Act, blk, name) -> Act:
        """timm ConvNeXtBlock: dw 7x7 (+bias) -> LayerNorm -> fc1 -> GELU -> fc2 -> * gamma -> + x."""
        T, d = self.code, cur.C
        N, H, W = cur.N, cur.H, cur.W
        dw = blk.conv_dw.weight                                                      # [d,1,7,7]
        taps = self.prep_w(dw, (1, 7, 7, d), (0, dw.stride(2), dw.stride(3), dw.stride(0))).view(49, d)
        taps_f = self.prep_w(dw, (1, 7, 7, d), (0, dw.stride(2), dw.stride(3), dw.stride(0)), flips=(0, 1, 1, 0)).view(49, d)
        fc1, fc2, gamma = blk.mlp.fc1, blk.mlp.fc2, blk.gamma
        for p_, n_ in ((blk.conv_dw.bias, "conv_dw.bias"), (blk.norm.weight, "norm.weight"), (blk.norm.bias, "norm.bias"), (fc1.bias, "fc1.bias"),
                       (fc2.bias, "fc2.bias"), (gamma, "gamma"), (fc2.weight, "fc2.weight")):
            _dense_vec(p_, f"{name}.{n_}")
        fused_mlp = T == L.BF16 and d in (96, 192) and FUSED_TRAIN_MLP and fc1.weight.is_contiguous() and fc2.weight.is_contiguous()   # (below)
        w1d = self.w_dgrad(fc1.weight)
        w1f, w2f = (None, None) if fused_mlp else (self.w_fwd(fc1.weight), self.w_fwd(fc2.weight))
        w2d = self.prep_w(fc2.weight, (4 * d, 1, 1, d), (fc2.weight.stride(1), 0, 0, fc2.weight.stride(0)), scale0=(gamma, 3)).view(4 * d, d)
        shift2 = self.prep_w(fc2.bias, (1, 1, 1, d), (0, 0, 0, 1), scale0=(gamma, 3), dtype=torch.float32).view(d)   # gamma * b2
        r, t = self.new(N, H, W, d), self.new(N, H, W, d)
        # bf16, d = 96 / 192: fc1 -> GELU -> fc2 as ONE launch that keeps only the fc1 pre-activation (mtbt_convnext_mlp_fused_train): the 4d-wide
        # activated tensor is neither written nor read back (stage 0 at batch 32: 2 x 629 MB per block) and is not kept for the backward --
        # the fc2 weight gradient re-applies GELU while it stages hpre.  Wider stages keep the two GEMMs (d = 384: the pair kernel has no
        # registers left for the store; d = 768: no fused kernel).  MTBT_TRAIN_FUSED_MLP=0 restores the two GEMMs everywhere.
        hpre = self.new(N, H, W, 4 * d)
        h = None if fused_mlp else self.new(N, H, W, 4 * d)
        y = self.new(N, H, W, d)
        self.fwd.dwconv_t(cur, taps, t, 7, bias=blk.conv_dw.bias, lnw=blk.norm.weight, lnb=blk.norm.bias, eps=blk.norm.eps, raw=r,
                          name=name + ".conv_dw+norm")
        # bf16: GELU as x * Phi(x) with the polynomial Phi of the inference path (|error| <= 2.3e-4, below bf16 resolution) and, in the
        # backward, the EXACT derivative of that polynomial -- no erf / exp in either epilogue (round 3: the erf forms made the epilogues
        # of fc1 and of fc2-dgrad VALU-bound at stage 2); the fp32 parity mode keeps the erf forms
        poly = T != L.F32
        if fused_mlp:
            nj = 4 * d // 32
            # staged row 16 b + 4 g + e of a 32-row chunk = hidden unit 8 g + 4 b + e (header): [chunk][b][g][e * d + column] over the row-major weight
            w1p = self.prep_w(fc1.weight, (nj, 2, 4, 4 * d), (32 * d, 4 * d, 8 * d, 1)).view(4 * d, d)
            b1p = self.prep_w(fc1.bias, (nj, 2, 4, 4), (32, 4, 8, 1), dtype=torch.float32, what="bias").view(4 * d)
            w2g = self.prep_w(fc2.weight, (d, 1, 1, 4 * d), (fc2.weight.stride(0), 0, 0, fc2.weight.stride(1)), scale0=(gamma, 0)).view(d, 4 * d)
            self.fwd.mlp_fused_train(t, cur, w1p, b1p, w2g, shift2, y, hpre, name + ".mlp(fused)")
        else:
            self.fwd.conv2(t, w1f, h, shift=fc1.bias, act=L.ACT_GELU_POLY if poly else L.ACT_GELU, y2=hpre, name=name + ".mlp.fc1")
            self.fwd.conv(h, w2f, y, scale=gamma, shift=shift2, res=cur, name=name + ".mlp.fc2")
        gtmp = torch.empty(d, 4 * d, dtype=torch.float32, device=self.device)
        ssum = torch.empty(d, dtype=torch.float32, device=self.device)
        one, zero = self.ones(d)

        def bwd():
            if not self.has_grad(y):
                return
            dy = self.G(y)
            # (sum_p dy and the fc1 bias gradient stay separate channel sums: in these GEMM-shaped weight gradients EVERY workgroup would
            #  carry the 25 % extra MFMAs of the fused form -- measured +2.0 ms against the 2.8 ms of the two sums)
            self.bwd.channel_sum(dy, ssum, name=name + ".sum_dy")
            if fused_mlp:
                self.bwd.wgrad(hpre, dy, gtmp, R=1, S=1, pad=0, x_act=L.ACT_GELU_POLY, name=name + ".fc2.wgrad")
            else:
                self.bwd.wgrad(h, dy, gtmp, R=1, S=1, pad=0, name=name + ".fc2.wgrad")
            dW2, dg, db2 = self.pg(fc2.weight), self.pg(gamma), self.pg(fc2.bias)
            self.bwd.raw(self.lib.mtbt_scale_grad, (0, gtmp.data_ptr(), fc2.weight.data_ptr(), gamma.data_ptr(), fc2.bias.data_ptr(), ssum.data_ptr(),
                                                    dW2.data_ptr(), dg.data_ptr(), db2.data_ptr(), d, 4 * d, 0), name + ".fc2.scale_grad",
                         keep=(gtmp, ssum, dW2, dg, db2), reads=[gtmp, ssum], writes=[dW2, dg, db2])
            d_hpre = self.bwd.new(N, H, W, 4 * d, T)
            # (d fc1.bias = sum_p d_hpre comes out of the same launch: column sums in the epilogue instead of a pass over the 4d-wide tensor)
            fused_db = self.bwd.colsum_ok(4 * d) and FUSED_BN_STATS
            self.bwd.conv2(dy, w2d, d_hpre, act=L.ACT_DGELU_POLY if poly else L.ACT_DGELU, res=hpre, name=name + ".fc2.dgrad*gelu'", colsum=self.pg(fc1.bias).view(-1) if fused_db else None)
            self.bwd.wgrad(t, d_hpre, self.pg(fc1.weight), R=1, S=1, pad=0, name=name + ".fc1.wgrad")
            if not fused_db:
                self.bwd.channel_sum(d_hpre, self.pg(fc1.bias), name=name + ".fc1.dbias")
            d_t = self.bwd.new(N, H, W, d, T)
            self.bwd.conv2(d_hpre, w1d, d_t, name=name + ".fc1.dgrad")
            self.bwd.release(d_hpre)
            d_r = self.bwd.new(N, H, W, d, T)
            self.bwd.ln_backward_params(r, d_t, blk.norm.weight, blk.norm.eps, d_r, False, self.pg(blk.norm.weight), self.pg(blk.norm.bias), name + ".norm.bwd")
            self.bwd.release(d_t)
            self.bwd.dw_wgrad(cur, d_r, self.pg(dw), 7, name + ".conv_dw.wgrad", dbias=self.pg(blk.conv_dw.bias))
            # residual: d cur = d y + dwconv^T(d_r) -- grad(y)'s buffer BECOMES grad(cur), the depthwise dgrad accumulates into it
            if self.has_grad(cur):
                raise NotImplementedError("a ConvNeXt block input with a second consumer")
            self.alias_grad(cur, y)
            gc = self.G(cur)
            self.bwd.dwconv_t(d_r, taps_f, gc, 7, scale=one, shift=zero, res=gc, name=name + ".conv_dw.dgrad")
            self.bwd.release(d_r)
        self.tape.append(bwd)
        return y

    # -- BiFPN (main_model.py:176-296) --
    def neck(self, c3, c4, c5):
        nk = self.m.neck
        p3 = self.conv_bn_act(c3, nk.p3_proj, None, L.ACT_SILU, "neck.p3_proj")
        p4 = self.conv_bn_act(c4, nk.p4_proj, None, L.ACT_SILU, "neck.p4_proj")
        p5 = self.conv_bn_act(c5, nk.p5_proj, None, L.ACT_SILU, "neck.p5_proj")
        for ui, u in enumerate(nk.bifpn_units):
            nm = f"neck.bifpn_units.{ui}"
            # normalised fusion weights on the device, transposed: node j's weights are wn[j*n .. j*n+n)
            wn1 = torch.zeros(4, dtype=torch.float32, device=self.device)
            wn2 = torch.zeros(6, dtype=torch.float32, device=self.device)
            dwn1, dwn2 = torch.zeros_like(wn1), torch.zeros_like(wn2)
            w1, w2 = _dense_vec(u.w1, nm + ".w1"), _dense_vec(u.w2, nm + ".w2")
            self.fwd.raw(self.lib.mtbt_bifpn_norm_weights, (w1.data_ptr(), 2, C.c_float(u.eps), wn1.data_ptr()), nm + ".w1.norm", keep=(w1, wn1), writes=[wn1])
            self.fwd.raw(self.lib.mtbt_bifpn_norm_weights, (w2.data_ptr(), 3, C.c_float(u.eps), wn2.data_ptr()), nm + ".w2.norm", keep=(w2, wn2), writes=[wn2])

            def norm_bwd(u=u, w1=w1, w2=w2, dwn1=dwn1, dwn2=dwn2, nm=nm):
                if not getattr(self, "_fuse_touched", {}).get(nm):
                    return
                g1, g2 = self.pg(u.w1), self.pg(u.w2)
                self.bwd.raw(self.lib.mtbt_bifpn_norm_weights_backward, (w1.data_ptr(), 2, C.c_float(u.eps), dwn1.data_ptr(), g1.data_ptr(), 0),
                             nm + ".w1.norm.bwd", keep=(w1, dwn1, g1), reads=[dwn1], writes=[g1])
                self.bwd.raw(self.lib.mtbt_bifpn_norm_weights_backward, (w2.data_ptr(), 3, C.c_float(u.eps), dwn2.data_ptr(), g2.data_ptr(), 0),
                             nm + ".w2.norm.bwd", keep=(w2, dwn2, g2), reads=[dwn2], writes=[g2])
            self.tape.append(norm_bwd)

            def node(inputs, wn, dwn, col, n, modes, like, conv, cf, tag, nm=nm):
                s = self.new(like.N, like.H, like.W, like.C)
                a = self.fwd.fuse(inputs, [0.0] * len(inputs), modes, s, name=f"{nm}.{tag}.fuse")
                a.wgt_dev = wn.data_ptr() + 4 * col * n
                self.fwd.launches[-1].keep += (wn,)
                self.fwd.launches[-1].reads += (_region(wn),)

                def bwd():
                    if not self.has_grad(s):
                        return
                    self.__dict__.setdefault("_fuse_touched", {})[nm] = True
                    ds = self.G(s)
                    nbytes = self.lib.mtbt_bifpn_fuse_backward_workspace_bytes()
                    for i, (xin, mode) in enumerate(zip(inputs, modes)):
                        gx = self.G(xin)
                        acc = self.acc(xin)
                        # the same tensor may enter a node twice (p5_out: w*p5 + w*p5 + ..., main_model.py:236-240): its weight gradients are separate
                        args = (ds.ptr, xin.ptr, mode, wn.data_ptr() + 4 * (col * n + i), gx.ptr, int(acc), dwn.data_ptr() + 4 * (col * n + i), 0,
                                s.N, s.H, s.W, s.C, self.code, self.bwd._ws(nbytes), nbytes)
                        self.bwd.raw(self.lib.mtbt_bifpn_fuse_backward, args, f"{nm}.{tag}.fuse.bwd{i}", keep=(ds.buf, xin.buf, wn, gx.buf, dwn),
                                     reads=[ds, xin, wn] + ([gx] if acc else []), writes=[gx, dwn, self.bwd.cur_ws])
                        self.bwd.est(2.0 * s.N * s.H * s.W * s.C * ESIZE[self.code])
                    self.done(s)
                self.tape.append(bwd)
                dd = self.dw_pointwise(s, conv, f"{nm}.{tag}_conv")
                return self.c2f(dd, cf, f"{nm}.{tag}_cf")

            p4_td = node([p4, p5], wn1, dwn1, 0, 2, [L.RES_ID, L.RES_UP_BILINEAR], p4, u.p4_td_conv, u.p4_td_cf, "p4_td")
            p3_td = node([p3, p4_td], wn1, dwn1, 1, 2, [L.RES_ID, L.RES_UP_BILINEAR], p3, u.p3_td_conv, u.p3_td_cf, "p3_td")
            p4_out = node([p4, p4_td, p3_td], wn2, dwn2, 0, 3, [L.RES_ID, L.RES_ID, L.RES_DOWN_MEAN], p4, u.p4_out_conv, u.p4_out_cf, "p4_out")
            p5_out = node([p5, p5, p4_out], wn2, dwn2, 1, 3, [L.RES_ID, L.RES_ID, L.RES_DOWN_MEAN], p5, u.p5_out_conv, u.p5_out_cf, "p5_out")
            p3, p4, p5 = p3_td, p4_out, p5_out
        return p3, p4, p5

    # -- neck of the oldest variant (src/model.py:27-93): lateral Convs, WeightedAdd (ADDS its normalised weights), DWConv 3x3 nodes --
    def neck_v0(self, f3, f4, f5):
        nk = self.m.neck
        p3 = self.conv_bn_act(f3, nk.lat3, None, L.ACT_SILU, "neck.lat3")
        p4 = self.conv_bn_act(f4, nk.lat4, None, L.ACT_SILU, "neck.lat4")
        p5 = self.conv_bn_act(f5, nk.lat5, None, L.ACT_SILU, "neck.lat5")
        for ui, u in enumerate(nk.units):
            nm = f"neck.units.{ui}"

            def node(add, inputs, modes, like: Act, key, nm=nm, u=u):
                # y = sum_i (w~_i + resample_i(x_i)), w~ = relu(w) / (sum relu(w) + eps): the weights are parameters, normalised on the device
                n = len(inputs)
                w = _dense_vec(add.w, f"{nm}.add_{key}.w")
                wn = torch.zeros(n, dtype=torch.float32, device=self.device)
                self.fwd.raw(self.lib.mtbt_wadd_norm_weights, (w.data_ptr(), n, C.c_float(add.eps), wn.data_ptr()), f"{nm}.add_{key}.norm", keep=(w, wn),
                             reads=[w], writes=[wn])
                s_ = self.new(like.N, like.H, like.W, like.C)
                a = self.fwd.fuse(inputs, [0.0] * n, modes, s_, bug=True, name=f"{nm}.add_{key}")
                a.wgt_dev = wn.data_ptr()
                self.fwd.launches[-1].keep += (wn,)
                self.fwd.launches[-1].reads += (_region(wn),)

                def bwd():
                    if not self.has_grad(s_):
                        return
                    ds = self.G(s_)
                    # d w: the node adds s / (s + eps) to EVERY element, so d w_j = [w_j > 0] eps / (s + eps)^2 sum(dy)
                    cs = torch.zeros(s_.C, dtype=torch.float32, device=self.device)
                    self.bwd.channel_sum(ds, cs, name=f"{nm}.add_{key}.dysum")
                    gw = self.pg(add.w)
                    self.bwd.raw(self.lib.mtbt_wadd_norm_weights_backward, (w.data_ptr(), n, C.c_float(add.eps), cs.data_ptr(), s_.C, gw.data_ptr(), 0),
                                 f"{nm}.add_{key}.norm.bwd", keep=(w, cs, gw), reads=[w, cs], writes=[gw])
                    for i, (xin, mode) in enumerate(zip(inputs, modes)):
                        gx = self.G(xin)
                        acc = self.acc(xin)
                        args = (ds.ptr, xin.ptr, mode, gx.ptr, int(acc), s_.N, s_.H, s_.W, s_.C, self.code)
                        self.bwd.raw(self.lib.mtbt_resample_backward, args, f"{nm}.add_{key}.bwd{i}", keep=(ds.buf, xin.buf, gx.buf),
                                     reads=[ds, xin] + ([gx] if acc else []), writes=[gx])
                        self.bwd.est(2.0 * s_.N * s_.H * s_.W * s_.C * ESIZE[self.code])
                    self.done(s_)
                self.tape.append(bwd)
                return self.dw_bn_act(s_, u.conv[key], f"{nm}.conv.{key}")

            p4_td = node(u.add_p4_td, [p4, p5], [L.RES_ID, L.RES_UP_NEAREST], p4, "p4_td")
            p3_td = node(u.add_p3_td, [p3, p4_td], [L.RES_ID, L.RES_UP_NEAREST], p3, "p3_td")
            p4_out = node(u.add_p4_out, [p4, p4_td, p3_td], [L.RES_ID, L.RES_ID, L.RES_MAXPOOL], p4, "p4_out")
            p5_out = node(u.add_p5_out, [p5, p4_out], [L.RES_ID, L.RES_MAXPOOL], p5, "p5_out")
            p3, p4, p5 = p3_td, p4_out, p5_out
        return p3, p4, p5

    # -- heads [ultralytics Detect / Segment / Proto] --
    def _f32(self, *shape) -> torch.Tensor:
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def det_branch(self, feats, head, tag, key):
        """cv2 (box) and cv3 (cls) of every level write side by side into one [N,h,w,no] fp32 map; its gradient arrives as an fp32 map of
        the same layout (`self.d_in[key][i]`) and is split into two dense compute-dtype operands."""
        maps, dmaps = [], []
        nb = 4 * head.reg_max
        for i, f in enumerate(feats):
            full = Act.of(self._f32(f.N, f.H, f.W, head.no))
            dfull = Act.of(self._f32(f.N, f.H, f.W, head.no))
            holder = {}

            def split(f=f, dfull=dfull, holder=holder, i=i):
                """(once per backward plan) dense gradient operands of the two output convs"""
                if key not in self.active:
                    return None
                if "box" not in holder:
                    box = self.bwd.new(f.N, f.H, f.W, nb, self.code)
                    cls = self.bwd.new(f.N, f.H, f.W, CLS_PAD if head.nc % 8 else head.nc, self.code)
                    P = f.H * f.W
                    self.bwd.copy_strided(dfull.ptr, L.F32, dfull.bs, dfull.ld, box, f.N, P, nb, nb, (dfull.buf,), f"{tag}.{i}.dbox")
                    self.bwd.copy_strided(dfull.ptr + 4 * nb, L.F32, dfull.bs, dfull.ld, cls, f.N, P, head.nc, cls.C, (dfull.buf,), f"{tag}.{i}.dcls")
                    holder["box"], holder["cls"] = box, cls
                return holder
            sq = head.cv2[i]
            t1 = self.conv_bn_act(f, sq[0], None, L.ACT_SILU, f"{tag}.cv2.{i}.0")
            t2 = self.conv_bn_act(t1, sq[1], None, L.ACT_SILU, f"{tag}.cv2.{i}.1")
            self.conv_out(t2, sq[2], full.slice(0, nb), (lambda split=split: (split() or {}).get("box")), f"{tag}.cv2.{i}.2")
            sq = head.cv3[i]
            d1 = self.dw_bn_act(f, sq[0][0], f"{tag}.cv3.{i}.0.0")
            u1 = self.conv_bn_act(d1, sq[0][1], None, L.ACT_SILU, f"{tag}.cv3.{i}.0.1")
            d2 = self.dw_bn_act(u1, sq[1][0], f"{tag}.cv3.{i}.1.0")
            u2 = self.conv_bn_act(d2, sq[1][1], None, L.ACT_SILU, f"{tag}.cv3.{i}.1.1")
            self.conv_out(u2, sq[2], full.slice(nb, head.nc), (lambda split=split: (split() or {}).get("cls")), f"{tag}.cv3.{i}.2")
            self._holders.append(holder)
            maps.append(full)
            dmaps.append(dfull)
        self.d_in[key] = dmaps
        return maps

    def seg_extras(self, feats, head):
        N = feats[0].N
        A = sum(f.H * f.W for f in feats)
        mc = self._f32(N, A, head.nm)
        dmc = self._f32(N, A, head.nm)
        off = 0
        for i, f in enumerate(feats):
            sq = head.cv4[i]
            t1 = self.conv_bn_act(f, sq[0], None, L.ACT_SILU, f"segment.cv4.{i}.0")
            t2 = self.conv_bn_act(t1, sq[1], None, L.ACT_SILU, f"segment.cv4.{i}.1")
            lvl = Act(mc, off * head.nm, N, f.H, f.W, head.nm, head.nm, A * head.nm)
            holder = {}

            def dsrc(f=f, off=off, holder=holder, i=i):
                if "mc" not in self.active:
                    return None
                if "d" not in holder:
                    d = self.bwd.new(f.N, f.H, f.W, head.nm, self.code)
                    self.bwd.copy_strided(dmc.data_ptr() + 4 * off * head.nm, L.F32, A * head.nm, head.nm, d, f.N, f.H * f.W, head.nm, head.nm, (dmc,),
                                          f"segment.cv4.{i}.dmc")
                    holder["d"] = d
                return holder["d"]
            self.conv_out(t2, sq[2], lvl, dsrc, f"segment.cv4.{i}.2")
            self._holders.append(holder)
            off += f.H * f.W
        self.d_in["mc"] = dmc
        # Proto on P3: Conv 3x3 -> ConvTranspose2d(2, 2, bias) -> Conv 3x3 -> Conv 1x1 (each Conv = conv + BN + SiLU)
        pr, f = head.proto, feats[0]
        t1 = self.conv_bn_act(f, pr.cv1, None, L.ACT_SILU, "segment.proto.cv1")
        up = self.new(f.N, 2 * f.H, 2 * f.W, pr.upsample.out_channels)
        wt = pr.upsample.weight                                               # [Cin, Cout, 2, 2]
        Ci, Co = wt.shape[0], wt.shape[1]
        wf = self.prep_w(wt, (2, 2, Co, Ci), (wt.stride(2), wt.stride(3), wt.stride(1), wt.stride(0))).view(4 * Co, Ci)   # rows (dy,dx,co)
        wd = self.prep_w(wt, (Ci, 2, 2, Co), (wt.stride(0), wt.stride(2), wt.stride(3), wt.stride(1))).view(Ci, 4 * Co)   # 2x2 / stride-2 conv of dY
        ub = _dense_vec(pr.upsample.bias, "segment.proto.upsample.bias")
        bias4 = self.prep_w(ub, (1, 1, 4, Co), (0, 0, 0, 1), dtype=torch.float32).view(4 * Co)
        self.fwd.conv(t1, wf, up, shift=bias4, out_mode=L.OUT_CONVT2X2, name="segment.proto.upsample")

        def up_bwd():
            if not self.has_grad(up):
                return
            dy = self.G(up)
            # dW[ci][dy][dx][co] = sum_p X[p][ci] * dY[2p + (dy,dx)][co]: the 2x2 / stride-2 weight gradient with the operand roles swapped
            self.bwd.wgrad(dy, t1, self.pg(wt), R=2, S=2, pad=0, stride=2, name="segment.proto.upsample.wgrad")
            self.bwd.channel_sum(dy, self.pg(ub), name="segment.proto.upsample.dbias")
            g1 = self.G(t1)
            acc = self.acc(t1)
            self.bwd.conv2(dy, wd, g1, R=2, S=2, stride=2, pad=0, res=g1 if acc else None, name="segment.proto.upsample.dgrad")
            self.done(up)
        self.tape.append(up_bwd)
        t2 = self.conv_bn_act(up, pr.cv2, None, L.ACT_SILU, "segment.proto.cv2")
        pT = self.conv_bn_act(t2, pr.cv3, None, L.ACT_SILU, "segment.proto.cv3")
        # the module hands out fp32 prototypes; their gradient arrives in the compute dtype (d_in["protos"]) and IS grad(pT)
        if self.code == L.F32:
            protos = pT
        else:
            protos = Act.of(self._f32(f.N, 2 * f.H, 2 * f.W, head.nm))
            self.fwd.cast(pT, protos, name="segment.proto.cast")
        self.d_in["protos"] = torch.zeros(tuple(pT.buf.shape), dtype=self.dt, device=self.device)

        def protos_seed():
            if "protos" not in self.active:
                return
            self.gmap[self._gkey(pT)] = gb = _GradBuf(self.d_in["protos"], pT.C)
            gb.init = [(0, pT.C)]
            gb.left = 1 << 30                                               # plan input: never returned to the pool
        self.tape.append(protos_seed)
        return mc, protos

    def cls_head(self, n5: Act):
        fc = self.m.cls_fc
        w, b = _dense_vec(fc.weight, "cls_fc.weight"), _dense_vec(fc.bias, "cls_fc.bias")
        logits = self._f32(n5.N, fc.out_features)
        dlog = self._f32(n5.N, fc.out_features)
        self.fwd.gap_fc(n5, w, b, logits, name="cls_pool+cls_fc")
        pool_ws = self._f32(n5.N, n5.C)
        self.d_in["logits"] = dlog

        def bwd():
            if "logits" not in self.active:
                return
            g5 = self.G(n5)
            acc = self.acc(n5)
            dW, db = self.pg(fc.weight), self.pg(fc.bias)
            args = (n5.ptr, dlog.data_ptr(), w.data_ptr(), g5.ptr, int(acc), dW.data_ptr(), db.data_ptr(), 0, pool_ws.data_ptr(), n5.N, n5.H * n5.W, n5.C,
                    fc.out_features, self.code)
            self.bwd.raw(self.lib.mtbt_gap_fc_backward, args, "cls_fc.bwd", keep=(n5.buf, dlog, w, g5.buf, dW, db, pool_ws),
                         reads=[n5, dlog] + ([g5] if acc else []), writes=[g5, dW, db, pool_ws])
        self.tape.append(bwd)
        return logits

    # ------------------------------------------------------------------------------------------------------------------
    def _lower_forward(self):
        self.d_in: Dict[str, object] = {}
        self._holders: List[dict] = []
        self.active = set()
        m = self.m
        bb = m.backbone
        f3, f4, f5 = self.features(bb.body)
        if hasattr(bb, "c2f_p3"):
            c3 = self.c2f(f3, bb.c2f_p3, "backbone.c2f_p3")
            c4 = self.c2f(f4, bb.c2f_p4, "backbone.c2f_p4")
            c5 = self.c2f(f5, bb.c2f_p5, "backbone.c2f_p5")
            n3, n4, n5 = self.neck(c3, c4, c5)
        else:                       # the src/model.py variant: no adaptors, its own neck
            n3, n4, n5 = self.neck_v0(f3, f4, f5)
        feats = [n3, n4, n5]
        self.det_maps = self.det_branch(feats, m.detect, "detect", "det") if hasattr(m, "detect") else None
        self.seg_maps = self.det_branch(feats, m.segment, "segment", "seg")
        self.mc, self.protos = self.seg_extras(feats, m.segment)
        self.logits = self.cls_head(n5)

    def backward_plan(self, active: Sequence[str]) -> TPlan:
        """The backward launch plan for the set of outputs that carry a gradient (built once per set)."""
        key = tuple(sorted(active))
        hit = self._bwd_cache.get(key)
        if hit is not None:
            return hit
        self.active = set(active)
        self.bwd = TPlan(self.device, self.ws)
        self.bwd.pool.reuse = True
        self.gmap: Dict[int, _GradBuf] = {}
        self.__dict__["_fuse_touched"] = {}
        self._touched = set()
        for h in self._holders:
            h.clear()
        with torch.no_grad():
            for fn in reversed(self.tape):
                fn()
        plan = self.bwd
        plan.written = sorted(self._touched)      # parameters this plan produces a gradient for (the others stay None / zero: SURVEY F13)
        self._bwd_cache[key] = plan
        return plan

    # ------------------------------------------------------------------------------------------------------------------
    # execution
    # ------------------------------------------------------------------------------------------------------------------
    def check_params(self):
        if [p.data_ptr() for p in self.m.parameters()] != self.param_ptrs:
            raise RuntimeError("parameter storage moved since the training plan was lowered (re-lower: model._train_plans.clear())")

    def issue(self, plan: TPlan, marks=None):
        """Spread a plan over the engine's lanes (independent launches -- weight gradients beside the input-gradient chain, the small
        pyramid levels side by side; bit-identical to single-stream execution: no atomics anywhere, tests/test_gpu_train.py), or, with
        MTBT_TRAIN_LANES=0 (read when the plan is built; `reload_env()`), issue it on the current stream.  Measured at batch 32: 80.0 -> 79.3 ms per step."""
        if self.use_lanes:
            return plan.run(marks=marks)
        return plan.run(stream=torch.cuda.current_stream(self.device).cuda_stream, marks=marks)

    def run_forward(self, x: torch.Tensor):
        self.x.copy_(x)
        self.issue(self.fwd)
        self.generation += 1
        if self.train_bns:
            torch._foreach_add_([bn.num_batches_tracked for bn in self.train_bns if bn.num_batches_tracked is not None], 1)
            for bn in self.train_bns:                        # the kernels rewrote running_mean / running_var: plans that folded them are stale
                bn.__dict__["_mtbt_epoch"] = bn.__dict__.get("_mtbt_epoch", 0) + 1

    def run_backward(self, active: Sequence[str]):
        plan = self.backward_plan(active)
        self.issue(plan)
        return plan


def _bn_mode_sig(model) -> Tuple[bool, ...]:
    return tuple(m.training for m in model.modules() if isinstance(m, nn.BatchNorm2d))


class _TrainFn(torch.autograd.Function):
    """`forward(x, "train")` as ONE autograd node: the reference's `total_loss.backward()` (running_main_v3.py:445 via Lightning) lands here
    with the gradients of the head outputs and leaves with one gradient per parameter."""

    @staticmethod
    def forward(ctx, tp: TrainPlan, n_det: int, x: torch.Tensor, *params):
        tp.run_forward(x)
        ctx.tp, ctx.gen, ctx.n_det = tp, tp.generation, n_det
        ctx.set_materialize_grads(False)
        outs = []
        for maps in ([tp.det_maps] if tp.det_maps is not None else []) + [tp.seg_maps]:
            outs += [m.nchw().clone() for m in maps]
        outs += [tp.mc.permute(0, 2, 1).clone(), tp.protos.nchw().clone(), tp.logits.clone()]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        tp: TrainPlan = ctx.tp
        if ctx.gen != tp.generation:
            raise RuntimeError("backward through a forward(x, 'train') whose kept activations were overwritten by a later forward of the same shape")
        nd = ctx.n_det
        groups = {"det": gs[:nd], "seg": gs[nd:nd + 3], "mc": gs[nd + 3:nd + 4], "protos": gs[nd + 4:nd + 5], "logits": gs[nd + 5:nd + 6]}
        active = []
        for key, gl in groups.items():
            if not gl or all(g is None for g in gl):
                continue
            active.append(key)
            if key in ("det", "seg"):
                for d, g in zip(tp.d_in[key], gl):
                    if g is None:
                        d.buf.zero_()
                    else:
                        d.nchw().copy_(g)
            elif key == "mc":
                tp.d_in["mc"].copy_(gl[0].permute(0, 2, 1))
            elif key == "protos":
                tp.d_in["protos"].copy_(gl[0].permute(0, 2, 3, 1))
            else:
                tp.d_in["logits"].copy_(gl[0])
        plan = tp.run_backward(active)
        written = set(plan.written)
        # The gradients leave as views of FRESH copies of the gradient buckets (one flat copy per bucket), never of the persistent arena:
        # AccumulateGrad keeps a returned tensor as `.grad` without copying when its strides match the parameter's (the stem weight, 1x1
        # depthwise weights, the class bias), and the next backward pass rewrites the arena in place -- `.grad += new` would then add a
        # buffer to itself (gradient accumulation, `zero_grad(set_to_none=False)`, two losses through one forward).
        snap = tp.arena.snapshot_views([n for n, p in tp.m.named_parameters() if p.requires_grad and n in written])
        grads = tuple((tp._gview[n](snap[n]) if tp._gview[n] is not None else snap[n]) if n in snap else None for n, p in tp.m.named_parameters())
        return (None, None, None) + grads


def train_forward(model, x: torch.Tensor):
    """`forward(x, "train")` with autograd history: returns (det maps | None, seg maps, mc, protos, logits) as fresh tensors."""
    from .engine import code_of
    if not x.is_cuda:
        raise RuntimeError("ConvNeXtBiFPNYOLO (HIP) needs CUDA/HIP tensors on an MI355X; there is no CPU path")
    if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
        raise ValueError(f"expected [B,3,S,S] with S a multiple of 32, got {tuple(x.shape)}")
    if model.compute_dtype == torch.float16:
        raise NotImplementedError("float16 is an inference arithmetic mode (BASELINE configs[4]); train in bfloat16 or float32")
    cache = model.__dict__.setdefault("_train_plans", {})
    key = (tuple(x.shape), model.compute_dtype, x.device.index, _bn_mode_sig(model))
    tp = cache.get(key)
    if tp is None:
        tp = cache[key] = TrainPlan(model, x.shape, x.device, code_of(model.compute_dtype))
    tp.check_params()
    nd = 3 if tp.det_maps is not None else 0
    outs = _TrainFn.apply(tp, nd, x, *model.parameters())
    det = list(outs[:nd]) if nd else None
    seg = list(outs[nd:nd + 3])
    return det, seg, outs[nd + 3], outs[nd + 4], outs[nd + 5]
