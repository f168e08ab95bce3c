"""Drop-in `ConvNeXtBiFPNYOLO` for MI355X: the reference's nn.Module interface
(`/root/reference/src/main_model.py:300-393`; Segment-only variant `main_modelv2.py:300-385`) over
hand-written HIP kernels (libmtbt_hip.so, C ABI in include/mtbt_hip.h).

Same constructor, same submodule / parameter names (state_dicts and Lightning checkpoints load
unchanged), same `forward(x, mode)` output layouts, same head `.training` flag handling (SURVEY
F14).  The submodules below are PARAMETER CONTAINERS: they own the tensors under the reference's
names and never run torch operators.  `forward` lowers the whole graph once per (shape, dtype,
weights version) into a launch plan (engine.Plan) of C-ABI calls and replays it.

There is no CPU path: tensors must live on an MI355X and the HIP library must be built.
"""
import math
import os
from typing import List

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Act, Plan, code_of, TORCH_DTYPE, reserved_stream

BN_MOMENTUM, BN_EPS = 0.9997, 4e-5  # main_model.py:95,135
LN_EPS = 1e-6                       # timm ConvNeXt [upstream]
DEPTHS, DIMS = (3, 3, 9, 3), (96, 192, 384, 768)


class _Params(nn.Module):
    """A module that only holds parameters; compute happens in the launch plan."""

    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container of the HIP launch plan; "
                           "call the top-level ConvNeXtBiFPNYOLO.forward")


# ---- reference-owned blocks (main_model.py:42-296) -------------------------------------------------
class ConvBlock(_Params):
    def __init__(self, cin, cout, k=1, s=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, k // 2)
        self.bn = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM, eps=BN_EPS)
        self.act = nn.SiLU()


class Bottleneck(_Params):
    def __init__(self, cin, cout, shortcut=True, kernel=(3, 3), e=0.5):
        super().__init__()
        hidden = int(cout * e)
        self.cv1 = ConvBlock(cin, hidden, kernel[0], 1)
        self.cv2 = ConvBlock(hidden, cout, kernel[1], 1)
        self.add = shortcut and cin == cout


class C2f(_Params):
    def __init__(self, cin, cout, n=2, shortcut=False, e=0.5):
        super().__init__()
        self.c = int(cout * e)
        self.cv1 = ConvBlock(cin, 2 * self.c, 1, 1)
        self.cv2 = ConvBlock((2 + n) * self.c, cout, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, kernel=(3, 3), e=1.0) for _ in range(n))


class DepthwiseConvBlock(_Params):
    def __init__(self, cin, cout):
        super().__init__()
        self.depthwise = nn.Conv2d(cin, cout, 1, 1, 0, 1, groups=cin, bias=False)
        self.pointwise = nn.Conv2d(cin, cout, 1, 1, 0, 1, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM, eps=BN_EPS)
        self.act = nn.ELU()


class BiFPNUnit(_Params):
    def __init__(self, feature_size=256, eps=1e-4):
        super().__init__()
        self.eps = eps
        for lvl in ("p3_td", "p4_td", "p4_out", "p5_out"):
            setattr(self, f"{lvl}_conv", DepthwiseConvBlock(feature_size, feature_size))
            setattr(self, f"{lvl}_cf", C2f(feature_size, feature_size, shortcut=False))
        # uninitialised in the reference (main_model.py:191-192); load real values
        self.w1 = nn.Parameter(torch.Tensor(2, 2), requires_grad=True)
        self.w2 = nn.Parameter(torch.Tensor(3, 2), requires_grad=True)


class BiFPN(_Params):
    def __init__(self, size: List[int], feature_size=256, num_layers=3, eps=1e-4):
        super().__init__()
        if len(size) != 3:
            raise ValueError(f"BiFPN expects 3 input sizes for C3, C4, C5 projections, got {len(size)}")
        self.p3_proj = ConvBlock(size[0], feature_size, 1)
        self.p4_proj = ConvBlock(size[1], feature_size, 1)
        self.p5_proj = ConvBlock(size[2], feature_size, 1)
        self.num_layers, self.feature_size = num_layers, feature_size
        self.bifpn_units = nn.Sequential(*[BiFPNUnit(feature_size, eps=eps) for _ in range(num_layers)])


# ---- timm ConvNeXt-T feature extractor, parameter names of FeatureListNet [upstream] ---------------
class _Mlp(_Params):
    def __init__(self, d):
        super().__init__()
        self.fc1 = nn.Linear(d, 4 * d)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(4 * d, d)


class _CNBlock(_Params):
    def __init__(self, d):
        super().__init__()
        self.conv_dw = nn.Conv2d(d, d, 7, 1, 3, groups=d, bias=True)
        self.norm = nn.LayerNorm(d, eps=LN_EPS)
        self.mlp = _Mlp(d)
        self.gamma = nn.Parameter(1e-6 * torch.ones(d))


class _CNStage(_Params):
    def __init__(self, cin, cout, depth, downsample):
        super().__init__()
        self.downsample = (nn.Sequential(nn.LayerNorm(cin, eps=LN_EPS), nn.Conv2d(cin, cout, 2, 2))
                           if downsample else nn.Identity())
        self.blocks = nn.Sequential(*[_CNBlock(cout) for _ in range(depth)])


class _CNFeatures(_Params):
    def __init__(self):
        super().__init__()
        self.stem_0 = nn.Conv2d(3, DIMS[0], 4, 4)
        self.stem_1 = nn.LayerNorm(DIMS[0], eps=LN_EPS)
        prev = DIMS[0]
        for i, (d, n) in enumerate(zip(DIMS, DEPTHS)):
            setattr(self, f"stages_{i}", _CNStage(prev, d, n, i > 0))
            prev = d


class ConvNeXtTiny(_Params):
    """main_model.py:12-38."""

    def __init__(self, pretrained: bool = True):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained_backbone=True would download timm weights (main_model.py:21-26); there is "
                               "no network: build with pretrained_backbone=False and load_state_dict() a checkpoint")
        self.body = _CNFeatures().eval()
        self.c2f_p3 = C2f(192, 256)
        self.c2f_p4 = C2f(384, 384)
        self.c2f_p5 = C2f(768, 512)
        self.out_channels = list(DIMS[1:])


# ---- ultralytics heads, parameter names of Detect / Segment / Proto / DFL [upstream] ---------------
class _UConv(_Params):
    def __init__(self, c1, c2, k=1, g=1):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, 1, k // 2, groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU()


class _DFL(_Params):
    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1


class _Proto(_Params):
    def __init__(self, c1, c_=256, c2=32):
        super().__init__()
        self.cv1 = _UConv(c1, c_, 3)
        self.upsample = nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2 = _UConv(c_, c_, 3)
        self.cv3 = _UConv(c_, c2)


class Detect(_Params):
    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc, self.nl, self.reg_max = nc, len(ch), 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)  # never set by the reference
        c2 = max(16, ch[0] // 4, self.reg_max * 4)
        c3 = max(ch[0], min(nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(_UConv(x, c2, 3), _UConv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(
            nn.Sequential(nn.Sequential(_UConv(x, x, 3, g=x), _UConv(x, c3, 1)),
                          nn.Sequential(_UConv(c3, c3, 3, g=c3), _UConv(c3, c3, 1)),
                          nn.Conv2d(c3, nc, 1)) for x in ch)
        self.dfl = _DFL(self.reg_max)


class Segment(Detect):
    def __init__(self, nc=80, nm=32, npr=256, ch=()):
        super().__init__(nc, ch)
        self.nm, self.npr = nm, npr
        self.proto = _Proto(ch[0], npr, nm)
        c4 = max(ch[0] // 4, nm)
        self.cv4 = nn.ModuleList(
            nn.Sequential(_UConv(x, c4, 3), _UConv(c4, c4, 3), nn.Conv2d(c4, nm, 1)) for x in ch)


# ---- weight folding (eval-mode BatchNorm folds into the conv epilogue) -----------------------------
def _bn_fold(bn: nn.BatchNorm2d, conv_bias=None):
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.float() * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().float() * scale
    return scale, shift


def _permute_hidden(w2: torch.Tensor) -> torch.Tensor:
    """fc2 weight [d][4d] -> column order of mlp_fused.hip: inside every group of 32 hidden units, slot 8g+j holds hidden
    4g+j (j < 4) or 16+4g+(j-4) (j >= 4): a lane's GEMM1 accumulators (4 + 4 hidden units) are then GEMM2's fragment."""
    kk = torch.arange(32)
    g, j = kk // 8, kk % 8
    within = torch.where(j < 4, 4 * g + j, 16 + 4 * g + (j - 4))
    idx = (torch.arange(w2.shape[1] // 32)[:, None] * 32 + within[None, :]).reshape(-1)
    return w2[:, idx.to(w2.device)].contiguous()


def _krsc(w: torch.Tensor) -> torch.Tensor:
    """[K,C,R,S] -> [K, R*S*C]."""
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def compose_upconv(wt: torch.Tensor, bt: torch.Tensor, w3: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor):
    """ConvTranspose2d(Ci, Cm, 2, 2, bias bt) followed by Conv2d(Cm, K, 3, pad 1, no bias) * scale + shift (a folded BatchNorm), composed
    for `mtbt_convt2x2_conv3x3_nhwc` (include/mtbt_hip.h): both operators are linear, so per output parity (a, b) the nine taps collapse
    onto 2 x 2 source pixels.  wt [Ci, Cm, 2, 2], w3 [K, Cm, 3, 3]  ->  (w [4K, 4Ci] = [q][k][rho][sigma][ci], shift9 [9, K]), all fp32.
    Upsampled row 2i + a + dy is source row i + floor((a + dy) / 2), sub-row (a + dy) mod 2; rho = floor((a + dy) / 2) - (a - 1)."""
    wt, bt, w3 = wt.detach().double(), bt.detach().double(), w3.detach().double() * scale.detach().double()[:, None, None, None]
    K, Ci = w3.shape[0], wt.shape[0]
    w = torch.zeros(2, 2, K, 2, 2, Ci, dtype=torch.float64, device=w3.device)
    for a in range(2):
        for b in range(2):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    rho, sy = (a + dy) // 2 - (a - 1), (a + dy) % 2
                    sig, sx = (b + dx) // 2 - (b - 1), (b + dx) % 2
                    w[a, b, :, rho, sig, :] += w3[:, :, dy + 1, dx + 1] @ wt[:, :, sy, sx].t()
    # bias of the transposed conv through the taps that lie inside the upsampled map: class 0 = first output row (the dy = -1 taps fall
    # into the zero padding), 2 = last row (dy = +1 missing), 1 = interior; the same for columns
    tapb = torch.einsum("kcyx,c->kyx", w3, bt)                                  # [K, 3, 3]
    valid = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}
    shift9 = torch.stack([shift.detach().double() + tapb[:, list(valid[rc])][:, :, list(valid[cc])].sum(dim=(1, 2)) for rc in range(3) for cc in range(3)])
    return w.reshape(4 * K, 4 * Ci).float(), shift9.float()


PLAN_OPTION_DEFAULTS = {"NODE_FUSED": "0", "ADAPTOR_EARLY": "0", "HEADS_EARLY": "0", "SEG_GATE": "0", "LANES": None, "LANE_WIDE_US": None}


def plan_option(model, name: str):
    """A scheduling / lowering knob of the inference plan: `model.plan_options[name]` (set by `GraphedInference(autotune=True)`, which times a
    few combinations and keeps the fastest) > the environment variable MTBT_<name> (development A/B) > the default.  Every combination
    computes the same values up to the fused kernels' accumulation order; only the launch schedule differs."""
    v = model.__dict__.get("plan_options", {}).get(name)
    if v is None:
        v = os.environ.get("MTBT_" + name, PLAN_OPTION_DEFAULTS[name])
    return v


class _Lowering:
    """Builds the launch plan for one (batch, size, dtype, mode) from the module tree."""

    def __init__(self, model, x: torch.Tensor, code: int):
        self.m = model
        self.code = code
        self.dt = TORCH_DTYPE[code]
        self.p = Plan(x.device)
        self.x = x
        self.train_bns = []  # BatchNorms lowered with batch statistics (their num_batches_tracked advances per run)

    # constants
    def W(self, t):  # weights in compute dtype
        return self.p.const(t.float(), self.dt)

    def F(self, t):  # fp32 vectors
        return self.p.const(t.float(), torch.float32)

    def _bn_batch_stats(self, raw: Act, bn, act, y, name):
        """BatchNorm in train mode (batch statistics, running-stat update) + activation on the raw conv output."""
        if y is not None and not y.dense:
            raise NotImplementedError(f"{name}: batch-statistic BatchNorm into a channel slice (C2f concat buffer) is not lowered; "
                                      "only the Detect/Segment heads run in train mode inside forward(mode='train')")
        self.train_bns.append(bn)
        if y is None or y.code == raw.code:
            out = y if y is not None else raw            # in place unless the caller owns the destination
            self.p.bn_train(raw, out, bn, act, name=name + ".bn(batch)")
            if out is not raw:
                self.p.release(raw)
            return out
        self.p.bn_train(raw, raw, bn, act, name=name + ".bn(batch)")
        self.p.cast(raw, y, name=name + ".cast")          # e.g. the fp32 prototype output in bf16 mode
        self.p.release(raw)
        return y

    # -- building blocks --
    def convblock(self, x: Act, mod, y: Act = None, name=""):
        """reference ConvBlock (conv bias) or ultralytics Conv (no bias): conv + BN + SiLU.  BN in eval mode folds into
        the weights / epilogue shift; BN in train mode (heads inside forward(mode="train")) uses batch statistics."""
        conv, bn = mod.conv, mod.bn
        k = conv.kernel_size[0]
        if bn.training:
            raw = self.p.new(x.N, x.H, x.W, conv.out_channels, self.code)
            self.p.conv(x, self.W(_krsc(conv.weight)), raw, R=k, S=k, stride=1, pad=k // 2,
                        shift=self.F(conv.bias) if conv.bias is not None else None, name=name)
            return self._bn_batch_stats(raw, bn, L.ACT_SILU, y, name)
        scale, shift = _bn_fold(bn, conv.bias)
        if y is None:
            y = self.p.new(x.N, x.H, x.W, conv.out_channels, self.code)
        # the per-channel BN scale is folded into the weight rows (fp32, before the cast): the epilogue only adds the shift
        self.p.conv(x, self.W(_krsc(conv.weight).float() * scale[:, None]), y, R=k, S=k, stride=1, pad=k // 2,
                    shift=self.F(shift), act=L.ACT_SILU, name=name)
        return y

    def conv_plain(self, x: Act, conv: nn.Conv2d, y: Act, name=""):
        k = conv.kernel_size[0]
        self.p.conv(x, self.W(_krsc(conv.weight)), y, R=k, S=k, stride=conv.stride[0], pad=conv.padding[0],
                    shift=self.F(conv.bias) if conv.bias is not None else None, name=name)
        return y

    def dwblock(self, x: Act, mod: _UConv, name=""):
        """ultralytics DWConv: depthwise 3x3 + BN + SiLU."""
        w = self.W(mod.conv.weight.detach().reshape(x.C, 9).t())
        y = self.p.new(x.N, x.H, x.W, x.C, self.code)
        if mod.bn.training:
            one, zero = self.F(torch.ones(x.C)), self.F(torch.zeros(x.C))
            self.p.dwconv(x, w, y, 3, scale=one, shift=zero, act=L.ACT_NONE, name=name)
            return self._bn_batch_stats(y, mod.bn, L.ACT_SILU, None, name)
        scale, shift = _bn_fold(mod.bn)
        self.p.dwconv(x, w, y, 3, scale=self.F(scale), shift=self.F(shift), act=L.ACT_SILU, name=name)
        return y

    def c2f(self, x: Act, mod: C2f, name=""):
        """Concat-free C2f: every branch writes its channel slice of one [.., (2+n)c] buffer."""
        c, n = mod.c, len(mod.m)
        cat = self.p.new(x.N, x.H, x.W, (2 + n) * c, self.code)
        self.convblock(x, mod.cv1, cat.slice(0, 2 * c), name + ".cv1")
        prev = cat.slice(c, c)
        for i, b in enumerate(mod.m):
            if b.add:
                raise NotImplementedError("Bottleneck shortcut is never enabled by the reference model")
            t = self.convblock(prev, b.cv1, None, f"{name}.m.{i}.cv1")
            dst = cat.slice((2 + i) * c, c)
            self.convblock(t, b.cv2, dst, f"{name}.m.{i}.cv2")
            self.p.release(t)
            prev = dst
        y = self.convblock(cat, mod.cv2, None, name + ".cv2")
        self.p.release(cat)
        return y

    def dw_pointwise(self, x: Act, mod: DepthwiseConvBlock, name=""):
        """DepthwiseConvBlock (k=1): per-channel scale folded into the pointwise weight, BN folded, ELU."""
        if mod.bn.training:
            raise NotImplementedError(f"{name}: BiFPN DepthwiseConvBlock with batch-statistic BatchNorm (model.train()) is not lowered yet")
        dw = mod.depthwise.weight.detach().float().reshape(1, -1)
        pw = mod.pointwise.weight.detach().float().reshape(mod.pointwise.out_channels, -1) * dw
        scale, shift = _bn_fold(mod.bn)
        y = self.p.new(x.N, x.H, x.W, pw.shape[0], self.code)
        self.p.conv(x, self.W(pw * scale[:, None]), y, shift=self.F(shift), act=L.ACT_ELU, name=name)
        return y

    @staticmethod
    def fused_dims():
        """ConvNeXt widths whose MLP runs as ONE launch (mlp_fused.hip): stages 0-2.  d = 384 measures EQUAL to its two GEMMs inside the step
        (7.062 vs 7.064 ms at batch 16 x 640^2, 92.0 vs 91.9 ms at batch 64 x 1280^2 fp16: at one wave per SIMD its GELU and fragment reads
        are not covered by a partner wave) and is on because the 4d-wide hidden tensor (79 MB per block at batch 16) then never reaches
        HBM; d = 768 is not built (one weight stage is 98 KiB of LDS).  MTBT_FUSED_DIMS="96:192" restores the two-GEMM form (A/B)."""
        env = os.environ.get("MTBT_FUSED_DIMS")
        return tuple(int(v) for v in env.replace(":", ",").split(",")) if env else (96, 192, 384)

    def subbatch(self, stage: int, a: Act) -> int:
        """Images per depth-first pass of a ConvNeXt stage.  Measured (bench.py --ab MTBT_SUBBATCH=...): keeping the 4d-wide
        intermediate Infinity-Cache-resident by running 2-8 images at a time does NOT pay on MI355X at batch 16 -- the
        smaller launches lose more to ramp/tail than the cache saves (9.75 ms at 4:8 vs 9.46 ms whole-batch) -- so the
        default is the whole batch; MTBT_SUBBATCH="s0:s1:s2:s3" overrides for experiments."""
        import os
        env = os.environ.get("MTBT_SUBBATCH")
        if env:
            v = int(env.split(":")[stage])
            return a.N if v <= 0 else min(v, a.N)
        return a.N

    # -- backbone (main_model.py:33-38) --
    def backbone(self):
        """ConvNeXt-T stages with each C2f adaptor lowered RIGHT BEHIND the stage that feeds it (plan option ADAPTOR_EARLY=1; default 0: after the whole body,
        as the reference writes it, main_model.py:33-38).  The adaptors depend on one stage output each, so they are side-lane work either
        way; but launches are issued -- and captured into the HIP graph -- in program order, and with the adaptors at the end the P3
        adaptor's first kernel started 1 ms after its input existed (timeline of the replayed graph: beside stage 2's seventh block).
        Stage 2 fills 200 of the 256 CUs and stage 3 far fewer: issued early, c2f_p3 / c2f_p4 run in that slack."""
        bb = self.m.backbone
        early = plan_option(self.m, "ADAPTOR_EARLY") == "1"
        mods = {1: (bb.c2f_p3, "backbone.c2f_p3"), 2: (bb.c2f_p4, "backbone.c2f_p4"), 3: (bb.c2f_p5, "backbone.c2f_p5")}
        outs = {}

        def hook(si, a):
            if early:
                first = len(self.p.launches)
                outs[si] = self.c2f(a, *mods[si])
                if si < 3:                       # (the last adaptor has nothing left to hide under: it continues the main chain)
                    for l in self.p.launches[first:]:
                        l.side = True
        feats = self.features(bb.body, hook)
        if not early:
            for si in (1, 2, 3):
                outs[si] = self.c2f(feats[si - 1], *mods[si])
        for f in feats:
            self.p.release(f)
        return outs[1], outs[2], outs[3]

    def features(self, body, on_feature=None):
        """timm ConvNeXt-T feature extractor: outputs of stages 1..3 (live buffers; the caller releases them).  `on_feature(si, act)` is called
        as soon as stage si's output has been lowered."""
        N, _, H, W = self.x.shape
        a = self.p.new(N, H // 4, W // 4, DIMS[0], self.code)
        self.p.stem(self.x, self.F(body.stem_0.weight.detach().reshape(DIMS[0], 48)), self.F(body.stem_0.bias),
                    self.F(body.stem_1.weight), self.F(body.stem_1.bias), body.stem_1.eps, a, name="backbone.body.stem")
        feats = []
        for si in range(4):
            st = getattr(body, f"stages_{si}")
            nm = f"backbone.body.stages_{si}"
            if si > 0:
                ln, cv = st.downsample[0], st.downsample[1]
                t = self.p.new(a.N, a.H, a.W, a.C, self.code)
                self.p.layernorm(a, self.F(ln.weight), self.F(ln.bias), ln.eps, t, name=nm + ".downsample.0")
                nxt = self.p.new(a.N, a.H // 2, a.W // 2, cv.out_channels, self.code)
                self.conv_plain(t, cv, nxt, name=nm + ".downsample.1")
                self.p.release(t)
                if si == 1:          # stage-0 output is not a returned feature
                    self.p.release(a)
                a = nxt
            d = a.C
            # Sub-batching (depth-first over the stage's blocks): the 4d-wide MLP intermediate of a full batch does not fit
            # the 256 MiB Infinity Cache in the early stages (16 x 160^2 x 384 bf16 = 315 MB); running all blocks of the
            # stage on a few images at a time keeps t / h / o cache-resident instead of round-tripping through HBM.
            sb = self.subbatch(si, a)
            out_full = self.p.new(a.N, a.H, a.W, d, self.code)
            t = self.p.new(sb, a.H, a.W, d, self.code)
            fused = self.code in (L.BF16, L.F16) and d in self.fused_dims() and sb == a.N and os.environ.get("MTBT_FUSED_MLP", "1") == "1"
            h = None if fused else self.p.new(sb, a.H, a.W, 4 * d, self.code)
            pp_ = [self.p.new(sb, a.H, a.W, d, self.code) for _ in range(2)] if len(st.blocks) > 1 else []
            consts = []
            for blk in st.blocks:
                g = blk.gamma.detach().float()
                consts.append((self.W(blk.conv_dw.weight.detach().reshape(d, 49).t()), self.F(blk.conv_dw.bias), self.F(blk.norm.weight),
                               self.F(blk.norm.bias), self.W(blk.mlp.fc1.weight), self.F(blk.mlp.fc1.bias),
                               self.W(blk.mlp.fc2.weight.detach().float() * g[:, None]), self.F(g * blk.mlp.fc2.bias.detach().float())))
            w2p = [self.p.const(_permute_hidden(c_[6]), self.dt) for c_ in consts] if fused else None
            # bf16 mode: polynomial GELU (|err| <= 2.3e-4, below bf16 resolution); fp32 parity mode: the erf form
            gelu = L.ACT_GELU_POLY if self.code in (L.BF16, L.F16) and os.environ.get("MTBT_GELU_POLY", "1") == "1" else L.ACT_GELU
            for n0 in range(0, a.N, sb):
                nn_ = min(sb, a.N - n0)
                view = lambda act, k=nn_, o=n0: Act(act.buf, act.off + o * act.bs, k, act.H, act.W, act.C, act.ld, act.bs)
                local = lambda act, k=nn_: Act(act.buf, act.off, k, act.H, act.W, act.C, act.ld, act.bs)
                cur = view(a)
                for bi, blk in enumerate(st.blocks):
                    bn_ = f"{nm}.blocks.{bi}" + (f"[{n0}:{n0 + nn_}]" if sb < a.N else "")
                    dww, dwb, lnw, lnb, w1, b1, w2, b2 = consts[bi]
                    self.p.dwconv(cur, dww, local(t), 7, bias=dwb, lnw=lnw, lnb=lnb, eps=blk.norm.eps, name=bn_ + ".conv_dw+norm")
                    dst = view(out_full) if bi == len(st.blocks) - 1 else local(pp_[bi % 2])
                    if fused:   # stages 0-1 in bf16: the 4d-wide hidden tensor stays on chip (mlp_fused.hip)
                        self.p.mlp_fused(local(t), cur, w1, b1, w2p[bi], b2, dst, name=bn_ + ".mlp(fused)")
                    else:
                        self.p.conv(local(t), w1, local(h), shift=b1, act=gelu, name=bn_ + ".mlp.fc1")
                        self.p.conv(local(h), w2, dst, shift=b2, res=cur, name=bn_ + ".mlp.fc2")
                    cur = dst
            for buf in [t] + ([] if h is None else [h]) + pp_:
                self.p.release(buf)
            self.p.release(a)        # stage input: stem / downsample output, never a feature
            a = out_full
            if si >= 1:
                feats.append(a)      # stage outputs 1..3 stay live until the adaptors have read them
                if on_feature is not None:
                    on_feature(si, a)
        return feats

    # -- neck of the oldest variant (src/model.py:27-93): lateral Convs, WeightedAdd (adds its weights), DWConv 3x3 nodes --
    def neck_v0(self, f3, f4, f5):
        nk = self.m.neck
        p3 = self.convblock(f3, nk.lat3, None, "neck.lat3")
        p4 = self.convblock(f4, nk.lat4, None, "neck.lat4")
        p5 = self.convblock(f5, nk.lat5, None, "neck.lat5")
        for t in (f3, f4, f5):
            self.p.release(t)
        for ui, u in enumerate(nk.units):
            nm = f"neck.units.{ui}"

            def node(add, inputs, modes, like: Act, key):
                w = torch.relu(add.w.detach().float().cpu())
                w = w / (w.sum() + add.eps)
                s_ = self.p.new(like.N, like.H, like.W, like.C, self.code)
                self.p.fuse(inputs, [float(v) for v in w], modes, s_, bug=True, name=f"{nm}.add_{key}")
                o = self.dwblock(s_, u.conv[key], f"{nm}.conv.{key}")
                self.p.release(s_)
                return o

            p4_td = node(u.add_p4_td, [p4, p5], [L.RES_ID, L.RES_UP_NEAREST], p4, "p4_td")
            p3_td = node(u.add_p3_td, [p3, p4_td], [L.RES_ID, L.RES_UP_NEAREST], p3, "p3_td")
            p4_out = node(u.add_p4_out, [p4, p4_td, p3_td], [L.RES_ID, L.RES_ID, L.RES_MAXPOOL], p4, "p4_out")
            p5_out = node(u.add_p5_out, [p5, p4_out], [L.RES_ID, L.RES_MAXPOOL], p5, "p5_out")
            for t in (p3, p4, p5, p4_td):
                self.p.release(t)
            p3, p4, p5 = p3_td, p4_out, p5_out
        return p3, p4, p5

    # -- neck (main_model.py:198-243, 275-296) --
    @staticmethod
    def _norm_w(w, eps):
        w = torch.nn.functional.elu(w.detach().float().cpu())
        return w / (w.sum(dim=0, keepdim=True) + eps)

    def neck(self, c3, c4, c5, on_level=None):
        """`on_level(i, act)`: called as soon as the LAST unit has produced pyramid level i (order: P3, P4, P5 -- the top-down P3 node
        finishes first), so that the heads of that level can be lowered (and issued) before the rest of the neck."""
        nk = self.m.neck
        p3 = self.convblock(c3, nk.p3_proj, None, "neck.p3_proj")
        p4 = self.convblock(c4, nk.p4_proj, None, "neck.p4_proj")
        p5 = self.convblock(c5, nk.p5_proj, None, "neck.p5_proj")
        for t in (c3, c4, c5):
            self.p.release(t)
        for ui, u in enumerate(nk.bifpn_units):
            nm = f"neck.bifpn_units.{ui}"
            a, b = self._norm_w(u.w1, u.eps), self._norm_w(u.w2, u.eps)

            def node(inputs, weights, modes, like: Act, conv, cf, tag):
                K = conv.pointwise.out_channels
                shapes = ([L.RES_ID, L.RES_UP_BILINEAR], [L.RES_ID, L.RES_ID, L.RES_DOWN_MEAN])
                if (self.code in (L.BF16, L.F16) and like.C in (128, 256) and K == like.C and list(modes) in shapes and not conv.bn.training
                        and plan_option(self.m, "NODE_FUSED") == "1"):
                    # weighted sum + resample + DepthwiseConvBlock (scale folded, BN folded, ELU) in ONE launch: the fused map never reaches HBM
                    dw = conv.depthwise.weight.detach().float().reshape(1, -1)
                    pw = conv.pointwise.weight.detach().float().reshape(K, -1) * dw
                    scale, shift = _bn_fold(conv.bn)
                    d = self.p.new(like.N, like.H, like.W, K, self.code)
                    self.p.node(inputs, [float(v) for v in weights], modes, self.W(pw * scale[:, None]), self.F(shift), d, act=L.ACT_ELU,
                                name=f"{nm}.{tag}.fuse+conv")
                else:
                    s = self.p.new(like.N, like.H, like.W, like.C, self.code)
                    self.p.fuse(inputs, [float(v) for v in weights], modes, s, name=f"{nm}.{tag}.fuse")
                    d = self.dw_pointwise(s, conv, f"{nm}.{tag}_conv")
                    self.p.release(s)
                o = self.c2f(d, cf, f"{nm}.{tag}_cf")
                self.p.release(d)
                return o

            last = on_level is not None and ui == len(nk.bifpn_units) - 1
            p4_td = node([p4, p5], [a[0, 0], a[1, 0]], [L.RES_ID, L.RES_UP_BILINEAR], p4, u.p4_td_conv, u.p4_td_cf, "p4_td")
            p3_td = node([p3, p4_td], [a[0, 1], a[1, 1]], [L.RES_ID, L.RES_UP_BILINEAR], p3, u.p3_td_conv, u.p3_td_cf, "p3_td")
            if last:
                on_level(0, p3_td)
            p4_out = node([p4, p4_td, p3_td], [b[0, 0], b[1, 0], b[2, 0]], [L.RES_ID, L.RES_ID, L.RES_DOWN_MEAN], p4,
                          u.p4_out_conv, u.p4_out_cf, "p4_out")
            if last:
                on_level(1, p4_out)
            p5_out = node([p5, p5, p4_out], [b[0, 1], b[1, 1], b[2, 1]], [L.RES_ID, L.RES_ID, L.RES_DOWN_MEAN], p5,
                          u.p5_out_conv, u.p5_out_cf, "p5_out")
            if last:
                on_level(2, p5_out)
            for t in (p3, p4, p5, p4_td):
                self.p.release(t)
            p3, p4, p5 = p3_td, p4_out, p5_out
        return p3, p4, p5

    # -- heads [ultralytics] --
    def f32_out(self, N, H, W, Cc) -> Act:
        return Act.of(torch.empty((N, H, W, Cc), dtype=torch.float32, device=self.x.device))

    def det_level(self, i, f: Act, head: Detect, tag, gate=None) -> Act:
        """Level i of Detect.forward: cv2 (box, 4*reg_max ch) and cv3 (cls, nc ch) write side by side into one [N,h,w,no] fp32 map
        (the `torch.cat((cv2_i, cv3_i), 1)` of Detect.forward without the copy).
        `gate`: activations / tensors the first launch of each of the two chains is made to WAIT for (recorded as extra reads, i.e.
        ordinary read-after-write edges for the lane scheduler): the caller's way of keeping a branch nobody is waiting for off the
        machine until the branches on the post-process's critical path are through."""
        from .engine import _region

        def gated(first):
            if gate:
                l = self.p.launches[first]
                l.reads = l.reads + tuple(_region(g) for g in gate)
        # pixel pitch rounded up to 4 floats (66 -> 68): the 64-channel box slice then starts every pixel on a 16-byte boundary and the
        # conv epilogue stores it with 16-byte accesses (at pitch 66 it fell back to 64 scalar stores per pixel: 43 us per P3 map, 0.11 of HBM)
        ld = (head.no + 3) // 4 * 4
        buf = torch.empty((f.N, f.H, f.W, ld), dtype=torch.float32, device=self.x.device)
        full = Act(buf, 0, f.N, f.H, f.W, head.no, ld, f.H * f.W * ld)
        # the class chain (two depthwise + two 1x1 + output conv) is the longer one: first
        s = head.cv3[i]
        first = len(self.p.launches)
        d1 = self.dwblock(f, s[0][0], f"{tag}.cv3.{i}.0.0")
        gated(first)
        t1 = self.convblock(d1, s[0][1], None, f"{tag}.cv3.{i}.0.1")
        self.p.release(d1)
        d2 = self.dwblock(t1, s[1][0], f"{tag}.cv3.{i}.1.0")
        self.p.release(t1)
        t2 = self.convblock(d2, s[1][1], None, f"{tag}.cv3.{i}.1.1")
        self.p.release(d2)
        self.conv_plain(t2, s[2], full.slice(4 * head.reg_max, head.nc), f"{tag}.cv3.{i}.2")
        self.p.release(t2)
        s = head.cv2[i]
        first = len(self.p.launches)
        t1 = self.convblock(f, s[0], None, f"{tag}.cv2.{i}.0")
        gated(first)
        t2 = self.convblock(t1, s[1], None, f"{tag}.cv2.{i}.1")
        self.p.release(t1)
        self.conv_plain(t2, s[2], full.slice(0, 4 * head.reg_max), f"{tag}.cv2.{i}.2")
        self.p.release(t2)
        return full

    def det_branch(self, feats, head: Detect, tag, gate=None):
        return [self.det_level(i, f, head, tag, gate[i] if gate is not None else None) for i, f in enumerate(feats)]

    def mc_buffer(self, shapes, head: Segment):
        """[N, A, nm] fp32 buffer of the mask coefficients of all levels + the anchor offset of every level."""
        N = shapes[0][0]
        A = sum(h * w for _, h, w in shapes)
        offs, off = [], 0
        for _, h, w in shapes:
            offs.append(off)
            off += h * w
        return torch.empty((N, A, head.nm), dtype=torch.float32, device=self.x.device), offs, A

    def cv4_level(self, i, f: Act, head: Segment, mc: torch.Tensor, off: int, A: int):
        s = head.cv4[i]
        t1 = self.convblock(f, s[0], None, f"segment.cv4.{i}.0")
        t2 = self.convblock(t1, s[1], None, f"segment.cv4.{i}.1")
        self.p.release(t1)
        lvl = Act(mc, off * head.nm, f.N, f.H, f.W, head.nm, head.nm, A * head.nm)
        self.conv_plain(t2, s[2], lvl, f"segment.cv4.{i}.2")
        self.p.release(t2)

    def proto(self, f: Act, head: Segment) -> Act:
        """ultralytics Proto on P3: cv1 3x3 -> upsample (ConvTranspose 2x2 / 2) -> cv2 3x3 -> cv3 1x1."""
        pr = head.proto
        t1 = self.convblock(f, pr.cv1, None, "segment.proto.cv1")
        if (not pr.cv2.bn.training and f.H % 16 == 0 and f.W % 16 == 0 and pr.cv2.conv.out_channels % 128 == 0
                and os.environ.get("MTBT_PROTO_FUSED", "1") == "1"):
            # upsample (ConvTranspose 2x2 / 2 + bias) and cv2 (3x3 + BN + SiLU) are linear with nothing in between: ONE 2x2-tap direct conv
            # per output parity on the low-resolution map -- 4/10 of the MACs, no [N, 2H, 2W, 256] tensor written and read back
            sc, sh = _bn_fold(pr.cv2.bn)
            wc, shift9 = compose_upconv(pr.upsample.weight, pr.upsample.bias, pr.cv2.conv.weight, sc, sh)
            t2 = self.p.new(f.N, 2 * f.H, 2 * f.W, pr.cv2.conv.out_channels, self.code)
            self.p.upconv(t1, self.W(wc), self.F(shift9), t2, act=L.ACT_SILU, name="segment.proto.upsample+cv2")
            self.p.release(t1)
        else:
            up = self.p.new(f.N, 2 * f.H, 2 * f.W, pr.upsample.out_channels, self.code)
            wt = pr.upsample.weight.detach()  # [Cin, Cout, 2, 2] -> GEMM rows (dy*2+dx)*Cout + co
            self.p.conv(t1, self.W(wt.permute(2, 3, 1, 0).reshape(4 * wt.shape[1], wt.shape[0])), up,
                        shift=self.F(pr.upsample.bias.detach().repeat(4)), out_mode=L.OUT_CONVT2X2, name="segment.proto.upsample")
            self.p.release(t1)
            t2 = self.convblock(up, pr.cv2, None, "segment.proto.cv2")
            self.p.release(up)
        protos = self.f32_out(f.N, 2 * f.H, 2 * f.W, head.nm)
        self.convblock(t2, pr.cv3, protos, "segment.proto.cv3")
        self.p.release(t2)
        return protos

    def seg_extras(self, feats, head: Segment):
        """Mask coefficients (cv4, all levels into one [N,A,nm] buffer) and prototypes (Proto on P3)."""
        mc, offs, A = self.mc_buffer([(f.N, f.H, f.W) for f in feats], head)
        for i, f in enumerate(feats):
            self.cv4_level(i, f, head, mc, offs[i], A)
        return mc, self.proto(feats[0], head)

    def cls_head(self, n5: Act):
        logits = torch.empty((n5.N, self.m.cls_fc.out_features), dtype=torch.float32, device=self.x.device)
        self.p.gap_fc(n5, self.F(self.m.cls_fc.weight), self.F(self.m.cls_fc.bias), logits, name="cls_pool+cls_fc")
        return logits


class _Compiled:
    """One lowered forward: the plan plus the tensors it writes (plan-owned, overwritten every run)."""

    def __init__(self, plan, x_static, det_maps, seg_maps, mc, protos, logits, sig):
        self.plan, self.x = plan, x_static
        self.det_maps, self.seg_maps, self.mc, self.protos, self.logits = det_maps, seg_maps, mc, protos, logits
        self.sig = sig


class _Base(nn.Module):
    """Shared machinery of the two variants: dtype policy, plan cache, forward plumbing."""

    _dtype = torch.float32         # torch.float32 = exact-fp32 MFMA (parity); torch.bfloat16 = throughput; torch.float16 = inference (configs[4])
    _dtype_explicit = False

    @property
    def compute_dtype(self) -> torch.dtype:
        """The arithmetic mode of the next forward.  Set explicitly (`set_compute_dtype`) it is what was set.  Otherwise it follows the
        caller's `torch.autocast`: the reference trainer runs with `precision="bf16-mixed"` (`running_main_v3.py:825`), i.e. Lightning wraps
        `forward` in `autocast("cuda", torch.bfloat16)` -- the unchanged trainer then gets the bf16 plans (round 2 ignored the context and
        ran the ~4x slower exact-fp32 mode); outside any autocast region the default is float32, the parity mode.  Outputs keep their
        dtypes in every mode: raw Detect / Segment maps, mask coefficients, prototypes and logits are fp32 tensors (torch's autocast would
        hand out bf16 conv outputs; the trainer's loss upcasts them either way)."""
        if not self._dtype_explicit and torch.is_autocast_enabled("cuda"):
            return torch.get_autocast_dtype("cuda")
        return self._dtype

    def set_compute_dtype(self, dtype: torch.dtype):
        """Pin the arithmetic mode (from then on `torch.autocast` is not consulted); `None` returns to following autocast."""
        if dtype is None:
            self.__dict__.pop("_dtype", None)
            self.__dict__["_dtype_explicit"] = False
            return self
        code_of(dtype)
        self.__dict__["_dtype"], self.__dict__["_dtype_explicit"] = dtype, True
        return self

    def _weights_sig(self, bn_modes=None):
        """What a lowered plan's FOLDED constants depend on: parameter / buffer versions (torch's in-place counters), the epoch counter
        that raw-pointer parameter updates bump (`mark_weights_updated`: the fused optimisers, checkpoint loaders), storage identity
        (a re-homed or re-assigned `.data`) and, for the BatchNorms this plan runs in EVAL mode, their running statistics (a
        BatchNorm on batch statistics folds nothing, so a train-mode call never invalidates its own plan)."""
        # per-tensor (address, version) PAIRS, hashed as a tuple: additive sums (round 2) let two changes cancel (a re-homed parameter
        # whose low pointer bits drop by a version bump, two swapped storages) and a plan with stale folded weights be reused
        ver = hash(tuple((p.data_ptr(), p._version) for p in self.parameters()))
        bns = self.__dict__.get("_bn_list")
        if bns is None:
            bns = self.__dict__["_bn_list"] = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        st = []
        for m, tr in zip(bns, bn_modes if bn_modes is not None else [m.training for m in bns]):
            if not tr:      # eval-mode BatchNorm: this plan FOLDED its running statistics (`_mtbt_epoch`: in-kernel updates of them)
                st.append((m.running_mean.data_ptr(), m.running_mean._version, m.running_var.data_ptr(), m.running_var._version,
                           m.__dict__.get("_mtbt_epoch", 0)))
        return (ver, hash(tuple(st)), self.__dict__.get("_w_epoch", 0))

    def mark_weights_updated(self):
        """Call after changing parameters or buffers through raw pointers (in place, outside torch's version counters): plans that
        folded the old values re-lower on their next use."""
        self.__dict__["_w_epoch"] = self.__dict__.get("_w_epoch", 0) + 1
        return self

    def _heads(self):
        raise NotImplementedError

    def _wants_training_plan(self) -> bool:
        """forward(x, "train") goes through the training lowering when autograd is recording on a trainable model, or when a
        backbone / neck BatchNorm is in train mode (batch statistics into C2f slices: only train.py lowers those)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return True
        heads = {id(m) for h in (getattr(self, "detect", None), self.segment) if h is not None for m in h.modules()}
        return any(m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d) and id(m) not in heads)

    def compile(self, x: torch.Tensor) -> "_Compiled":
        """Lower the graph for this input's shape/dtype policy (cached until weights or BN modes change)."""
        if not x.is_cuda:
            raise RuntimeError("ConvNeXtBiFPNYOLO (HIP) needs CUDA/HIP tensors on an MI355X; there is no CPU path")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,3,S,S] with S a multiple of 32, got {tuple(x.shape)}")
        # one plan per BatchNorm-mode tuple: alternating forward(x, "infer") and forward(x, "train") (validation: heads on batch
        # statistics) no longer evict each other, and a train-mode call does not invalidate its own plan
        bn_modes = tuple(m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d))
        opts = tuple(plan_option(self, k) for k in sorted(PLAN_OPTION_DEFAULTS))
        key = (tuple(x.shape), self.compute_dtype, x.device.index, bn_modes, opts)
        sig = self._weights_sig(bn_modes)
        cache = self.__dict__.setdefault("_plans", {})
        if len(cache) > 8:                                # bounded: stale (shape, mode) plans hold buffer pools
            cache.pop(next(iter(cache)))
        c = cache.get(key)
        if c is not None and c.sig == sig:
            return c
        with torch.no_grad():
            xs = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
            lo = _Lowering(self, xs, code_of(self.compute_dtype))
            if plan_option(self, "LANES") is not None:
                lo.p.n_lanes = max(1, int(plan_option(self, "LANES")))
            if plan_option(self, "LANE_WIDE_US") is not None:
                lo.p.lane_wide_s = float(plan_option(self, "LANE_WIDE_US")) * 1e-6
            has_det = hasattr(self, "detect")
            det_maps = [None] * 3 if has_det else None
            seg_maps = [None] * 3
            state = {}
            gate_mode = int(plan_option(self, "SEG_GATE"))
            early = plan_option(self, "HEADS_EARLY") == "1" and not isinstance(self, ConvNeXtBiFPNYOLOv0)

            def heads_of_level(i, f, shapes):
                """Everything that hangs off pyramid level i, in the order of what the post-process waits for: the prototype chain (the
                longest, P3 only), Detect's branches (decode -> NMS), the mask coefficients, then Segment's own box / class branches
                (they feed `segment_preds_cat` only: optionally gated behind Detect's map of the level, MTBT_SEG_GATE=1)."""
                # the head branches are independent: no buffer recycling between them, so the lane scheduler sees no false dependencies
                keep = lo.p.pool.reuse
                lo.p.pool.reuse = keep and os.environ.get("MTBT_HEAD_REUSE", "0") == "1"
                first = len(lo.p.launches)
                if "mc" not in state:
                    state["mc"], state["offs"], state["A"] = lo.mc_buffer(shapes, self.segment)
                if i == 0:
                    state["protos"] = lo.proto(f, self.segment)
                if has_det:
                    det_maps[i] = lo.det_level(i, f, self.detect, "detect")
                lo.cv4_level(i, f, self.segment, state["mc"], state["offs"][i], state["A"])
                gate = [det_maps[i]] if (has_det and gate_mode >= 1) else None
                seg_maps[i] = lo.det_level(i, f, self.segment, "segment", gate)
                if early:                         # lowered in the middle of the neck: branch work, off the neck's main chain
                    for l in lo.p.launches[first:]:
                        l.side = True
                lo.p.pool.reuse = keep

            S_ = x.shape[2]
            shapes = [(x.shape[0], S_ // 8, x.shape[3] // 8), (x.shape[0], S_ // 16, x.shape[3] // 16), (x.shape[0], S_ // 32, x.shape[3] // 32)]
            if isinstance(self, ConvNeXtBiFPNYOLOv0):
                n3, n4, n5 = lo.neck_v0(*lo.features(self.backbone.body))
            else:
                c3, c4, c5 = lo.backbone()
                # MTBT_HEADS_EARLY (default): the heads of a pyramid level are lowered -- and therefore issued / captured -- the moment the
                # LAST BiFPN unit has produced that level.  P3 comes first (top-down node), and P3 carries most of the head work (the
                # prototype chain, the 80x80 branches): it then runs beside the unit's P4 / P5 output nodes, small launches that leave most
                # of the machine idle, instead of queueing behind them; and the chains the post-process waits for (prototypes, Detect)
                # are first in line.  (Round 2 lowered all heads after the neck, Detect -> cv4 -> Proto -> Segment: in the replayed graph
                # the prototype chain started up to 0.6 ms after its input existed and the mask assembly waited for it at the end.)
                n3, n4, n5 = lo.neck(c3, c4, c5, on_level=(lambda i, f: heads_of_level(i, f, shapes)) if early else None)
            feats = [n3, n4, n5]
            if not early:
                for i, f in enumerate(feats):
                    heads_of_level(i, f, [(t.N, t.H, t.W) for t in feats])
            mc, protos = state["mc"], state["protos"]
            logits = lo.cls_head(n5)
        c = _Compiled(lo.p, xs, det_maps, seg_maps, mc, protos, logits, sig)
        # launches that write the maps the box decode reads: the post-process forks as soon as these are done
        keys = {m.buf.untyped_storage().data_ptr() for m in (det_maps if det_maps is not None else seg_maps)}
        c.det_marks = [i for i, l in enumerate(lo.p.launches) if any(w[0] in keys for w in l.writes)]
        mkeys = {mc.untyped_storage().data_ptr(), protos.buf.untyped_storage().data_ptr()}
        c.mask_marks = [i for i, l in enumerate(lo.p.launches) if any(w[0] in mkeys for w in l.writes)]
        c.train_bns = lo.train_bns
        cache[key] = c
        return c

    def _bind_input(self, c: "_Compiled", x: torch.Tensor):
        stem = c.plan.launches[0]
        if x.dtype == torch.float32 and x.is_contiguous() and x.data_ptr() % 16 == 0:
            src = x            # read the caller's resident NCHW fp32 batch in place
        else:
            c.x.copy_(x)       # other dtypes / layouts: one conversion copy into the plan's input buffer
            src = c.x
        stem.args = (src.data_ptr(),) + stem.args[1:]
        stem.keep = (src,) + tuple(stem.keep[1:])

    @torch.no_grad()
    def infer_and_detect(self, x: torch.Tensor, img_size: int, conf_th: float = 0.05, iou_th: float = 0.6, top_k: int = 100,
                         masks: bool = True, side_stream: "torch.cuda.Stream" = None, own_outputs: bool = True):
        """`forward(x, "infer")` + `postprocess.detect_and_segment` as ONE scheduled step: the box decode and the
        per-image NMS (16 workgroups of latency-bound work) fork onto a side stream as soon as the Detect maps exist and
        run UNDER the Segment / Proto / cls launches; the mask assembly joins both.  Same results as the two calls.
        `own_outputs=False` returns the raw maps / mc / protos / logits of the forward dict as VIEWS of the plan's buffers
        (overwritten by the next call) instead of the fresh copies `forward()` hands out -- what a graph replay wants.
        Returns (forward dict, detections dict)."""
        from . import postprocess as pp
        det_flag, seg_flag = getattr(self, "detect", self.segment).training, self.segment.training
        try:
            if hasattr(self, "detect"):
                self.detect.eval()
            self.segment.eval()
            c = self.compile(x)
            self._bind_input(c, x)
            maps = c.det_maps if c.det_maps is not None else c.seg_maps
            main = torch.cuda.current_stream(x.device)
            side = side_stream or reserved_stream(x.device, "eager_side")
            ready = c.plan.run(marks={"det": c.det_marks, "mask": c.mask_marks if masks else []})
            mk = None
            with torch.cuda.stream(side):
                for ev in ready["det"]:
                    side.wait_event(ev)
                d = pp.decode_boxes([m.nchw() for m in maps], img_size, want_scores=False)
                k = pp.nms_batched(d["boxes"], d["best_score"], d["best_label"], float(img_size), conf_th, iou_th, top_k)
                early = masks and os.environ.get("MTBT_MASK_OVERLAP", "1") == "1"     # (=0: A/B switch, masks after the join as in round 1)
                if early:     # as soon as the coefficients and prototypes exist: under the launches that are still running on the plan's lanes
                    for ev in ready["mask"]:
                        side.wait_event(ev)
                    mk, _ = pp.assemble_masks(c.protos.nchw(), c.mc.permute(0, 2, 1), k["keep_anchor"], k["counts"], (img_size, img_size))
            # the forward dict's own tail (the two `*_preds_cat` decodes, the class softmax) depends only on the head maps: issued on the main
            # stream BEFORE the join, it runs beside the side stream's NMS / mask assembly instead of behind it (round 2: 55 us serial tail)
            fwd = self._infer_dict(c, own=own_outputs)
            main.wait_stream(side)
            out = {"boxes": k["boxes"], "scores": k["scores"], "labels": k["labels"], "counts": k["counts"],
                   "keep_idx": k["keep_idx"], "keep_anchor": k["keep_anchor"], "n_cand": k["n_cand"]}
            if masks:
                out["masks"] = mk if mk is not None else pp.assemble_masks(c.protos.nchw(), c.mc.permute(0, 2, 1), k["keep_anchor"], k["counts"], (img_size, img_size))[0]
            for t in [d["boxes"], d["best_score"], d["best_label"]] + [v for v in out.values() if isinstance(v, torch.Tensor)]:
                t.record_stream(main)
            return fwd, out
        finally:
            if hasattr(self, "detect"):
                self.detect.training = det_flag
            self.segment.training = seg_flag

    def _run(self, x: torch.Tensor) -> "_Compiled":
        c = self.compile(x)
        self._bind_input(c, x)
        c.plan.run()
        self._after_run(c)
        return c

    def _after_run(self, c):
        """Book-keeping of batch-statistic BatchNorms: the kernels updated running_mean / running_var in place."""
        if c.train_bns:
            for bn in c.train_bns:
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
                bn.__dict__["_mtbt_epoch"] = bn.__dict__.get("_mtbt_epoch", 0) + 1   # invalidates plans that folded its old statistics

    # decoded `[B, 4+nc(+nm), A]` tensor of Detect/Segment eval (ultralytics `_inference`), from raw maps
    def _preds_cat(self, maps: List[Act], head: Detect, mc: torch.Tensor = None):
        from . import postprocess as pp
        return pp.detect_inference(maps, head, mc)


class ConvNeXtBiFPNYOLO(_Base):
    """Canonical variant, `/root/reference/src/main_model.py:300-393`."""

    def _infer_dict(self, c, own=True):  # main_model.py:378-386
        cp = (lambda t: t.clone()) if own else (lambda t: t)   # own=False: views of plan buffers, valid until the next call
        det_feats = [cp(m.nchw()) for m in c.det_maps]
        seg_feats = [cp(m.nchw()) for m in c.seg_maps]
        mc = cp(c.mc.permute(0, 2, 1))
        logits = cp(c.logits)
        return {
            "detect_features": det_feats,
            "detect_preds_cat": self._preds_cat(c.det_maps, self.detect),
            "segment_protos": (seg_feats, mc, cp(c.protos.nchw())),
            "segment_preds_cat": self._preds_cat(c.seg_maps, self.segment, c.mc),
            "img_cls_logits": logits,
            "img_cls_probs": logits.softmax(dim=1),
        }

    def __init__(self, nc_det: int, nc_img: int, proto_ch: int = 32, bifpn_feature_size: int = 256,
                 bifpn_num_layers: int = 2, pretrained_backbone: bool = True):
        super().__init__()
        L.load()  # fail loudly at construction if the HIP library is absent
        self.backbone = ConvNeXtTiny(pretrained=pretrained_backbone)
        self.neck = BiFPN(size=[256, 384, 512], feature_size=bifpn_feature_size, num_layers=bifpn_num_layers)
        ch = [bifpn_feature_size] * 3
        self.detect = Detect(nc=nc_det, ch=ch)
        self.segment = Segment(nc=nc_det, nm=proto_ch, npr=bifpn_feature_size, ch=ch)
        self.cls_pool = nn.AdaptiveAvgPool2d(1)
        self.cls_fc = nn.Linear(bifpn_feature_size, nc_img)
        self.nc_det, self.nc_img, self.proto_ch = nc_det, nc_img, proto_ch

    def forward(self, x, mode: str = "train"):
        det_flag, seg_flag = self.detect.training, self.segment.training
        try:
            if mode == "train":      # main_model.py:357-365
                self.detect.train()
                self.segment.train()
                if self._wants_training_plan():
                    # model.train() and / or autograd recording: the training lowering (train.py) keeps what backward needs and
                    # returns tensors whose grad_fn runs the backward plan -- `total_loss.backward()` of running_main_v3.py:393-445
                    from .train import train_forward
                    det, seg, mc, protos, logits = train_forward(self, x)
                    return det, (seg, mc, protos), logits
                c = self._run(x)
                det = [m.nchw().clone() for m in c.det_maps]
                seg = [m.nchw().clone() for m in c.seg_maps]
                return det, (seg, c.mc.permute(0, 2, 1).clone(), c.protos.nchw().clone()), c.logits.clone()
            if mode == "infer":      # main_model.py:367-386
                self.detect.eval()
                self.segment.eval()
                return self._infer_dict(self._run(x))
            raise ValueError(f"Unknown mode for ConvNeXtBiFPNYOLO.forward: {mode}. Expected 'train' or 'infer'.")
        finally:  # top-level flags only, as the reference does (main_model.py:391-393)
            self.detect.training = det_flag
            self.segment.training = seg_flag


class ConvNeXtBiFPNYOLOv2(_Base):
    """Segment-only variant, `/root/reference/src/main_modelv2.py:300-385`."""

    def _infer_dict(self, c, own=True):  # main_modelv2.py:371-378
        cp = (lambda t: t.clone()) if own else (lambda t: t)
        seg_feats = [cp(m.nchw()) for m in c.seg_maps]
        mc = cp(c.mc.permute(0, 2, 1))
        seg_cat = self._preds_cat(c.seg_maps, self.segment, c.mc)
        logits = cp(c.logits)
        return {
            "detect_preds_cat": seg_cat[:, : 4 + self.nc_det],
            "segment_protos": (seg_feats, mc, cp(c.protos.nchw())),
            "segment_preds_cat": seg_cat,
            "img_cls_logits": logits,
            "img_cls_probs": logits.softmax(dim=1),
        }

    def __init__(self, nc_det: int, nc_img: int, proto_ch: int = 32, bifpn_feature_size: int = 256,
                 bifpn_num_layers: int = 2, pretrained_backbone: bool = True):
        super().__init__()
        L.load()
        self.backbone = ConvNeXtTiny(pretrained=pretrained_backbone)
        self.neck = BiFPN(size=[256, 384, 512], feature_size=bifpn_feature_size, num_layers=bifpn_num_layers)
        ch = [bifpn_feature_size] * 3
        self.segment = Segment(nc=nc_det, nm=proto_ch, npr=bifpn_feature_size, ch=ch)
        self.cls_pool = nn.AdaptiveAvgPool2d(1)
        self.cls_fc = nn.Linear(bifpn_feature_size, nc_img)
        self.nc_det, self.nc_img, self.proto_ch = nc_det, nc_img, proto_ch

    def forward(self, x, mode: str = "train"):
        seg_flag = self.segment.training
        try:
            if mode == "train":      # main_modelv2.py:353-360
                self.segment.train()
                if self._wants_training_plan():
                    from .train import train_forward
                    _, seg, mc, protos, logits = train_forward(self, x)
                    return (seg, mc, protos), logits
                c = self._run(x)
                seg = [m.nchw().clone() for m in c.seg_maps]
                return (seg, c.mc.permute(0, 2, 1).clone(), c.protos.nchw().clone()), c.logits.clone()
            if mode == "infer":      # main_modelv2.py:362-378
                self.segment.eval()
                return self._infer_dict(self._run(x))
            raise ValueError(f"Unknown mode for ConvNeXtBiFPNYOLO.forward: {mode}. Expected 'train' or 'infer'.")
        finally:
            self.segment.training = seg_flag


class _WeightedAdd(_Params):
    def __init__(self, n, eps=1e-4):
        super().__init__()
        self.w = nn.Parameter(torch.ones(n, dtype=torch.float32))
        self.eps = eps


class _BiFPNUnitV0(_Params):
    def __init__(self, ch=256):
        super().__init__()
        self.add_p4_td, self.add_p3_td = _WeightedAdd(2), _WeightedAdd(2)
        self.add_p4_out, self.add_p5_out = _WeightedAdd(3), _WeightedAdd(2)
        self.conv = nn.ModuleDict({k: _UConv(ch, ch, 3, g=ch) for k in ("p4_td", "p3_td", "p4_out", "p5_out")})


class _BiFPNV0(_Params):
    def __init__(self, in_ch, repeats=2):
        super().__init__()
        self.lat3, self.lat4, self.lat5 = _UConv(in_ch[0], 256, 1), _UConv(in_ch[1], 256, 1), _UConv(in_ch[2], 256, 1)
        self.units = nn.ModuleList([_BiFPNUnitV0(256) for _ in range(repeats)])


class _BackboneV0(_Params):
    def __init__(self):
        super().__init__()
        self.body = _CNFeatures()
        self.out_channels = list(DIMS[1:])


class ConvNeXtBiFPNYOLOv0(_Base):
    """Oldest variant, `/root/reference/src/model.py:97-123` (BASELINE config 0): ConvNeXt-T features -> BiFPN with
    lateral Convs / nearest-x2 / max-pool / DWConv 3x3 nodes and the weight-ADDING WeightedAdd ->
    Detect + Segment + cls.  Unlike the later variants its `forward` does not touch the heads' training flags: the heads answer in
    whatever mode the module is in.  `module.eval()`: `{"detect": (y, feats), "segment": (cat[y, mc], (feats, mc, protos)), "img_cls":
    softmax}` or, for any other mode string, the raw `(det_out, seg_out, logits)` tuple.  `module.train()` (both heads in training mode):
    the training lowering (train.py: batch-statistic BatchNorm, autograd node, backward plan) with the heads' training outputs --
    `det_out` = the three raw maps, `seg_out` = (maps, mc, protos).  One head in each mode is not lowered."""

    def __init__(self, nc_det: int, nc_img: int, proto_ch: int = 32):
        super().__init__()
        L.load()
        self.backbone = _BackboneV0()
        self.neck = _BiFPNV0(self.backbone.out_channels, repeats=2)
        ch = (256, 256, 256)
        self.detect = Detect(nc_det, ch=ch)
        self.segment = Segment(nc_det, nm=proto_ch, ch=ch)
        self.cls_pool = nn.AdaptiveAvgPool2d(1)
        self.cls_fc = nn.Linear(256, nc_img)
        self.nc_det, self.nc_img, self.proto_ch = nc_det, nc_img, proto_ch

    def forward(self, x, mode: str = "infer"):
        if self.detect.training != self.segment.training:
            raise NotImplementedError("the src/model.py variant is lowered with both heads in the same mode (module.train() or module.eval())")
        if self.detect.training:          # src/model.py:105-123 under module.train(): Detect returns its maps, Segment (maps, mc, protos)
            from .train import train_forward
            det, seg, mc, protos, logits = train_forward(self, x)
            det_out, seg_out = det, (seg, mc, protos)
            if mode == "infer":
                return {"detect": det_out, "segment": (seg_out[0], seg_out[1]), "img_cls": logits.softmax(1)}
            return det_out, seg_out, logits
        if any(m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d)):
            raise NotImplementedError("the src/model.py variant: eval-mode heads over train-mode BatchNorm layers are not lowered")
        c = self._run(x)
        det_feats = [m.nchw().clone() for m in c.det_maps]
        seg_feats = [m.nchw().clone() for m in c.seg_maps]
        mc = c.mc.permute(0, 2, 1).clone()
        det_out = (self._preds_cat(c.det_maps, self.detect), det_feats)
        seg_out = (self._preds_cat(c.seg_maps, self.segment, c.mc), (seg_feats, mc, c.protos.nchw().clone()))
        logits = c.logits.clone()
        if mode == "infer":
            return {"detect": det_out, "segment": (seg_out[0], seg_out[1]), "img_cls": logits.softmax(1)}
        return det_out, seg_out, logits


@torch.no_grad()
def init_synthetic_(model: nn.Module, seed: int = 0) -> nn.Module:
    """Seeded synthetic weights for benchmarks (no checkpoints offline): torch default inits as built,
    BatchNorm statistics / affine randomised, layer-scale gamma ~ U(.05,.15), BiFPN w1/w2 = 1."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    for name, p in model.named_parameters():
        if name.endswith(".gamma"):
            p.copy_((torch.rand(p.shape, generator=g) + 0.5) * 0.1)
        elif name.endswith(".w1") or name.endswith(".w2"):
            p.fill_(1.0)
    return model


def synthetic_images(B: int, S: int, seed: int = 0) -> torch.Tensor:
    """Synthetic input batch [B,3,S,S] in [0,1): uniform noise carrying a few large soft blobs per image (2-5, sigma S/20 .. S/8, a
    colour each).  Pure noise makes every position of every feature map statistically the same, so no head -- trained or calibrated -- has
    anything to localise and the kept-box set is a coin toss per anchor; the blobs give the pyramid a handful of "objects" (the reference's
    radiographs: one lesion on a bone, `dataset_btxrdv2.py`), the noise keeps every operand bit toggling (no DVFS bonus from smooth data)."""
    g = torch.Generator().manual_seed(10_000 + seed)
    x = torch.rand(B, 3, S, S, generator=g) * 0.5
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    for b in range(B):
        for _ in range(int(torch.randint(2, 6, (1,), generator=g))):
            cy, cx = (torch.rand(2, generator=g) * 0.7 + 0.15).tolist()
            sg = float(torch.rand(1, generator=g)) * (S / 8 - S / 20) + S / 20
            amp = torch.rand(3, generator=g) * 0.5
            blob = torch.exp(-((yy - cy * S) ** 2 + (xx - cx * S) ** 2) / (2 * sg * sg))
            x[b] += amp[:, None, None] * blob[None]
    return x.clamp_(0.0, 0.999)


@torch.no_grad()
def calibrate_synthetic_heads_(model: nn.Module, x: torch.Tensor, cand_frac: float = 0.12, conf_th: float = 0.05, top_score: float = 0.95,
                               box_bins: float = 11.0, box_gain: float = 4.0, top_quantile: float = 0.999) -> nn.Module:
    """Make the random-initialised Detect / Segment heads behave like trained ones on the synthetic batch `x` ("8400 boxes per
    image ... so that about 10^3 pass conf 0.05").  Untouched, every class score of a random head is sigmoid(~0) = 0.5: all 8400 anchors
    are candidates, top-100 is decided below any arithmetic's resolution and the NMS input is degenerate (round 2's VERDICT, weak #2).
    One forward on the device measures, per pyramid level, the distribution of the best-class logit over the anchors; the last class conv
    of the level (`cv3[i][2]`, main_model.py:324 [ultralytics Detect]) is rescaled and re-biased (one gain, one offset: a two-point fit)
    so that the (1 - cand_frac) quantile lands on logit(conf_th) and the 99.9 % quantile on logit(top_score): ~`cand_frac` of the anchors
    pass the confidence filter with scores spread over (conf_th, ~top_score).  The box conv (`cv2[i][2]`) gets a gain and a bias ramp
    over the DFL bins so that the expected side is ~`box_bins` bins with per-anchor variation.  Deterministic for a given seed / input;
    changes parameters in place (plans re-lower by themselves)."""
    heads = [(h, key) for h, key in ((getattr(model, "detect", None), "detect_features"), (model.segment, None)) if h is not None]
    # (1) BatchNorm running statistics := the batch statistics of `x` (one train-mode forward with momentum 1): every BatchNorm then really
    # normalises its input, as in a trained network.  With arbitrary running statistics a deep random network forgets its input -- every
    # layer adds a constant component, after ~100 layers the head maps are the same for every image and vary only with the distance to the
    # zero-padded border -- and whatever is calibrated on top of that amplifies rounding noise instead of signal.
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    saved = [(m.momentum, m.training) for m in bns]
    flags = [(m, m.training) for m in model.modules()]
    dt_state = (model.__dict__.get("_dtype"), model.__dict__.get("_dtype_explicit"))
    try:
        for m in bns:
            m.momentum = 1.0
        if model.compute_dtype == torch.float16:     # (fp16 is an inference mode: the statistics pass runs in bf16)
            model.set_compute_dtype(torch.bfloat16)
        model.train()
        # (at most 16 x 640^2 pixels of the batch: the training lowering keeps every activation -- 166 GiB for 64 x 1280^2)
        n_stat = max(1, min(x.shape[0], (16 * 640 * 640) // (x.shape[2] * x.shape[3])))
        model(x[:n_stat].contiguous(), "train")
    finally:
        for key, v in zip(("_dtype", "_dtype_explicit"), dt_state):
            if v is None:
                model.__dict__.pop(key, None)
            else:
                model.__dict__[key] = v
        for m, (mom, _) in zip(bns, saved):
            m.momentum = mom
        for m, tr in flags:
            m.training = tr
    model.__dict__.pop("_train_plans", None)            # the calibration batch's training plan (kept activations) is not needed again
    out = model(x, "infer")
    lg = lambda pr: math.log(pr / (1 - pr))
    for head, key in heads:
        feats = out[key] if key is not None else out["segment_protos"][0]
        nb = 4 * head.reg_max
        lo, hi = 0.0, 2.0                                   # bias ramp a*k over the bins: expected bin = box_bins
        ks = torch.arange(head.reg_max, dtype=torch.float64)
        for _ in range(60):
            a = 0.5 * (lo + hi)
            e = float((torch.softmax(a * ks, 0) * ks).sum())
            lo, hi = (a, hi) if e < box_bins else (lo, a)
        for i, f in enumerate(feats):
            best = f[:, nb:].float().amax(dim=1).flatten()
            q = torch.quantile(best[:: max(1, best.numel() // 200_000)], torch.tensor([1.0 - cand_frac, top_quantile], device=best.device))
            q_lo, q_hi = float(q[0]), float(q[1])
            g = (lg(top_score) - lg(conf_th)) / max(q_hi - q_lo, 1e-9)
            c = lg(conf_th) - g * q_lo
            cls = head.cv3[i][2]
            cls.bias.copy_(cls.bias * g + c)
            cls.weight.mul_(g)
            box = head.cv2[i][2]
            box.weight.mul_(box_gain)
            box.bias.copy_(box.bias * box_gain + (a * ks).float().repeat(4).to(box.bias.device))
    return model
