"""Oracle restatement of the reference-owned conv blocks and the C2f-BiFPN neck.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameter names equal the reference's so a
state_dict moves between the two unchanged.  Each class cites the reference lines it follows;
all of `/root/reference/src/main_model.py`.
"""
from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_MOMENTUM = 0.9997  # main_model.py:95,135 (TF-style value used with torch semantics)
BN_EPS = 4e-5


def same_pad(k: int, p=None, d: int = 1) -> int:
    """main_model.py:105-110 `autopad` for integer kernels."""
    if d > 1:
        k = d * (k - 1) + 1
    return k // 2 if p is None else p


class ConvBlock(nn.Module):
    """Conv2d(bias=True) -> BatchNorm2d(eps 4e-5, momentum .9997) -> SiLU.  main_model.py:113-141."""

    def __init__(self, cin, cout, k=1, s=1, p=None, d=1, g=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, same_pad(k, p, d), d, g)
        self.bn = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM, eps=BN_EPS)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    """Two ConvBlocks, optional residual.  main_model.py:42-59."""

    def __init__(self, cin, cout, shortcut=True, groups=1, kernel=(3, 3), e=0.5):
        super().__init__()
        hidden = int(cout * e)
        self.cv1 = ConvBlock(cin, hidden, kernel[0], 1)
        self.cv2 = ConvBlock(hidden, cout, kernel[1], 1, g=groups)
        self.add = shortcut and cin == cout

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C2f(nn.Module):
    """cv1 1x1 -> chunk 2 -> n Bottlenecks chained on the last chunk -> cat -> cv2 1x1.
    main_model.py:144-173."""

    def __init__(self, cin, cout, n=2, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(cout * e)
        self.cv1 = ConvBlock(cin, 2 * self.c, 1, 1)
        self.cv2 = ConvBlock((2 + n) * self.c, cout, 1)
        self.m = nn.ModuleList(
            Bottleneck(self.c, self.c, shortcut, groups=g, kernel=(3, 3), e=1.0) for _ in range(n)
        )

    def forward(self, x):
        parts = list(self.cv1(x).chunk(2, 1))
        for blk in self.m:
            parts.append(blk(parts[-1]))
        return self.cv2(torch.cat(parts, 1))


class DepthwiseConvBlock(nn.Module):
    """depthwise (k=1 by default => per-channel scale, no bias) -> pointwise 1x1 (no bias) -> BN -> ELU.
    main_model.py:62-102."""

    def __init__(self, cin, cout, kernel_size=1, stride=1, padding=0, dilation=1):
        super().__init__()
        self.depthwise = nn.Conv2d(cin, cout, kernel_size, stride, padding, dilation, groups=cin, bias=False)
        self.pointwise = nn.Conv2d(cin, cout, 1, 1, 0, 1, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout, momentum=BN_MOMENTUM, eps=BN_EPS)
        self.act = nn.ELU()

    def forward(self, x):
        return self.act(self.bn(self.pointwise(self.depthwise(x))))


def _up2(x):
    # main_model.py:211-213: F.interpolate(scale_factor=2, mode="bilinear") (align_corners=False)
    return F.interpolate(x, scale_factor=2, mode="bilinear")


def _down2(x):
    # main_model.py:231: F.interpolate(scale_factor=0.5, mode="bilinear") == exact 2x2 mean
    return F.interpolate(x, scale_factor=0.5, mode="bilinear")


class BiFPNUnit(nn.Module):
    """One top-down + bottom-up pass over (P3,P4,P5).  main_model.py:176-243.

    w1 (2,2) / w2 (3,2) are created UNINITIALISED by the reference (`torch.Tensor(2,2)`, :191-192,
    SURVEY F7).  The oracle does the same; every test loads explicit values.
    """

    def __init__(self, feature_size=256, eps=1e-4):
        super().__init__()
        self.eps = eps
        self.p3_td_conv = DepthwiseConvBlock(feature_size, feature_size)
        self.p3_td_cf = C2f(feature_size, feature_size, shortcut=False)
        self.p4_td_conv = DepthwiseConvBlock(feature_size, feature_size)
        self.p4_td_cf = C2f(feature_size, feature_size, shortcut=False)
        self.p4_out_conv = DepthwiseConvBlock(feature_size, feature_size)
        self.p4_out_cf = C2f(feature_size, feature_size, shortcut=False)
        self.p5_out_conv = DepthwiseConvBlock(feature_size, feature_size)
        self.p5_out_cf = C2f(feature_size, feature_size, shortcut=False)
        self.w1 = nn.Parameter(torch.Tensor(2, 2), requires_grad=True)
        self.w2 = nn.Parameter(torch.Tensor(3, 2), requires_grad=True)

    def _norm(self, w):
        # main_model.py:194-196
        w = F.elu(w)
        return w / (w.sum(dim=0, keepdim=True) + self.eps)

    def forward(self, feats: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        if len(feats) != 3:
            raise ValueError(f"BiFPNBlock expects 3 input feature levels, got {len(feats)}")
        p3, p4, p5 = feats
        a, b = self._norm(self.w1), self._norm(self.w2)
        # top-down (main_model.py:205-220)
        p4_td = self.p4_td_cf(self.p4_td_conv(a[0, 0] * p4 + a[1, 0] * _up2(p5)))
        p3_td = self.p3_td_cf(self.p3_td_conv(a[0, 1] * p3 + a[1, 1] * _up2(p4_td)))
        # bottom-up (main_model.py:222-241); note p5 enters its own sum twice (:237-238)
        p4_out = self.p4_out_cf(self.p4_out_conv(b[0, 0] * p4 + b[1, 0] * p4_td + b[2, 0] * _down2(p3_td)))
        p5_out = self.p5_out_cf(self.p5_out_conv(b[0, 1] * p5 + b[1, 1] * p5 + b[2, 1] * _down2(p4_out)))
        return [p3_td, p4_out, p5_out]


class BiFPN(nn.Module):
    """Three 1x1 ConvBlock projections then `num_layers` BiFPNUnits.  main_model.py:246-296."""

    def __init__(self, size: List[int], feature_size=256, num_layers=3, eps=1e-4):
        super().__init__()
        if len(size) != 3:
            raise ValueError(f"BiFPN expects 3 input sizes for C3, C4, C5 projections, got {len(size)}")
        self.p3_proj = ConvBlock(size[0], feature_size, 1)
        self.p4_proj = ConvBlock(size[1], feature_size, 1)
        self.p5_proj = ConvBlock(size[2], feature_size, 1)
        self.num_layers = num_layers
        self.feature_size = feature_size
        self.bifpn_units = nn.Sequential(*[BiFPNUnit(feature_size, eps=eps) for _ in range(num_layers)])

    def forward(self, inputs):
        if len(inputs) != 3:
            raise ValueError(f"BiFPN expects 3 input feature maps (from backbone C2f), got {len(inputs)}")
        c3, c4, c5 = inputs
        feats = [self.p3_proj(c3), self.p4_proj(c4), self.p5_proj(c5)]
        for unit in self.bifpn_units:
            feats = unit(feats)
        return feats
