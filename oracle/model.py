"""Oracle restatement of the three `ConvNeXtBiFPNYOLO` variants.  TEST INFRASTRUCTURE.

  ConvNeXtBiFPNYOLO      canonical, `/root/reference/src/main_model.py:300-393`
  ConvNeXtBiFPNYOLOv2    Segment-only, `/root/reference/src/main_modelv2.py:300-385`

Same constructor, same submodule names (=> same state_dict keys), same `forward(x, mode)` output
layouts and the same head `.training` flag handling as the reference.
"""
import torch
import torch.nn as nn

from .blocks import BiFPN, C2f
from .convnext import ConvNeXtTinyFeatures
from .heads import Detect, Segment


class ConvNeXtTiny(nn.Module):
    """main_model.py:12-38: timm features (P3 192, P4 384, P5 768) + three C2f adaptors."""

    def __init__(self, pretrained: bool = True):
        super().__init__()
        if pretrained:
            # main_model.py:21-26 would download weights by name; there is no network here.
            raise RuntimeError("oracle: pretrained ConvNeXt weights are not available offline; "
                               "pass pretrained_backbone=False and load a state_dict")
        self.body = ConvNeXtTinyFeatures().eval()  # main_model.py:27
        self.c2f_p3 = C2f(192, 256)
        self.c2f_p4 = C2f(384, 384)
        self.c2f_p5 = C2f(768, 512)
        self.out_channels = self.body.channels()

    def forward(self, x):
        p3, p4, p5 = self.body(x)
        return self.c2f_p3(p3), self.c2f_p4(p4), self.c2f_p5(p5)


class ConvNeXtBiFPNYOLO(nn.Module):
    def __init__(self, nc_det, nc_img, proto_ch=32, bifpn_feature_size=256, bifpn_num_layers=2,
                 pretrained_backbone=True):
        super().__init__()
        self.backbone = ConvNeXtTiny(pretrained=pretrained_backbone)
        self.neck = BiFPN(size=[256, 384, 512], feature_size=bifpn_feature_size, num_layers=bifpn_num_layers)
        ch = [bifpn_feature_size] * 3
        self.detect = Detect(nc=nc_det, ch=ch)
        self.segment = Segment(nc=nc_det, nm=proto_ch, npr=bifpn_feature_size, ch=ch)
        self.cls_pool = nn.AdaptiveAvgPool2d(1)
        self.cls_fc = nn.Linear(bifpn_feature_size, nc_img)
        self.nc_det, self.nc_img, self.proto_ch = nc_det, nc_img, proto_ch

    def forward(self, x, mode: str = "train"):
        n3, n4, n5 = self.neck(self.backbone(x))
        heads_in = [n3, n4, n5]
        det_flag, seg_flag = self.detect.training, self.segment.training
        try:
            if mode == "train":  # main_model.py:357-365
                self.detect.train()
                self.segment.train()
                det = self.detect(list(heads_in))
                seg = self.segment(list(heads_in))
                logits = self.cls_fc(self.cls_pool(n5).flatten(1))
                return det, seg, logits
            if mode == "infer":  # main_model.py:367-386
                self.detect.eval()
                self.segment.eval()
                det_cat, det_feats = self.detect(list(heads_in))
                seg_cat, protos = self.segment(list(heads_in))
                logits = self.cls_fc(self.cls_pool(n5).flatten(1))
                return {
                    "detect_features": det_feats,
                    "detect_preds_cat": det_cat,
                    "segment_protos": protos,
                    "segment_preds_cat": seg_cat,
                    "img_cls_logits": logits,
                    "img_cls_probs": logits.softmax(dim=1),
                }
            raise ValueError(f"Unknown mode for ConvNeXtBiFPNYOLO.forward: {mode}. Expected 'train' or 'infer'.")
        finally:  # only the top-level flags are restored (main_model.py:391-393)
            self.detect.training = det_flag
            self.segment.training = seg_flag


class ConvNeXtBiFPNYOLOv2(nn.Module):
    """main_modelv2.py: no separate Detect head; det preds are a slice of the Segment output."""

    def __init__(self, nc_det, nc_img, proto_ch=32, bifpn_feature_size=256, bifpn_num_layers=2,
                 pretrained_backbone=True):
        super().__init__()
        self.backbone = ConvNeXtTiny(pretrained=pretrained_backbone)
        self.neck = BiFPN(size=[256, 384, 512], feature_size=bifpn_feature_size, num_layers=bifpn_num_layers)
        ch = [bifpn_feature_size] * 3
        self.segment = Segment(nc=nc_det, nm=proto_ch, npr=bifpn_feature_size, ch=ch)
        self.cls_pool = nn.AdaptiveAvgPool2d(1)
        self.cls_fc = nn.Linear(bifpn_feature_size, nc_img)
        self.nc_det, self.nc_img, self.proto_ch = nc_det, nc_img, proto_ch

    def forward(self, x, mode: str = "train"):
        n3, n4, n5 = self.neck(self.backbone(x))
        heads_in = [n3, n4, n5]
        seg_flag = self.segment.training
        try:
            if mode == "train":  # main_modelv2.py:353-360
                self.segment.train()
                seg = self.segment(list(heads_in))
                return seg, self.cls_fc(self.cls_pool(n5).flatten(1))
            if mode == "infer":  # main_modelv2.py:362-378
                self.segment.eval()
                seg_cat, protos = self.segment(list(heads_in))
                logits = self.cls_fc(self.cls_pool(n5).flatten(1))
                return {
                    "detect_preds_cat": seg_cat[:, : 4 + self.nc_det],
                    "segment_protos": protos,
                    "segment_preds_cat": seg_cat,
                    "img_cls_logits": logits,
                    "img_cls_probs": logits.softmax(dim=1),
                }
            raise ValueError(f"Unknown mode for ConvNeXtBiFPNYOLO.forward: {mode}. Expected 'train' or 'infer'.")
        finally:
            self.segment.training = seg_flag


@torch.no_grad()
def randomize_(model: nn.Module, seed: int = 0) -> nn.Module:
    """Seeded synthetic weights per SURVEY 8(d): default inits, BN running stats randomised,
    layer-scale gamma ~ U(.05,.15), BiFPN w1/w2 = 1 (F7).  Used by tests and bench alike."""
    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
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


# ---------------------------------------------------------------------------------------------------
# Oldest variant: /root/reference/src/model.py (BASELINE config 0).  ConvNeXt-T features straight into a
# BiFPN with ultralytics lateral Convs, DWConv 3x3 nodes, nearest-x2 / max-pool resampling and the
# WeightedAdd that ADDS its weights (`sum(w_i + f)`, src/model.py:33-36).
# ---------------------------------------------------------------------------------------------------
import torch.nn.functional as _F

from .heads import Conv as _UConv, DWConv as _UDWConv


class WeightedAdd(nn.Module):
    def __init__(self, n, eps=1e-4):
        super().__init__()
        self.w = nn.Parameter(torch.ones(n, dtype=torch.float32))
        self.eps = eps

    def forward(self, feats):
        w = _F.relu(self.w)
        w = w / (w.sum() + self.eps)
        return sum(w_i + f for w_i, f in zip(w, feats))  # sic: adds the weight (src/model.py:36)


class BiFPNUnitV0(nn.Module):
    """src/model.py:39-71."""

    def __init__(self, ch=256):
        super().__init__()
        self.add_p4_td = WeightedAdd(2)
        self.add_p3_td = WeightedAdd(2)
        self.add_p4_out = WeightedAdd(3)
        self.add_p5_out = WeightedAdd(2)
        self.conv = nn.ModuleDict({k: _UDWConv(ch, ch, k=3, s=1) for k in ("p4_td", "p3_td", "p4_out", "p5_out")})

    def forward(self, p3, p4, p5):
        p4_td = self.conv["p4_td"](self.add_p4_td([p4, _F.interpolate(p5, scale_factor=2, mode="nearest")]))
        p3_td = self.conv["p3_td"](self.add_p3_td([p3, _F.interpolate(p4_td, scale_factor=2, mode="nearest")]))
        p4_out = self.conv["p4_out"](self.add_p4_out([p4, p4_td, _F.max_pool2d(p3_td, 2)]))
        p5_out = self.conv["p5_out"](self.add_p5_out([p5, _F.max_pool2d(p4_out, 2)]))
        return p3_td, p4_out, p5_out


class BiFPNV0(nn.Module):
    """src/model.py:74-93."""

    def __init__(self, in_ch=(96, 192, 384), repeats=2):
        super().__init__()
        self.lat3 = _UConv(in_ch[0], 256, 1, 1)
        self.lat4 = _UConv(in_ch[1], 256, 1, 1)
        self.lat5 = _UConv(in_ch[2], 256, 1, 1)
        self.units = nn.ModuleList([BiFPNUnitV0(256) for _ in range(repeats)])

    def forward(self, feats):
        p3, p4, p5 = feats
        p3, p4, p5 = self.lat3(p3), self.lat4(p4), self.lat5(p5)
        for layer in self.units:
            p3, p4, p5 = layer(p3, p4, p5)
        return p3, p4, p5


class _BackboneV0(nn.Module):
    """src/model.py:13-23 (timm features only; `pretrained=True` there would need the network)."""

    def __init__(self):
        super().__init__()
        self.body = ConvNeXtTinyFeatures()
        self.out_channels = self.body.channels()

    def forward(self, x):
        return self.body(x)


class ConvNeXtBiFPNYOLOv0(nn.Module):
    """src/model.py:97-123."""

    def __init__(self, nc_det: int, nc_img: int, proto_ch: int = 32):
        super().__init__()
        self.backbone = _BackboneV0()
        self.neck = BiFPNV0(self.backbone.out_channels, repeats=2)
        ch = (256, 256, 256)
        self.detect = Detect(nc_det, ch=ch)
        self.segment = Segment(nc_det, nm=proto_ch, ch=ch)
        self.cls_pool = nn.AdaptiveAvgPool2d(1)
        self.cls_fc = nn.Linear(256, nc_img)

    def forward(self, x, mode: str = "infer"):
        p3, p4, p5 = self.neck(self.backbone(x))
        det_out = self.detect([p3, p4, p5])
        seg_out = self.segment([p3, p4, p5])
        img_logits = self.cls_fc(self.cls_pool(p5).flatten(1))
        if mode == "infer":
            return {"detect": det_out, "segment": (seg_out[0], seg_out[1]), "img_cls": img_logits.softmax(1)}
        return det_out, seg_out, img_logits
