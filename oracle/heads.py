"""Oracle restatement of the ultralytics heads the reference instantiates stand-alone at
`/root/reference/src/main_model.py:324-328` (`Detect(nc, ch)`, `Segment(nc, nm, npr, ch)`).

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned**: ultralytics is third-party, not
vendored, not installed, version unpinned (`src/requirements.txt:279`).  Restated from the published
modules `ultralytics/nn/modules/{conv,block,head}.py` + `utils/tal.py`, the >=8.3 ("non-legacy")
generation the reference's checkpoint name `yolo11m-seg.pt` (`src/model.py:131`) implies:

  Conv    = Conv2d(bias=False, pad=k//2) + BatchNorm2d(default eps 1e-5, momentum 0.1) + SiLU
  DWConv  = Conv with groups=gcd(c1,c2)
  Detect  : per level  cv2 = Conv(c,c2,3) Conv(c2,c2,3) Conv2d(c2,64,1),   c2 = max(16, ch0//4, 64)
                       cv3 = [DWConv(c,c,3) Conv(c,c3,1)] [DWConv(c3,c3,3) Conv(c3,c3,1)] Conv2d(c3,nc,1),
                             c3 = max(ch0, min(nc,100))
            train: list of cat(cv2_i, cv3_i);  eval: (cat(dbox*stride, sigmoid(cls)) [B,4+nc,A], list)
            `stride` stays zeros(nl) unless a caller sets it.
  Segment : Detect + cv4 = Conv(c,c4,3) Conv(c4,c4,3) Conv2d(c4,nm,1), c4 = max(ch0//4, nm) + Proto on P3
  Proto   : Conv(c,npr,3) -> ConvTranspose2d(npr,npr,2,2,bias) -> Conv(npr,npr,3) -> Conv(npr,nm,1)
  DFL     : softmax over reg_max bins, expectation with arange(reg_max) (a frozen 1x1 conv)
"""
import math

import torch
import torch.nn as nn


class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, g=1):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class DWConv(Conv):
    def __init__(self, c1, c2, k=1, s=1):
        super().__init__(c1, c2, k, s, g=math.gcd(c1, c2))


class DFL(nn.Module):
    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        b, _, a = x.shape
        return self.conv(x.view(b, 4, self.c1, a).transpose(2, 1).softmax(1)).view(b, 4, a)


class Proto(nn.Module):
    def __init__(self, c1, c_=256, c2=32):
        super().__init__()
        self.cv1 = Conv(c1, c_, 3)
        self.upsample = nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2 = Conv(c_, c_, 3)
        self.cv3 = Conv(c_, c2)

    def forward(self, x):
        return self.cv3(self.cv2(self.upsample(self.cv1(x))))


def make_anchors(feats, strides, offset=0.5):
    """Anchor centres (grid units) and per-anchor stride, levels concatenated (utils/tal.py)."""
    pts, st = [], []
    for f, s in zip(feats, strides):
        h, w = f.shape[2:]
        sx = torch.arange(w, dtype=f.dtype, device=f.device) + offset
        sy = torch.arange(h, dtype=f.dtype, device=f.device) + offset
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((gx, gy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=f.dtype, device=f.device))
    return torch.cat(pts), torch.cat(st)


def dist2bbox_xywh(distance, anchor_points, dim=1):
    lt, rb = distance.chunk(2, dim)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)


class Detect(nn.Module):
    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2 = max(16, ch[0] // 4, self.reg_max * 4)
        c3 = max(ch[0], min(nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch
        )
        self.cv3 = nn.ModuleList(
            nn.Sequential(
                nn.Sequential(DWConv(x, x, 3), Conv(x, c3, 1)),
                nn.Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)),
                nn.Conv2d(c3, nc, 1),
            )
            for x in ch
        )
        self.dfl = DFL(self.reg_max)

    def _levels(self, x):
        return [torch.cat((self.cv2[i](x[i]), self.cv3[i](x[i])), 1) for i in range(self.nl)]

    def _inference(self, feats):
        b = feats[0].shape[0]
        x_cat = torch.cat([f.view(b, self.no, -1) for f in feats], 2)
        anchors, strides = (t.transpose(0, 1) for t in make_anchors(feats, self.stride, 0.5))
        box, cls = x_cat.split((self.reg_max * 4, self.nc), 1)
        dbox = dist2bbox_xywh(self.dfl(box), anchors.unsqueeze(0), dim=1) * strides
        return torch.cat((dbox, cls.sigmoid()), 1)

    def forward(self, x):
        feats = self._levels(x)
        if self.training:
            return feats
        return self._inference(feats), feats


class Segment(Detect):
    def __init__(self, nc=80, nm=32, npr=256, ch=()):
        super().__init__(nc, ch)
        self.nm = nm
        self.npr = npr
        self.proto = Proto(ch[0], npr, nm)
        c4 = max(ch[0] // 4, nm)
        self.cv4 = nn.ModuleList(
            nn.Sequential(Conv(x, c4, 3), Conv(c4, c4, 3), nn.Conv2d(c4, nm, 1)) for x in ch
        )

    def forward(self, x):
        p = self.proto(x[0])
        b = p.shape[0]
        mc = torch.cat([self.cv4[i](x[i]).view(b, self.nm, -1) for i in range(self.nl)], 2)
        feats = self._levels(x)
        if self.training:
            return feats, mc, p
        y = self._inference(feats)
        return torch.cat([y, mc], 1), (feats, mc, p)
