"""Oracle restatement of the timm ConvNeXt-Tiny feature extractor the reference builds at
`/root/reference/src/main_model.py:21-26`:

    timm.create_model("convnext_tiny.in12k_ft_in1k", features_only=True, out_indices=(1, 2, 3))

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned**: timm is a third-party package
that is neither vendored in /root/reference nor installed here, and the reference pins no version
(`src/requirements.txt:259` is the bare name).  What follows restates the published architecture
(timm `models/convnext.py`, Liu et al. 2022): stem Conv 4x4/4 + LayerNorm2d; stages with depths
(3,3,9,3) and dims (96,192,384,768); stages 1..3 start with LayerNorm2d + Conv 2x2/2; block =
depthwise 7x7 (pad 3, bias) -> channels-last LayerNorm(eps 1e-6) -> Linear d->4d -> GELU(erf)
-> Linear 4d->d -> * gamma (layer scale, init 1e-6) -> + input.  `features_only` wraps the net in a
FeatureListNet with flattened child names (`stem_0`, `stem_1`, `stages_0` ... `stages_3`) and, with
out_indices=(1,2,3), returns the outputs of stages 1..3 with no final norm.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

DEPTHS = (3, 3, 9, 3)
DIMS = (96, 192, 384, 768)
LN_EPS = 1e-6


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel axis of an NCHW tensor."""

    def __init__(self, c, eps=LN_EPS):
        super().__init__(c, eps=eps)

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class Mlp(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.fc1 = nn.Linear(d, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, d)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class ConvNeXtBlock(nn.Module):
    def __init__(self, d, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(d, d, 7, 1, 3, groups=d, bias=True)
        self.norm = nn.LayerNorm(d, eps=LN_EPS)
        self.mlp = Mlp(d, 4 * d)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(d))

    def forward(self, x):
        y = self.conv_dw(x).permute(0, 2, 3, 1)
        y = self.mlp(self.norm(y)).permute(0, 3, 1, 2)
        return y.mul(self.gamma.reshape(1, -1, 1, 1)) + x


class ConvNeXtStage(nn.Module):
    def __init__(self, cin, cout, depth, downsample: bool):
        super().__init__()
        if downsample:
            self.downsample = nn.Sequential(LayerNorm2d(cin), nn.Conv2d(cin, cout, 2, 2, bias=True))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[ConvNeXtBlock(cout) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class ConvNeXtTinyFeatures(nn.Module):
    """The FeatureListNet view: children stem_0, stem_1, stages_0..3; returns stages 1,2,3."""

    def __init__(self):
        super().__init__()
        self.stem_0 = nn.Conv2d(3, DIMS[0], 4, 4, bias=True)
        self.stem_1 = LayerNorm2d(DIMS[0])
        prev = DIMS[0]
        for i, (d, n) in enumerate(zip(DIMS, DEPTHS)):
            setattr(self, f"stages_{i}", ConvNeXtStage(prev, d, n, downsample=i > 0))
            prev = d

    def channels(self):
        return list(DIMS[1:])

    def forward(self, x):
        x = self.stem_1(self.stem_0(x))
        outs = []
        for i in range(4):
            x = getattr(self, f"stages_{i}")(x)
            if i >= 1:
                outs.append(x)
        return outs
