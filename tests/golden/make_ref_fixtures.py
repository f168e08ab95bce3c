"""Generate `ref_blocks.pt`: golden input/output vectors from the REAL reference classes.

Run in the build container only (needs /root/reference; never runs on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ref_fixtures.py

`/root/reference/src/main_model.py` imports timm and ultralytics at module scope; neither is
installed.  Empty placeholder modules are seeded into `sys.modules` for those names only so that the
file imports; every class exercised below (ConvBlock, Bottleneck, C2f, DepthwiseConvBlock, BiFPNUnit,
BiFPN, autopad) is pure-torch reference code that runs unmodified.  The same is done for
`running_main_v3.py` (Lightning / torchmetrics / torchvision / wandb / seaborn placeholders) to reach the
pure functions `batch_bbox_iou` and `dist2bbox`.  The fixture holds data only: seeded inputs,
state_dicts and the reference's outputs.
"""
import os
import sys
import types

import torch

REF_SRC = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_blocks.pt")


class _Anything:
    """Placeholder for absent third-party symbols: subclassable, callable, attribute-transparent."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


def _placeholder(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package so sub-imports resolve
    m.__getattr__ = lambda attr: type(attr, (_Anything,), {})
    sys.modules[name] = m
    return m


def import_reference():
    for name in [
        "timm", "ultralytics", "ultralytics.nn", "ultralytics.nn.modules", "ultralytics.nn.modules.conv",
        "ultralytics.nn.modules.head", "ultralytics.nn.modules.block", "ultralytics.utils",
        "ultralytics.utils.torch_utils", "ultralytics.utils.ops",
        "pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.loggers", "lightning",
        "torchmetrics", "torchmetrics.classification", "torchmetrics.detection", "torchmetrics.segmentation",
        "torchvision", "torchvision.ops", "wandb", "seaborn", "cv2", "matplotlib", "matplotlib.pyplot",
        "dataset_btxrdv2", "multitask_logging", "sklearn", "sklearn.metrics", "PIL", "PIL.Image",
    ]:
        try:
            __import__(name)
        except Exception:
            _placeholder(name)
    sys.path.insert(0, REF_SRC)
    import main_model as ref_model  # noqa
    try:
        import running_main_v3 as ref_train  # noqa
    except Exception as e:  # pure functions are optional extras
        print("running_main_v3 not importable:", repr(e))
        ref_train = None
    return ref_model, ref_train


@torch.no_grad()
def randomize(m, gen):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=gen) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=gen) + 0.5)
            mod.weight.copy_(torch.rand(mod.num_features, generator=gen) * 0.5 + 0.75)
            mod.bias.copy_(torch.randn(mod.num_features, generator=gen) * 0.1)
    for n, p in m.named_parameters():
        if n.endswith("w1") or n.endswith("w2"):
            p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
    return m


def main():
    ref, ref_train = import_reference()
    gen = torch.Generator().manual_seed(1234)
    cases = {}

    def run(name, module, inputs, train=False):
        torch.manual_seed(7)
        randomize(module, gen)
        sd = {k: v.clone() for k, v in module.state_dict().items()}
        module.train(train)
        with torch.no_grad():
            out = module(*inputs)
        post = {k: v.clone() for k, v in module.state_dict().items()} if train else None
        outs = [o.clone() for o in out] if isinstance(out, (list, tuple)) else out.clone()
        cases[name] = {"state_dict": sd, "inputs": inputs, "output": outs, "train": train,
                       "state_dict_after": post}

    r = lambda *s: torch.randn(*s, generator=gen)
    torch.manual_seed(0)
    run("ConvBlock_3x3", ref.ConvBlock(16, 32, 3, 1), (r(2, 16, 12, 12),))
    run("ConvBlock_1x1", ref.ConvBlock(24, 16, 1), (r(2, 24, 9, 7),))
    run("ConvBlock_3x3_train", ref.ConvBlock(16, 32, 3, 1), (r(3, 16, 10, 10),), train=True)
    run("Bottleneck", ref.Bottleneck(16, 16, False, kernel=(3, 3), e=1.0), (r(2, 16, 10, 10),))
    run("Bottleneck_add", ref.Bottleneck(16, 16, True), (r(2, 16, 10, 10),))
    run("C2f", ref.C2f(24, 32), (r(2, 24, 12, 12),))
    run("DepthwiseConvBlock", ref.DepthwiseConvBlock(32, 32), (r(2, 32, 8, 8),))
    run("BiFPNUnit", ref.BiFPNUnit(32), ([r(2, 32, 16, 16), r(2, 32, 8, 8), r(2, 32, 4, 4)],))
    run("BiFPN", ref.BiFPN([16, 24, 32], 32, 2), ([r(1, 16, 16, 16), r(1, 24, 8, 8), r(1, 32, 4, 4)],))
    cases["autopad"] = {"inputs": [(1, None, 1), (3, None, 1), (7, None, 1), (3, None, 2), (3, 0, 1)],
                        "output": [ref.autopad(*a) for a in [(1, None, 1), (3, None, 1), (7, None, 1), (3, None, 2), (3, 0, 1)]]}

    if ref_train is not None:
        xy = torch.rand(40, 2, generator=gen) * 600
        wh = torch.rand(40, 2, generator=gen) * 80
        b1 = torch.cat([xy, xy + wh], 1)
        xy2 = torch.rand(5, 2, generator=gen) * 600
        wh2 = torch.rand(5, 2, generator=gen) * 200
        b2 = torch.cat([xy2, xy2 + wh2], 1)
        cases["batch_bbox_iou"] = {"inputs": (b1, b2), "output": ref_train.batch_bbox_iou(b1, b2)}
        cases["batch_bbox_iou_empty"] = {"inputs": (b1, b2[:0]), "output": ref_train.batch_bbox_iou(b1, b2[:0])}
        d = torch.rand(3, 50, 4, generator=gen) * 10
        a = torch.rand(3, 50, 2, generator=gen) * 80
        cases["dist2bbox_xyxy"] = {"inputs": (d, a), "output": ref_train.dist2bbox(d, a, "xyxy")}
        cases["dist2bbox_xywh"] = {"inputs": (d, a), "output": ref_train.dist2bbox(d, a, "xywh")}
        cases["constants"] = {"CONF_TH": ref_train.CONF_TH, "NMS_IOU": ref_train.NMS_IOU, "TOP_K": ref_train.TOP_K}

    # src/model.py (oldest variant): WeightedAdd is pure torch and ADDS its weights
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_model_v0", os.path.join(REF_SRC, "model.py"))
    v0 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(v0)
    for n in (2, 3):
        wa = v0.WeightedAdd(n)
        with torch.no_grad():
            wa.w.copy_(torch.rand(n, generator=gen) + 0.25)
            feats = [r(2, 8, 6, 5) for _ in range(n)]
            cases[f"WeightedAdd_{n}"] = {"w": wa.w.detach().clone(), "inputs": feats, "output": wa(feats).clone()}

    torch.save(cases, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", list(cases))


if __name__ == "__main__":
    main()
