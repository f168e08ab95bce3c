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

    # MultiTaskLitModel._multitask_loss (running_main_v3.py:232-387), called unbound on a stand-in `self` that carries exactly
    # what the method reads (hparams, the four torch loss modules of :189-192, the 1x1 projector of :186, reg_max)
    if ref_train is not None:
        import torch.nn as nn
        S, B, NC = 64, 3, 2

        def make_self(training, smoothing):
            hp = types.SimpleNamespace(img_size=S, nc_det=NC, proto_ch=32, loss_weight_seg=1.0, loss_weight_box_iou=2.0,
                                       loss_weight_dfl=1.5, loss_weight_cls_det=0.5, loss_weight_img_cls=1.0,
                                       iou_match_thresh=0.5, det_label_smoothing=smoothing)
            proj = nn.Conv2d(32, 1, kernel_size=1)
            with torch.no_grad():
                proj.weight.copy_(torch.randn(1, 32, 1, 1, generator=torch.Generator().manual_seed(5)) * 0.3)
                proj.bias.fill_(-0.2)
            return types.SimpleNamespace(hparams=hp, img_cls_loss_fn=nn.CrossEntropyLoss(), seg_loss_fn=nn.BCEWithLogitsLoss(),
                                         det_cls_loss_fn=nn.BCEWithLogitsLoss(reduction="sum"),
                                         det_dfl_loss_fn=nn.CrossEntropyLoss(reduction="none"), seg_proto_projector=proj,
                                         reg_max=16, project=None, training=training, temp_matched_preds_for_cm=[],
                                         seg_logits_for_logging=None), proj

        # GT boxes (batch_idx, cls, cx, cy, w, h) normalised; image 1 has no box.  The raw maps are steered so that some
        # anchors really overlap a GT box by more than 0.5 (random maps alone would give no positive match).
        gt = torch.tensor([[0, 1, 0.30, 0.35, 0.40, 0.30], [0, 0, 0.70, 0.70, 0.30, 0.35], [2, 1, 0.50, 0.50, 0.60, 0.50]])
        det = []
        for h in (8, 4, 2):
            m = r(B, 64 + NC, h, h) * 0.5
            stride = S / h
            for row in gt:
                b, cx, cy, w_, h_ = int(row[0]), row[2] * S, row[3] * S, row[4] * S, row[5] * S
                for yy in range(h):
                    for xx in range(h):
                        ax, ay = (xx + 0.5) * stride, (yy + 0.5) * stride
                        ltrb = torch.tensor([ax - (cx - w_ / 2), ay - (cy - h_ / 2), (cx + w_ / 2) - ax, (cy + h_ / 2) - ay]) / stride
                        if ltrb.min() > 0.3 and ltrb.max() < 14.0:
                            for k in range(4):   # peak the 16-bin distribution around the target distance
                                bins = torch.arange(16.0)
                                m[b, 16 * k:16 * k + 16, yy, xx] += 6.0 * torch.exp(-0.5 * (bins - ltrb[k]) ** 2 / 0.3)
            det.append(m)
        protos = r(B, 32, 16, 16)
        logits = r(B, 2)
        gt_masks = (torch.rand(B, 1, S, S, generator=gen) > 0.6).float()
        gt_cls = torch.tensor([0, 1, 1])
        for name, training, smoothing in (("loss_train", True, 0.1), ("loss_train_nosmooth", True, 0.0), ("loss_eval", False, 0.1)):
            fake, proj = make_self(training, smoothing)
            with torch.no_grad():
                out = ref_train.MultiTaskLitModel._multitask_loss(fake, [d.clone() for d in det], (None, None, protos.clone()),
                                                                  logits.clone(), gt.clone(), gt_masks.clone(), gt_cls.clone())
            cases[name] = {"det": det, "protos": protos, "logits": logits, "gt_boxes": gt, "gt_masks": gt_masks, "gt_cls": gt_cls,
                           "proj_w": proj.weight.detach().clone(), "proj_b": proj.bias.detach().clone(), "training": training,
                           "smoothing": smoothing, "img_size": S, "nc_det": NC,
                           "output": [o.detach().clone() if isinstance(o, torch.Tensor) else torch.tensor(float(o)) for o in out]}
            print(name, [round(float(o), 5) for o in cases[name]["output"]])

    torch.save(cases, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", list(cases))


if __name__ == "__main__":
    main()
