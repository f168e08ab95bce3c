"""Checkpoint helpers with the reference's names.

`load_pretrained_heads` is what `running_main_v3.py:38` imports next to the model
(`/root/reference/src/main_model.py:399-603`): it copies the parameters of a pretrained ultralytics Detect / Segment head
into `model.detect` / `model.segment`, name by name, wherever the shapes agree (parameters only -- BatchNorm running
statistics are not touched, exactly like the reference's `named_parameters()` walk), and reports how many tensors moved.

The reference unpickles the YOLO checkpoint through ultralytics (`YOLO(path).model`).  That package is not available here
and a pickled model object is never loaded by this code: the source has to be a FLAT STATE DICT -- a dict, a `.safetensors`
file, or a file `torch.load(..., weights_only=True)` accepts -- with the checkpoint's own key names (`model.<N>.cv2.0.0.conv.weight`
...; produce one with `torch.save(YOLO(p).model.state_dict(), out)` wherever ultralytics is installed).  The head is the
highest-numbered `model.<N>.` block that owns the head's parameter names, as the reference takes the LAST Detect / Segment
module of the layer list.

`strip_lightning_prefix` maps a Lightning checkpoint's `state_dict` (`net.` prefix, `running_main_v3.py:181`) onto the model."""
import re
from typing import Dict, Optional, Union

import torch

StateSource = Union[str, Dict[str, torch.Tensor], None]


def _flat_state(src: StateSource) -> Optional[Dict[str, torch.Tensor]]:
    if src is None:
        return None
    if isinstance(src, dict):
        sd = src
    elif str(src).endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(str(src))
    else:
        try:
            sd = torch.load(str(src), map_location="cpu", weights_only=True)
        except Exception as e:  # a pickled ultralytics model object
            raise RuntimeError(f"{src}: not a plain state dict (weights_only load refused: {e}).  Convert it where ultralytics "
                               "is installed: torch.save(YOLO(path).model.state_dict(), 'heads_state.pt')") from e
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise RuntimeError("expected a flat {name: tensor} state dict")
    return sd


def _head_prefix(sd: Dict[str, torch.Tensor], need: str) -> Optional[str]:
    """Prefix ('model.23.') of the last layer block that has `<prefix><need>`; '' if the dict is already head-relative."""
    if need in sd:
        return ""
    best = None
    for k in sd:
        m = re.match(r"^((?:model\.)+(\d+)\.)" + re.escape(need) + "$", k)
        if m and (best is None or int(m.group(2)) > best[0]):
            best = (int(m.group(2)), m.group(1))
    return None if best is None else best[1]


def _copy_params(sd, prefix, dst_mod, sub, label):
    """Reference `copy_named_params`: every parameter of dst_mod.<sub> whose name exists in the source with the same shape."""
    copied = total = 0
    mod = dst_mod
    for part in filter(None, sub.split(".")):
        mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
    for name, p in mod.named_parameters():
        total += 1
        key = f"{prefix}{sub + '.' if sub else ''}{name}"
        src = sd.get(key)
        if src is None:
            print(f"    Param '{name}' not found in source module {label}.")
        elif tuple(src.shape) != tuple(p.shape):
            print(f"    Shape mismatch for {label} param '{name}': src {tuple(src.shape)}, dst {tuple(p.shape)}")
        else:
            p.copy_(src.to(device=p.device, dtype=p.dtype))
            copied += 1
    return copied, total


@torch.no_grad()
def load_pretrained_heads(model, detect_ckpt_path: StateSource = None, segment_ckpt_path: StateSource = None):
    """Same call and RETURN VALUE as the reference's (`main_model.py:400-402`, `:603`): the model itself -- its callers rebind it
    (`net = load_pretrained_heads(net, ...)`, `main_model.py:645`).  The per-head counts {'detect': (copied, total), 'segment': ...} are
    left on `model._head_load_report`."""
    report = {"detect": (0, 0), "segment": (0, 0)}
    det_sd = _flat_state(detect_ckpt_path)
    if getattr(model, "detect", None) is None:
        print("Destination model has no 'detect' attribute or it's None.")
    elif det_sd is None or _head_prefix(det_sd, "cv2.0.0.conv.weight") is None:
        print(f"No source Detect head found or loaded from {detect_ckpt_path} to copy to model.detect.")
    else:
        report["detect"] = _copy_params(det_sd, _head_prefix(det_sd, "cv2.0.0.conv.weight"), model.detect, "", "Detect Head")
        print(f"Detect head          : {report['detect'][0]}/{report['detect'][1]} tensors copied from {detect_ckpt_path}")
    seg_sd = _flat_state(segment_ckpt_path)
    if getattr(model, "segment", None) is None:
        print("Destination model has no 'segment' attribute or it's None.")
    elif seg_sd is None or _head_prefix(seg_sd, "proto.cv1.conv.weight") is None:
        print(f"No source Segment head found or loaded from {segment_ckpt_path} to copy to model.segment.")
    else:
        pre = _head_prefix(seg_sd, "proto.cv1.conv.weight")
        c_tot = t_tot = 0
        seg = model.segment                                                 # the reference's order: cv4, proto, cv2, cv3
        subs = ([f"cv4.{i}" for i in range(len(seg.cv4))] + ["proto"] + [f"cv2.{i}" for i in range(len(seg.cv2))]
                + [f"cv3.{i}" for i in range(len(seg.cv3))])
        for sub in subs:
            c, t = _copy_params(seg_sd, pre, model.segment, sub, f"Segment.{sub}")
            c_tot, t_tot = c_tot + c, t_tot + t
        report["segment"] = (c_tot, t_tot)
        print(f"Segment head         : {c_tot}/{t_tot} tensors copied from {segment_ckpt_path}")
    total_c, total_t = report["detect"][0] + report["segment"][0], report["detect"][1] + report["segment"][1]
    print(f"\nHead-weight summary  : {total_c}/{total_t} tensors copied overall.")
    model.__dict__["_head_load_report"] = report
    if hasattr(model, "mark_weights_updated"):
        model.mark_weights_updated()
    return model


def strip_lightning_prefix(state_dict: Dict[str, torch.Tensor], prefix: str = "net.") -> Dict[str, torch.Tensor]:
    """Lightning `.ckpt['state_dict']` -> model state_dict (drops the trainer's own entries such as `seg_proto_projector.*`)."""
    return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
