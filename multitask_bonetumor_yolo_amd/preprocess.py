"""Input pipeline of a batch on the device: what `BTXRDDataset.__getitem__` + `collate_fn`
(`/root/reference/src/dataset_btxrdv2.py:109-166`, `:168-245`, `:261-284`) do per sample on the CPU with cv2.

`letterbox_batch` is the image work (one HIP launch per <= 32 images, no CPU path); `transform_yolo_labels` and
`collate_boxes` are the label arithmetic, which is a handful of Python-float operations per box and stays on the host
exactly as the reference writes it.
"""
import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib as L


def letterbox_batch(images: Sequence[torch.Tensor], masks: Optional[Sequence[Optional[torch.Tensor]]] = None, img_size: int = 640):
    """images: decoded BGR uint8 [H0, W0, 3] CUDA tensors (any sizes); masks: uint8 [H0, W0] CUDA tensors or None.
    Returns (imgs [B,3,S,S] f32 RGB in [0,1], masks [B,1,S,S] f32 {0,1}, scales list[float]) -- `img_t`, `mask_t` and
    `scale` of dataset_btxrdv2.py:153-166 for every sample, stacked like `collate_fn` (:264-265)."""
    lib = L.load()
    B = len(images)
    if B == 0:
        raise ValueError("letterbox_batch: empty batch")
    dev = images[0].device
    descs = (L.RawImage * B)()
    keep = []
    for i, im in enumerate(images):
        if not im.is_cuda:
            raise RuntimeError("letterbox_batch: expected CUDA/HIP tensors on an MI355X (no CPU path)")
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError("letterbox_batch: images must be uint8 [H, W, 3] (BGR, as cv2.imread returns them)")
        if im.stride(2) != 1 or im.stride(1) != 3:
            im = im.contiguous()
        keep.append(im)
        d = descs[i]
        d.bgr, d.height, d.width, d.row_stride = im.data_ptr(), im.shape[0], im.shape[1], im.stride(0)
        mk = masks[i] if masks is not None else None
        if mk is not None:
            if mk.dtype != torch.uint8 or tuple(mk.shape) != tuple(im.shape[:2]) or not mk.is_cuda:
                raise ValueError("letterbox_batch: mask must be a CUDA uint8 [H, W] tensor of the image's size")
            if mk.stride(1) != 1:
                mk = mk.contiguous()
            keep.append(mk)
            d.mask, d.mask_row_stride = mk.data_ptr(), mk.stride(0)
        else:
            d.mask, d.mask_row_stride = None, 0
    out = torch.empty(B, 3, img_size, img_size, device=dev, dtype=torch.float32)
    out_m = torch.empty(B, 1, img_size, img_size, device=dev, dtype=torch.float32)
    scales = (C.c_double * B)()
    L.check(lib.mtbt_letterbox_batch(descs, B, img_size, out.data_ptr(), out_m.data_ptr(), scales,
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtbt_letterbox_batch")
    for t in keep:   # the launch is asynchronous: keep the sources alive until the stream has consumed them
        t.record_stream(torch.cuda.current_stream(dev))
    return out, out_m, [float(s) for s in scales]


def transform_yolo_labels(rows: Sequence[Sequence[float]], W0: int, H0: int, scale: float, img_size: int) -> List[List[float]]:
    """YOLO-txt rows (cls, xc, yc, w, h normalised to the ORIGINAL image) -> the reference's per-sample `det_rows`
    [0.0, cls, xc, yc, w, h] normalised to the letterboxed S x S image (dataset_btxrdv2.py:173-245): boxes with
    non-positive size, under one pixel after scaling, or under 1/S after clamping to [0, 1] are dropped."""
    out = []
    min_norm = 1.0 / img_size
    clip = lambda v: min(max(v, 0.0), 1.0)
    for r in rows:
        if len(r) < 5:
            continue
        cls, xc, yc, w, h = (float(v) for v in r[:5])
        if w <= 0 or h <= 0:
            continue
        axc, ayc, aw, ah = xc * W0, yc * H0, w * W0, h * H0
        x1, y1, x2, y2 = (axc - aw / 2) * scale, (ayc - ah / 2) * scale, (axc + aw / 2) * scale, (ayc + ah / 2) * scale
        fw, fh = x2 - x1, y2 - y1
        if fw < 1.0 or fh < 1.0:
            continue
        cxn, cyn, wn, hn = ((x1 + x2) / 2) / img_size, ((y1 + y2) / 2) / img_size, fw / img_size, fh / img_size
        nx1, ny1, nx2, ny2 = clip(cxn - wn / 2), clip(cyn - hn / 2), clip(cxn + wn / 2), clip(cyn + hn / 2)
        cw, ch = nx2 - nx1, ny2 - ny1
        if cw < min_norm or ch < min_norm:
            continue
        out.append([0.0, cls, (nx1 + nx2) / 2, (ny1 + ny2) / 2, cw, ch])
    return out


def collate_boxes(per_sample_rows: Sequence[Sequence[Sequence[float]]], device=None) -> torch.Tensor:
    """`collate_fn` (:267-281): stamp the batch index into column 0 and concatenate -> [N, 6] float32."""
    rows = []
    for i, sample in enumerate(per_sample_rows):
        for r in sample:
            rows.append([float(i)] + [float(v) for v in r[1:6]])
    t = torch.tensor(rows, dtype=torch.float32) if rows else torch.zeros((0, 6), dtype=torch.float32)
    return t.to(device) if device is not None else t
