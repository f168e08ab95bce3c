"""Oracle restatement of the reference's per-sample input pipeline (`/root/reference/src/dataset_btxrdv2.py:109-166`
image work, `:168-245` labels, `:261-284` collate).

TEST INFRASTRUCTURE (CPU, numpy).  PARITY UNPINNED for the resize arithmetic: the reference calls `cv2.resize`,
`cv2.copyMakeBorder`, `cv2.cvtColor` (OpenCV; `opencv-python`, version not pinned in src/requirements.txt) and cv2 is
not installed here, the reference holds no image fixtures or tests for this path, so no vector produced by cv2 itself
anchors this file.  What is restated is OpenCV's published 8-bit algorithm (modules/imgproc/src/resize.cpp):

  INTER_LINEAR, 8-bit: per destination index  f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s;
      s < 0 -> (s, f) = (0, 0);  s >= size - 1 -> (size - 1, 0);  coefficients short(cvRound((1 - f) * 2048)), short(cvRound(f * 2048))
      (cvRound = round half to even);  horizontal pass in int32: r = S[s] * a0 + S[s + 1] * a1;
      vertical pass: dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
      (the exact-2x case OpenCV routes to INTER_AREA gives the same integers: (a + b + c + d + 2) >> 2)
  INTER_NEAREST: s = min(floor(d * scale), size - 1)
  scale = 1.0 / (double(dst_size) / src_size)

tests/test_oracle_golden.py checks the properties that hold for cv2 by construction (identity at equal size, 2x2 box
average at exact halving, constants preserved, agreement with float bilinear within 1 level); tests/test_gpu_preprocess.py
checks the HIP kernel against this file bit for bit.
"""
import numpy as np


def _linear_taps(dst: int, src: int):
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int32)
    c1 = np.rint(f * np.float32(2048.0)).astype(np.int32)
    return s, np.minimum(s + 1, src - 1), c0, c1


def resize_linear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 [H, W, C]."""
    H0, W0 = img.shape[:2]
    sx0, sx1, a0, a1 = _linear_taps(new_w, W0)
    sy0, sy1, b0, b1 = _linear_taps(new_h, H0)
    src = img.astype(np.int32)
    rows = src[:, sx0] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]          # [H0, new_w, C]
    r0, r1 = rows[sy0] >> 4, rows[sy1] >> 4
    out = (((b0[:, None, None] * r0) >> 16) + ((b1[:, None, None] * r1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_nearest_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_NEAREST) for uint8 [H, W]."""
    H0, W0 = img.shape[:2]
    sx = np.minimum(np.floor(np.arange(new_w) * (1.0 / (float(new_w) / W0))).astype(np.int64), W0 - 1)
    sy = np.minimum(np.floor(np.arange(new_h) * (1.0 / (float(new_h) / H0))).astype(np.int64), H0 - 1)
    return img[sy][:, sx]


def letterbox(img_bgr: np.ndarray, mask: np.ndarray, S: int):
    """`_letterbox` (:109-134) followed by :157-166.  Returns (img_t [3,S,S] float32 RGB, mask_t [1,S,S] float32, scale)."""
    H0, W0 = img_bgr.shape[:2]
    scale = S / max(H0, W0)
    new_w, new_h = max(1, int(W0 * scale)), max(1, int(H0 * scale))
    canvas = np.full((S, S, 3), 114, dtype=np.uint8)                       # copyMakeBorder(BORDER_CONSTANT, 114), top-left
    canvas[:new_h, :new_w] = resize_linear_u8(img_bgr, new_w, new_h)
    mcanvas = np.zeros((S, S), dtype=np.uint8)
    if mask is not None:
        mcanvas[:new_h, :new_w] = resize_nearest_u8(mask, new_w, new_h)
    img_t = (canvas[:, :, ::-1].astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)     # BGR2RGB, /255, HWC->CHW
    mask_t = ((mcanvas.astype(np.float32) / np.float32(255.0)) > 0.5).astype(np.float32)[None]
    return np.ascontiguousarray(img_t), mask_t, scale


def yolo_labels(rows, W0, H0, scale, S):
    """:173-245, written the way the reference writes it (np.clip on Python floats)."""
    out = []
    for r in rows:
        if len(r) < 5:
            continue
        cls_label, xc, yc, w, h = map(float, r[:5])
        if w <= 0 or h <= 0:
            continue
        abs_xc, abs_yc, abs_w, abs_h = xc * W0, yc * H0, w * W0, h * H0
        sx1, sy1 = (abs_xc - abs_w / 2) * scale, (abs_yc - abs_h / 2) * scale
        sx2, sy2 = (abs_xc + abs_w / 2) * scale, (abs_yc + abs_h / 2) * scale
        fw, fh = sx2 - sx1, sy2 - sy1
        if fw < 1.0 or fh < 1.0:
            continue
        xcn, ycn, wn, hn = ((sx1 + sx2) / 2) / S, ((sy1 + sy2) / 2) / S, fw / S, fh / S
        x1, y1 = np.clip(xcn - wn / 2, 0.0, 1.0), np.clip(ycn - hn / 2, 0.0, 1.0)
        x2, y2 = np.clip(xcn + wn / 2, 0.0, 1.0), np.clip(ycn + hn / 2, 0.0, 1.0)
        cw, ch = x2 - x1, y2 - y1
        if cw < 1.0 / S or ch < 1.0 / S:
            continue
        out.append([0.0, float(cls_label), float((x1 + x2) / 2), float((y1 + y2) / 2), float(cw), float(ch)])
    return out
