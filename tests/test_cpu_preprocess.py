"""Input-pipeline oracle (oracle/preprocess.py) and the host-side label arithmetic of the package.

cv2 is absent, so the resize restatement is PARITY UNPINNED against cv2 itself; these tests hold it to the properties that
OpenCV's 8-bit algorithm has by construction and to an independent float bilinear (torch) within one grey level."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import preprocess as O
from multitask_bonetumor_yolo_amd.preprocess import collate_boxes, transform_yolo_labels


def _img(h, w, seed=0, c=3):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, c), dtype=np.uint8)


def test_resize_identity_and_constant():
    a = _img(37, 53)
    assert np.array_equal(O.resize_linear_u8(a, 53, 37), a)
    c = np.full((20, 31, 3), 201, np.uint8)
    assert np.all(O.resize_linear_u8(c, 77, 45) == 201) and np.all(O.resize_linear_u8(c, 9, 7) == 201)


def test_resize_exact_halving_is_box_average():
    a = _img(64, 96, 1).astype(np.int32)
    want = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(O.resize_linear_u8(a.astype(np.uint8), 48, 32), want.astype(np.uint8))


def test_resize_close_to_float_bilinear():
    for (h, w, nh, nw) in [(40, 60, 100, 150), (100, 150, 40, 60), (33, 71, 64, 20), (5, 1, 11, 7)]:
        a = _img(h, w, h)
        t = torch.from_numpy(a).permute(2, 0, 1)[None].float()
        ref = F.interpolate(t, size=(nh, nw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        got = O.resize_linear_u8(a, nw, nh).astype(np.float32)
        assert np.abs(got - ref).max() <= 1.0   # 11-bit coefficients + two truncating shifts: under one level


def test_resize_nearest_integer_ratio():
    m = _img(30, 20, 2, c=1)[:, :, 0]
    assert np.array_equal(O.resize_nearest_u8(m, 10, 15), m[::2, ::2])
    assert np.array_equal(O.resize_nearest_u8(m, 40, 60), np.repeat(np.repeat(m, 2, 0), 2, 1))


def test_letterbox_layout():
    img, mask = _img(100, 50, 3), (_img(100, 50, 4, c=1)[:, :, 0] > 128).astype(np.uint8) * 255
    x, m, scale = O.letterbox(img, mask, 64)
    assert x.shape == (3, 64, 64) and m.shape == (1, 64, 64) and scale == 0.64
    assert np.all(x[:, :, 32:] == np.float32(114) / np.float32(255)) and np.all(m[:, :, 32:] == 0)      # right of the resized image
    small = O.resize_linear_u8(img, 32, 64)
    assert np.array_equal(x[0, :, :32], small[:, :, 2].astype(np.float32) / np.float32(255))               # channel 0 = R = BGR[2]
    assert set(np.unique(m)) <= {0.0, 1.0}


def test_label_transform_matches_oracle_and_collate():
    rng = np.random.default_rng(5)
    rows = [[float(rng.integers(0, 2)), *rng.uniform(0.0, 1.0, 2), *rng.uniform(-0.05, 0.6, 2)] for _ in range(200)]
    rows += [[1, 0.5], [0, 0.999, 0.999, 0.5, 0.5], [1, 0.5, 0.5, 0.0005, 0.3]]     # malformed, clamped at the border, sub-pixel
    W0, H0, S = 1234, 777, 640
    scale = S / max(H0, W0)
    a, b = transform_yolo_labels(rows, W0, H0, scale, S), O.yolo_labels(rows, W0, H0, scale, S)
    assert len(a) == len(b) and 0 < len(a) < len(rows)
    assert np.array_equal(np.array(a), np.array(b))
    assert all(0.0 <= r[2] <= 1.0 and r[4] >= 1.0 / S for r in a)
    det = collate_boxes([a[:3], [], a[3:5]])
    assert det.shape == (5, 6) and det[:, 0].tolist() == [0, 0, 0, 2, 2] and det.dtype == torch.float32
    assert collate_boxes([[], []]).shape == (0, 6)
