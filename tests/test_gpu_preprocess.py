"""HIP input pipeline (mtbt_letterbox_batch) vs oracle/preprocess.py: bit-exact, since everything past the coefficient
set-up is integer arithmetic and the /255 is one correctly rounded fp32 division."""
import numpy as np
import pytest
import torch

from oracle import preprocess as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sample(h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), rng.integers(0, 256, size=(h, w), dtype=np.uint8)


@pytest.mark.parametrize("S,sizes", [
    (640, [(480, 640), (1000, 700), (640, 640), (1280, 960), (37, 91), (3000, 11)]),   # down, up, identity, exact 2x, tiny, 1-pixel-wide result
    (64, [(1, 5), (5, 1), (64, 63), (129, 127), (2, 2)]),
])
def test_letterbox_matches_oracle(S, sizes):
    from multitask_bonetumor_yolo_amd import preprocess as P
    imgs, masks = zip(*[_sample(h, w, i) for i, (h, w) in enumerate(sizes)])
    x, m, scales = P.letterbox_batch([torch.from_numpy(a).to(DEV) for a in imgs], [torch.from_numpy(a).to(DEV) for a in masks], S)
    torch.cuda.synchronize()
    assert x.shape == (len(sizes), 3, S, S) and m.shape == (len(sizes), 1, S, S)
    for i, (a, k) in enumerate(zip(imgs, masks)):
        rx, rm, rs = O.letterbox(a, k, S)
        assert scales[i] == rs
        assert np.array_equal(x[i].cpu().numpy(), rx), f"image {i} {sizes[i]}"
        assert np.array_equal(m[i].cpu().numpy(), rm), f"mask {i} {sizes[i]}"


def test_letterbox_many_images_strided_rows_and_missing_mask():
    from multitask_bonetumor_yolo_amd import preprocess as P
    S = 32
    wide = torch.from_numpy(np.random.default_rng(9).integers(0, 256, size=(40, 90, 3), dtype=np.uint8)).to(DEV)
    imgs = [wide[:, 10 * (i % 5): 10 * (i % 5) + 20 + i] for i in range(35)]     # views with a 270-byte row stride; > 32 images
    masks = [None if i % 2 else (wide[:, :, 0] > 99).to(torch.uint8)[:, 10 * (i % 5): 10 * (i % 5) + 20 + i] * 255 for i in range(35)]
    x, m, _ = P.letterbox_batch(imgs, masks, S)
    torch.cuda.synchronize()
    for i in range(35):
        rx, rm, _ = O.letterbox(imgs[i].cpu().numpy(), None if masks[i] is None else masks[i].cpu().numpy(), S)
        assert np.array_equal(x[i].cpu().numpy(), rx) and np.array_equal(m[i].cpu().numpy(), rm), i


def test_letterbox_rejects_bad_input():
    from multitask_bonetumor_yolo_amd import preprocess as P
    with pytest.raises(RuntimeError):
        P.letterbox_batch([torch.zeros(4, 4, 3, dtype=torch.uint8)], None, 64)          # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        P.letterbox_batch([torch.zeros(4, 4, 3, device=DEV)], None, 64)                 # not uint8
    with pytest.raises(RuntimeError):
        P.letterbox_batch([torch.zeros(4, 4, 3, dtype=torch.uint8, device=DEV)], None, 62)   # S % 4
