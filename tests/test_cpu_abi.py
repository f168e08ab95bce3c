"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU, exports
every symbol include/mtbt_hip.h declares, and its entry points reject bad arguments before any launch."""
import ctypes as C
import os
import re

import pytest

from multitask_bonetumor_yolo_amd import _lib as L
from multitask_bonetumor_yolo_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    B.build()  # no-op when up to date; hipcc cross-compiles without a GPU
    return L.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mtbt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtbt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    decl = declared_symbols()
    assert decl == sorted(L.SYMBOLS), (decl, sorted(L.SYMBOLS))
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/mtbt_hip.h but not exported"


def test_version_and_arch(lib):
    assert lib.mtbt_abi_version() == L.ABI_VERSION == 5
    assert lib.mtbt_target_arch() == b"gfx950"


def test_struct_layouts_match_header(tmp_path):
    """Compile include/mtbt_hip.h with gcc and compare sizeof / offsetof with the ctypes mirrors: a drifted
    mirror would silently corrupt arguments."""
    import subprocess
    probes = {
        "mtbt_conv_args": (L.ConvArgs, ["x", "res", "x_batch_stride", "x_pixel_stride", "N", "K", "Ho", "dtype", "tile_hint"]),
        "mtbt_fuse_args": (L.FuseArgs, ["x", "wgt", "resample", "n_in", "y", "N", "add_weight_bug"]),
        "mtbt_decode_args": (L.DecodeArgs, ["map", "h", "map_pixel_stride", "stride", "n_levels", "xywh", "boxes", "preds_cat", "cat_stride"]),
        "mtbt_loss_args": (L.LossArgs, ["map", "h", "img_size", "gt_xyxy", "iou_thresh", "training", "seg_logits", "seg_bias", "seg_n", "img_gt",
                                        "n_img_classes", "w_img", "workspace", "workspace_bytes", "out"]),
        "mtbt_raw_image": (L.RawImage, ["bgr", "mask", "height", "width", "row_stride", "mask_row_stride"]),
        "mtbt_node_args": (L.NodeArgs, ["fuse", "w", "shift", "y", "y_pixel_stride", "K", "act"]),
        "mtbt_upconv_args": (L.UpconvArgs, ["x", "shift", "x_batch_stride", "x_pixel_stride", "N", "K", "dtype", "act"]),
        "mtbt_mask_args": (L.MaskArgs, ["protos", "coeff_batch_stride", "gather_idx", "bias", "N", "Wout", "logits", "masks"]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mtbt_hip.h"', 'int main(void){']
    for cname, (_, fields) in probes.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append('return 0;}')
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, (ct, fields) in probes.items():
        assert int(out[cname]) == C.sizeof(ct), cname
        for f in fields:
            assert int(out[f"{cname}.{f}"]) == getattr(ct, f).offset, f"{cname}.{f}"


def test_entry_points_reject_bad_arguments_without_launching(lib):
    assert lib.mtbt_conv2d_nhwc(None, None) == -1
    assert lib.mtbt_conv2d_nhwc(C.byref(L.ConvArgs()), None) == -1
    assert lib.mtbt_bifpn_fuse(None, None) == -1
    assert lib.mtbt_decode_boxes(None, None) == -1
    assert lib.mtbt_mask_assemble(None, None) == -1
    assert lib.mtbt_stem_conv4x4_ln(None, None, None, None, None, 1e-6, None, 1, 8, 8, 96, 0, None) == -1
    assert lib.mtbt_dwconv_nhwc(None, None, None, None, None, 0.0, None, None, 0, None, 1, 8, 8, 96, 7, 0, None) == -1
    assert lib.mtbt_layernorm_nhwc(None, None, None, 0.0, None, 1, 96, 0, None) == -1
    assert lib.mtbt_gap_fc(None, None, None, None, 1, 1, 8, 1, 0, None) == -1
    assert lib.mtbt_cast(None, None, 1, 0, 1, None) == -1
    assert lib.mtbt_nms_batched(None, None, None, 1, 1, 0.05, 0.6, 640.0, 10, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.mtbt_nms_workspace_bytes(0, 10) == 0
    assert lib.mtbt_nms_workspace_bytes(2, 8400) >= 2 * (8400 * 20 + 16384 * 8)
    a = L.ConvArgs()
    a.x = a.w = a.y = 16  # non-null dummies: shape validation must fire first
    a.N, a.H, a.W, a.C, a.K, a.R, a.S, a.stride, a.pad, a.Ho, a.Wo, a.dtype, a.out_dtype = 1, 8, 8, 40, 8, 1, 1, 1, 0, 8, 8, L.BF16, L.BF16
    assert lib.mtbt_conv2d_nhwc(C.byref(a), None) == -1  # C % 32 != 0 for bf16


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libmtbt_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()
