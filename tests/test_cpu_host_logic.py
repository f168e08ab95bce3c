"""CPU checks of the host logic around the kernels: parameter-name parity with the oracle (state_dicts
interchange), BatchNorm folding and weight packing against plain torch, error behaviour of the module, the
tile/shard helpers, and the world_size-2 gloo path of the timing contract."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, ConvNeXtBiFPNYOLOv0, ConvNeXtBiFPNYOLOv2
from multitask_bonetumor_yolo_amd import model as M
from multitask_bonetumor_yolo_amd.dist_utils import shard_range
from oracle.model import ConvNeXtBiFPNYOLO as OracleModel
from oracle.model import ConvNeXtBiFPNYOLOv2 as OracleModelV2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_names_and_shapes_match_oracle():
    for ours, theirs in ((ConvNeXtBiFPNYOLO, OracleModel), (ConvNeXtBiFPNYOLOv2, OracleModelV2)):
        a = ours(2, 3, pretrained_backbone=False).state_dict()
        b = theirs(2, 3, pretrained_backbone=False).state_dict()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    from oracle.model import ConvNeXtBiFPNYOLOv0 as OracleModelV0
    a, b = ConvNeXtBiFPNYOLOv0(3, 2).state_dict(), OracleModelV0(3, 2).state_dict()
    assert list(a) == list(b) and all(a[k].shape == b[k].shape for k in a)


def test_reference_attributes_present():
    m = ConvNeXtBiFPNYOLO(nc_det=2, nc_img=2, proto_ch=32, pretrained_backbone=False)
    assert m.detect.reg_max == 16 and m.detect.nc == 2 and m.detect.no == 66 and m.segment.nm == 32
    assert torch.equal(m.detect.stride, torch.zeros(3))
    assert (m.nc_det, m.nc_img, m.proto_ch) == (2, 2, 32)
    assert m.neck.bifpn_units[0].w1.shape == (2, 2) and m.neck.bifpn_units[0].w2.shape == (3, 2)


def test_errors_without_gpu():
    m = ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)
    with pytest.raises(RuntimeError, match="pretrained"):
        ConvNeXtBiFPNYOLO(2, 2)  # pretrained_backbone=True would need the network
    with pytest.raises(ValueError, match="Unknown mode"):
        m(torch.rand(1, 3, 64, 64), "eval")
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.eval()(torch.rand(1, 3, 64, 64), "infer")
    with pytest.raises(RuntimeError, match="parameter container"):
        m.backbone.c2f_p3(torch.rand(1, 192, 8, 8))


def test_bn_fold_and_krsc_packing_equal_torch():
    g = torch.Generator().manual_seed(0)
    blk = M.ConvBlock(8, 12, 3)
    with torch.no_grad():
        blk.bn.running_mean.copy_(torch.randn(12, generator=g))
        blk.bn.running_var.copy_(torch.rand(12, generator=g) + 0.5)
        blk.bn.weight.copy_(torch.rand(12, generator=g) + 0.5)
        blk.bn.bias.copy_(torch.randn(12, generator=g))
    x = torch.randn(2, 8, 6, 6, generator=g)
    ref = F.batch_norm(F.conv2d(x, blk.conv.weight, blk.conv.bias, 1, 1), blk.bn.running_mean, blk.bn.running_var, blk.bn.weight,
                       blk.bn.bias, False, 0.0, blk.bn.eps)
    scale, shift = M._bn_fold(blk.bn, blk.conv.bias)
    wp = M._krsc(blk.conv.weight)  # [K, R*S*C], C fastest
    # implicit GEMM on the host: im2col in (r, s, c) order
    cols = F.unfold(x, 3, padding=1).view(2, 8, 9, 36).permute(0, 3, 2, 1).reshape(2, 36, 72)  # [n, pix, (rs)c]
    out = (cols @ wp.t()).permute(0, 2, 1).reshape(2, 12, 6, 6) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    assert torch.allclose(out, ref, atol=1e-5)


def test_conv_transpose_packing_equals_torch():
    g = torch.Generator().manual_seed(1)
    up = nn.ConvTranspose2d(6, 4, 2, 2, 0, bias=True)
    x = torch.randn(1, 6, 3, 5, generator=g)
    ref = up(x)
    wt = up.weight.detach()
    packed = wt.permute(2, 3, 1, 0).reshape(4 * 4, 6)  # rows (dy*2+dx)*Cout + co  (model.py lowering)
    y = torch.einsum("rc,nchw->nrhw", packed, x) + up.bias.detach().repeat(4).view(1, -1, 1, 1)
    out = torch.zeros_like(ref)
    for q in range(4):
        out[:, :, (q >> 1)::2, (q & 1)::2] = y[:, q * 4:(q + 1) * 4]
    assert torch.allclose(out, ref, atol=1e-5)


def test_depthwise_scale_folds_into_pointwise():
    g = torch.Generator().manual_seed(2)
    blk = M.DepthwiseConvBlock(8, 8)
    x = torch.randn(1, 8, 4, 4, generator=g)
    ref = blk.pointwise(blk.depthwise(x))
    dw = blk.depthwise.weight.detach().reshape(1, -1)
    pw = blk.pointwise.weight.detach().reshape(8, -1) * dw
    assert torch.allclose(F.conv2d(x, pw.view(8, 8, 1, 1)), ref, atol=1e-6)


def test_shard_range():
    for n, w in [(16, 1), (16, 8), (17, 8), (3, 8), (0, 2)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - s for s, e in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_gloo_world_size_2_timing_contract(tmp_path):
    """Two CPU ranks run the sharded-timing helper: disjoint shards covering the batch, identical MAX time."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, time, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from multitask_bonetumor_yolo_amd.dist_utils import shard_range, timed_steps, gather_counts, world\n"
        "dist.init_process_group('gloo')\n"
        "r, w = world()\n"
        "s, e = shard_range(33, r, w)\n"
        "t = timed_steps(lambda: time.sleep(0.01 * (r + 1)), 3, lambda: None)\n"
        "cs = gather_counts(torch.arange(s, e))\n"
        "assert torch.equal(torch.cat(cs), torch.arange(33))\n"
        "print(f'RANK{r} {s} {e} {t:.6f}', flush=True)\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    rows = sorted(line.split() for o in outs for line in o.splitlines() if line.startswith("RANK"))
    assert (rows[0][1], rows[0][2], rows[1][1], rows[1][2]) == ("0", "17", "17", "33")
    assert rows[0][3] == rows[1][3] and float(rows[0][3]) >= 0.06  # MAX over ranks: the slow rank's 3 x 20 ms


def test_plan_dependencies_and_lanes(monkeypatch):
    """engine.Plan: data dependencies from the recorded read / write regions (incl. channel slices of a concat buffer and
    pool-recycled buffers) and the lane assignment of independent branches.  No launch is executed (CPU tensors)."""
    from multitask_bonetumor_yolo_amd import _lib as L
    from multitask_bonetumor_yolo_amd.engine import Plan
    monkeypatch.setenv("MTBT_LANES", "3")
    p = Plan(torch.device("cpu"))
    w = torch.zeros(64, 64, dtype=torch.bfloat16)
    a = p.new(1, 8, 8, 64, L.BF16)
    cat = p.new(1, 8, 8, 128, L.BF16)
    p.conv(a, w, cat.slice(0, 64), name="l0")           # 0: a -> cat[0:64]
    p.conv(a, w, cat.slice(64, 64), name="l1")          # 1: a -> cat[64:128]   (independent of 0: disjoint slices)
    b = p.new(1, 8, 8, 64, L.BF16)
    p.conv(cat.slice(0, 64), w, b, name="l2")           # 2: reads slice 0 -> depends on 0 only
    c = p.new(1, 8, 8, 64, L.BF16)
    p.conv(cat.slice(64, 64), w, c, name="l3")          # 3: reads slice 1 -> depends on 1 only
    d = p.new(1, 8, 8, 64, L.BF16)
    p.fuse([b, c], [0.5, 0.5], [L.RES_ID, L.RES_ID], d, name="l4")   # 4: joins 2 and 3
    p.release(b)
    e = p.new(1, 8, 8, 64, L.BF16)                      # recycles b's storage
    assert e.buf is b.buf
    p.conv(a, w, e, name="l5")                          # 5: write-after-read on b (read by 4) and write-after-write (2)
    p.conv(cat, w[:, :64].repeat(1, 2).contiguous(), c, name="l6")   # 6: whole cat (0, 1) ; overwrites c (read by 4, written by 3)
    deps = p.dependencies()
    assert deps == [[], [], [0], [1], [2, 3], [2, 4], [0, 1, 3, 4]]
    s = p.schedule()
    assert s.lane[0] != s.lane[1]                       # the two independent convs run on different lanes
    assert s.lane[2] == s.lane[0] and s.lane[3] == s.lane[1]   # chains stay on their lane
    for i in range(len(deps)):                          # every cross-lane dependency is covered by an event wait
        for j in deps[i]:
            if s.lane[j] != s.lane[i]:
                covered = any(s.records[k] in s.waits[m] for k in range(j, i) if s.lane[k] == s.lane[j] and s.records[k] >= 0
                              for m in range(k + 1, i + 1) if s.lane[m] == s.lane[i])
                assert covered, (i, j)


def test_load_pretrained_heads_from_flat_state_dict(tmp_path, capsys):
    """main_model.py:399-603 semantics on a flat state dict with ultralytics key names: last head block wins, name + shape
    matching, parameters only, the reference's report lines."""
    from multitask_bonetumor_yolo_amd import load_pretrained_heads, strip_lightning_prefix
    from oracle.heads import Detect as ODetect, Segment as OSegment
    torch.manual_seed(0)
    src_det, src_seg = ODetect(nc=2, ch=[256] * 3), OSegment(nc=2, nm=32, npr=256, ch=[256] * 3)
    other = ODetect(nc=5, ch=[256] * 3)                                   # an earlier block with another class count
    sd_det = {f"model.9.{k}": v for k, v in other.state_dict().items()}
    sd_det.update({f"model.22.{k}": v for k, v in src_det.state_dict().items()})
    sd_seg = {f"model.model.23.{k}": v for k, v in src_seg.state_dict().items()}
    path = tmp_path / "seg_state.pt"
    torch.save(sd_seg, path)
    m = ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)
    bn_before = m.detect.cv2[0][0].bn.running_var.clone()
    ret = load_pretrained_heads(m, detect_ckpt_path=sd_det, segment_ckpt_path=str(path))
    assert ret is m                                                       # the reference returns the model; callers rebind it (main_model.py:603, :645)
    rep = m._head_load_report
    n_det, n_seg = len(list(m.detect.named_parameters())), len(list(m.segment.named_parameters()))
    # (the reference walks Segment as cv4, proto, cv2, cv3 -- `dfl.conv.weight` is not part of its Segment report)
    assert rep["detect"] == (n_det, n_det) and rep["segment"] == (n_seg - 1, n_seg - 1)
    assert torch.equal(m.detect.cv3[1][2].weight, src_det.cv3[1][2].weight) and torch.equal(m.segment.proto.upsample.weight, src_seg.proto.upsample.weight)
    assert torch.equal(m.detect.cv2[0][0].bn.running_var, bn_before)     # buffers untouched, like named_parameters() in the reference
    out = capsys.readouterr().out
    assert f"Detect head          : {n_det}/{n_det} tensors copied" in out and "Segment head" in out
    assert f"Head-weight summary  : {n_det + n_seg - 1}/{n_det + n_seg - 1} tensors copied overall." in out
    # class-count mismatch: only the nc-dependent tensors are skipped
    rep2 = load_pretrained_heads(ConvNeXtBiFPNYOLO(3, 2, pretrained_backbone=False), detect_ckpt_path=sd_det)._head_load_report
    assert rep2["detect"][0] == n_det - 6 and "Shape mismatch" in capsys.readouterr().out
    # a pickled object is refused with instructions, never unpickled
    import pickle
    bad = tmp_path / "model_object.pt"
    with open(bad, "wb") as f:
        pickle.dump({"model": object}, f)
    with pytest.raises(RuntimeError, match="not a plain state dict"):
        load_pretrained_heads(m, detect_ckpt_path=str(bad))
    lit = {"net.cls_fc.weight": torch.zeros(2, 256), "seg_proto_projector.weight": torch.zeros(1, 32, 1, 1)}
    assert list(strip_lightning_prefix(lit)) == ["cls_fc.weight"]


def test_compose_upconv_equals_convtranspose_then_conv3x3():
    """model.compose_upconv (host side of mtbt_convt2x2_conv3x3_nhwc): ultralytics Proto `upsample` -> `cv2` (main_model.py:326-328) composed into
    four 2x2-tap convolutions (one per output parity) + nine border-class shift vectors reproduces torch's ConvTranspose2d(2, 2, bias) ->
    Conv2d(3x3, pad 1) * scale + shift, border rows and columns included."""
    import torch.nn.functional as F
    from multitask_bonetumor_yolo_amd.model import compose_upconv
    g = torch.Generator().manual_seed(0)
    Ci, Cm, K, H, W = 5, 7, 6, 4, 6
    wt, bt, w3 = torch.randn(Ci, Cm, 2, 2, generator=g), torch.randn(Cm, generator=g), torch.randn(K, Cm, 3, 3, generator=g)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    x = torch.randn(2, Ci, H, W, generator=g)
    ref = F.conv2d(F.conv_transpose2d(x, wt, bt, stride=2), w3, padding=1) * sc[None, :, None, None] + sh[None, :, None, None]
    w, s9 = compose_upconv(wt, bt, w3, sc, sh)
    assert tuple(w.shape) == (4 * K, 4 * Ci) and tuple(s9.shape) == (9, K)
    w = w.view(2, 2, K, 2, 2, Ci)
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            acc = sum(torch.einsum("kc,nchw->nkhw", w[a, b, :, r, s, :], xp[:, :, a + r:a + r + H, b + s:b + s + W]) for r in range(2) for s in range(2))
            rc = torch.tensor([0 if (a == 0 and i == 0) else (2 if (a == 1 and i == H - 1) else 1) for i in range(H)])
            cc = torch.tensor([0 if (b == 0 and j == 0) else (2 if (b == 1 and j == W - 1) else 1) for j in range(W)])
            cls = rc[:, None] * 3 + cc[None, :]                                     # [H, W]
            out[:, :, a::2, b::2] = acc + s9[cls].permute(2, 0, 1)[None]
    assert (out - ref).abs().max().item() < 2e-4
    assert (s9[4] - s9[0]).abs().max().item() > 1e-3                                # the border classes really differ


def test_plan_option_precedence(monkeypatch):
    """model.plan_options (what GraphedInference(autotune=True) writes) > MTBT_<NAME> > default (model.plan_option)."""
    from multitask_bonetumor_yolo_amd import model as M

    class Dummy:
        pass
    m = Dummy()
    monkeypatch.delenv("MTBT_SEG_GATE", raising=False)
    assert M.plan_option(m, "SEG_GATE") == M.PLAN_OPTION_DEFAULTS["SEG_GATE"] == "0"
    monkeypatch.setenv("MTBT_SEG_GATE", "2")
    assert M.plan_option(m, "SEG_GATE") == "2"
    m.plan_options = {"SEG_GATE": "1"}
    assert M.plan_option(m, "SEG_GATE") == "1"
    assert M.plan_option(m, "LANES") is None              # scheduler knobs default to "not set" (engine.Plan reads its own defaults)
    m.plan_options = {"SEG_GATE": None}
    assert M.plan_option(m, "SEG_GATE") == "2"             # an explicit None falls through to the environment


def test_conv_tile_rules_follow_the_round_structure():
    """`mtbt_conv_kernel_choice` (no launch, no GPU): the tile rules of round 3, measured per launch inside a captured chain
    (profiles/r03_chain_tune.txt) -- 128-byte K-steps wherever the row allows; 300 .. 512 workgroups of the LARGEST tile that yields that many
    (one round on the 512 workgroup slots) before twice as many half tiles; 513 .. 1023 avoided; 64-pixel tiles for the heads' narrow outputs;
    the streaming kernel only for the 64 -> 32 coefficient convs; 3x3 on 16-aligned maps on the direct kernel."""
    import ctypes as C
    from multitask_bonetumor_yolo_amd import _lib as L
    lib = L.load()

    def choice(N, H, W, Cin, K, k=1, stride=1, act=L.ACT_SILU, dtype=L.BF16, out_dtype=None, shift=True):
        a = L.ConvArgs()
        a.x, a.w, a.y = 0x10000, 0x20000, 0x30000              # never dereferenced by the query
        a.shift = 0x40000 if shift else None
        pad = k // 2 if stride == 1 else 0
        a.N, a.H, a.W, a.C, a.K, a.R, a.S, a.stride, a.pad = N, H, W, Cin, K, k, k, stride, pad
        a.Ho, a.Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        a.x_pixel_stride, a.x_batch_stride = Cin, H * W * Cin
        a.y_pixel_stride, a.y_batch_stride = K, a.Ho * a.Wo * K
        a.dtype, a.out_dtype, a.act, a.out_mode = dtype, dtype if out_dtype is None else out_dtype, act, L.OUT_NHWC
        out = (C.c_int32 * 4)()
        assert lib.mtbt_conv_kernel_choice(C.byref(a), out) == 0
        return tuple(out)

    # 1x1 at 40 x 40 (25 600 pixels): K = 256 -> 400 tiles of 128 x 128 = one round; K = 384 would be 600 (a second, mostly empty round) -> 128 x 64
    assert choice(16, 40, 40, 256, 256) == (0, 128, 128, 1)
    assert choice(16, 40, 40, 512, 256) == (0, 128, 128, 1)
    assert choice(16, 40, 40, 384, 384) == (0, 128, 64, 1)
    assert choice(16, 40, 40, 768, 384) == (0, 128, 64, 1)
    # 20 x 20 (6 400 pixels): 64 x 64 tiles (400 workgroups) for K = 256, 128 x 64 (400) for K = 512, 128 x 128 once that gives 300
    assert choice(16, 20, 20, 256, 256) == (0, 64, 64, 1)
    assert choice(16, 20, 20, 768, 512) == (0, 128, 64, 1)
    assert choice(16, 20, 20, 3072, 768, act=L.ACT_NONE) == (0, 128, 128, 1)
    # large grids: 128 x 128, and 128-byte K-steps also for the short rows the round-2 rule gave 64-byte steps
    assert choice(16, 80, 80, 256, 256) == (0, 128, 128, 1)
    assert choice(16, 80, 80, 192, 256) == (0, 128, 128, 1)
    assert choice(16, 160, 160, 96, 192, k=2, stride=2, act=L.ACT_NONE)[3] == 0        # 192-byte rows: not a multiple of 128
    # the heads: the 64 -> 64 box convs on 64-pixel implicit-GEMM tiles, outputs of at most 32 channels (class / coefficient convs, bias only) on
    # the streaming kernel; with a scale vector (not a head conv) the narrow shape stays on the implicit GEMM
    assert choice(16, 80, 80, 64, 64, act=L.ACT_NONE, out_dtype=L.F32) == (0, 64, 64, 1)
    assert choice(16, 40, 40, 256, 2, act=L.ACT_NONE, out_dtype=L.F32)[0] == 2
    assert choice(16, 20, 20, 64, 32, act=L.ACT_NONE, out_dtype=L.F32)[0] == 2
    assert choice(16, 160, 160, 256, 32, act=L.ACT_SILU, out_dtype=L.F32) == (0, 32, 64, 1)      # Proto cv3: activation -> not the streaming kernel
    # 3x3: the direct kernel with the LDS-resident halo on 16-aligned maps (row-reuse form), the implicit GEMM elsewhere
    assert choice(16, 80, 80, 128, 128, k=3)[:2] == (1, 128) and choice(16, 80, 80, 128, 128, k=3)[3] == 0
    assert choice(16, 80, 80, 256, 64, k=3)[:2] == (1, 64)
    assert choice(16, 40, 40, 128, 128, k=3) == (0, 128, 64, 1)
    assert choice(16, 20, 20, 256, 256, k=3) == (0, 64, 64, 1)
    assert choice(16, 80, 80, 192, 384, k=2, stride=2, act=L.ACT_NONE) == (0, 128, 64, 1)   # 600 tiles of 128 x 128 -> half tiles
    # fp32 (the parity mode) follows the same rules with 32-element rows
    assert choice(2, 40, 40, 256, 256, dtype=L.F32)[0] == 0
