"""CPU oracle for the multitask YOLO hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU fp32 restatement of the reference's forward graph
(`/root/reference/src/main_model.py:12-393`, `main_modelv2.py`, `model.py`) and of the
post-process the reference's trainer runs on its outputs
(`/root/reference/src/running_main_v3.py:71-110`, `:251-257`, `:518-552`;
`/root/reference/src/test_model.py:80-85`).

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` -- as the checker / reported baseline, never as the thing measured or shipped.
The product package (`multitask_bonetumor_yolo_amd`) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * reference-owned blocks (ConvBlock, Bottleneck, C2f, DepthwiseConvBlock, BiFPNUnit, BiFPN,
    batch_bbox_iou, dist2bbox) are pinned bit-for-bit against the real reference classes by the
    fixtures in `tests/golden/ref_blocks.pt` (made by `tests/golden/make_ref_fixtures.py`).
  * third-party blocks (timm ConvNeXt-T, ultralytics Conv/DWConv/Detect/Segment/Proto/DFL,
    torchvision.ops.nms) are absent from this image and from /root/reference, their versions are
    unpinned by the reference (`src/requirements.txt` holds names only) and the reference holds no
    golden vectors for them: those parts are restated from the published algorithms and are
    **parity unpinned**.
"""
