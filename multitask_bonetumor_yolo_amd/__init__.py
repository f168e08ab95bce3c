"""MI355X-native multitask YOLO hot path: drop-in ConvNeXtBiFPNYOLO over hand-written HIP kernels.

    from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO          # == reference main_model.ConvNeXtBiFPNYOLO
    from multitask_bonetumor_yolo_amd import postprocess                 # decode / NMS / masks on the GPU
    from multitask_bonetumor_yolo_amd import preprocess                  # letterbox / BGR->RGB / /255 of a batch on the GPU
    from multitask_bonetumor_yolo_amd import multitask_loss              # == MultiTaskLitModel._multitask_loss (value)

The HIP library (csrc/libmtbt_hip.so, C ABI in include/mtbt_hip.h) is built by
`python -m multitask_bonetumor_yolo_amd.build`; nothing here falls back to the CPU.
"""
from . import postprocess, preprocess  # noqa: F401
from .loss import multitask_loss  # noqa: F401
from .checkpoints import load_pretrained_heads, strip_lightning_prefix  # noqa: F401
from .graphed import GraphedInference  # noqa: F401
from .metrics import MeanAveragePrecision, SegmentationMetrics  # noqa: F401
from .model import ConvNeXtBiFPNYOLO, ConvNeXtBiFPNYOLOv0, ConvNeXtBiFPNYOLOv2, calibrate_synthetic_heads_, init_synthetic_, synthetic_images  # noqa: F401
