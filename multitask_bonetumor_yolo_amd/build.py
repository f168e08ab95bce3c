"""Build libmtbt_hip.so (gfx950) in-tree with hipcc.  `python -m multitask_bonetumor_yolo_amd.build`.

hipcc cross-compiles without a GPU.  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libmtbt_hip.so")
SOURCES = ["conv_igemm.hip", "conv_igemm_bf16.hip", "conv_igemm_bf16_n.hip", "conv_direct_bf16.hip", "conv_igemm_f32.hip", "conv_igemm_f32_n.hip", "conv_direct_f32.hip", "conv_igemm_f16.hip", "conv_igemm_f16_n.hip", "conv_direct_f16.hip", "dwconv.hip", "dwconv_bf16.hip", "dwconv_f16.hip", "dwconv_f32.hip", "pointwise.hip", "postprocess.hip", "mask_mfma.hip", "bn_train.hip", "mlp_fused.hip", "upconv_fused.hip", "node_gemm.hip", "pw_stream.hip", "loss.hip", "preprocess.hip", "metrics.hip", "optim.hip", "wgrad.hip", "pointwise_bwd.hip", "train_ops.hip", "resample_bwd.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
         "-fno-gpu-rdc", "-Wno-unused-value"]
if os.environ.get("MTBT_CONV_ABLATION"):  # development build: MTBT_CONV_DEBUG ablation bits live in the conv K loop
    FLAGS.append("-DMTBT_CONV_ABLATION")


# per-file flags: the post-process must round like the CPU reference (separate multiply / add; hipcc's default
# -ffp-contract=fast fuses them in the backend even across `#pragma clang fp contract(off)`)
EXTRA_FLAGS = {"postprocess.hip": ["-ffp-contract=off"], "loss.hip": ["-ffp-contract=off"], "preprocess.hip": ["-ffp-contract=off"], "optim.hip": ["-ffp-contract=off"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    headers = [os.path.join(INCLUDE, "mtbt_hip.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        if force or _newer(obj, [src] + headers):
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
