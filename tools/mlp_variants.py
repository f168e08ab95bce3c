"""Ablation timing of the fused ConvNeXt MLP (tools/probes/mlp_variants.hip; `--build` compiles it with hipcc).  One box, one process."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "probes", "libmlpv.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "multitask_bonetumor_yolo_amd", "csrc"), "-Wno-unused-value", os.path.join(ROOT, "tools", "probes", "mlp_variants.hip"), "-o", SO])
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
from multitask_bonetumor_yolo_amd.model import _permute_hidden
lib = C.CDLL(SO)
lib.mlp_variant.restype = C.c_int
lib.mlp_variant.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_long, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
NAMES = {0: "full", 0x2000: "pair, halves in opposite phase order", 0x2001: "pair-ord no GELU", 0x2008: "pair-ord no DMA", 0x1020: "pair no stores", 0x1040: "pair no t loads", 0x1010: "pair no residual", 0x1070: "pair no loads/stores", 0x1001: "pair no GELU", 0x1006: "pair no MFMA", 0x1008: "pair no weight DMA", 0x100e: "pair no MFMA no DMA", 0x100f: "pair skeleton", 0x1000: "pair form (2 waves per 32 pixels, 8 waves)", 0x800: "pipelined chunk (GEMM1 block 1 beside GELU 0)", 0x400: "resident 8 waves + prefetch", 0x500: "resident 12 waves + prefetch", 0x401: "resident8+pref no GELU", 0x406: "resident8+pref no MFMA", 0x407: "resident8+pref no MFMA no GELU", 0x100: "resident 16 waves", 0x200: "resident 12 waves", 0x300: "resident 8 waves", 0x201: "resident12 no GELU", 0x206: "resident12 no MFMA", 0x207: "resident12 no MFMA no GELU", 0x220: "resident12 no stores", 0x250: "resident12 no loads (t, res)", 0x277: "resident12 skeleton: no loads/stores/MFMA/GELU", 0x247: "resident12 stores only", 1: "no GELU", 2: "no GEMM2", 4: "no GEMM1", 8: "no weight DMA", 6: "no MFMA", 7: "no MFMA, no GELU", 15: "skeleton (no DMA / MFMA / GELU)", 32: "no stores"}
ONLY = [int(v, 0) for a in sys.argv if a.startswith("--only=") for v in a[7:].split(",")]
if ONLY:
    NAMES = {k: v for k, v in NAMES.items() if k in ONLY}
for (M, D) in [(25600, 384), (102400, 192), (409600, 96)]:
    t = torch.randn(M, D, device=dev).bfloat16(); res = torch.randn(M, D, device=dev).bfloat16()
    w1 = (torch.randn(4 * D, D, device=dev) / D ** 0.5).bfloat16()
    w2 = _permute_hidden((torch.randn(D, 4 * D, device=dev) / (4 * D) ** 0.5 * 0.1).bfloat16()).contiguous()
    b1, b2 = torch.randn(4 * D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
    y = torch.empty_like(t)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):
        for dbg, name in NAMES.items():
            args = (dbg, t.data_ptr(), res.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), M, D, s)
            if (0x100 <= dbg < 0x800 and D != 96) or (dbg >= 0x800 and D == 96) or (dbg >= 0x2000 and D != 384):
                continue
            for _ in range(3):
                assert lib.mlp_variant(*args) == 0
            if dbg >= 0x100 and rep == 0:      # the resident form against the streaming kernel, same operands
                yr = y.clone(); args0 = (0,) + args[1:]; lib.mlp_variant(*args0); torch.cuda.synchronize()
                print(f"   max |resident - streaming| = {(yr.float() - y.float()).abs().max().item():.3e}", flush=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.mlp_variant(*args)
            e1.record(); torch.cuda.synchronize()
            print(f"D={D} M={M} {name:34s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
