"""Time the depthwise 7x7 + LayerNorm variants of tools/probes/dw_variants.hip (built by this script with hipcc when run with --build)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "probes", "libdwv.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "multitask_bonetumor_yolo_amd", "csrc"), "-Wno-unused-value", os.path.join(ROOT, "tools", "probes", "dw_variants.hip"), "-o", SO])
    sys.exit(0)
import torch
VARIANTS = [int(a) for a in sys.argv[1:] if a.isdigit()]
lib = C.CDLL(SO)
lib.dw_variant.restype = C.c_int
lib.dw_variant.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
for (N, H, Cc) in [(16, 160, 96), (16, 80, 192)]:
    x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    w = (torch.randn(49, Cc, device="cuda") / 7).bfloat16()
    b, lw, lb = torch.randn(Cc, device="cuda") * 0.1, torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
    ref = None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):
        for v in (VARIANTS or range(13)):
            y = torch.empty_like(x)
            rc = lib.dw_variant(v, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, y.data_ptr(), N, H, H, Cc, s)
            if rc == -100:
                continue
            assert rc == 0, (v, rc)
            for _ in range(3):
                lib.dw_variant(v, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, y.data_ptr(), N, H, H, Cc, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.dw_variant(v, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, y.data_ptr(), N, H, H, Cc, s)
            e1.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"C={Cc} H={H} variant {v}: {us:7.1f} us  {2 * x.numel() * 2 / us / 1e6:6.2f} TB/s  equal={torch.equal(y, ref)} maxdiff={(y.float() - ref.float()).abs().max().item():.3g}", flush=True)
