"""Phase ablation of the x4 mask-assembly kernel (tools/probes/mask_variants.hip; `--build` compiles it).  One box, one process."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "probes", "libmaskv.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "multitask_bonetumor_yolo_amd", "csrc"), "-Wno-unused-value", os.path.join(ROOT, "tools", "probes", "mask_variants.hip"), "-o", SO])
    sys.exit(0)
import torch
lib = C.CDLL(SO)
lib.mask_variant.restype = C.c_int
lib.mask_variant.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]
dev = "cuda:0"
N, K, A, hp = 16, 100, 8400, 160
protos = torch.randn(N, hp, hp, 32, device=dev)
coeff = torch.randn(N, A, 32, device=dev)
gather = torch.stack([torch.randperm(A, device=dev)[:K] for _ in range(N)]).int().contiguous()
counts = torch.full((N,), K, dtype=torch.int32, device=dev)
masks = torch.empty(N, K, 4 * hp, 4 * hp, dtype=torch.uint8, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "full", 16: "full, scalar taps (round-3 form)", 17: "scalar taps, no stores", 1: "no stores", 2: "no upsample phase", 3: "no upsample, no stores", 4: "no MFMA phase", 8: "no patch loads", 14: "skeleton (barriers, coefficient loads)"}
for rep in range(2):
    for dbg, name in names.items():
        args = (dbg, protos.data_ptr(), coeff.data_ptr(), A * 32, 32, 1, gather.data_ptr(), counts.data_ptr(), N, K, hp, hp, masks.data_ptr(), s)
        for _ in range(3):
            assert lib.mask_variant(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.mask_variant(*args)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{name:40s} {us:7.1f} us   ({masks.numel() / us / 1e6:5.2f} TB/s of mask bytes)", flush=True)
