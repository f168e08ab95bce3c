"""Tile shapes of the one-chunk depthwise 7x7 kernel (scale / shift form, chunks as grid rows) on the small maps (tools/probes/dw_variants.hip,
built by `python tools/dw_variants.py --build`).  One box, one process."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", "libdwv.so"))
lib.dw_variant_nl.restype = C.c_int
lib.dw_variant_nl.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]
for (N, H, Cc) in [(16, 40, 384), (16, 20, 768), (16, 80, 192), (32, 40, 384)]:
    x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    w = (torch.randn(49, Cc, device="cuda") / 7).bfloat16()
    sc, sh = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = None
    for rep in range(2):
        for v in range(6):
            y = torch.empty_like(x)
            args = (v, x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), N, H, H, Cc, s)
            rc = lib.dw_variant_nl(*args)
            if rc == -100 or (v in (2, 5) and H % 8):
                continue
            assert rc == 0, (v, rc)
            for _ in range(3): lib.dw_variant_nl(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): lib.dw_variant_nl(*args)
            e1.record(); torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            print(f"N={N} C={Cc} H={H} variant {v}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us  equal={torch.equal(y, ref)}", flush=True)

lib.dw_variant_ln3.restype = C.c_int
lib.dw_variant_ln3.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
for (N, H, Cc) in [(16, 40, 384), (32, 40, 384)]:
    x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    w = (torch.randn(49, Cc, device="cuda") / 7).bfloat16()
    b, lw, lb = torch.randn(Cc, device="cuda") * 0.1, torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = None
    for rep in range(2):
        for v in range(5):
            y = torch.empty_like(x)
            args = (v, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, y.data_ptr(), N, H, H, Cc, s)
            assert lib.dw_variant_ln3(*args) == 0
            for _ in range(3): lib.dw_variant_ln3(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): lib.dw_variant_ln3(*args)
            e1.record(); torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            print(f"LN form N={N} C={Cc} H={H} variant {v}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us  equal={torch.equal(y, ref)}", flush=True)

lib.dw_variant_lnt.restype = C.c_int
lib.dw_variant_lnt.argtypes = lib.dw_variant_ln3.argtypes
for (N, H, Cc) in [(16, 160, 96), (16, 80, 192), (16, 40, 384), (16, 20, 768), (32, 160, 96), (32, 80, 192), (32, 20, 768), (64, 40, 768)]:
    x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    w = (torch.randn(49, Cc, device="cuda") / 7).bfloat16()
    b, lw, lb = torch.randn(Cc, device="cuda") * 0.1, torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = None
    for rep in range(2):
        for v in range(9):
            if (v in (1, 3, 8) and H % 16) or (v == 5 and H % 8):
                continue
            y = torch.empty_like(x)
            args = (v, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, y.data_ptr(), N, H, H, Cc, s)
            if lib.dw_variant_lnt(*args) != 0:
                continue
            for _ in range(3): lib.dw_variant_lnt(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): lib.dw_variant_lnt(*args)
            e1.record(); torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            print(f"LN tiles N={N} C={Cc} H={H} variant {v}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us  max diff {(y.float() - ref.float()).abs().max().item():.4f}", flush=True)
