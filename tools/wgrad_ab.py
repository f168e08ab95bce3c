"""A/B of the weight-gradient entry point between builds of the library in one process (batch-32 shapes of the training step)."""
import ctypes as C, os, sys, torch
libs = sys.argv[1:]
shapes = [("fc1.s0 96->384 @160", 32, 160, 96, 384, 1), ("fc2.s0 384->96 @160", 32, 160, 384, 96, 1), ("fc1.s1 192->768 @80", 32, 80, 192, 768, 1),
          ("fc1.s2 384->1536 @40", 32, 40, 384, 1536, 1), ("fc2.s3 3072->768 @20", 32, 20, 3072, 768, 1), ("c2f cv2 512->256 @80", 32, 80, 512, 256, 1),
          ("head 256->64 3x3 @80", 32, 80, 256, 64, 3), ("c2f 192->192 3x3 @40", 32, 40, 192, 192, 3), ("c2f 256->256 3x3 @20", 32, 20, 256, 256, 3),
          ("out 64->64 @80", 32, 80, 64, 64, 1), ("proto.cv3 256->32 @160", 32, 160, 256, 32, 1)]
for rep in range(2):
    for path in libs:
        lib = C.CDLL(path)
        lib.mtbt_conv_wgrad_workspace_bytes.restype = C.c_int64
        lib.mtbt_conv_wgrad_workspace_bytes.argtypes = [C.c_int] * 7
        lib.mtbt_conv_wgrad.restype = C.c_int
        lib.mtbt_conv_wgrad.argtypes = [C.c_void_p] * 3 + [C.c_int] * 9 + [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        row = []
        for name, N, H, Cc, K, k in shapes:
            x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
            dy = torch.randn(N, H, H, K, device="cuda").bfloat16()
            out = torch.empty(K, k * k * Cc, device="cuda")
            nb = lib.mtbt_conv_wgrad_workspace_bytes(N, H, H, Cc, K, k, k)
            ws = torch.empty(nb // 4, device="cuda")
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            call = lambda: lib.mtbt_conv_wgrad(x.data_ptr(), dy.data_ptr(), out.data_ptr(), N, H, H, Cc, K, k, k, k // 2, 1, H * H * Cc, Cc, H * H * K, K, 1, 0, ws.data_ptr(), nb, s)
            for _ in range(2):
                assert call() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                call()
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 8 * 1e3)
        print(os.path.basename(path)[:22].ljust(22), " ".join(f"{v:7.1f}" for v in row), f" sum {sum(row):8.1f} us", flush=True)
print("columns:", " | ".join(n for n, *_ in shapes))
