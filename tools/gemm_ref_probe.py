"""How fast does the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) run the plain 1x1-conv shapes of the model?  A yardstick for
conv_igemm on the same shapes (printed next to it).  Run on the GPU box."""
import sys, time, torch
sys.path.insert(0, ".")
dev = torch.device("cuda:0")

def t_us(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

shapes = [("stage2 fc1", 16 * 40 * 40, 384, 1536), ("stage2 fc2", 16 * 40 * 40, 1536, 384), ("stage3 fc1", 16 * 20 * 20, 768, 3072),
          ("stage3 fc2", 16 * 20 * 20, 3072, 768), ("stage1 fc1", 16 * 80 * 80, 192, 768), ("stage0 fc1", 16 * 160 * 160, 96, 384),
          ("bifpn 1x1 p3", 16 * 80 * 80, 128, 128), ("bifpn 1x1 p4", 16 * 40 * 40, 128, 128), ("head 1x1 p3", 16 * 80 * 80, 128, 128)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us_mm = t_us(lambda: torch.matmul(a, w.t(), out=out))
    us_lin = t_us(lambda: torch.nn.functional.linear(a, w, bias))
    fl = 2.0 * M * N * K
    print(f"{name:14s} M={M:6d} K={K:5d} N={N:5d}  matmul {us_mm:7.1f} us {fl/us_mm/1e6:7.1f} TF/s   linear+bias {us_lin:7.1f} us {fl/us_lin/1e6:7.1f} TF/s", flush=True)
