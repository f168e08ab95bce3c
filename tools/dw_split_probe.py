"""ConvNeXt conv_dw + norm: the fused multi-chunk kernel against (one-chunk depthwise kernel over grid rows) + (LayerNorm kernel), stages 0-3 of
the 640x640 batch-16 forward.  One box, one process."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import _lib as L
lib = L.load()
dev = "cuda:0"
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t_us(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (N, H, Cc) in [(16, 160, 96), (16, 80, 192), (16, 40, 384), (16, 20, 768)]:
    x = torch.randn(N, H, H, Cc, device=dev).bfloat16()
    w = (torch.randn(49, Cc, device=dev) / 7).bfloat16()
    b, lw, lb = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    one, y1, y2, tmp = torch.ones(Cc, device=dev), torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    fused = lambda: lib.mtbt_dwconv_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), C.c_float(1e-6), None, None, 0, y1.data_ptr(), N, H, H, Cc, 7, 1, S())
    dw = lambda: lib.mtbt_dwconv_nhwc(x.data_ptr(), w.data_ptr(), None, None, None, C.c_float(0.0), one.data_ptr(), b.data_ptr(), 0, tmp.data_ptr(), N, H, H, Cc, 7, 1, S())
    ln = lambda: lib.mtbt_layernorm_nhwc(tmp.data_ptr(), lw.data_ptr(), lb.data_ptr(), C.c_float(1e-6), y2.data_ptr(), N * H * H, Cc, 1, S())
    assert fused() == 0 and dw() == 0 and ln() == 0
    torch.cuda.synchronize()
    d = (y1.float() - y2.float()).abs()
    print(f"C={Cc:3d} H={H:3d}: fused {t_us(fused):6.1f} us | split dw {t_us(dw):6.1f} + LN {t_us(ln):5.1f} us | max diff {d.max().item():.3g}, mean {d.mean().item():.2g}", flush=True)
