"""Is the XB = 4 LayerNorm-form tile (variants 3 / 4 of dw_variant_ln3) the same function as the product tile?  Against an fp32 torch reference."""
import ctypes as C, os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", "libdwv.so"))
lib.dw_variant_ln3.restype = C.c_int
lib.dw_variant_ln3.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
torch.manual_seed(0)
N, H, Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 40, 384
x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
w = (torch.randn(49, Cc, device="cuda") / 7).bfloat16()
b, lw, lb = torch.randn(Cc, device="cuda") * 0.1, torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().t().reshape(Cc, 1, 7, 7), b, padding=3, groups=Cc).permute(0, 2, 3, 1)
ref = F.layer_norm(ref, (Cc,), lw, lb, 1e-6)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = {}
for v in range(5):
    y = torch.empty_like(x)
    assert lib.dw_variant_ln3(v, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, y.data_ptr(), N, H, H, Cc, s) == 0
    torch.cuda.synchronize()
    outs[v] = y.float()
    d = (y.float() - ref).abs()
    print(f"variant {v}: max |y - fp32 ref| {d.max().item():.4f} mean {d.mean().item():.5f}; vs variant 0: max {(outs[v] - outs[0]).abs().max().item():.4f}, "
          f"differing elements {(outs[v] != outs[0]).float().mean().item():.5f}")
    if v and not torch.equal(outs[v], outs[0]):
        idx = (outs[v] != outs[0]).nonzero()
        print("   first differing (n, y, x, c):", idx[:6].tolist(), " images:", sorted(set(idx[:, 0].tolist()))[:20], " rows:", sorted(set(idx[:, 1].tolist()))[:12],
              " cols:", sorted(set(idx[:, 2].tolist()))[:12], " channel range:", idx[:, 3].min().item(), idx[:, 3].max().item())
