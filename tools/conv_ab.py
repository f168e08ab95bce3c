"""A/B of one convolution between two builds of libmtbt_hip.so inside ONE process on one box (MFMA-bound kernels are clock / power sensitive:
numbers from different boxes or runs are not comparable).  usage: python tools/conv_ab.py libA.so libB.so [rounds]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multitask_bonetumor_yolo_amd import _lib as L  # noqa: E402


class OldConvArgs(C.Structure):
    _fields_ = L.ConvArgs._fields_[:-1] if L.ConvArgs._fields_[-1][0] == "y2" else L.ConvArgs._fields_


def run(path, shapes, rounds):
    lib = C.CDLL(path)
    lib.mtbt_conv2d_nhwc.restype = C.c_int
    lib.mtbt_conv2d_nhwc.argtypes = [C.c_void_p, C.c_void_p]
    out = []
    for (N, H, W, Cc, K, k, act) in shapes:
        x = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
        w = (torch.randn(K, k * k * Cc, device="cuda") / (k * k * Cc) ** 0.5).bfloat16()
        y = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
        sh = torch.zeros(K, device="cuda")
        a = L.ConvArgs()
        a.x, a.w, a.y, a.shift = x.data_ptr(), w.data_ptr(), y.data_ptr(), sh.data_ptr()
        a.x_batch_stride = a.y_batch_stride = H * W * Cc if Cc == K else 0
        a.x_batch_stride, a.y_batch_stride = H * W * Cc, H * W * K
        a.x_pixel_stride, a.y_pixel_stride = Cc, K
        a.N, a.H, a.W, a.C, a.K, a.R, a.S, a.stride, a.pad, a.Ho, a.Wo = N, H, W, Cc, K, k, k, 1, k // 2, H, W
        a.dtype, a.out_dtype, a.act = 1, 1, act
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            assert lib.mtbt_conv2d_nhwc(C.byref(a), s) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rounds):
            lib.mtbt_conv2d_nhwc(C.byref(a), s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / rounds * 1e3
        out.append((us, 2.0 * N * H * W * K * k * k * Cc / (us * 1e-6) / 1e12))
    return out


if __name__ == "__main__":
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    shapes = [(16, 160, 160, 256, 256, 3, 1), (16, 80, 80, 256, 256, 3, 1), (16, 80, 80, 128, 128, 3, 1), (16, 40, 40, 384, 1536, 1, 3), (16, 80, 80, 64, 64, 3, 1)]
    for rep in range(3):
        for p in libs:
            r = run(p, shapes, rounds)
            print(os.path.basename(p), " | ".join(f"{us:7.1f} us {tf:6.0f} TF/s" for us, tf in r), flush=True)
