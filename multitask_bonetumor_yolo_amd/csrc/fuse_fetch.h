// BiFPN fusion-node input fetch, shared by the stand-alone fusion kernel (pointwise.hip) and the fused node kernel (node_gemm.hip):
// 8 consecutive channels of output pixel (n, y, x) of input i under its resampling mode, fp32, in torch's association order.
//   mode 0 identity; 1 bilinear x2 up (align_corners=False): src = (dst+.5)/2-.5 clamped at 0, i0 = floor, i1 = min(i0+1, last),
//   l0h*(l0w*v00+l1w*v01) + l1h*(l0w*v10+l1w*v11); 2 exact 2x2 mean (= bilinear x0.5); 3 nearest x2 up; 4 2x2 max
#pragma once
#include "common.h"

namespace {

struct FuseP {
  const void* x[3];
  float wgt[3];
  int resample[3];
  int n_in;
  void* y;
  int N, H, W, C;
  int bug;
  const float* wgt_dev;  // non-null: the weights are read from device memory (training: they are parameters)
};

template <typename T>
__device__ __forceinline__ void fuse_fetch(const T* src, int mode, int n, int y, int x, int H, int W, int C, int c0, float (&o)[8]) {
  if (mode == 0) {
    ld8<T>(src + (((long)n * H + y) * W + x) * C + c0, o);
  } else if (mode == 1) {  // bilinear x2 up, source is H/2 x W/2
    const int Hs = H >> 1, Ws = W >> 1;
    float sy = (y + 0.5f) * 0.5f - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = (x + 0.5f) * 0.5f - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
    float a[8], b[8], c[8], d[8];
    const T* base = src + (long)n * Hs * Ws * C + c0;
    ld8<T>(base + ((long)y0 * Ws + x0) * C, a);
    ld8<T>(base + ((long)y0 * Ws + x1) * C, b);
    ld8<T>(base + ((long)y1 * Ws + x0) * C, c);
    ld8<T>(base + ((long)y1 * Ws + x1) * C, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ly0 * (lx0 * a[e] + lx1 * b[e]) + ly1 * (lx0 * c[e] + lx1 * d[e]);
  } else if (mode == 2 || mode == 4) {  // x0.5: source is 2H x 2W; 2x2 mean (bilinear) or max
    const int Ws = W << 1;
    float a[8], b[8], c[8], d[8];
    const T* base = src + (long)n * (H << 1) * Ws * C + c0;
    ld8<T>(base + ((long)(2 * y) * Ws + 2 * x) * C, a);
    ld8<T>(base + ((long)(2 * y) * Ws + 2 * x + 1) * C, b);
    ld8<T>(base + ((long)(2 * y + 1) * Ws + 2 * x) * C, c);
    ld8<T>(base + ((long)(2 * y + 1) * Ws + 2 * x + 1) * C, d);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = (mode == 2) ? 0.5f * (0.5f * a[e] + 0.5f * b[e]) + 0.5f * (0.5f * c[e] + 0.5f * d[e])
                         : fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
  } else {  // nearest x2 up
    const int Hs = H >> 1, Ws = W >> 1;
    ld8<T>(src + (((long)n * Hs + (y >> 1)) * Ws + (x >> 1)) * C + c0, o);
  }
}

}  // namespace
