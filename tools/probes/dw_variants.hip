// Development probe: tile / occupancy / tap-placement variants of the depthwise 7x7 + LayerNorm kernel behind one entry point, timed by
// tools/dw_variants.py on one box in one process.  Includes the product source so that the variants are the same code.
#include "../../multitask_bonetumor_yolo_amd/csrc/dwconv.inc"

#define V(ID, TH, TW, MAXCH, XB, OCC, REGT) \
  case ID: return launch_dw<bf16_t, 7, true, TH, TW, MAXCH, XB, 0, false, OCC, REGT>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);

extern "C" int dw_variant(int id, const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps, void* y, int N, int H,
                          int W, int C, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (C + 127) / 128;
  if (nch == 1) {
    switch (id) {
      V(0, 4, 16, 1, 8, 2, true)     // product default
      V(1, 4, 16, 1, 8, 3, false)    // taps in LDS, 3 waves / SIMD
      V(2, 4, 16, 1, 8, 4, false)
      V(3, 4, 8, 1, 8, 2, true)      // half-width tile (2 waves per workgroup), taps in registers
      V(4, 4, 8, 1, 8, 4, false)
      V(5, 2, 16, 1, 8, 4, false)    // two output rows per workgroup
      V(6, 2, 16, 1, 8, 2, true)
      V(7, 4, 8, 1, 4, 4, false)     // XB = 4: 4 waves per 4x8 tile, fewer accumulators
      V(8, 4, 16, 1, 4, 3, false)
      // no LayerNorm (y = conv * lnw + lnb): activation switch at run time / compiled out / compiled out + no bounds branches; 12 = LayerNorm, no bounds branches
      case 9: return launch_dw<bf16_t, 7, false, 4, 16, 1, 8, -1, false>(x, w, nullptr, nullptr, nullptr, 0.f, lnw, lnb, 0, y, nullptr, nullptr, N, H, W, C, s);
      case 10: return launch_dw<bf16_t, 7, false, 4, 16, 1, 8, 0, false>(x, w, nullptr, nullptr, nullptr, 0.f, lnw, lnb, 0, y, nullptr, nullptr, N, H, W, C, s);
      case 11: return launch_dw<bf16_t, 7, false, 4, 16, 1, 8, 0, true>(x, w, nullptr, nullptr, nullptr, 0.f, lnw, lnb, 0, y, nullptr, nullptr, N, H, W, C, s);
      case 12: return launch_dw<bf16_t, 7, true, 4, 16, 1, 8, 0, true>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);
    }
  } else if (nch == 2) {
    switch (id) {
      V(0, 4, 16, 2, 8, 2, false)
      V(1, 4, 16, 2, 8, 3, false)
      V(3, 4, 8, 2, 8, 2, false)
      V(4, 4, 8, 2, 8, 4, false)
      V(5, 2, 16, 2, 8, 4, false)
      V(7, 4, 8, 2, 4, 4, false)
      V(8, 4, 16, 2, 4, 3, false)
      case 9: return launch_dw<bf16_t, 7, false, 4, 16, 2, 8, -1, false>(x, w, nullptr, nullptr, nullptr, 0.f, lnw, lnb, 0, y, nullptr, nullptr, N, H, W, C, s);
      case 10: return launch_dw<bf16_t, 7, false, 4, 16, 2, 8, 0, false>(x, w, nullptr, nullptr, nullptr, 0.f, lnw, lnb, 0, y, nullptr, nullptr, N, H, W, C, s);
      case 11: return launch_dw<bf16_t, 7, false, 4, 16, 2, 8, 0, true>(x, w, nullptr, nullptr, nullptr, 0.f, lnw, lnb, 0, y, nullptr, nullptr, N, H, W, C, s);
      case 12: return launch_dw<bf16_t, 7, true, 4, 16, 2, 8, 0, true>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);
    }
  }
  return -100;
}
