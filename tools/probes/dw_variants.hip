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

// scale / shift form (no LayerNorm), any width (one-chunk kernel, chunks as grid rows): tile shapes for the small maps of stages 2-3
extern "C" int dw_variant_nl(int id, const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W, int C, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define NL(ID, TH, TW, XB, FULLV) case ID: return launch_dw<bf16_t, 7, false, TH, TW, 1, XB, 0, FULLV>(x, w, nullptr, nullptr, nullptr, 0.f, scale, shift, 0, y, nullptr, nullptr, N, H, W, C, s);
  switch (id) {
    NL(0, 4, 16, 8, false)   // product tile
    NL(1, 4, 8, 8, false)    // half-width tile: 2 waves per workgroup, 36 KiB halo -> 4 workgroups per CU
    NL(2, 4, 8, 8, true)     // ... without bounds branches (W % 8 == 0)
    NL(3, 2, 16, 8, false)   // two output rows
    NL(4, 4, 8, 4, false)    // XB = 4: 4 waves per 4x8 tile
    NL(5, 4, 8, 4, true)
  }
#undef NL
  return -100;
}

// LayerNorm form on the 3-chunk stage (C = 384, 40x40 maps): tile shapes
extern "C" int dw_variant_ln3(int id, const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps, void* y, int N, int H,
                              int W, int C, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define LN3(ID, TH, TW, XB, FULLV) case ID: return launch_dw<bf16_t, 7, true, TH, TW, 3, XB, 0, FULLV>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);
  switch (id) {
    LN3(0, 4, 16, 8, false)  // product
    LN3(1, 4, 8, 8, false)
    LN3(2, 4, 8, 8, true)
    LN3(3, 4, 8, 4, false)
    LN3(4, 4, 8, 4, true)
  }
#undef LN3
  return -100;
}

// LayerNorm form, every chunk count: product tile (4x16, XB = 8; 4x8, XB = 4 for six chunks) against XB = 4 tiles
template <int MAXCH>
static int ln_tiles(int id, const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps, void* y, int N, int H, int W, int C,
                    hipStream_t s) {
#define LNT(ID, TH, TW, XB, FULLV) case ID: return launch_dw<bf16_t, 7, true, TH, TW, MAXCH, XB, 0, FULLV>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);
  switch (id) {
    LNT(0, 4, 16, 8, false)
    LNT(1, 4, 16, 8, true)
    LNT(2, 4, 16, 4, false)
    LNT(3, 4, 16, 4, true)
    LNT(4, 4, 8, 4, false)
    LNT(5, 4, 8, 4, true)
    // one wave per SIMD (up to 512 registers: no spills in the six-chunk kernel)
    case 6: return launch_dw<bf16_t, 7, true, 4, 8, MAXCH, 4, 0, false, 1>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);
    case 7: return launch_dw<bf16_t, 7, true, 4, 16, MAXCH, 8, 0, false, 1>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);
    case 8: return launch_dw<bf16_t, 7, true, 4, 16, MAXCH, 8, 0, true, 1>(x, w, bias, lnw, lnb, eps, nullptr, nullptr, 0, y, nullptr, nullptr, N, H, W, C, s);
  }
#undef LNT
  return -100;
}
extern "C" int dw_variant_lnt(int id, const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps, void* y, int N, int H,
                              int W, int C, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (C + 127) / 128;
  if (nch == 1) return ln_tiles<1>(id, x, w, bias, lnw, lnb, eps, y, N, H, W, C, s);
  if (nch == 2) return ln_tiles<2>(id, x, w, bias, lnw, lnb, eps, y, N, H, W, C, s);
  if (nch == 3) return ln_tiles<3>(id, x, w, bias, lnw, lnb, eps, y, N, H, W, C, s);
  if (nch == 6 && id >= 4 && id != 7 && id != 8) return ln_tiles<6>(id, x, w, bias, lnw, lnb, eps, y, N, H, W, C, s);
  return -100;
}
