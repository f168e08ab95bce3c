// C entry points of the depthwise convolution: validation and dispatch to the per-storage-type translation units
// (dwconv_bf16.hip, dwconv_f16.hip, dwconv_f32.hip; kernel in dwconv.inc).
#include "common.h"
#include "dwconv_args.h"

// w: [k*k][C] in the activation dtype (bf16 taps in bf16 mode, like every other conv's weights).
static int dwconv_entry(const void* x, const void* w, const float* bias, const float* ln_w, const float* ln_b, float ln_eps, const float* scale,
                        const float* shift, int act, void* y, void* raw, const void* res, int N, int H, int W, int C, int ksize, int dtype,
                        void* stream) {
  if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C > 768) return MTBT_EINVAL;
  if (ksize != 3 && ksize != 7) return MTBT_EINVAL;
  const bool ln = ln_w != nullptr;
  if (ln && (!ln_b || !bias || res)) return MTBT_EINVAL;
  if (!ln && (!scale || !shift || raw)) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(w) || (raw && !aligned16(raw)) || (res && !aligned16(res))) return MTBT_EALIGN;
  if ((long)(W + 64) * C * 4 >= 0x7fff0000L || (long)H * W * C >= 0x7fff0000L) return MTBT_EINVAL;  // 32-bit offsets in a row / an image
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DwArgs a{x, w, bias, ln_w, ln_b, ln_eps, scale, shift, act, y, raw, res, N, H, W, C, ksize};
  if (dtype == MTBT_BF16) return mtbt_dw_run_bf16(a, s);
  if (dtype == MTBT_F16) return mtbt_dw_run_f16(a, s);
  if (dtype == MTBT_F32) return mtbt_dw_run_f32(a, s);
  return MTBT_EINVAL;
}

extern "C" int mtbt_dwconv_nhwc(const void* x, const void* w, const float* bias, const float* ln_w, const float* ln_b,
                                float ln_eps, const float* scale, const float* shift, int act, void* y, int N, int H,
                                int W, int C, int ksize, int dtype, void* stream) {
  return dwconv_entry(x, w, bias, ln_w, ln_b, ln_eps, scale, shift, act, y, nullptr, nullptr, N, H, W, C, ksize, dtype, stream);
}

// Training variants: `raw` (LayerNorm form only) also receives the LayerNorm INPUT conv + bias (what the LayerNorm backward needs);
// `res` (scale / shift form only) is added after the activation and may alias y (gradient accumulation of the depthwise dgrad).
extern "C" int mtbt_dwconv_nhwc_train(const void* x, const void* w, const float* bias, const float* ln_w, const float* ln_b,
                                      float ln_eps, const float* scale, const float* shift, int act, void* y, void* raw, const void* res,
                                      int N, int H, int W, int C, int ksize, int dtype, void* stream) {
  return dwconv_entry(x, w, bias, ln_w, ln_b, ln_eps, scale, shift, act, y, raw, res, N, H, W, C, ksize, dtype, stream);
}
