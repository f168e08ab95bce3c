// fp16 instantiations of the implicit-GEMM / direct 3x3 conv (see conv_igemm.inc): v_mfma_f32_16x16x32_f16, fp32 accumulate, fp16 storage with
// saturating stores -- BASELINE configs[4] ("fp16 MFMA conv path").  Two LDS stages only (the depth the heuristic always picks).
#include "conv_igemm.inc"
#include "conv3x3_direct.inc"

int mtbt_conv_dispatch_f16(const ConvP& p, int TC, int TP, int wide, int nbuf, hipStream_t s) {
  (void)nbuf;
  return wide ? dispatch_tile<f16_t, 128, 2>(p, TC, TP, s) : dispatch_tile<f16_t, 64, 2>(p, TC, TP, s);
}

int mtbt_conv3x3_direct_f16(const ConvP& p, int TC, hipStream_t s) {
  if (TC == (128 | 0x1000)) return launch_direct3x3_rr<f16_t, 128>(p, s);
  if (TC == (64 | 0x1000)) return launch_direct3x3_rr<f16_t, 64>(p, s);
  if (TC == 128) return launch_direct3x3<f16_t, 128>(p, s);
  if (TC == 64) return launch_direct3x3<f16_t, 64>(p, s);
  return MTBT_EINVAL;
}
