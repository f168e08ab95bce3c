// fp32 (exact-fp32 MFMA, parity mode) instantiations of the direct 3x3 convolution kernels (conv3x3_direct.inc).
#include "conv3x3_direct.inc"

int mtbt_conv3x3_direct_f32(const ConvP& p, int TC, hipStream_t s) {
  if (TC == (128 | 0x1000)) return launch_direct3x3_rr<float, 128>(p, s);
  if (TC == (64 | 0x1000)) return launch_direct3x3_rr<float, 64>(p, s);
  if (TC == 128) return launch_direct3x3<float, 128>(p, s);
  if (TC == 64) return launch_direct3x3<float, 64>(p, s);
  return MTBT_EINVAL;
}
