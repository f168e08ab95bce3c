// fp32 (exact-fp32 MFMA, parity mode) instantiations of the implicit-GEMM conv (see conv_igemm.inc).
#include "conv_igemm.inc"

int mtbt_conv_dispatch_f32(const ConvP& p, int TC, int TP, int wide, int nbuf, hipStream_t s) {
  if (wide) return nbuf >= 3 ? dispatch_tile<float, 128, 3>(p, TC, TP, s) : dispatch_tile<float, 128, 2>(p, TC, TP, s);
  return nbuf >= 3 ? dispatch_tile<float, 64, 3>(p, TC, TP, s) : dispatch_tile<float, 64, 2>(p, TC, TP, s);
}
