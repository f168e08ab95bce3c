// bf16 instantiations of the implicit-GEMM conv (see conv_igemm.inc).
#include "conv_igemm.inc"

int mtbt_conv_dispatch_bf16(const ConvP& p, int TC, int TP, int wide, int nbuf, hipStream_t s) {
  if (wide) {
    if (nbuf == 2) return dispatch_tile<bf16_t, 128, 2>(p, TC, TP, s);
    if (nbuf == 3) return dispatch_tile<bf16_t, 128, 3>(p, TC, TP, s);
    return dispatch_tile<bf16_t, 128, 4>(p, TC, TP, s);
  }
  if (nbuf == 2) return dispatch_tile<bf16_t, 64, 2>(p, TC, TP, s);
  if (nbuf == 3) return dispatch_tile<bf16_t, 64, 3>(p, TC, TP, s);
  return dispatch_tile<bf16_t, 64, 4>(p, TC, TP, s);
}
