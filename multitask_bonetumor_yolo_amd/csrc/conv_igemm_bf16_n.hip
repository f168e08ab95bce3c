// bf16 instantiations of the implicit-GEMM conv, 64-byte K-steps (kernel in conv_igemm.inc).
#include "conv_igemm.inc"

int mtbt_conv_dispatch_bf16_narrow(const ConvP& p, int TC, int TP, hipStream_t s) { return dispatch_tile<bf16_t, 64, 2>(p, TC, TP, s); }
