// fp16 (v_mfma_f32_16x16x32_f16, saturating stores: BASELINE configs[4]) instantiations of the implicit-GEMM conv, 64-byte K-steps (kernel in conv_igemm.inc).
#include "conv_igemm.inc"

int mtbt_conv_dispatch_f16_narrow(const ConvP& p, int TC, int TP, hipStream_t s) { return dispatch_tile<f16_t, 64, 2>(p, TC, TP, s); }
