// fp16 (v_mfma_f32_16x16x32_f16, saturating stores: BASELINE configs[4]) instantiations of the implicit-GEMM conv, 128-byte K-steps (kernel in conv_igemm.inc).  One translation unit per (storage type,
// K-step width) and one for the direct 3x3 kernels: the build compiles them in parallel.
#include "conv_igemm.inc"

int mtbt_conv_dispatch_f16_wide(const ConvP& p, int TC, int TP, hipStream_t s) { return dispatch_tile<f16_t, 128, 2>(p, TC, TP, s); }
