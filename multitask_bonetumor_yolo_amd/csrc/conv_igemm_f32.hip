// fp32 (exact-fp32 MFMA, parity mode) instantiations of the implicit-GEMM conv, 128-byte K-steps (kernel in conv_igemm.inc).  One translation unit per (storage type,
// K-step width) and one for the direct 3x3 kernels: the build compiles them in parallel.
#include "conv_igemm.inc"

int mtbt_conv_dispatch_f32_wide(const ConvP& p, int TC, int TP, hipStream_t s) { return dispatch_tile<float, 128, 2>(p, TC, TP, s); }
