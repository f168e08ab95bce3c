// Depthwise convolution kernels, bf16 storage (kernel in dwconv.inc).
#include "dwconv.inc"

int mtbt_dw_run_bf16(const DwArgs& a, hipStream_t s) { return dw_run<bf16_t, 16, true>(a, s); }
