// Depthwise convolution kernels, f16 storage (kernel in dwconv.inc).
#include "dwconv.inc"

int mtbt_dw_run_f16(const DwArgs& a, hipStream_t s) { return dw_run<f16_t, 16, true>(a, s); }
