// Depthwise convolution kernels, f32 storage (kernel in dwconv.inc).
#include "dwconv.inc"

int mtbt_dw_run_f32(const DwArgs& a, hipStream_t s) { return dw_run<float, 8, false>(a, s); }
