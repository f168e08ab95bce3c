// Arguments of one depthwise-convolution launch, handed from the C entry (dwconv.hip) to the per-storage-type translation units.
#pragma once
#include <hip/hip_runtime.h>

struct DwArgs {
  const void* x; const void* w; const float* bias; const float* ln_w; const float* ln_b; float ln_eps;
  const float* scale; const float* shift; int act; void* y; void* raw; const void* res;
  int N, H, W, C, ksize;
};

int mtbt_dw_run_bf16(const DwArgs& a, hipStream_t s);
int mtbt_dw_run_f16(const DwArgs& a, hipStream_t s);
int mtbt_dw_run_f32(const DwArgs& a, hipStream_t s);
