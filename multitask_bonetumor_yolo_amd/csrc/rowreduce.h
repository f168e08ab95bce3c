// Deterministic two-level reductions over the pixel rows of an NHWC tensor, shared by the backward kernels
// (pointwise_bwd.hip, train_ops.hip): level 1 = per-workgroup partial rows (row groups added in a fixed order through LDS),
// level 2 = one wave per output element over the workgroup partials.
#pragma once
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 256;   // pixels per workgroup in the first level

// A thread owns one 16-byte chunk (8 channels) of a pixel; with fewer than 256 chunks per pixel the workgroup's threads split into
// row groups that walk the workgroup's ROWS_PER_BLOCK pixels in parallel and are then added in row-group order through LDS
// (deterministic).  `fetch(p, ch, v)` returns false when pixel p contributes nothing.
template <typename F>
__device__ __forceinline__ void rows_reduce(long p0, long p1, int chunks, float* __restrict__ dst /* [chunks*8] of this workgroup */, F fetch) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  if (chunks >= 256) {
    for (int ch = tid; ch < chunks; ch += 256) {
      float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (long p = p0; p < p1; ++p) {
        float v[8];
        if (fetch(p, ch, v)) {
#pragma unroll
          for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) dst[ch * 8 + k] = s[k];
    }
    return;
  }
  const int rpp = 256 / chunks, rg = tid / chunks, ch = tid - rg * chunks;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rg < rpp) {
    for (long p = p0 + rg; p < p1; p += rpp) {
      float v[8];
      if (fetch(p, ch, v)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(rg * chunks + ch) * 8 + k] = s[k];
  }
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      for (int g = 0; g < rpp; ++g) t += red[(g * chunks + ch) * 8 + k];
      dst[ch * 8 + k] = t;
    }
  }
}

// second level: one wave per output element; its lanes stride over the workgroup partials and are combined by a butterfly (fixed
// order: deterministic).  A serial loop per element took longer than the first level once there were ~1000 partial rows.
__global__ __launch_bounds__(256) void channel_sum_final(const float* __restrict__ partial, int blocks, int C, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;                     // whole waves leave together
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += partial[(long)b * C + c];
  s = wave_sum(s);
  if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

// the same over a [blocks][pitch] partial matrix, columns [c0, c0 + C)
__global__ __launch_bounds__(256) void channel_sum_final_pitch(const float* __restrict__ partial, int blocks, int pitch, int c0, int C, float* __restrict__ out,
                                                               int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += partial[(long)b * pitch + c0 + c];
  s = wave_sum(s);
  if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

}  // namespace
