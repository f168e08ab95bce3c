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

// Pre-reduction of a TALL partial matrix, in place.  The one-wave-per-column second levels read a column with one 4-byte access per row:
// every access is its own 64-byte sector, which 15 other waves fetch again -- at the 3 200 .. 51 200 partial rows the conv epilogues write
// for the 80^2 / 160^2 maps of a batch-32 step that is 16x the matrix in L2 traffic and 200+ dependent trips per lane (round 3: the second
// levels took 12 - 15 us on average, 3.3 ms per step together).  Here a workgroup is 16 row lanes x 16 columns (a wave reads four whole
// sectors per instruction), owns `per` consecutive rows and leaves their column sums in ITS OWN first row -- the only row of the matrix it
// may overwrite without a reader still waiting for it.  The caller then runs the second level over S rows of pitch per * pitch.
__global__ __launch_bounds__(256) void colsum_tile_kernel(float* __restrict__ partial, int rows, int pitch, int c0, int cols, int per) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  if (c < cols) {
    const float* col = partial + c0 + c;
    int r = r0 + rl;
    for (; r + 48 < r1; r += 64) {
      const float v0 = col[(long)r * pitch], v1 = col[(long)(r + 16) * pitch], v2 = col[(long)(r + 32) * pitch], v3 = col[(long)(r + 48) * pitch];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; r < r1; r += 16) s += col[(long)r * pitch];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    partial[(long)r0 * pitch + c0 + c] = t;
  }
}

// host side: reduce `partial` ([rows][pitch], columns [c0, c0 + cols)) in place when it is tall; returns the (rows, pitch) the second level
// is to use.  Deterministic (fixed segment boundaries and orders).
static inline void colsum_prereduce(float* partial, long& rows, int& pitch, int c0, int cols, hipStream_t s) {
  if (rows <= 768 || rows > 0x7fffffffL) return;
  long S = (rows + 255) / 256;
  if (S > 48) S = 48;
  const long per = (rows + S - 1) / S;
  S = (rows + per - 1) / per;
  if (per * pitch > 0x7fffffffL) return;
  hipLaunchKernelGGL(colsum_tile_kernel, dim3((unsigned)((cols + 15) / 16), (unsigned)S), dim3(256), 0, s, partial, (int)rows, pitch, c0, cols, (int)per);
  rows = S;
  pitch = (int)(per * pitch);
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
