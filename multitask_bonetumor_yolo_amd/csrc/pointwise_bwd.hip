// Pointwise pieces of the backward pass, HBM-bound:
//   mtbt_act_backward   dz = dy * act'(z) for the activations of the forward epilogues (SiLU: ConvBlock / ultralytics Conv,
//                       main_model.py:136; ELU: DepthwiseConvBlock, :96; GELU erf form: timm Mlp) -- z is the PRE-activation,
//                       which a training forward keeps (bf16) next to the activated output
//   mtbt_channel_sum    out[c] (+)= sum over pixels of dz[p][c] (* u[p][c] if a second operand is given): the gradient of a conv
//                       bias / BatchNorm shift, and of a BatchNorm scale / layer-scale (product with the normalised input); two fixed-order levels (per-workgroup partial rows in the workspace, then one pass), deterministic
// fp32 arithmetic with libm exp / erf; 16-byte accesses.
#include <cmath>

#include "common.h"
#include "rowreduce.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void act_backward_kernel(const T* __restrict__ dy, const T* __restrict__ z, T* __restrict__ dz, long n8, int act) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float a[8], b[8];
    ld8(dy + i * 8, a);
    ld8(z + i * 8, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] *= act_grad(b[k], act);
    st8(dz + i * 8, a);
  }
}

// out[p][c] = a[c] * x1[p][c] + b[c] * x2[p][c] + d[c]: the elementwise pass of a batch-statistic BatchNorm backward,
//   dx = (gamma / sigma) * (dy - mean(dy) - xhat * mean(dy * xhat)),  xhat = (u - beta) / gamma
// written on (dy, u) with per-channel coefficients the host forms from the two channel sums.
template <typename T>
__global__ __launch_bounds__(256) void channel_affine2_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ a,
                                                              const float* __restrict__ b, const float* __restrict__ d, T* __restrict__ out,
                                                              long P, int C) {
  const int chunks = C >> 3;
  const long n8 = P * chunks;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    float u[8], v[8];
    ld8(x1 + i * 8, u);
    ld8(x2 + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = a[ch * 8 + k] * u[k] + b[ch * 8 + k] * v[k] + d[ch * 8 + k];
    st8(out + i * 8, u);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void channel_sum_partial(const T* __restrict__ x, const T* __restrict__ x2, long P, int C, int ld, int ld2,
                                                           float* __restrict__ partial) {
  const long p0 = (long)blockIdx.x * ROWS_PER_BLOCK, p1 = min(P, p0 + ROWS_PER_BLOCK);
  rows_reduce(p0, p1, C >> 3, partial + (long)blockIdx.x * C, [&](long p, int ch, float (&v)[8]) {
    ld8(x + p * ld + ch * 8, v);
    if (x2) {
      float u[8];
      ld8(x2 + p * ld2 + ch * 8, u);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= u[k];
    }
    return true;
  });
}

// Depthwise weight gradient: dW[tap][c] = sum_p dy[p][c] * x[p shifted by the tap][c] (zero outside the image).  grid.y = filter ROW:
// a thread keeps the KS taps of that row for its 8 channels, so dy is read once and x KS times per row (8 instead of 14 tensor reads
// per row of a 7x7 filter).  Row groups as in rows_reduce; partial: [workgroup][k*k*C].  A halo-tile version that forms all taps from
// one staged tile is the follow-up.
template <typename T, int KS>
__global__ __launch_bounds__(256) void dw_wgrad_partial(const T* __restrict__ dy, const T* __restrict__ x, int N, int H, int W, int C,
                                                        float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x, chunks = C >> 3, fr = blockIdx.y, r = fr - KS / 2;
  const bool par = chunks < 256;
  const int rpp = par ? 256 / chunks : 1, rg = par ? tid / chunks : 0, ch0 = par ? tid - rg * chunks : tid;
  const long P = (long)N * H * W;
  const long p0 = (long)blockIdx.x * ROWS_PER_BLOCK, p1 = min(P, p0 + ROWS_PER_BLOCK);
  float* dst = partial + (long)blockIdx.x * KS * KS * C + (long)fr * KS * C;
  for (int ch = ch0; ch < chunks; ch += (par ? chunks : 256)) {     // row-group mode: exactly one trip for every thread
    float s[KS][8];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[t][k] = 0.f;
    if (rg < rpp) {
      for (long p = p0 + rg; p < p1; p += rpp) {
        const int n = (int)(p / ((long)H * W)), rem = (int)(p - (long)n * H * W);
        const int y = rem / W + r, xx = rem % W;
        if ((unsigned)y >= (unsigned)H) continue;
        float a[8];
        ld8(dy + p * C + ch * 8, a);
        const T* xr = x + (((long)n * H + y) * W) * C + ch * 8;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          const int ix = xx + t - KS / 2;
          if ((unsigned)ix < (unsigned)W) {
            float b[8];
            ld8(xr + (long)ix * C, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[t][k] += a[k] * b[k];
          }
        }
      }
    }
    if (!par) {
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[(long)t * C + ch * 8 + k] = s[t][k];
    } else {
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        if (rg < rpp) {
#pragma unroll
          for (int k = 0; k < 8; ++k) red[(rg * chunks + ch) * 8 + k] = s[t][k];
        }
        __syncthreads();
        if (rg == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float v = 0.f;
            for (int g = 0; g < rpp; ++g) v += red[(g * chunks + ch) * 8 + k];
            dst[(long)t * C + ch * 8 + k] = v;
          }
        }
        __syncthreads();
      }
    }
  }
}

}  // namespace

extern "C" int mtbt_act_backward(const void* dy, const void* z, void* dz, int64_t n, int act, int dtype, void* stream) {
  if (!dy || !z || !dz || n < 0 || n % 8 || act < MTBT_ACT_NONE || act > MTBT_ACT_GELU_POLY) return MTBT_EINVAL;
  if (!aligned16(dy) || !aligned16(z) || !aligned16(dz)) return MTBT_EALIGN;
  if (n == 0) return MTBT_OK;
  const long n8 = n / 8;
  long blocks = (n8 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MTBT_BF16)
    hipLaunchKernelGGL(act_backward_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)z, (bf16_t*)dz, n8, act);
  else if (dtype == MTBT_F32)
    hipLaunchKernelGGL(act_backward_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dy, (const float*)z, (float*)dz, n8, act);
  else
    return MTBT_EINVAL;
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int64_t mtbt_channel_sum_workspace_bytes(int64_t pixels, int C) {
  if (pixels <= 0 || C <= 0) return 0;
  return ((pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) * (int64_t)C * (int64_t)sizeof(float);
}

extern "C" int mtbt_channel_sum(const void* x, const void* x2, int64_t pixels, int C, int32_t pixel_stride, int32_t pixel_stride2, int dtype,
                                float* out, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !out || !workspace || pixels <= 0 || C <= 0 || C % 8 || pixel_stride < C || pixel_stride % 8) return MTBT_EINVAL;
  if (x2 && (pixel_stride2 < C || pixel_stride2 % 8)) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(workspace) || (x2 && !aligned16(x2))) return MTBT_EALIGN;
  if (workspace_bytes < mtbt_channel_sum_workspace_bytes(pixels, C)) return MTBT_EWORKSPACE;
  const long blocks = (pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  if (dtype == MTBT_BF16)
    hipLaunchKernelGGL(channel_sum_partial<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)x2, (long)pixels, C, pixel_stride, pixel_stride2, partial);
  else if (dtype == MTBT_F32)
    hipLaunchKernelGGL(channel_sum_partial<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, (const float*)x2, (long)pixels, C, pixel_stride, pixel_stride2, partial);
  else
    return MTBT_EINVAL;
  hipLaunchKernelGGL(channel_sum_final, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, partial, (int)blocks, C, out, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_channel_affine2(const void* x1, const void* x2, const float* a, const float* b, const float* d, void* out, int64_t pixels,
                                    int C, int dtype, void* stream) {
  if (!x1 || !x2 || !a || !b || !d || !out || pixels <= 0 || C <= 0 || C % 8) return MTBT_EINVAL;
  if (!aligned16(x1) || !aligned16(x2) || !aligned16(out)) return MTBT_EALIGN;
  const long n8 = pixels * (C / 8);
  long blocks = (n8 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MTBT_BF16)
    hipLaunchKernelGGL(channel_affine2_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x1, (const bf16_t*)x2, a, b, d, (bf16_t*)out,
                       (long)pixels, C);
  else if (dtype == MTBT_F32)
    hipLaunchKernelGGL(channel_affine2_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x1, (const float*)x2, a, b, d, (float*)out,
                       (long)pixels, C);
  else
    return MTBT_EINVAL;
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int64_t mtbt_dwconv_wgrad_workspace_bytes(int N, int H, int W, int C, int ksize) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0) return 0;
  return (((int64_t)N * H * W + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) * (int64_t)ksize * ksize * C * (int64_t)sizeof(float);
}

extern "C" int mtbt_dwconv_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int ksize, int dtype, int accumulate,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !dy || !dw || !workspace || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (ksize != 3 && ksize != 7)) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(workspace)) return MTBT_EALIGN;
  if (workspace_bytes < mtbt_dwconv_wgrad_workspace_bytes(N, H, W, C, ksize)) return MTBT_EWORKSPACE;
  const long blocks = ((long)N * H * W + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  const dim3 grid((unsigned)blocks, (unsigned)ksize);
#define DWW(T, KSV) hipLaunchKernelGGL((dw_wgrad_partial<T, KSV>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, N, H, W, C, partial)
  if (dtype == MTBT_BF16) { if (ksize == 7) DWW(bf16_t, 7); else DWW(bf16_t, 3); }
  else if (dtype == MTBT_F32) { if (ksize == 7) DWW(float, 7); else DWW(float, 3); }
  else return MTBT_EINVAL;
#undef DWW
  const int n = ksize * ksize * C;
  hipLaunchKernelGGL(channel_sum_final, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, partial, (int)blocks, n, dw, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
