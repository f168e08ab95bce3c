// BatchNorm2d forward with BATCH statistics (module in train mode), NHWC, fused activation.
// Needed for drop-in parity of `forward(mode="train")`: the reference flips the Detect/Segment heads to train
// mode for that call (main_model.py:358-359), so their BatchNorms normalise with batch statistics
// and update running_mean / running_var (momentum, unbiased variance) even under Lightning validation.
//
// Deterministic two-pass statistics (sum -> mean, then sum (x-mean)^2 -> biased variance): per-workgroup
// partials in a workspace, reduced in a fixed order by a finalize kernel; no atomics.  HBM-bound, small tensors.
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 256;

// PASS 0: partial[b][c] = sum_x ; PASS 1: partial[b][c] = sum (x - mean[c])^2
template <typename T, int PASS>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, long pixels, int C, const float* __restrict__ mean,
                                                         float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [G][C]
  const int CH8 = C >> 3, G = 256 / CH8;
  const int tid = threadIdx.x, cg = tid % CH8, g = tid / CH8;
  const long r0 = (long)blockIdx.x * ROWS_PER_BLOCK;
  const long r1 = r0 + ROWS_PER_BLOCK < pixels ? r0 + ROWS_PER_BLOCK : pixels;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mu[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (g < G) {
    if (PASS == 1) ld8<float>(mean + cg * 8, mu);
    for (long r = r0 + g; r < r1; r += G) {
      float v[8];
      ld8<T>(x + r * C + cg * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[e] - mu[e]; acc[e] += PASS == 0 ? v[e] : d * d; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[g * C + cg * 8 + e] = acc[e];
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int k = 0; k < G; ++k) s += red[k * C + c];
    partial[(long)blockIdx.x * C + c] = s;
  }
}

// PASS 0: mean[c] = sum_b partial / pixels.  PASS 1: var[c] (biased) and the running-statistics update.
template <int PASS>
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int nblocks, int C, long pixels, float* __restrict__ mean,
                                   float* __restrict__ var, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += partial[(long)b * C + c];
  if (PASS == 0) {
    mean[c] = s / (float)pixels;
  } else {
    const float v = s / (float)pixels;
    var[c] = v;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    if (running_var) {
      const float unbiased = pixels > 1 ? s / (float)(pixels - 1) : v;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
}

template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ var,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act, long pixels, int C,
                                int y_ld) {
  const int CH8 = C >> 3;
  const long total = pixels * CH8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CH8);
    const long pix = i / CH8;
    float v[8], mu[8], va[8], ga[8], be[8];
    ld8<T>(x + i * 8, v);
    ld8<float>(mean + cg * 8, mu);
    ld8<float>(var + cg * 8, va);
    ld8<float>(gamma + cg * 8, ga);
    ld8<float>(beta + cg * 8, be);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_apply((v[e] - mu[e]) * rsqrtf(va[e] + eps) * ga[e] + be[e], act);
    st8<T>(y + pix * y_ld + cg * 8, v);
  }
}

__global__ void bn_copy_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ mean, float* __restrict__ var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { mean[c] = rm[c]; var[c] = rv[c]; }
}

}  // namespace

extern "C" int64_t mtbt_bn_train_workspace_bytes(int64_t pixels, int C) {
  if (pixels <= 0 || C <= 0) return 0;
  const int64_t nb = (pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (nb * C + 2 * (int64_t)C) * (int64_t)sizeof(float);
}

// General form (training lowering): x dense [pixels][C]; y rows of y_pixel_stride elements (a channel slice of a C2f concat buffer, or
// dense; may alias x when dense).  use_running != 0: normalise with running_mean / running_var (module in eval mode; nothing is updated).
// stats [2*C] receives the (mean, biased variance) the normalisation used -- what mtbt_bn_backward_nhwc reads.
extern "C" int mtbt_bn_forward_nhwc(const void* x, void* y, int32_t y_pixel_stride, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype, int use_running,
                                    float* stats, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !y || !gamma || !beta || !stats || pixels <= 0 || C <= 0 || C % 8 || C > 2048 || y_pixel_stride < C || y_pixel_stride % 8) return MTBT_EINVAL;
  if (dtype != MTBT_F32 && dtype != MTBT_BF16) return MTBT_EINVAL;
  if (act < MTBT_ACT_NONE || act > MTBT_ACT_GELU_POLY) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta) || !aligned16(stats)) return MTBT_EALIGN;
  if (use_running && (!running_mean || !running_var)) return MTBT_EINVAL;
  const long nb = (pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (nb > 0x7fffffffL) return MTBT_EINVAL;
  const int CH8 = C / 8;
  if (CH8 > 256) return MTBT_EINVAL;
  float* mean = stats;
  float* var = stats + C;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned fb = (unsigned)((C + 255) / 256);
  const long total = pixels * CH8;
  long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (use_running) {
    hipLaunchKernelGGL(bn_copy_stats_kernel, dim3(fb), dim3(256), 0, s, running_mean, running_var, mean, var, C);
  } else {
    if (!workspace || !aligned16(workspace)) return MTBT_EALIGN;
    if (workspace_bytes < nb * C * (int64_t)sizeof(float)) return MTBT_EWORKSPACE;
    float* partial = reinterpret_cast<float*>(workspace);
    const size_t lds = (size_t)(256 / CH8) * C * sizeof(float);
#define BN_STATS(T)                                                                                                            \
    hipLaunchKernelGGL((bn_partial_kernel<T, 0>), dim3((unsigned)nb), dim3(256), lds, s, (const T*)x, (long)pixels, C, nullptr, partial); \
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(fb), dim3(256), 0, s, partial, (int)nb, C, (long)pixels, mean, var, nullptr, nullptr, 0.f); \
    hipLaunchKernelGGL((bn_partial_kernel<T, 1>), dim3((unsigned)nb), dim3(256), lds, s, (const T*)x, (long)pixels, C, mean, partial);   \
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(fb), dim3(256), 0, s, partial, (int)nb, C, (long)pixels, mean, var, running_mean,   \
                       running_var, momentum);
    if (dtype == MTBT_F32) { BN_STATS(float) } else { BN_STATS(bf16_t) }
#undef BN_STATS
  }
  if (dtype == MTBT_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float>), dim3((unsigned)g), dim3(256), 0, s, (const float*)x, (float*)y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride);
  else
    hipLaunchKernelGGL((bn_apply_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

// x, y: dense NHWC [pixels][C] in `dtype` (y may alias x).  gamma/beta/running_*: fp32 [C] (running_* may be NULL).
// workspace: >= mtbt_bn_train_workspace_bytes; on return its last 2*C floats hold the batch mean and biased variance.
extern "C" int mtbt_bn_train_nhwc(const void* x, void* y, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  if (!workspace || pixels <= 0 || C <= 0) return MTBT_EINVAL;
  if (workspace_bytes < mtbt_bn_train_workspace_bytes(pixels, C)) return MTBT_EWORKSPACE;
  const long nb = (pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  float* stats = reinterpret_cast<float*>(workspace) + nb * C;
  return mtbt_bn_forward_nhwc(x, y, C, gamma, beta, running_mean, running_var, momentum, eps, act, pixels, C, dtype, 0, stats, workspace,
                              workspace_bytes, stream);
}
