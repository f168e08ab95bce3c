// BatchNorm2d forward with BATCH statistics (module in train mode), NHWC, fused activation.
// Needed for drop-in parity of `forward(mode="train")`: the reference flips the Detect/Segment heads to train
// mode for that call (main_model.py:358-359), so their BatchNorms normalise with batch statistics
// and update running_mean / running_var (momentum, unbiased variance) even under Lightning validation.
//
// Deterministic statistics: per-workgroup (mean, M2) partials in a workspace, combined exactly in a fixed order; no atomics.
#include "common.h"
#include "rowreduce.h"

namespace {

// ---- single-pass statistics: every workgroup reduces its ROWS_PER_BLOCK rows to a per-channel (mean_b, M2_b)
// with sums SHIFTED by the block's first row (no cancellation for |mean| >> sigma); the final kernel combines the blocks exactly
// (Chan et al.): mean = sum n_b mean_b / N, M2 = sum (M2_b + n_b (mean_b - mean)^2), one WAVE per channel striding over the blocks.
// One read of x for the statistics instead of two, and no serial loop over thousands of partial rows. ----
template <typename T>
__global__ __launch_bounds__(256) void bn_block_stats_kernel(const T* __restrict__ x, long pixels, int C, float* __restrict__ partial /* [blocks][2C] */) {
  __shared__ float red[256 * 16];
  const int CH8 = C >> 3, G = 256 / CH8;
  const int tid = threadIdx.x, cg = tid % CH8, g = tid / CH8;
  const long r0 = (long)blockIdx.x * ROWS_PER_BLOCK;
  const long r1 = r0 + ROWS_PER_BLOCK < pixels ? r0 + ROWS_PER_BLOCK : pixels;
  const float nb = (float)(r1 - r0);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sh[8];
  if (g < G) {
    ld8<T>(x + r0 * C + cg * 8, sh);
    for (long r = r0 + g; r < r1; r += G) {
      float v[8];
      ld8<T>(x + r * C + cg * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[e] - sh[e]; s[e] += d; q[e] += d * d; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[(g * CH8 + cg) * 16 + e] = s[e]; red[(g * CH8 + cg) * 16 + 8 + e] = q[e]; }
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float ts = 0.f, tq = 0.f;
      for (int k = 0; k < G; ++k) { ts += red[(k * CH8 + cg) * 16 + e]; tq += red[(k * CH8 + cg) * 16 + 8 + e]; }
      const float md = ts / nb;                                    // mean of the shifted values
      partial[(long)blockIdx.x * 2 * C + cg * 8 + e] = sh[e] + md;            // block mean
      partial[(long)blockIdx.x * 2 * C + C + cg * 8 + e] = tq - ts * md;      // block M2 = sum d^2 - (sum d)^2 / n
    }
  }
}

__global__ __launch_bounds__(256) void bn_combine_stats_kernel(const float* __restrict__ partial, int nblocks, int C, long pixels, float* __restrict__ mean,
                                                               float* __restrict__ var, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float momentum) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;                                              // whole waves leave together
  const long last = pixels - (long)(nblocks - 1) * ROWS_PER_BLOCK;   // rows of the last block
  float sm = 0.f;
  for (int b = lane; b < nblocks; b += 64) sm += (b == nblocks - 1 ? (float)last : (float)ROWS_PER_BLOCK) * partial[(long)b * 2 * C + c];
  const float mu = wave_sum(sm) / (float)pixels;
  float m2 = 0.f;
  for (int b = lane; b < nblocks; b += 64) {
    const float d = partial[(long)b * 2 * C + c] - mu;
    m2 += partial[(long)b * 2 * C + C + c] + (b == nblocks - 1 ? (float)last : (float)ROWS_PER_BLOCK) * d * d;
  }
  m2 = wave_sum(m2);
  if (lane == 0) {
    mean[c] = mu;
    var[c] = m2 / (float)pixels;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (pixels > 1 ? m2 / (float)(pixels - 1) : m2 / (float)pixels);
  }
}

// y = act((x - mean) * rstd * gamma + beta).  FIXED (256 % (C / 8) == 0): a thread's 8-channel piece is the same in every trip of the
// grid-stride loop, so the four parameter vectors and the rsqrt are taken ONCE per thread and four rows are in flight per trip; ACT = the
// activation as a compile-time constant or -1 for the run-time `act` (round 3: the general loop re-read 4 x 32 B of parameters and held one
// 16-byte load per trip -- 2.5 TB/s over the step's BatchNorm family).
template <typename T, int ACT, bool FIXED>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ var,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act, long pixels, int C,
                                int y_ld) {
  const int CH8 = C >> 3;
  const long total = pixels * CH8;
  if constexpr (FIXED) {
    const int cg = threadIdx.x % CH8;
    const int sh = 31 - __builtin_clz(CH8);              // CH8 is a power of two here: pixel = piece >> sh (a 64-bit division per piece otherwise)
    float mu[8], rs[8], ga[8], be[8];
    ld8<float>(mean + cg * 8, mu);
    ld8<float>(var + cg * 8, rs);
    ld8<float>(gamma + cg * 8, ga);
    ld8<float>(beta + cg * 8, be);
#pragma unroll
    for (int e = 0; e < 8; ++e) rs[e] = rsqrtf(rs[e] + eps);
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < total; i += 4 * stride) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld8<T>(x + (i + u * stride) * 8, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = act_apply((v[u][e] - mu[e]) * rs[e] * ga[e] + be[e], ACT < 0 ? act : ACT);
        st8<T>(y + ((i + u * stride) >> sh) * y_ld + cg * 8, v[u]);
      }
    }
    for (; i < total; i += stride) {
      float v[8];
      ld8<T>(x + i * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_apply((v[e] - mu[e]) * rs[e] * ga[e] + be[e], ACT < 0 ? act : ACT);
      st8<T>(y + (i >> sh) * y_ld + cg * 8, v);
    }
    return;
  }
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

template <typename T>
void launch_bn_apply(const void* x, void* y, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act, long pixels,
                     int C, int y_ld, hipStream_t s) {
  const int CH8 = C >> 3;
  const bool fixed = 256 % CH8 == 0;
  const long total = pixels * CH8;
  long g = (total + (fixed ? 1023 : 255)) / (fixed ? 1024 : 256);      // FIXED: four pieces per thread and trip, at most 8 workgroups per CU
  const long cap = fixed ? 2048 : 8192;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
#define BA(A, F) hipLaunchKernelGGL((bn_apply_kernel<T, A, F>), dim3((unsigned)g), dim3(256), 0, s, (const T*)x, (T*)y, mean, var, gamma, beta, eps, act, pixels, C, y_ld)
  if (!fixed) BA(-1, false);
  else if (act == MTBT_ACT_SILU) BA(MTBT_ACT_SILU, true);
  else if (act == MTBT_ACT_ELU) BA(MTBT_ACT_ELU, true);
  else if (act == MTBT_ACT_NONE) BA(MTBT_ACT_NONE, true);
  else BA(-1, true);
#undef BA
}

// batch statistics straight from the conv epilogue's PARTIAL rows ([rows][pitch]: sum (x - s) in [0, C), sum (x - s)^2 in [C, 2C)): one wave per
// channel, lanes stride over the rows, butterfly (fixed order) -- the second level of the column sums and the statistics in one launch
__global__ __launch_bounds__(256) void bn_stats_from_partials_kernel(const float* __restrict__ partial, int rows, int pitch, const float* __restrict__ shift,
                                                                     long pixels, int C, float* __restrict__ mean, float* __restrict__ var,
                                                                     float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;                                   // whole waves leave together
  float s1 = 0.f, s2 = 0.f;
  for (int r = lane; r < rows; r += 64) { s1 += partial[(long)r * pitch + c]; s2 += partial[(long)r * pitch + C + c]; }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) {
    const float n = (float)pixels;
    const float sh = shift ? shift[c] : 0.f;            // (may BE running_mean[c]: read before it is updated below)
    const float m1 = s1 / n;
    const float mu = sh + m1;
    const float m2 = fmaxf(s2 - n * m1 * m1, 0.f);
    mean[c] = mu;
    var[c] = m2 / n;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (pixels > 1 ? m2 / (float)(pixels - 1) : m2 / n);
  }
}

// batch statistics from the column sums the producing conv's epilogue accumulated (mtbt_conv_args.colsum): sum (x - s), sum (x - s)^2
__global__ void bn_stats_from_sums_kernel(const float* __restrict__ sums, const float* __restrict__ shift, long pixels, int C, float* __restrict__ mean,
                                          float* __restrict__ var, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float n = (float)pixels;
  const float sh = shift ? shift[c] : 0.f;          // (may BE running_mean[c]: read before it is updated below)
  const float m1 = sums[c] / n;
  const float mu = sh + m1;
  const float m2 = fmaxf(sums[C + c] - n * m1 * m1, 0.f);   // sum (x - mu)^2
  mean[c] = mu;
  var[c] = m2 / n;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (pixels > 1 ? m2 / (float)(pixels - 1) : m2 / n);
}

__global__ void bn_copy_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ mean, float* __restrict__ var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { mean[c] = rm[c]; var[c] = rv[c]; }
}

}  // namespace

extern "C" int64_t mtbt_bn_train_workspace_bytes(int64_t pixels, int C) {
  if (pixels <= 0 || C <= 0) return 0;
  const int64_t nb = (pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (nb * 2 * C + 2 * (int64_t)C) * (int64_t)sizeof(float);
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
    if (workspace_bytes < nb * 2 * C * (int64_t)sizeof(float)) return MTBT_EWORKSPACE;
    float* partial = reinterpret_cast<float*>(workspace);
    if (dtype == MTBT_F32) hipLaunchKernelGGL(bn_block_stats_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)x, (long)pixels, C, partial);
    else hipLaunchKernelGGL(bn_block_stats_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)x, (long)pixels, C, partial);
    hipLaunchKernelGGL(bn_combine_stats_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, partial, (int)nb, C, (long)pixels, mean, var, running_mean,
                       running_var, momentum);
  }
  if (dtype == MTBT_F32) launch_bn_apply<float>(x, y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride, s);
  else launch_bn_apply<bf16_t>(x, y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride, s);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

// The same with the statistics taken from column sums (mtbt_conv_args.colsum of the conv that produced x): sums [2C] = sum (x - shift),
// sum (x - shift)^2 over the `pixels` rows; shift [C] or NULL, and it may alias running_mean (each channel is read before it is updated).
extern "C" int mtbt_bn_forward_sums_nhwc(const void* x, void* y, int32_t y_pixel_stride, const float* gamma, const float* beta, float* running_mean,
                                         float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype,
                                         const float* sums, const float* shift, float* stats, void* stream) {
  if (!x || !y || !gamma || !beta || !stats || !sums || pixels <= 0 || C <= 0 || C % 8 || C > 2048 || y_pixel_stride < C || y_pixel_stride % 8) return MTBT_EINVAL;
  if (dtype != MTBT_F32 && dtype != MTBT_BF16) return MTBT_EINVAL;
  if (act < MTBT_ACT_NONE || act > MTBT_ACT_GELU_POLY) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta) || !aligned16(stats)) return MTBT_EALIGN;
  const int CH8 = C / 8;
  if (CH8 > 256) return MTBT_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* mean = stats;
  float* var = stats + C;
  hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, sums, shift, (long)pixels, C, mean, var, running_mean,
                     running_var, momentum);
  const long total = pixels * CH8;
  long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (dtype == MTBT_F32) launch_bn_apply<float>(x, y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride, s);
  else launch_bn_apply<bf16_t>(x, y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride, s);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

// ... and from the conv's partial rows (mtbt_conv_colsum_layout gives rows / pitch): no separate second level.
extern "C" int mtbt_bn_forward_partials_nhwc(const void* x, void* y, int32_t y_pixel_stride, const float* gamma, const float* beta, float* running_mean,
                                             float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype,
                                             float* partial, int64_t rows, int32_t pitch, const float* shift, float* stats, void* stream) {
  if (!x || !y || !gamma || !beta || !stats || !partial || rows <= 0 || rows > 0x7fffffffL || pitch < 2 * C || pixels <= 0 || C <= 0 || C % 8 || C > 2048 ||
      y_pixel_stride < C || y_pixel_stride % 8)
    return MTBT_EINVAL;
  if (dtype != MTBT_F32 && dtype != MTBT_BF16) return MTBT_EINVAL;
  if (act < MTBT_ACT_NONE || act > MTBT_ACT_GELU_POLY) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta) || !aligned16(stats)) return MTBT_EALIGN;
  const int CH8 = C / 8;
  if (CH8 > 256) return MTBT_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* mean = stats;
  float* var = stats + C;
  long rr = rows;
  int pp = pitch;
  colsum_prereduce(partial, rr, pp, 0, 2 * C, s);      // tall matrices: folded in place first (rowreduce.h)
  hipLaunchKernelGGL(bn_stats_from_partials_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, partial, (int)rr, pp, shift, (long)pixels, C, mean, var,
                     running_mean, running_var, momentum);
  if (dtype == MTBT_F32) launch_bn_apply<float>(x, y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride, s);
  else launch_bn_apply<bf16_t>(x, y, mean, var, gamma, beta, eps, act, (long)pixels, C, y_pixel_stride, s);
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
  float* stats = reinterpret_cast<float*>(workspace) + nb * 2 * C;
  return mtbt_bn_forward_nhwc(x, y, C, gamma, beta, running_mean, running_var, momentum, eps, act, pixels, C, dtype, 0, stats, workspace,
                              workspace_bytes, stream);
}
