// Fused AdamW step over a flat fp32 bucket: the optimiser of the reference trainer,
// `torch.optim.AdamW(params, lr, weight_decay=0.0005)` (/root/reference/src/running_main_v3.py:732-734; betas (0.9, 0.999),
// eps 1e-8 by default), in torch's single-tensor operation order (torch/optim/adamw.py / adam.py `_single_tensor_adam`):
//     p      *= 1 - lr * weight_decay
//     m      += (g - m) * (1 - beta1)                      (Tensor.lerp_)
//     v       = v * beta2 + (1 - beta2) * g * g            (mul_ + addcmul_)
//     denom   = sqrt(v) / sqrt(1 - beta2^t) + eps
//     p      -= (lr / (1 - beta1^t)) * m / denom           (addcdiv_)
// One pass: 16 B read + 12 B written per parameter (p, g, m, v in; p, m, v out) -- HBM-bound, 16-byte accesses, no FMA
// contraction (-ffp-contract=off) so that the result follows torch's CPU rounding step by step (checked to 1e-6 relative over several
// steps, not bit for bit: torch's CUDA / foreach paths round differently from its CPU path as well).  The bias corrections are
// computed on the host in double like torch does (Python floats) and passed as fp32 scalars exactly where torch uses them.
#include <cmath>

#include "common.h"

namespace {

struct AdamP {
  float* p; const float* g; float* m; float* v;
  long n;
  float decay;        // 1 - lr * weight_decay
  float one_m_b1;     // 1 - beta1
  float b2, one_m_b2; // beta2, 1 - beta2
  float bc2_sqrt;     // sqrt(1 - beta2^t)
  float step_size;    // lr / (1 - beta1^t)
  float eps;
  const float* gscale; // optional device scalar multiplied into every gradient first (the clip coefficient of clip_grad_norm_)
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamP& q) {
  p = p * q.decay;
  m = fmaf(q.one_m_b1, g - m, m);                       // ATen's lerp is one fused multiply-add
  v = v * q.b2 + (q.one_m_b2 * g) * g;
  const float denom = sqrtf(v) / q.bc2_sqrt + q.eps;
  p = p + ((-q.step_size) * m) / denom;                 // addcdiv: self + value * t1 / t2, evaluated left to right
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamP q) {
  const long n4 = q.n >> 2;
  const float gs = q.gscale ? *q.gscale : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 p = reinterpret_cast<float4*>(q.p)[i], m = reinterpret_cast<float4*>(q.m)[i], v = reinterpret_cast<float4*>(q.v)[i];
    float4 g = reinterpret_cast<const float4*>(q.g)[i];
    if (q.gscale) { g.x *= gs; g.y *= gs; g.z *= gs; g.w *= gs; }
    adam1(p.x, g.x, m.x, v.x, q); adam1(p.y, g.y, m.y, v.y, q); adam1(p.z, g.z, m.z, v.z, q); adam1(p.w, g.w, m.w, v.w, q);
    reinterpret_cast<float4*>(q.p)[i] = p; reinterpret_cast<float4*>(q.m)[i] = m; reinterpret_cast<float4*>(q.v)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (q.n & 3)) {   // tail
    const long i = (n4 << 2) + threadIdx.x;
    adam1(q.p[i], q.gscale ? q.g[i] * gs : q.g[i], q.m[i], q.v[i], q);
  }
}

// torch.optim.SGD (single-tensor form): g += wd * p;  first step: buf = g, later: buf = momentum * buf + (1 - dampening) * g;
// g = nesterov ? g + momentum * buf : buf;  p -= lr * g.   (BASELINE configs[2] names SGD; the reference trainer itself uses AdamW.)
struct SgdP {
  float* p; const float* g; float* buf; long n;
  float lr, momentum, dampening, wd; int nesterov, first;
  const float* gscale;
};

__device__ __forceinline__ void sgd1(float& p, float g, float* buf, const SgdP& q) {
  if (q.wd != 0.f) g = g + q.wd * p;
  if (q.momentum != 0.f) {
    const float b = q.first ? g : q.momentum * (*buf) + (1.f - q.dampening) * g;
    *buf = b;
    g = q.nesterov ? g + q.momentum * b : b;
  }
  p = p - q.lr * g;
}

__global__ __launch_bounds__(256) void sgd_kernel(const SgdP q) {
  const float gs = q.gscale ? *q.gscale : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < q.n; i += (long)gridDim.x * 256) {
    float p = q.p[i];
    sgd1(p, q.gscale ? q.g[i] * gs : q.g[i], q.buf ? q.buf + i : nullptr, q);
    q.p[i] = p;
  }
}

// sum of squares of a flat fp32 buffer, deterministic: per-workgroup partials (fixed element -> thread mapping, LDS tree), then one
// workgroup over the partials; *out (+)= the sum.
__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sumsq_final(const float* __restrict__ partial, int n, float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = accumulate ? *out + red[0] : red[0];
}

// torch.nn.utils.clip_grad_norm_: total_norm = sqrt(sumsq); coef = clamp(max_norm / (total_norm + 1e-6), max = 1)
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef, float* __restrict__ norm_out) {
  const float norm = sqrtf(*sumsq);
  const float c = max_norm / (norm + 1e-6f);
  *coef = c > 1.f ? 1.f : c;
  if (norm_out) *norm_out = norm;
}

}  // namespace

extern "C" int mtbt_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return MTBT_EINVAL;
  if (n == 0) return MTBT_OK;
  if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq)) return MTBT_EALIGN;
  AdamP q;
  q.p = param; q.g = grad; q.m = exp_avg; q.v = exp_avg_sq; q.n = n;
  // torch computes these in Python floats (double) and hands the results to fp32 tensor ops
  const double b1 = (double)beta1, b2 = (double)beta2, l = (double)lr;
  q.decay = (float)(1.0 - l * (double)weight_decay);
  q.one_m_b1 = (float)(1.0 - b1);
  q.b2 = beta2;
  q.one_m_b2 = (float)(1.0 - b2);
  q.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(b2, (double)step));
  q.step_size = (float)(l / (1.0 - std::pow(b1, (double)step)));
  q.eps = eps;
  q.gscale = grad_scale;
  long blocks = ((n >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                             float weight_decay, int nesterov, int64_t step, const float* grad_scale, void* stream) {
  if (!param || !grad || n < 0 || step < 1 || (momentum != 0.f && !momentum_buf)) return MTBT_EINVAL;
  if (nesterov && (momentum <= 0.f || dampening != 0.f)) return MTBT_EINVAL;
  if (n == 0) return MTBT_OK;
  SgdP q;
  q.p = param; q.g = grad; q.buf = momentum_buf; q.n = n; q.lr = lr; q.momentum = momentum; q.dampening = dampening; q.wd = weight_decay;
  q.nesterov = nesterov; q.first = step == 1; q.gscale = grad_scale;
  long blocks = (n + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int64_t mtbt_sumsq_workspace_bytes(void) { return 1024 * (int64_t)sizeof(float); }

extern "C" int mtbt_sumsq(const float* g, int64_t n, float* out, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!g || !out || !workspace || n < 0 || workspace_bytes < mtbt_sumsq_workspace_bytes()) return MTBT_EINVAL;
  long blocks = (n + 256 * 16 - 1) / (256 * 16);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sumsq_partial, dim3((unsigned)blocks), dim3(256), 0, s, g, (long)n, reinterpret_cast<float*>(workspace));
  hipLaunchKernelGGL(sumsq_final, dim3(1), dim3(256), 0, s, reinterpret_cast<const float*>(workspace), (int)blocks, out, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream) {
  if (!sumsq || !coef || !(max_norm > 0.f)) return MTBT_EINVAL;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), sumsq, max_norm, coef, norm_out);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
