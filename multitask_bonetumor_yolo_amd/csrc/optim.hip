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
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 p = reinterpret_cast<float4*>(q.p)[i], m = reinterpret_cast<float4*>(q.m)[i], v = reinterpret_cast<float4*>(q.v)[i];
    const float4 g = reinterpret_cast<const float4*>(q.g)[i];
    adam1(p.x, g.x, m.x, v.x, q); adam1(p.y, g.y, m.y, v.y, q); adam1(p.z, g.z, m.z, v.z, q); adam1(p.w, g.w, m.w, v.w, q);
    reinterpret_cast<float4*>(q.p)[i] = p; reinterpret_cast<float4*>(q.m)[i] = m; reinterpret_cast<float4*>(q.v)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (q.n & 3)) {   // tail
    const long i = (n4 << 2) + threadIdx.x;
    adam1(q.p[i], q.g[i], q.m[i], q.v[i], q);
  }
}

}  // namespace

extern "C" int mtbt_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int64_t step, void* stream) {
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
  long blocks = ((n >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
