// Segmentation metric accumulators on the device.
//
// `validation_step` (/root/reference/src/running_main_v3.py:466-498) thresholds sigmoid(seg logits) at 0.5 and feeds the
// pixels to torchmetrics' binary F1 / precision / recall / accuracy / Dice accumulators (:198-203), and per image builds
// the single-instance record of the segmentation mAP: mask = prob > 0.5, score = sum(prob * mask) / (sum(mask) + 1e-6)
// (:480-497).  All of these are functions of four pixel counts per image plus that one sum, so one pass over
// (logits, gt) -- 8 bytes per pixel, HBM-bound -- produces per image {TP, FP, FN, TN} and sum(prob over predicted foreground);
// nothing is synchronised with the host until the caller asks for the numbers.
//
//   prediction  sigmoid(x) > 0.5 in fp32  <=>  x > 2^-24 (for 0 < x <= 2^-24, exp(-x) rounds to 1 and the quotient to 0.5)
//   target      masks_gt.int() > 0.5      <=>  trunc(t) >= 1
// Deterministic: fixed grid, per-workgroup partials, fixed-order second pass.
#include <cmath>

#include "common.h"

namespace {

constexpr int NB = 32;   // workgroups per image

struct Part { unsigned tp, fp, fn; float psum; };

__global__ __launch_bounds__(256) void seg_confusion_kernel(const float* __restrict__ logits, const float* __restrict__ gt, long n,
                                                            Part* __restrict__ part) {
  __shared__ Part red[4];
  const float4* x4 = reinterpret_cast<const float4*>(logits + (long)blockIdx.y * n);
  const float4* t4 = reinterpret_cast<const float4*>(gt + (long)blockIdx.y * n);
  unsigned tp = 0, fp = 0, fn = 0;
  float ps = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += (long)NB * 256) {
    const float4 x = x4[i], t = t4[i];
    const float xs[4] = {x.x, x.y, x.z, x.w}, ts[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool p = xs[k] > 5.9604644775390625e-08f, g = (int)ts[k] >= 1;
      tp += p && g; fp += p && !g; fn += !p && g;
      if (p) ps += 1.f / (1.f + expf(-xs[k]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64);
  }
  ps = wave_sum(ps);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = Part{tp, fp, fn, ps};
  __syncthreads();
  if (threadIdx.x == 0) {
    Part r{red[0].tp + red[1].tp + red[2].tp + red[3].tp, red[0].fp + red[1].fp + red[2].fp + red[3].fp,
           red[0].fn + red[1].fn + red[2].fn + red[3].fn, (red[0].psum + red[1].psum) + (red[2].psum + red[3].psum)};
    part[(long)blockIdx.y * NB + blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(64) void seg_confusion_finish(const Part* __restrict__ part, long n, long long* __restrict__ counts,
                                                           float* __restrict__ prob_sum, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  long long tp = 0, fp = 0, fn = 0;
  float ps = 0.f;
  for (int k = 0; k < NB; ++k) {
    const Part p = part[(long)b * NB + k];
    tp += p.tp; fp += p.fp; fn += p.fn; ps += p.psum;
  }
  counts[b * 4 + 0] = tp; counts[b * 4 + 1] = fp; counts[b * 4 + 2] = fn; counts[b * 4 + 3] = n - tp - fp - fn;
  prob_sum[b] = ps;
}

}  // namespace

extern "C" int64_t mtbt_seg_confusion_workspace_bytes(int B) { return (int64_t)(B > 0 ? B : 0) * NB * (int64_t)sizeof(Part); }

extern "C" int mtbt_seg_confusion(const float* logits, const float* gt, int B, int64_t n_per_image, int64_t* counts, float* prob_sum,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  if (!logits || !gt || !counts || !prob_sum || B < 0 || n_per_image <= 0 || n_per_image % 4 || n_per_image > 0xffffffffLL) return MTBT_EINVAL;
  if (B == 0) return MTBT_OK;
  if (!workspace || workspace_bytes < mtbt_seg_confusion_workspace_bytes(B)) return MTBT_EINVAL;
  if (!aligned16(logits) || !aligned16(gt) || !aligned16(workspace)) return MTBT_EALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(seg_confusion_kernel, dim3(NB, (unsigned)B), dim3(256), 0, s, logits, gt, (long)n_per_image, reinterpret_cast<Part*>(workspace));
  MTBT_LAUNCH_CHECK();
  hipLaunchKernelGGL(seg_confusion_finish, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, reinterpret_cast<const Part*>(workspace),
                     (long)n_per_image, reinterpret_cast<long long*>(counts), prob_sum, B);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
