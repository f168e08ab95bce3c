// Multitask loss VALUE on the device (`MultiTaskLitModel._multitask_loss`, running_main_v3.py:232-387):
// no per-image Python loop, no `.item()` synchronisation.  Forward only (the backward pass is not built yet).
//
//   det_loss_kernel   4 lanes per (image, anchor), one per box side: softmax expectation of the side's 16-bin
//                     distribution (+ its log-sum-exp for the DFL cross-entropy), box decode, IoU against the image's
//                     GT boxes (first maximum wins, like torch.max), positives = max IoU > threshold; per positive:
//                     1 - IoU, BCE-with-logits(sum) of the class logits against one-hot / label-smoothed targets,
//                     two-bin DFL cross-entropy.  Workgroup partial sums -> workspace (no atomics).
//   bce_kernel        sum of BCE-with-logits over the S x S segmentation logits (the 1x1 projector + bilinear resize is
//                     mtbt_mask_assemble's projector path), workgroup partials.
//   finalize_kernel   fixed-order reduction of the partials, image-classification cross-entropy, normalisation by the
//                     batch's positive count (batch size if none), weighted total.  One workgroup.
// Deterministic (fixed reduction orders); fp32 with libm exp / log (this file is built with -ffp-contract=off like the rest
// of the post-process so that the IoU matches torch's arithmetic).
#include <cmath>

#include "common.h"

namespace {

struct LossP {
  const float* map[3];
  int h[3], w[3], ld[3];
  int off[4];
  float stride[3];
  int n_levels, N, A, nc, reg_max;
  const float* gt_xyxy;   // [G][4]
  const int* gt_cls;      // [G]
  const int* gt_off;      // [N+1]
  float iou_thresh, smoothing;
  int training;
  float* partial;         // [blocks][5]: n_pos, sum(1-iou), sum(iou), sum(cls bce), sum(dfl)
};

__device__ __forceinline__ float iou_xyxy(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
  // running_main_v3.py:71-97
  const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float inter = iw * ih;
  const float a1 = (ax2 - ax1) * (ay2 - ay1), a2 = (bx2 - bx1) * (by2 - by1);
  return inter / (a1 + a2 - inter + 1e-7f);
}

__device__ __forceinline__ float bce_logits(float x, float t) {  // torch: max(x,0) - x*t + log(1 + exp(-|x|))
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void det_loss_kernel(const LossP p) {
  __shared__ float red[4][5];
  const long g = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int side = threadIdx.x & 3;
  const long total = (long)p.N * p.A;
  const bool live = g < total;
  const long gg = live ? g : 0;
  const int n = (int)(gg / p.A), a = (int)(gg - (long)n * p.A);
  int l = 0;
  if (p.n_levels > 1 && a >= p.off[1]) l = 1;
  if (p.n_levels > 2 && a >= p.off[2]) l = 2;
  const int cell = a - p.off[l];
  const int w = p.w[l], hw = p.h[l] * w;
  const int cy = cell / w, cx = cell - cy * w;
  const float* row = p.map[l] + ((long)n * hw + cell) * p.ld[l];
  const float* d = row + side * p.reg_max;

  // this side's distribution: expectation and log-sum-exp
  float m = -INFINITY;
  for (int i = 0; i < p.reg_max; ++i) m = fmaxf(m, d[i]);
  float s = 0.f;
  for (int i = 0; i < p.reg_max; ++i) s += expf(d[i] - m);
  float dist = 0.f;
  for (int i = 0; i < p.reg_max; ++i) dist += (expf(d[i] - m) / s) * (float)i;
  const float lse = m + logf(s);

  const int qbase = (threadIdx.x & 63) & ~3;
  const float st = p.stride[l];
  const float ax = (cx + 0.5f) * st, ay = (cy + 0.5f) * st;
  const float x1 = ax - __shfl(dist, qbase + 0, 64) * st, y1 = ay - __shfl(dist, qbase + 1, 64) * st;
  const float x2 = ax + __shfl(dist, qbase + 2, 64) * st, y2 = ay + __shfl(dist, qbase + 3, 64) * st;

  // match against this image's GT boxes (every lane of the group computes the same thing)
  const int g0 = p.gt_off[n], g1 = p.gt_off[n + 1];
  float best = -INFINITY;
  int bi = -1;
  for (int k = g0; k < g1; ++k) {
    const float4 b = *reinterpret_cast<const float4*>(p.gt_xyxy + 4 * k);
    const float v = iou_xyxy(x1, y1, x2, y2, b.x, b.y, b.z, b.w);
    if (v > best) { best = v; bi = k; }
  }
  const bool pos = live && bi >= 0 && best > p.iou_thresh;

  float v_cnt = 0.f, v_box = 0.f, v_iou = 0.f, v_cls = 0.f, v_dfl = 0.f;
  if (pos) {
    const float4 b = *reinterpret_cast<const float4*>(p.gt_xyxy + 4 * bi);
    // DFL target of this side (running_main_v3.py:351-367)
    const float apc = (side & 1) ? ay : ax;
    const float gtc = side == 0 ? b.x : side == 1 ? b.y : side == 2 ? b.z : b.w;
    float t = ((side < 2) ? (apc - gtc) : (gtc - apc)) / st;
    t = fminf(fmaxf(t, 0.f), (float)p.reg_max - 1.01f);
    int tl = (int)floorf(t);
    tl = min(max(tl, 0), p.reg_max - 1);
    const int tr = min(tl + 1, p.reg_max - 1);
    const float wl = (float)tr - t, wr = t - (float)tl;
    v_dfl = (lse - d[tl]) * wl + (lse - d[tr]) * wr;
    // class BCE: the group's lanes split the classes
    const int gc = p.gt_cls[bi];
    const bool smooth = p.smoothing > 0.f && p.training;
    for (int c = side; c < p.nc; c += 4) {
      const float tgt = smooth ? (c == gc ? 1.f - p.smoothing : p.smoothing / (float)(p.nc - 1)) : (c == gc ? 1.f : 0.f);
      v_cls += bce_logits(row[4 * p.reg_max + c], tgt);
    }
    if (side == 0) { v_cnt = 1.f; v_box = 1.f - best; v_iou = best; }
  }
  // workgroup reduction in a fixed order: wave sums, then 4 wave partials
  float vals[5] = {v_cnt, v_box, v_iou, v_cls, v_dfl};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const float r = wave_sum(vals[q]);
    if (lane == 0) red[wave][q] = r;
  }
  __syncthreads();
  if (threadIdx.x < 5) p.partial[(long)blockIdx.x * 5 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ bias,
                                                  long n, float* __restrict__ partial) {
  __shared__ float red[4];
  const float b = bias ? *bias : 0.f;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += bce_logits(x[i] + b, t[i]);
  const float r = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct FinP {
  const float* det_partial; int det_blocks;
  const float* seg_partial; int seg_blocks; long seg_n;
  const float* img_logits; const long long* img_gt; int N, n_img_classes;
  float w_seg, w_box, w_dfl, w_cls, w_img;
  float* out;  // total, seg, box, dfl, cls_det, img_cls, n_pos, mean matched IoU
};

__global__ __launch_bounds__(256) void finalize_kernel(const FinP p) {
  __shared__ float red[4][8];
  float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // 5 detection sums, seg sum, img CE sum
  for (int b = threadIdx.x; b < p.det_blocks; b += 256)
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] += p.det_partial[(long)b * 5 + q];
  for (int b = threadIdx.x; b < p.seg_blocks; b += 256) v[5] += p.seg_partial[b];
  for (int i = threadIdx.x; i < p.N; i += 256) {  // CrossEntropyLoss, mean over the batch
    const float* lg = p.img_logits + (long)i * p.n_img_classes;
    float m = -INFINITY;
    for (int c = 0; c < p.n_img_classes; ++c) m = fmaxf(m, lg[c]);
    float s = 0.f;
    for (int c = 0; c < p.n_img_classes; ++c) s += expf(lg[c] - m);
    v[6] += (m + logf(s)) - lg[p.img_gt[i]];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float r = wave_sum(v[q]);
    if (lane == 0) red[wave][q] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[7];
    for (int q = 0; q < 7; ++q) t[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
    const float n_pos = t[0];
    const float norm = n_pos > 0.f ? n_pos : (float)p.N;                 // running_main_v3.py:371
    const float box = t[1] / norm, cls = t[3] / norm, dfl = t[4] / norm;
    const float seg = p.seg_n > 0 ? t[5] / (float)p.seg_n : 0.f;
    const float img = t[6] / (float)p.N;
    p.out[0] = p.w_seg * seg + p.w_box * box + p.w_dfl * dfl + p.w_cls * cls + p.w_img * img;
    p.out[1] = seg; p.out[2] = box; p.out[3] = dfl; p.out[4] = cls; p.out[5] = img;
    p.out[6] = n_pos;
    p.out[7] = n_pos > 0.f ? t[2] / n_pos : 0.f;
  }
}

inline long det_blocks_of(long N, long A) { return (N * A + 63) / 64; }
inline long seg_blocks_of(long n) { long b = (n + 256 * 8 - 1) / (256 * 8); return b > 2048 ? 2048 : (b < 1 ? 1 : b); }

// ---- gradient of the weighted total with respect to the three head outputs (first operator of the backward pass) ----
// Same thread mapping as det_loss_kernel.  What autograd does on the reference's expressions:
//   box   d(1 - IoU)/d(corner) through batch_bbox_iou (:71-97; max / min pick the larger / smaller operand, clamp(min=0) passes
//         the gradient where its input is positive), corner = anchor -/+ dist * stride, dist = sum_j softmax(raw)_j * j
//         => d dist / d raw_j = p_j (j - dist)
//   DFL   two-bin cross-entropy: wl (p - onehot(tl)) + wr (p - onehot(tr))
//   class BCE-with-logits(sum): sigmoid(x) - target
// each scaled by its weight / (#positives of the batch, batch size if none) -- read from the forward's out[6].  Anchors that are
// not positives get zeros.  The positives mask and the matched GT index carry no gradient (comparison / argmax).
struct GradP {
  LossP l;
  float* dmap[3];
  int dld[3];
  const float* fwd_out;   // out[8] of mtbt_multitask_loss
  float w_box, w_dfl, w_cls;
};

__global__ __launch_bounds__(256) void det_loss_grad_kernel(const GradP q) {
  const LossP& p = q.l;
  const long g = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int side = threadIdx.x & 3;
  const long total = (long)p.N * p.A;
  const bool live = g < total;
  const long gg = live ? g : 0;
  const int n = (int)(gg / p.A), a = (int)(gg - (long)n * p.A);
  int l = 0;
  if (p.n_levels > 1 && a >= p.off[1]) l = 1;
  if (p.n_levels > 2 && a >= p.off[2]) l = 2;
  const int cell = a - p.off[l];
  const int w = p.w[l], hw = p.h[l] * w;
  const int cy = cell / w, cx = cell - cy * w;
  const float* row = p.map[l] + ((long)n * hw + cell) * p.ld[l];
  const float* d = row + side * p.reg_max;
  float* drow = q.dmap[l] + ((long)n * hw + cell) * q.dld[l];

  float m = -INFINITY;
  for (int i = 0; i < p.reg_max; ++i) m = fmaxf(m, d[i]);
  float s = 0.f;
  for (int i = 0; i < p.reg_max; ++i) s += expf(d[i] - m);
  float dist = 0.f;
  for (int i = 0; i < p.reg_max; ++i) dist += (expf(d[i] - m) / s) * (float)i;

  const int qbase = (threadIdx.x & 63) & ~3;
  const float st = p.stride[l];
  const float ax = (cx + 0.5f) * st, ay = (cy + 0.5f) * st;
  const float x1 = ax - __shfl(dist, qbase + 0, 64) * st, y1 = ay - __shfl(dist, qbase + 1, 64) * st;
  const float x2 = ax + __shfl(dist, qbase + 2, 64) * st, y2 = ay + __shfl(dist, qbase + 3, 64) * st;

  const int g0 = p.gt_off[n], g1 = p.gt_off[n + 1];
  float best = -INFINITY;
  int bi = -1;
  for (int k = g0; k < g1; ++k) {
    const float4 b = *reinterpret_cast<const float4*>(p.gt_xyxy + 4 * k);
    const float v = iou_xyxy(x1, y1, x2, y2, b.x, b.y, b.z, b.w);
    if (v > best) { best = v; bi = k; }
  }
  const bool pos = live && bi >= 0 && best > p.iou_thresh;
  if (!live) return;
  const int no = 4 * p.reg_max + p.nc;
  if (!pos) {
    for (int i = 0; i < p.reg_max; ++i) drow[side * p.reg_max + i] = 0.f;
    for (int c = side; c < p.nc; c += 4) drow[4 * p.reg_max + c] = 0.f;
    (void)no;
    return;
  }
  const float n_pos = q.fwd_out[6];
  const float inv = 1.f / (n_pos > 0.f ? n_pos : (float)p.N);
  const float4 b = *reinterpret_cast<const float4*>(p.gt_xyxy + 4 * bi);
  // d IoU / d (this lane's corner)
  const float iwr = fminf(x2, b.z) - fmaxf(x1, b.x), ihr = fminf(y2, b.w) - fmaxf(y1, b.y);
  const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
  const float inter = iw * ih;
  const float bw = x2 - x1, bh = y2 - y1;
  const float uni = bw * bh + (b.z - b.x) * (b.w - b.y) - inter + 1e-7f;
  float dinter, darea;   // with respect to the corner owned by `side`
  if (side == 0)      { dinter = (iwr > 0.f && x1 > b.x) ? -ih : 0.f; darea = -bh; }
  else if (side == 1) { dinter = (ihr > 0.f && y1 > b.y) ? -iw : 0.f; darea = -bw; }
  else if (side == 2) { dinter = (iwr > 0.f && x2 < b.z) ? ih : 0.f;  darea = bh; }
  else                { dinter = (ihr > 0.f && y2 < b.w) ? iw : 0.f;  darea = bw; }
  const float diou = (dinter * uni - inter * (darea - dinter)) / (uni * uni);
  const float dcorner = -q.w_box * inv * diou;                       // d total / d corner
  const float ddist = dcorner * (side < 2 ? -st : st);               // corner = anchor -/+ dist * stride
  // DFL target of this side
  const float apc = (side & 1) ? ay : ax;
  const float gtc = side == 0 ? b.x : side == 1 ? b.y : side == 2 ? b.z : b.w;
  float t = ((side < 2) ? (apc - gtc) : (gtc - apc)) / st;
  t = fminf(fmaxf(t, 0.f), (float)p.reg_max - 1.01f);
  int tl = (int)floorf(t);
  tl = min(max(tl, 0), p.reg_max - 1);
  const int tr = min(tl + 1, p.reg_max - 1);
  const float wl = (float)tr - t, wr = t - (float)tl;
  const float kd = q.w_dfl * inv;
  for (int i = 0; i < p.reg_max; ++i) {
    const float pj = expf(d[i] - m) / s;
    float gv = ddist * pj * ((float)i - dist) + kd * (wl + wr) * pj;
    if (i == tl) gv -= kd * wl;
    if (i == tr) gv -= kd * wr;
    drow[side * p.reg_max + i] = gv;
  }
  const int gc = p.gt_cls[bi];
  const bool smooth = p.smoothing > 0.f && p.training;
  for (int c = side; c < p.nc; c += 4) {
    const float tgt = smooth ? (c == gc ? 1.f - p.smoothing : p.smoothing / (float)(p.nc - 1)) : (c == gc ? 1.f : 0.f);
    const float x = row[4 * p.reg_max + c];
    drow[4 * p.reg_max + c] = q.w_cls * inv * (1.f / (1.f + expf(-x)) - tgt);
  }
}

// d total / d seg logit = w_seg / n * (sigmoid(x + bias) - target);  d total / d img logit = w_img / N * (softmax - onehot)
__global__ __launch_bounds__(256) void seg_img_grad_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ bias,
                                                           long n, float k_seg, float* __restrict__ dx, const float* __restrict__ img_logits,
                                                           const long long* __restrict__ img_gt, int N, int ncls, float k_img,
                                                           float* __restrict__ dimg) {
  const float b = bias ? *bias : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = k_seg * (1.f / (1.f + expf(-(x[i] + b))) - t[i]);
  if (blockIdx.x == 0 && dimg) {
    for (int i = threadIdx.x; i < N; i += 256) {
      const float* lg = img_logits + (long)i * ncls;
      float m = -INFINITY;
      for (int c = 0; c < ncls; ++c) m = fmaxf(m, lg[c]);
      float s = 0.f;
      for (int c = 0; c < ncls; ++c) s += expf(lg[c] - m);
      for (int c = 0; c < ncls; ++c) dimg[(long)i * ncls + c] = k_img * (expf(lg[c] - m) / s - (c == img_gt[i] ? 1.f : 0.f));
    }
  }
}

}  // namespace

// validated kernel parameters shared by the value and the gradient entry points
static int fill_loss_params(const mtbt_loss_args* a, LossP& p, int& A) {
  if (!a || !a->out || a->n_levels < 1 || a->n_levels > 3 || a->N <= 0 || a->nc <= 0 || a->reg_max <= 0 || a->reg_max > 64) return MTBT_EINVAL;
  if (!a->gt_xyxy || !a->gt_cls || !a->gt_off || !a->img_logits || !a->img_gt || a->n_img_classes <= 0) return MTBT_EINVAL;
  if ((a->seg_n > 0) != (a->seg_logits != nullptr && a->seg_targets != nullptr)) return MTBT_EINVAL;
  if (!aligned16(a->gt_xyxy)) return MTBT_EALIGN;
  A = 0;
  for (int l = 0; l < 3; ++l) {
    p.off[l] = A;
    if (l < a->n_levels) {
      if (!a->map[l] || a->h[l] <= 0 || a->w[l] <= 0 || a->map_pixel_stride[l] < 4 * a->reg_max + a->nc) return MTBT_EINVAL;
      p.map[l] = a->map[l]; p.h[l] = a->h[l]; p.w[l] = a->w[l]; p.ld[l] = a->map_pixel_stride[l];
      p.stride[l] = a->img_size / (float)a->w[l];                       // running_main_v3.py:266
      A += a->h[l] * a->w[l];
    } else { p.map[l] = nullptr; p.h[l] = p.w[l] = 1; p.ld[l] = 0; p.stride[l] = 0.f; }
  }
  p.off[3] = A;
  p.n_levels = a->n_levels; p.N = a->N; p.A = A; p.nc = a->nc; p.reg_max = a->reg_max;
  p.gt_xyxy = a->gt_xyxy; p.gt_cls = a->gt_cls; p.gt_off = a->gt_off;
  p.iou_thresh = a->iou_thresh; p.smoothing = a->label_smoothing; p.training = a->training;
  return MTBT_OK;
}

extern "C" int64_t mtbt_loss_workspace_bytes(int N, int A, int64_t seg_n) {
  if (N <= 0 || A <= 0 || seg_n < 0) return 0;
  return (det_blocks_of(N, A) * 5 + seg_blocks_of(seg_n)) * (int64_t)sizeof(float);
}

extern "C" int mtbt_multitask_loss(const mtbt_loss_args* a, void* stream) {
  if (!a || !a->workspace) return MTBT_EINVAL;
  LossP p;
  int A = 0;
  if (int rc = fill_loss_params(a, p, A)) return rc;
  const long db = det_blocks_of(a->N, A), sb = a->seg_n > 0 ? seg_blocks_of(a->seg_n) : 0;
  if (db > 0x7fffffffL) return MTBT_EINVAL;
  if (a->workspace_bytes < mtbt_loss_workspace_bytes(a->N, A, a->seg_n)) return MTBT_EWORKSPACE;
  float* det_partial = a->workspace;
  float* seg_partial = det_partial + db * 5;
  p.partial = det_partial;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(det_loss_kernel, dim3((unsigned)db), dim3(256), 0, s, p);
  if (sb > 0) hipLaunchKernelGGL(bce_kernel, dim3((unsigned)sb), dim3(256), 0, s, a->seg_logits, a->seg_targets, a->seg_bias, (long)a->seg_n, seg_partial);
  FinP f;
  f.det_partial = det_partial; f.det_blocks = (int)db; f.seg_partial = seg_partial; f.seg_blocks = (int)sb; f.seg_n = a->seg_n;
  f.img_logits = a->img_logits; f.img_gt = reinterpret_cast<const long long*>(a->img_gt); f.N = a->N; f.n_img_classes = a->n_img_classes;
  f.w_seg = a->w_seg; f.w_box = a->w_box; f.w_dfl = a->w_dfl; f.w_cls = a->w_cls; f.w_img = a->w_img;
  f.out = a->out;
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, f);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_multitask_loss_grad(const mtbt_loss_args* a, float* const* d_map, const int32_t* d_map_pixel_stride, float* d_seg_logits,
                                        float* d_img_logits, void* stream) {
  if (!a || !d_map || !d_map_pixel_stride) return MTBT_EINVAL;
  GradP q;
  int A = 0;
  if (int rc = fill_loss_params(a, q.l, A)) return rc;
  for (int l = 0; l < 3; ++l) {
    q.dmap[l] = nullptr; q.dld[l] = 0;
    if (l < a->n_levels) {
      if (!d_map[l] || d_map_pixel_stride[l] < 4 * a->reg_max + a->nc) return MTBT_EINVAL;
      q.dmap[l] = d_map[l]; q.dld[l] = d_map_pixel_stride[l];
    }
  }
  if (a->seg_n > 0 && !d_seg_logits) return MTBT_EINVAL;
  q.l.partial = nullptr;
  q.fwd_out = a->out; q.w_box = a->w_box; q.w_dfl = a->w_dfl; q.w_cls = a->w_cls;
  const long db = det_blocks_of(a->N, A);
  if (db > 0x7fffffffL) return MTBT_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(det_loss_grad_kernel, dim3((unsigned)db), dim3(256), 0, s, q);
  const long sb = a->seg_n > 0 ? seg_blocks_of(a->seg_n) : 1;
  hipLaunchKernelGGL(seg_img_grad_kernel, dim3((unsigned)sb), dim3(256), 0, s, a->seg_logits, a->seg_targets, a->seg_bias, (long)a->seg_n,
                     a->seg_n > 0 ? a->w_seg / (float)a->seg_n : 0.f, d_seg_logits, a->img_logits, reinterpret_cast<const long long*>(a->img_gt), a->N,
                     a->n_img_classes, a->w_img / (float)a->N, d_img_logits);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
