// Input pipeline on the device: the per-sample image work of `BTXRDDataset.__getitem__`
// (/root/reference/src/dataset_btxrdv2.py:109-166) for a whole batch in one launch:
//
//   _letterbox (:109-134)   scale = S / max(H0, W0); new = max(1, int(dim * scale));
//                           image  cv2.resize(INTER_LINEAR), mask  cv2.resize(INTER_NEAREST);
//                           copyMakeBorder top-left aligned: image pad (114,114,114), mask pad 0
//   :157-166                BGR -> RGB, float32 / 255, HWC -> CHW;  mask / 255 > 0.5 -> {0,1} float32, [1,S,S]
//
// cv2 is a third-party dependency that is absent here; the arithmetic restated below is OpenCV's published 8-bit
// path (modules/imgproc/src/resize.cpp; IPP is not used for 8-bit linear unless "not exact" IPP is enabled):
//   linear   fx = (float)((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx; sx < 0 -> (0, 0); sx >= W-1 -> (W-1, 0);
//            coefficients short(round_half_even((1-fx) * 2048)), short(round_half_even(fx * 2048));
//            horizontal pass in int:  r = S[sx] * a0 + S[sx+1] * a1;
//            vertical pass:           dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
//   nearest  sx = min(floor(dx * scale_x), W-1), same for y
// with scale_x = 1.0 / ((double)new_w / W0).  All of it is integer work except the coefficient set-up, which is done in
// the same double/float steps (this file is compiled with -ffp-contract=off so that (dx+0.5)*scale-0.5 is not fused).
//
// One thread produces 4 horizontally adjacent output pixels of one image: three float4 plane stores + one mask
// float4 store (the output, 16 B per pixel, is the HBM traffic that bounds the kernel); the <= 12 source bytes per pixel
// are gathered through L2.
#include "common.h"

namespace {

constexpr int MAX_IMAGES = 32;   // images per launch (descriptors travel as kernel arguments)

struct RawImage {
  const uint8_t* bgr;    // [H0][row_stride] bytes, 3 per pixel
  const uint8_t* mask;   // [H0][mask_stride] or null
  long row_stride, mask_stride;
  int H0, W0, new_h, new_w;
  double scale_x, scale_y;   // source step per output pixel (cv2's 1 / inv_scale)
};
struct Batch { RawImage im[MAX_IMAGES]; };

__device__ __forceinline__ void linear_tap(int d, double scale, int size, int& s0, int& s1, int& c0, int& c1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= size - 1) { f = 0.f; s = size - 1; }
  s0 = s;
  s1 = min(s + 1, size - 1);
  c0 = __float2int_rn((1.f - f) * 2048.f);
  c1 = __float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(256) void letterbox_kernel(const Batch b, int S, float* __restrict__ out_img, float* __restrict__ out_mask) {
  const RawImage& im = b.im[blockIdx.y];
  const int quads = S >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= S * quads) return;
  const int dy = q / quads, dx0 = (q - dy * quads) << 2;
  float r[4], g[4], bl[4], m[4];
  const float pad = __fdiv_rn(114.f, 255.f);
  const bool row_in = dy < im.new_h;
  int sy0 = 0, sy1 = 0, b0 = 0, b1 = 0, my = 0;
  if (row_in) {
    linear_tap(dy, im.scale_y, im.H0, sy0, sy1, b0, b1);
    my = min((int)floor(dy * im.scale_y), im.H0 - 1);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int dx = dx0 + i;
    r[i] = g[i] = bl[i] = pad;
    m[i] = 0.f;
    if (row_in && dx < im.new_w) {
      int sx0, sx1, a0, a1;
      linear_tap(dx, im.scale_x, im.W0, sx0, sx1, a0, a1);
      const uint8_t* p0 = im.bgr + sy0 * im.row_stride;
      const uint8_t* p1 = im.bgr + sy1 * im.row_stride;
      int v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int r0 = p0[sx0 * 3 + c] * a0 + p0[sx1 * 3 + c] * a1;
        const int r1 = p1[sx0 * 3 + c] * a0 + p1[sx1 * 3 + c] * a1;
        const int o = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v[c] = min(max(o, 0), 255);
      }
      bl[i] = __fdiv_rn((float)v[0], 255.f);
      g[i] = __fdiv_rn((float)v[1], 255.f);
      r[i] = __fdiv_rn((float)v[2], 255.f);
      if (im.mask) {
        const int mx = min((int)floor(dx * im.scale_x), im.W0 - 1);
        m[i] = im.mask[my * im.mask_stride + mx] >= 128 ? 1.f : 0.f;   // v / 255 > 0.5  <=>  v >= 128
      }
    }
  }
  const long plane = (long)S * S;
  float* o = out_img + (long)blockIdx.y * 3 * plane + (long)dy * S + dx0;
  *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
  *reinterpret_cast<float4*>(o + plane) = make_float4(g[0], g[1], g[2], g[3]);
  *reinterpret_cast<float4*>(o + 2 * plane) = make_float4(bl[0], bl[1], bl[2], bl[3]);
  if (out_mask) *reinterpret_cast<float4*>(out_mask + (long)blockIdx.y * plane + (long)dy * S + dx0) = make_float4(m[0], m[1], m[2], m[3]);
}

}  // namespace

extern "C" int mtbt_letterbox_batch(const mtbt_raw_image* images, int count, int img_size, float* out_images, float* out_masks,
                                    double* out_scales, void* stream) {
  if (!images || count < 0 || img_size <= 0 || img_size % 4 || !out_images) return MTBT_EINVAL;
  if (!aligned16(out_images) || (out_masks && !aligned16(out_masks))) return MTBT_EALIGN;
  const long plane = (long)img_size * img_size;
  for (int first = 0; first < count; first += MAX_IMAGES) {
    const int nb = count - first < MAX_IMAGES ? count - first : MAX_IMAGES;
    Batch b;
    for (int i = 0; i < nb; ++i) {
      const mtbt_raw_image& s = images[first + i];
      if (!s.bgr || s.height <= 0 || s.width <= 0 || s.row_stride < (int64_t)s.width * 3 || (s.mask && s.mask_row_stride < s.width) ||
          (long)s.height * s.row_stride >= 0x7fffffffL)
        return MTBT_EINVAL;
      RawImage& d = b.im[i];
      d.bgr = s.bgr; d.mask = s.mask; d.row_stride = s.row_stride; d.mask_stride = s.mask_row_stride;
      d.H0 = s.height; d.W0 = s.width;
      // dataset_btxrdv2.py:114-117 in the same double arithmetic as Python's floats
      const double scale = (double)img_size / (double)(s.height > s.width ? s.height : s.width);
      const int nw = (int)((double)s.width * scale), nh = (int)((double)s.height * scale);
      d.new_w = nw < 1 ? 1 : nw;
      d.new_h = nh < 1 ? 1 : nh;
      if (d.new_w > img_size || d.new_h > img_size) return MTBT_EINVAL;   // cannot happen for scale = S / max(H0, W0)
      d.scale_x = 1.0 / ((double)d.new_w / (double)s.width);
      d.scale_y = 1.0 / ((double)d.new_h / (double)s.height);
      if (out_scales) out_scales[first + i] = scale;
    }
    const unsigned gx = (unsigned)((plane / 4 + 255) / 256);
    hipLaunchKernelGGL(letterbox_kernel, dim3(gx, (unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), b, img_size,
                       out_images + (long)first * 3 * plane, out_masks ? out_masks + (long)first * plane : nullptr);
    MTBT_LAUNCH_CHECK();
  }
  return MTBT_OK;
}
