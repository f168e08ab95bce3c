// Weight gradient of a stride-1 k x k convolution, bf16 operands, fp32 result:
//     dW[k][r][s][c] = sum over pixels p = (n, y, x) of  dY[p][k] * X[n][y + r - pad][x + s - pad][c]
// The reduction runs over the NHWC-SLOW index, so both MFMA operands arrive transposed with respect to how the forward
// kernels read them.  gfx950's ds_read_b64_tr_b16 delivers a 4-row x 16-column block of 16-bit elements column-major to a
// 16-lane group, which is exactly one half of a 16x16x32 operand fragment, so the tiles are staged in their natural layout
// ([pixel][channel] rows, 16-byte global loads) and transposed by the read.
//
//   workgroup   one (64 output channels) x (64 input channels) tile of one filter tap, over one slice of the pixel range;
//               wave w owns output channels 16 w .. 16 w + 15 and all 64 input channels (4 accumulator fragments)
//   step        32 pixels: dY tile [32][64] and the tap-shifted X tile [32][64] (zeros outside the image) through registers
//               into LDS, two transposed reads per operand fragment, 4 MFMAs per wave
//   reduction   fp32 partial tiles per pixel slice, summed in slice order by a second kernel (deterministic; no atomics)
// First correct version: single-buffered, unswizzled LDS rows.  Its place in the plan and what comes next: DESIGN.md §7.
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TK = 64, TCH = 64, TPX = 32;

struct WgP {
  const bf16_t* x; const bf16_t* dy; float* partial;
  int N, H, W, C, K, R, S, pad;
  long x_bs, dy_bs; int ldx, ldy;
  long P, per;        // pixels, pixels per slice (multiple of TPX)
  int nsplit, ktiles, ctiles;
};

__device__ __forceinline__ s16x4 tr_read(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__global__ __launch_bounds__(256) void wgrad_kernel(const WgP p) {
  __shared__ __attribute__((aligned(16))) bf16_t sdy[TPX * TK];
  __shared__ __attribute__((aligned(16))) bf16_t sx[TPX * TCH];
  int b = blockIdx.x;
  const int ct = b % p.ctiles; b /= p.ctiles;
  const int kt = b % p.ktiles; b /= p.ktiles;
  const int taps = p.R * p.S;
  const int tap = b % taps;
  const int split = b / taps;
  const int r = tap / p.S, s = tap - r * p.S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = tid >> 3, chunk = tid & 7;       // staging: tile row (pixel of the step), 16-byte chunk (8 channels)
  const int HW = p.H * p.W;
  const long p0 = (long)split * p.per, p1 = min(p.P, p0 + p.per);
  const int kch = kt * TK + chunk * 8, cch = ct * TCH + chunk * 8;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const bf16_t* a_lo = sdy + (8 * g + q) * TK + 16 * wave + 4 * pp;      // rows 8g .. 8g+3 of the step (operand k index)
  const bf16_t* b_lo = sx + (8 * g + q) * TCH + 4 * pp;

  f32x4 acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (long pb = p0; pb < p1; pb += TPX) {
    const long pix = pb + row;
    uint4 vdy = {0u, 0u, 0u, 0u}, vx = {0u, 0u, 0u, 0u};
    if (pix < p1) {
      const int n = (int)(pix / HW), rem = (int)(pix - (long)n * HW);
      const int y = rem / p.W, xx = rem - y * p.W;
      if (kch < p.K) vdy = *reinterpret_cast<const uint4*>(p.dy + (long)n * p.dy_bs + (long)rem * p.ldy + kch);
      const int iy = y + r - p.pad, ix = xx + s - p.pad;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && cch < p.C)
        vx = *reinterpret_cast<const uint4*>(p.x + (long)n * p.x_bs + ((long)iy * p.W + ix) * p.ldx + cch);
    }
    __syncthreads();                                   // the previous step's fragment reads are done
    *reinterpret_cast<uint4*>(sdy + row * TK + chunk * 8) = vdy;
    *reinterpret_cast<uint4*>(sx + row * TCH + chunk * 8) = vx;
    __syncthreads();
    // every lane takes part in the transposed reads (EXEC must be full): no divergence from here to the MFMAs
    const s16x4 a0 = tr_read(a_lo), a1 = tr_read(a_lo + 4 * TK);
    const s16x8 A = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const s16x4 b0 = tr_read(b_lo + 16 * f), b1 = tr_read(b_lo + 16 * f + 4 * TCH);
      const s16x8 B = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[f], 0, 0, 0);
    }
  }
  // lane: output channels kt*64 + 16 w + 4 (lane / 16) + e, input channel ct*64 + 16 f + lane % 16
  const long RSC = (long)taps * p.C;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int c = ct * TCH + 16 * f + (lane & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = kt * TK + 16 * wave + 4 * (lane >> 4) + e;
      if (k < p.K && c < p.C) p.partial[((long)split * p.K + k) * RSC + (long)tap * p.C + c] = acc[f][e];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, long n, int nsplit, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? dw[i] : 0.f;
  for (int k = 0; k < nsplit; ++k) s += partial[(long)k * n + i];
  dw[i] = s;
}

int pick_split(int K, int C, int taps, long P) {
  const long tiles = (long)((K + TK - 1) / TK) * ((C + TCH - 1) / TCH) * taps;
  long ns = (2048 + tiles - 1) / tiles;                        // ~8 workgroups per CU over the whole launch
  const long maxs = (P + 8 * TPX - 1) / (8 * TPX);             // at least 8 steps per slice
  if (ns > maxs) ns = maxs;
  return (int)(ns < 1 ? 1 : ns);
}

}  // namespace

extern "C" int64_t mtbt_conv_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0) return 0;
  return (int64_t)pick_split(K, C, R * S, (long)N * H * W) * K * R * S * C * (int64_t)sizeof(float);
}

extern "C" int mtbt_conv_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int pad,
                               int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride, int dtype,
                               int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !dy || !dw || !workspace || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0) return MTBT_EINVAL;
  if (dtype != MTBT_BF16 || C % 8 || K % 8 || x_pixel_stride % 8 || dy_pixel_stride % 8 || x_batch_stride % 8 || dy_batch_stride % 8) return MTBT_EINVAL;
  if (2 * pad != R - 1 || 2 * pad != S - 1) return MTBT_EINVAL;          // stride-1 "same" convolutions
  if (!aligned16(x) || !aligned16(dy) || !aligned16(workspace)) return MTBT_EALIGN;
  if (workspace_bytes < mtbt_conv_wgrad_workspace_bytes(N, H, W, C, K, R, S)) return MTBT_EWORKSPACE;
  WgP p;
  p.x = reinterpret_cast<const bf16_t*>(x); p.dy = reinterpret_cast<const bf16_t*>(dy); p.partial = reinterpret_cast<float*>(workspace);
  p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.pad = pad;
  p.x_bs = x_batch_stride; p.dy_bs = dy_batch_stride; p.ldx = x_pixel_stride; p.ldy = dy_pixel_stride;
  p.P = (long)N * H * W;
  p.nsplit = pick_split(K, C, R * S, p.P);
  p.per = ((p.P + p.nsplit - 1) / p.nsplit + TPX - 1) / TPX * TPX;
  p.ktiles = (K + TK - 1) / TK; p.ctiles = (C + TCH - 1) / TCH;
  const long blocks = (long)p.nsplit * R * S * p.ktiles * p.ctiles;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  const long n = (long)K * R * S * C;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.partial, dw, n, p.nsplit, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
