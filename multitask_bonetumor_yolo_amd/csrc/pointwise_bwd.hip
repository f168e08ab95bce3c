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
    const int ch = (int)((unsigned)i % (unsigned)chunks);
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

// ------------------------------------------------------------------------------------------------------------------------------
// Depthwise weight gradient from an LDS-resident halo tile (the forward kernel's data flow, reduction instead of convolution):
//   dW[ky][kx][c] = sum over pixels (y, x) of dy[y][x][c] * X[y + ky - P][x + kx - P][c]
// Workgroup = 4 waves, one 128-channel chunk (lane = channel pair), PERSISTENT over 8 x 8-pixel tiles; wave w owns tile rows 2w, 2w+1.
// Per tile the (8+KS-1)^2 input halo and the 64 dy pixels of the chunk are staged in LDS (16-byte loads); a halo row segment of
// 8 + KS - 1 pairs is read once and feeds all KS horizontal taps of 8 output pixels (KS * 8 packed FMAs per KS + 7 LDS reads).
// The KS*KS tap accumulators (fp32 pairs) stay in registers across tiles; one partial row per WAVE is written at the end and the
// shared second level (channel_sum_final) adds the rows in a fixed order.
// ------------------------------------------------------------------------------------------------------------------------------
typedef float f32p __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ f32p ld_pair(const char* p);
template <> __device__ __forceinline__ f32p ld_pair<float>(const char* p) { return *reinterpret_cast<const f32p*>(p); }
template <> __device__ __forceinline__ f32p ld_pair<bf16_t>(const char* p) {
  const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
  return f32p{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}

// EIGHT waves (round 3): waves 0-3 accumulate the filter rows ky < NKY, waves 4-7 the rest, each group over the same tile rows (wave & 3 owns
// rows 2w, 2w + 1).  With all KS x KS accumulator pairs per wave (98 registers at 7 x 7, plus the one-tile-ahead staging registers) the
// kernel ran ONE wave per SIMD -- four per CU -- and its FMAs, LDS reads and load waits had nothing to overlap with.
template <typename T, int KS>
__global__ __launch_bounds__(512, 2) void dw_wgrad_tile_kernel(const T* __restrict__ dy, const T* __restrict__ x, int N, int H, int W, int C,
                                                            float* __restrict__ partial /* [slots][(KS*KS + 1) * C]: taps, then sum_p dy */) {
  constexpr int PAD = KS / 2, TS = 8, IW = TS + KS - 1, ES = (int)sizeof(T), PIXB = 128 * ES, EPC = 16 / ES, PARTS = PIXB / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xt = smem;                       // [IW*IW][128] T
  char* dt = smem + IW * IW * PIXB;      // [64][128] T
  constexpr int NKY = (KS + 1) / 2;                      // filter rows per wave group
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, tg = tid >> 8;
  const int ky0 = tg * NKY;
  const int chunks = (C + 127) / 128;
  const int chunk = blockIdx.x % chunks, slot = blockIdx.x / chunks, nslots = gridDim.x / chunks;
  const int cb = chunk * 128, cc = min(128, C - cb);
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const long ntiles = (long)N * tiles_y * tiles_x;
  const bool active = lane * 2 < cc;
  f32p acc[NKY * KS], accb = f32p{0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NKY * KS; ++t) acc[t] = f32p{0.f, 0.f};
  // Staging through REGISTERS, one tile ahead: all of a thread's 16-byte pieces of tile t + 1 (13 of the halo + 4 of dy for 7 x 7 bf16) are
  // requested before the FMAs of tile t and written to LDS behind them.  (Round 3: the first version loaded and stored piece by piece in
  // one loop -- every piece waited for its own round trip, 12 dependent latencies per tile: 24 us per tile where the FMAs need 1.3, 610 us
  // for a stage-0 launch that moves 315 MB.)
  constexpr int NHX = (IW * IW * PARTS + 511) / 512, NDY = (TS * TS * PARTS + 511) / 512;
  uint4 hx[NHX], hd[NDY];
  auto fetch = [&](long tl) {
    const unsigned utl = (unsigned)tl;                     // (32-bit tile arithmetic: ntiles < 2^31)
    const int n = (int)(utl / (unsigned)(tiles_x * tiles_y)), trem = (int)(utl - (unsigned)n * (unsigned)(tiles_x * tiles_y));
    const int ty = (int)((unsigned)trem / (unsigned)tiles_x), tx = trem - ty * tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
#pragma unroll
    for (int k = 0; k < NHX; ++k) {                         // halo: zeros outside the image / past the chunk's channels
      const int it = tid + k * 512;
      const int pix = it / PARTS, part = it - pix * PARTS;
      const int r = pix / IW, c = pix - r * IW;
      const int iy = y0 + r - PAD, ix = x0 + c - PAD;
      hx[k] = uint4{0u, 0u, 0u, 0u};
      if (it < IW * IW * PARTS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && part * EPC < cc)
        hx[k] = *reinterpret_cast<const uint4*>(x + (((long)n * H + iy) * W + ix) * C + cb + part * EPC);
    }
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      const int it = tid + k * 512;
      const int pix = it / PARTS, part = it - pix * PARTS;
      const int oy = y0 + pix / TS, ox = x0 + pix % TS;
      hd[k] = uint4{0u, 0u, 0u, 0u};
      if (it < TS * TS * PARTS && oy < H && ox < W && part * EPC < cc)
        hd[k] = *reinterpret_cast<const uint4*>(dy + (((long)n * H + oy) * W + ox) * C + cb + part * EPC);
    }
  };
  if (slot < ntiles) fetch(slot);
  for (long tl = slot; tl < ntiles; tl += nslots) {
    __syncthreads();                                       // the previous tile's readers are done
#pragma unroll
    for (int k = 0; k < NHX; ++k) {
      const int it = tid + k * 512;
      if (it < IW * IW * PARTS) *reinterpret_cast<uint4*>(xt + (it / PARTS) * PIXB + (it % PARTS) * 16) = hx[k];
    }
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      const int it = tid + k * 512;
      if (it < TS * TS * PARTS) *reinterpret_cast<uint4*>(dt + (it / PARTS) * PIXB + (it % PARTS) * 16) = hd[k];
    }
    __syncthreads();
    if (tl + nslots < ntiles) fetch(tl + nslots);          // in flight during this tile's FMAs
    if (active) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = 2 * wave + a;
        f32p g[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) { g[i] = ld_pair<T>(dt + (row * TS + i) * PIXB + lane * 2 * ES); accb += g[i]; }   // (bias gradient: sum_p dy; group 0's copy is used)
#pragma unroll
        for (int kk = 0; kk < NKY; ++kk) {
          if (ky0 + kk >= KS) break;                        // (wave-uniform: the second group has KS - NKY rows)
          f32p in[IW];
#pragma unroll
          for (int j = 0; j < IW; ++j) in[j] = ld_pair<T>(xt + ((row + ky0 + kk) * IW + j) * PIXB + lane * 2 * ES);
#pragma unroll
          for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int i = 0; i < TS; ++i) acc[kk * KS + kx] = __builtin_elementwise_fma(g[i], in[i + kx], acc[kk * KS + kx]);
        }
      }
    }
  }
  // fold the four waves of each group through LDS (waves 2, 3 -> 0, 1, then 1 -> 0; a fixed order) so that ONE partial row per
  // workgroup goes out: the second level reads a quarter of the rows
  f32p* red = reinterpret_cast<f32p*>(smem) + tg * (2 * (NKY * KS + 1) * 64);   // [group][2][NKY*KS + 1][64] pairs <= 59 KiB, inside the tile region
  constexpr int KK1 = NKY * KS + 1;
  __syncthreads();                                        // the last tile's readers are done
  if (wave >= 2) {
#pragma unroll
    for (int t = 0; t < NKY * KS; ++t) red[((wave - 2) * KK1 + t) * 64 + lane] = acc[t];
    red[((wave - 2) * KK1 + NKY * KS) * 64 + lane] = accb;
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int t = 0; t < NKY * KS; ++t) acc[t] += red[(wave * KK1 + t) * 64 + lane];
    accb += red[(wave * KK1 + NKY * KS) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int t = 0; t < NKY * KS; ++t) red[t * 64 + lane] = acc[t];
    red[NKY * KS * 64 + lane] = accb;
  }
  __syncthreads();
  // partial row `slot`: the workgroups of one slot (one per chunk) write disjoint column ranges of the same row, so every row is
  // complete without any zero fill
  float* dst = partial + (long)slot * (KS * KS + 1) * C;
  if (wave == 0 && active) {
#pragma unroll
    for (int kk = 0; kk < NKY; ++kk) {
      if (ky0 + kk >= KS) break;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx)
        *reinterpret_cast<f32p*>(dst + (long)((ky0 + kk) * KS + kx) * C + cb + lane * 2) = acc[kk * KS + kx] + red[(kk * KS + kx) * 64 + lane];
    }
    if (tg == 0) *reinterpret_cast<f32p*>(dst + (long)KS * KS * C + cb + lane * 2) = accb + red[NKY * KS * 64 + lane];
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

static int dw_wgrad_blocks(int N, int H, int W, int C) {
  const int chunks = (C + 127) / 128;
  const long tiles = (long)N * ((H + 7) / 8) * ((W + 7) / 8);
  long slots = 512 / chunks;                        // ~2 workgroups per CU over all chunks
  if (slots < 1) slots = 1;
  if (slots > tiles) slots = tiles;
  return (int)(slots * chunks);
}

extern "C" int64_t mtbt_dwconv_wgrad_workspace_bytes(int N, int H, int W, int C, int ksize) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0) return 0;
  return (int64_t)(dw_wgrad_blocks(N, H, W, C) / ((C + 127) / 128)) * (ksize * ksize + 1) * C * (int64_t)sizeof(float);
}

static int dw_wgrad_entry(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int ksize, int dtype, int accumulate,
                          void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !dy || !dw || !workspace || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (ksize != 3 && ksize != 7)) return MTBT_EINVAL;
  if (dtype != MTBT_BF16 && dtype != MTBT_F32) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(workspace)) return MTBT_EALIGN;
  if (workspace_bytes < mtbt_dwconv_wgrad_workspace_bytes(N, H, W, C, ksize)) return MTBT_EWORKSPACE;
  const int blocks = dw_wgrad_blocks(N, H, W, C);
  const int chunks = (C + 127) / 128;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  const int n = ksize * ksize * C;
  const int es = dtype == MTBT_F32 ? 4 : 2;
  const int iw = 8 + ksize - 1;
  const int lds = (iw * iw + 64) * 128 * es;
#define DWT(T, KSV)                                                                                                      \
  do {                                                                                                                  \
    if (int rc = mtbt_allow_lds(dw_wgrad_tile_kernel<T, KSV>, lds)) return rc;                                           \
    hipLaunchKernelGGL((dw_wgrad_tile_kernel<T, KSV>), dim3((unsigned)blocks), dim3(512), lds, s, (const T*)dy, (const T*)x, N, H, W, C, partial); \
  } while (0)
  if (dtype == MTBT_BF16) { if (ksize == 7) DWT(bf16_t, 7); else DWT(bf16_t, 3); }
  else { if (ksize == 7) DWT(float, 7); else DWT(float, 3); }
#undef DWT
  const int rows = blocks / chunks, pitch = n + C;
  hipLaunchKernelGGL(channel_sum_final_pitch, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, partial, rows, pitch, 0, n, dw, accumulate);
  if (dbias) hipLaunchKernelGGL(channel_sum_final_pitch, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, partial, rows, pitch, n, C, dbias, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_dwconv_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int ksize, int dtype, int accumulate,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  return dw_wgrad_entry(x, dy, dw, nullptr, N, H, W, C, ksize, dtype, accumulate, workspace, workspace_bytes, stream);
}

// The same plus the bias gradient dbias[c] (+)= sum_p dy[p][c] (`conv_dw.bias.grad`), from the dy tile the kernel stages anyway.
extern "C" int mtbt_dwconv_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int ksize, int dtype,
                                      int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!dbias) return MTBT_EINVAL;
  return dw_wgrad_entry(x, dy, dw, dbias, N, H, W, C, ksize, dtype, accumulate, workspace, workspace_bytes, stream);
}
