// Bandwidth-bound kernels of the forward path (NHWC, 16-byte accesses, fp32 arithmetic):
// ConvNeXt stem (4x4/4 conv + LayerNorm2d), LayerNorm over channels, BiFPN weighted fusion with resampling, GAP + Linear, casts.
#include "common.h"
#include "rowreduce.h"
#include "fuse_fetch.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Stem: y[n, oy, ox, :] = LN( W[Cout][3*4*4] . patch(n, oy, ox) + b )
// Block = Cout*G threads, PIX consecutive output pixels per pass; the 48-float patch of each pixel
// is staged in LDS with coalesced float4 reads of the NCHW image (4 floats = one kernel row).
// ------------------------------------------------------------------------------------------------
template <typename OT, int PIX>
__global__ void stem_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                            const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, OT* __restrict__ y,
                            OT* __restrict__ raw, int N, int H, int W, int Cout, int G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* patch = reinterpret_cast<float*>(smem);            // [PIX][48]
  float* outv = patch + PIX * 48;                            // [PIX][Cout]
  const int Ho = H / 4, Wo = W / 4;
  const long total = (long)N * Ho * Wo;
  const int tid = threadIdx.x;
  const int ch = tid % Cout, grp = tid / Cout;
  float wr[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) wr[i] = w[ch * 48 + i];
  const float bch = bias ? bias[ch] : 0.f;
  const int nwaves = blockDim.x >> 6, wave = tid >> 6, lane = tid & 63;

  for (long base = (long)blockIdx.x * PIX; base < total; base += (long)gridDim.x * PIX) {
    // stage patches: item = (pixel, c, ky) -> one float4
    for (int it = tid; it < PIX * 12; it += blockDim.x) {
      const int pix = it % PIX, cky = it / PIX;
      const long gp = base + pix;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gp < total) {
        const int n = (int)(gp / (Ho * Wo));
        const int rem = (int)(gp - (long)n * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const int c = cky >> 2, ky = cky & 3;
        v = *reinterpret_cast<const float4*>(x + (((long)n * 3 + c) * H + (oy * 4 + ky)) * W + ox * 4);
      }
      *reinterpret_cast<float4*>(patch + pix * 48 + cky * 4) = v;
    }
    __syncthreads();
    for (int pix = grp; pix < PIX; pix += G) {
      float acc = bch;
      const float* pp = patch + pix * 48;
#pragma unroll
      for (int i = 0; i < 48; ++i) acc = fmaf(wr[i], pp[i], acc);
      outv[pix * Cout + ch] = acc;
    }
    __syncthreads();
    // LayerNorm over Cout per pixel: one wave per pixel
    for (int pix = wave; pix < PIX; pix += nwaves) {
      const long gp = base + pix;
      if (gp >= total) break;
      const float* o = outv + pix * Cout;
      float s = 0.f;
      for (int c = lane; c < Cout; c += 64) s += o[c];
      const float mean = wave_sum(s) / Cout;
      float q = 0.f;
      for (int c = lane; c < Cout; c += 64) { const float d = o[c] - mean; q += d * d; }
      const float rstd = rsqrtf(wave_sum(q) / Cout + eps);
      for (int c = lane; c < Cout; c += 64) st_elem<OT>(y + gp * Cout + c, (o[c] - mean) * rstd * lnw[c] + lnb[c]);
      if (raw) for (int c = lane; c < Cout; c += 64) st_elem<OT>(raw + gp * Cout + c, o[c]);   // training: the LayerNorm input
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Stem on the MFMA (bf16 output): the 4x4/4 patchify conv is a [Cout x 48] x [48 x pixels] GEMM.
// A wave owns 16 consecutive output pixels of a row per iteration: its B fragment (8 consecutive k =
// two kernel rows (c, ky), (c, ky+1) x 4 kx) is two float4 reads of the NCHW image, converted to bf16
// in registers -- no LDS at all; the Cout x 64 (zero-padded) weight fragments live in registers for the
// wave's whole grid-stride loop.  LayerNorm2d: a pixel's channels sit in the 4 lanes {n, n+16, n+32, n+48}
// x 4 registers x Cout/16 fragments, so two xor-shuffles finish each reduction.
// ------------------------------------------------------------------------------------------------
template <typename HT> struct StemCvt;   // fp32 -> 16-bit MFMA operand element
template <> struct StemCvt<bf16_t> { typedef bf16x8 vec; static __device__ __forceinline__ __bf16 cv(float v) { return (__bf16)v; } };
template <> struct StemCvt<f16_t> { typedef f16x8 vec; static __device__ __forceinline__ _Float16 cv(float v) { return (_Float16)v; } };
template <typename HT> __device__ __forceinline__ uint32_t pk16(float a, float b);
template <> __device__ __forceinline__ uint32_t pk16<bf16_t>(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }
template <> __device__ __forceinline__ uint32_t pk16<f16_t>(float a, float b) { return pk_h2(a, b); }

template <int FCH, typename HT>  // Cout / 16; bf16_t or f16_t output
__global__ __launch_bounds__(256, 3) void stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ lnw,
                                                        const float* __restrict__ lnb, float eps, HT* __restrict__ y,
                                                        HT* __restrict__ raw, int N, int H, int W) {
  constexpr int Cout = FCH * 16;
  const int lane = threadIdx.x & 63, nq = lane & 15, q = lane >> 4;
  const int Ho = H >> 2, Wo = W >> 2;
  const long groups = (long)N * Ho * (Wo >> 4);
  // weight fragments: A[m = ch][k], k = 32 ks + 8 q + j, zero for k >= 48.  Operands are built as PACKED 16-bit pairs (pk16): assembled element
  // by element as a vector of __bf16, and with no register bound (the compiler hoisted the bias / LayerNorm vectors of all six fragments out of
  // the loop), the kernel needed 216 registers and ran two waves per SIMD -- a quarter of the waves it takes to cover its load latency: 80 us
  // for 157 MB (round 3).  Bounded to three waves per SIMD now (four spill 40 registers).
  uint4 afr[FCH][2];
#pragma unroll
  for (int f = 0; f < FCH; ++f)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int k0 = ks * 32 + q * 8;                     // (a lane's 8 consecutive k of one weight row: two 16-byte loads, or zeros past k = 48)
      float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
      // fragment row f * 16 + nq holds weight row 32 (f / 2) + 8 (nq / 4) + 4 (f % 2) + nq % 4: a lane's accumulators of the fragment pair (2 m, 2 m + 1)
      // are then the 8 consecutive channels 32 m + 8 q .. + 7 -- 16-byte stores (the same row order as conv_epilogue.h PERM / mlp_fused.hip)
      const int wr = 32 * (f >> 1) + 8 * (nq >> 2) + 4 * (f & 1) + (nq & 3);
      if (k0 < 48) { w0 = *reinterpret_cast<const float4*>(w + wr * 48 + k0); w1 = *reinterpret_cast<const float4*>(w + wr * 48 + k0 + 4); }
      afr[f][ks] = uint4{pk16<HT>(w0.x, w0.y), pk16<HT>(w0.z, w0.w), pk16<HT>(w1.x, w1.y), pk16<HT>(w1.z, w1.w)};
    }
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  // the image rows of a group are requested one group AHEAD (the wave's next 16 pixels land while it normalises and stores these)
  float4 pre[2][2];
  auto fetch = [&](long g) {
    const int xg = (int)(g % (Wo >> 4));
    const long ny = g / (Wo >> 4);
    const int oy = (int)(ny % Ho), n = (int)(ny / Ho);
    const int ox = xg * 16 + nq;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kc = ks * 4 + q;  // 8-wide k chunk: channel kc/2, kernel rows 2*(kc&1), +1
      pre[ks][0] = pre[ks][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kc < 6) {
        const float* src = x + (((long)n * 3 + (kc >> 1)) * H + (oy * 4 + (kc & 1) * 2)) * W + ox * 4;
        pre[ks][0] = *reinterpret_cast<const float4*>(src);
        pre[ks][1] = *reinterpret_cast<const float4*>(src + W);
      }
    }
  };
  if (wave_id < groups) fetch(wave_id);
  for (long g = wave_id; g < groups; g += nwaves) {
    const int xg = (int)(g % (Wo >> 4));
    const long ny = g / (Wo >> 4);
    const int oy = (int)(ny % Ho), n = (int)(ny / Ho);
    const int ox = xg * 16 + nq;
    uint4 bfr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float4 r0 = pre[ks][0], r1 = pre[ks][1];
      bfr[ks] = uint4{pk16<HT>(r0.x, r0.y), pk16<HT>(r0.z, r0.w), pk16<HT>(r1.x, r1.y), pk16<HT>(r1.z, r1.w)};
    }
    if (g + nwaves < groups) fetch(g + nwaves);
    // (compiler barrier: without it the bias / LayerNorm vectors of all six fragments -- loop-invariant loads -- are hoisted out of the loop, 72
    //  registers that the allocator then SPILLS to fit four waves per SIMD; read per group they are L1 hits)
    asm volatile("" ::: "memory");
    f32x4 acc[FCH];
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < FCH; ++f) {
      acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[f] = mfma_16x16x32<HT>(afr[f][0], bfr[0], acc[f]);
      acc[f] = mfma_16x16x32<HT>(afr[f][1], bfr[1], acc[f]);
      const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * (f >> 1) + 8 * q + 4 * (f & 1));
      acc[f][0] += bv.x; acc[f][1] += bv.y; acc[f][2] += bv.z; acc[f][3] += bv.w;
      s += (acc[f][0] + acc[f][1]) + (acc[f][2] + acc[f][3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / Cout);
    float v = 0.f;
#pragma unroll
    for (int f = 0; f < FCH; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[f][r] - mean; v += d * d; }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const float rstd = rsqrtf(v * (1.0f / Cout) + eps);
    static_assert(FCH % 2 == 0, "fragment pairs");
    if (raw) {   // training: the LayerNorm input
      HT* ro = raw + (((long)n * Ho + oy) * Wo + ox) * Cout + q * 8;
#pragma unroll
      for (int m = 0; m < FCH / 2; ++m)
        *reinterpret_cast<u32x4*>(ro + m * 32) = u32x4{pk16<HT>(acc[2 * m][0], acc[2 * m][1]), pk16<HT>(acc[2 * m][2], acc[2 * m][3]),
                                                        pk16<HT>(acc[2 * m + 1][0], acc[2 * m + 1][1]), pk16<HT>(acc[2 * m + 1][2], acc[2 * m + 1][3])};
    }
    HT* yo = y + (((long)n * Ho + oy) * Wo + ox) * Cout + q * 8;
#pragma unroll
    for (int m = 0; m < FCH / 2; ++m) {
      const float* gwp = lnw + m * 32 + q * 8;
      const float* gbp = lnb + m * 32 + q * 8;
      const float4 w0 = *reinterpret_cast<const float4*>(gwp), w1 = *reinterpret_cast<const float4*>(gwp + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(gbp), b1 = *reinterpret_cast<const float4*>(gbp + 4);
      const f32x4 a0 = acc[2 * m], a1 = acc[2 * m + 1];
      u32x4 o;
      o.x = pk16<HT>((a0[0] - mean) * rstd * w0.x + b0.x, (a0[1] - mean) * rstd * w0.y + b0.y);
      o.y = pk16<HT>((a0[2] - mean) * rstd * w0.z + b0.z, (a0[3] - mean) * rstd * w0.w + b0.w);
      o.z = pk16<HT>((a1[0] - mean) * rstd * w1.x + b1.x, (a1[1] - mean) * rstd * w1.y + b1.y);
      o.w = pk16<HT>((a1[2] - mean) * rstd * w1.z + b1.z, (a1[3] - mean) * rstd * w1.w + b1.w);
      *reinterpret_cast<u32x4*>(yo + m * 32) = o;
    }
  }
}

// LayerNorm over C, values held in registers between the two passes.  A pixel is handled by a group of LP lanes (LP = 16,
// 32 or 64: the power of two covering C/8 eight-channel pieces), so a wave normalises 64/LP pixels at once -- with one
// wave per pixel the 96-channel downsample norm used 12 of 64 lanes (104 us for 157 MB).  MAXV = pieces per lane.
template <int LP> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int MAXV, int LP>
__global__ void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                 float eps, T* __restrict__ y, long pixels, int C) {
  constexpr int PPW = 64 / LP;                       // pixels per wave
  const int lane = threadIdx.x & 63, gl = lane % LP;
  const long pix = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PPW + lane / LP;
  const bool live = pix < pixels;                    // (whole groups: the shuffles below stay inside a group)
  const int CH8 = C >> 3;
  float v[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
    if (live && ch < CH8) {
      ld8<T>(x + pix * C + ch * 8, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = group_sum<LP>(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
    if (live && ch < CH8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(group_sum<LP>(q) / C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
    if (live && ch < CH8) {
      float gw[8], gb[8], o[8];
      ld8<float>(w + ch * 8, gw);
      ld8<float>(b + ch * 8, gb);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * gw[e] + gb[e];
      st8<T>(y + pix * C + ch * 8, o);
    }
  }
}

// LayerNorm backward over the channels of a pixel (timm ConvNeXt `norm` / downsample LayerNorm2d), same lane-group layout as
// the forward: recompute mean / rstd from x, g = dy * gamma,
//     dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  xhat = (x - mean) * rstd
// and optionally store xhat (for d gamma = sum_pixels dy * xhat through mtbt_channel_sum; d beta = sum_pixels dy).
template <typename T, int MAXV, int LP>
__global__ void layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ w, float eps,
                                     T* __restrict__ dx, T* __restrict__ xhat, long pixels, int C, int accumulate) {
  constexpr int PPW = 64 / LP;
  const int lane = threadIdx.x & 63, gl = lane % LP;
  const long pix = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PPW + lane / LP;
  const bool live = pix < pixels;
  const int CH8 = C >> 3;
  float v[MAXV][8], g[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
    if (live && ch < CH8) {
      ld8<T>(x + pix * C + ch * 8, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = group_sum<LP>(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
    if (live && ch < CH8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(group_sum<LP>(q) / C + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
    if (live && ch < CH8) {
      float gw[8];
      ld8<float>(w + ch * 8, gw);
      ld8<T>(dy + pix * C + ch * 8, g[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (v[i][e] - mean) * rstd;       // xhat
        g[i][e] *= gw[e];
        s1 += g[i][e];
        s2 += g[i][e] * v[i][e];
      }
    }
  }
  const float m1 = group_sum<LP>(s1) / C, m2 = group_sum<LP>(s2) / C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
    if (live && ch < CH8) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - m1 - v[i][e] * m2);
      if (accumulate) {   // dx already holds another consumer's gradient (a ConvNeXt stage output feeds the next stage AND its adaptor)
        float old[8];
        ld8<T>(dx + pix * C + ch * 8, old);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += old[e];
      }
      st8<T>(dx + pix * C + ch * 8, o);
      if (xhat) st8<T>(xhat + pix * C + ch * 8, v[i]);
    }
  }
}

// LayerNorm backward WITH the parameter gradients: a wave walks LNB_ITERS consecutive groups of its 64/LP pixels and keeps, per lane,
// the running sums of dy * xhat (d gamma) and dy (d beta) of its channels; at the end the pixel groups of the wave are folded by
// shuffles and ONE partial row [d beta | d gamma] per wave goes to the workspace (second level: channel_sum_final_pitch).  Saves the
// xhat tensor (a write and two reads) and two channel-sum passes over dy per LayerNorm.
constexpr int LNB_ITERS = 64;   // at most; fewer when the tensor is small (keep >= ~4096 waves in flight)
static inline int lnb_iters(long pixels, int ppw) {
  long it = pixels / ((long)ppw * 4096);
  return (int)(it < 1 ? 1 : (it > LNB_ITERS ? LNB_ITERS : it));
}

template <typename T, int MAXV, int LP>
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ w, float eps,
                                                                   T* __restrict__ dx, long pixels, int C, int accumulate, float* __restrict__ partial,
                                                                   int iters) {
  constexpr int PPW = 64 / LP;
  const int lane = threadIdx.x & 63, gl = lane % LP;
  const long wave_id = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int CH8 = C >> 3;
  float sg[MAXV][8], sb[MAXV][8], gw[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sg[i][e] = 0.f; sb[i][e] = 0.f; gw[i][e] = 0.f; }
    if (ch < CH8) ld8<float>(w + ch * 8, gw[i]);
  }
  for (int it = 0; it < iters; ++it) {
    const long pix = (wave_id * iters + it) * PPW + lane / LP;
    const bool live = pix < pixels;                     // (whole groups)
    float v[MAXV][8], g[MAXV][8], d[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int ch = gl + i * LP;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = 0.f; d[i][e] = 0.f; }
      if (live && ch < CH8) {
        ld8<T>(x + pix * C + ch * 8, v[i]);
        ld8<T>(dy + pix * C + ch * 8, d[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mean = group_sum<LP>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int ch = gl + i * LP;
      if (live && ch < CH8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = v[i][e] - mean; q += t * t; }
      }
    }
    const float rstd = rsqrtf(group_sum<LP>(q) / C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int ch = gl + i * LP;
      if (live && ch < CH8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[i][e] = (v[i][e] - mean) * rstd;       // xhat
          sg[i][e] += d[i][e] * v[i][e];
          sb[i][e] += d[i][e];
          g[i][e] = d[i][e] * gw[i][e];
          s1 += g[i][e];
          s2 += g[i][e] * v[i][e];
        }
      }
    }
    const float m1 = group_sum<LP>(s1) / C, m2 = group_sum<LP>(s2) / C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int ch = gl + i * LP;
      if (live && ch < CH8) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - m1 - v[i][e] * m2);
        if (accumulate) {
          float old[8];
          ld8<T>(dx + pix * C + ch * 8, old);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += old[e];
        }
        st8<T>(dx + pix * C + ch * 8, o);
      }
    }
  }
  // fold the wave's 64/LP pixel groups (same channels in lanes gl, gl + LP, ...), then the workgroup's four waves through LDS in
  // wave order, and write ONE row per workgroup (the second level reads a quarter of the rows)
  extern __shared__ __attribute__((aligned(16))) float lnb_red[];   // [4][2 * C]
  float* row = lnb_red + (threadIdx.x >> 6) * 2 * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = gl + i * LP;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = sb[i][e], b = sg[i][e];
#pragma unroll
      for (int o = LP; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (lane < LP && ch < CH8) { row[ch * 8 + e] = a; row[C + ch * 8 + e] = b; }
    }
  }
  __syncthreads();
  float* out = partial + (long)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256) out[c] = ((lnb_red[c] + lnb_red[2 * C + c]) + lnb_red[4 * C + c]) + lnb_red[6 * C + c];
}

// ------------------------------------------------------------------------------------------------
// BiFPN fusion node: thread = (output pixel, 8-channel chunk).
// Bilinear x2 (align_corners=False): src = (dst+.5)/2-.5 clamped at 0; i0=floor, i1=min(i0+1,last).
// Same association as torch: l0h*(l0w*v00+l1w*v01) + l1h*(l0w*v10+l1w*v11).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void fuse_kernel(const FuseP p) {
  const int CH8 = p.C >> 3;
  const long total = (long)p.N * p.H * p.W * CH8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int chunk, x, y;
    long pix, ny, nn;
    divmod_u32(idx, CH8, pix, chunk);
    divmod_u32(pix, p.W, ny, x);
    divmod_u32(ny, p.H, nn, y);
    const int n = (int)nn;
    float acc[8], t[8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i >= p.n_in) break;
      fuse_fetch<T>(reinterpret_cast<const T*>(p.x[i]), p.resample[i], n, y, x, p.H, p.W, p.C, chunk * 8, t);
      const float wi = p.wgt_dev ? p.wgt_dev[i] : p.wgt[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float term = p.bug ? (wi + t[e]) : (wi * t[e]);
        acc[e] = (i == 0) ? term : acc[e] + term;
      }
    }
    st8<T>(reinterpret_cast<T*>(p.y) + pix * p.C + chunk * 8, acc);
  }
}

// GAP over HW then Linear(C, nout); one block per image.
template <typename T>
__global__ void gap_fc_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                              float* __restrict__ y, int HW, int C, int nout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = reinterpret_cast<float*>(smem);  // [G][C]
  float* mean = part + (blockDim.x / (C >> 3)) * C;
  const int CH8 = C >> 3, G = blockDim.x / CH8;
  const int tid = threadIdx.x, n = blockIdx.x;
  const int chunk = tid % CH8, g = tid / CH8;
  if (g < G) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
    for (int p = g; p < HW; p += G) {
      ld8<T>(x + ((long)n * HW + p) * C + chunk * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[g * C + chunk * 8 + e] = s[e];
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < G; ++k) s += part[k * C + c];
    mean[c] = s / HW;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
  for (int o = wave; o < nout; o += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += mean[c] * w[o * C + c];
    s = wave_sum(s);
    if (lane == 0) y[n * nout + o] = s + (b ? b[o] : 0.f);
  }
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    st_elem<D>(d + i, ld_elem<S>(s + i));
}

inline unsigned grid_for(long work, int block) {
  long g = (work + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

static int stem_entry(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float ln_eps, void* y, void* raw,
                      int N, int H, int W, int Cout, int out_dtype, void* stream) {
  if (!x || !w || !ln_w || !ln_b || !y || N <= 0 || H % 4 || W % 4 || H <= 0 || W <= 0) return MTBT_EINVAL;
  if (Cout <= 0 || Cout > 512 || Cout % 32) return MTBT_EINVAL;
  if (!aligned16(x)) return MTBT_EALIGN;
  constexpr int PIX = 32;
  const int G = Cout <= 128 ? 2 : 1;
  const int threads = Cout * G;
  const long total = (long)N * (H / 4) * (W / 4);
  const unsigned blocks = (unsigned)((total + PIX - 1) / PIX > 4096 ? 4096 : (total + PIX - 1) / PIX);
  const size_t lds = (size_t)PIX * (48 + Cout) * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((out_dtype == MTBT_BF16 || out_dtype == MTBT_F16) && Cout == 96 && (W / 4) % 16 == 0 && bias && aligned16(w) && aligned16(bias) && aligned16(ln_w) &&
      aligned16(ln_b) && aligned16(y) && aligned16(raw)) {
    const long groups = (long)N * (H / 4) * (W / 64);
    const unsigned nb = (unsigned)((groups + 3) / 4 > 2048 ? 2048 : (groups + 3) / 4);
    if (out_dtype == MTBT_BF16)
      hipLaunchKernelGGL((stem_mfma_kernel<6, bf16_t>), dim3(nb), dim3(256), 0, s, x, w, bias, ln_w, ln_b, ln_eps, (bf16_t*)y, (bf16_t*)raw, N, H, W);
    else
      hipLaunchKernelGGL((stem_mfma_kernel<6, f16_t>), dim3(nb), dim3(256), 0, s, x, w, bias, ln_w, ln_b, ln_eps, (f16_t*)y, (f16_t*)raw, N, H, W);
    MTBT_LAUNCH_CHECK();
    return MTBT_OK;
  }
  if (out_dtype == MTBT_F32)
    hipLaunchKernelGGL((stem_kernel<float, PIX>), dim3(blocks), dim3(threads), lds, s, x, w, bias, ln_w, ln_b, ln_eps, (float*)y, (float*)raw, N, H, W, Cout, G);
  else if (out_dtype == MTBT_BF16)
    hipLaunchKernelGGL((stem_kernel<bf16_t, PIX>), dim3(blocks), dim3(threads), lds, s, x, w, bias, ln_w, ln_b, ln_eps, (bf16_t*)y, (bf16_t*)raw, N, H, W, Cout, G);
  else if (out_dtype == MTBT_F16)
    hipLaunchKernelGGL((stem_kernel<f16_t, PIX>), dim3(blocks), dim3(threads), lds, s, x, w, bias, ln_w, ln_b, ln_eps, (f16_t*)y, (f16_t*)raw, N, H, W, Cout, G);
  else return MTBT_EINVAL;
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_stem_conv4x4_ln(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                                    float ln_eps, void* y, int N, int H, int W, int Cout, int out_dtype, void* stream) {
  return stem_entry(x, w, bias, ln_w, ln_b, ln_eps, y, nullptr, N, H, W, Cout, out_dtype, stream);
}

// Training variant: `raw` [N,H/4,W/4,Cout] (out_dtype) also receives the LayerNorm2d INPUT (conv + bias).
extern "C" int mtbt_stem_conv4x4_ln_train(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                                          float ln_eps, void* y, void* raw, int N, int H, int W, int Cout, int out_dtype, void* stream) {
  if (!raw) return MTBT_EINVAL;
  return stem_entry(x, w, bias, ln_w, ln_b, ln_eps, y, raw, N, H, W, Cout, out_dtype, stream);
}

extern "C" int mtbt_layernorm_nhwc(const void* x, const float* w, const float* b, float eps, void* y, int64_t pixels,
                                   int C, int dtype, void* stream) {
  if (!x || !w || !b || !y || pixels <= 0 || C <= 0 || C % 8 || C > 2048) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(w) || !aligned16(b)) return MTBT_EALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int CH8 = C / 8;
  const int LP = CH8 <= 16 ? 16 : (CH8 <= 32 ? 32 : 64);   // lanes per pixel
  const long blocks = (pixels + 4 * (64 / LP) - 1) / (4 * (64 / LP));
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
#define LN_LAUNCH(T, MAXV, LPV) \
  hipLaunchKernelGGL((layernorm_kernel<T, MAXV, LPV>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)x, w, b, eps, (T*)y, (long)pixels, C)
#define LN_BY_C(T) \
  do { if (LP == 16) LN_LAUNCH(T, 1, 16); else if (LP == 32) LN_LAUNCH(T, 1, 32); else if (CH8 <= 64) LN_LAUNCH(T, 1, 64); \
       else if (CH8 <= 128) LN_LAUNCH(T, 2, 64); else LN_LAUNCH(T, 4, 64); } while (0)
  if (dtype == MTBT_F32) LN_BY_C(float);
  else if (dtype == MTBT_BF16) LN_BY_C(bf16_t);
  else if (dtype == MTBT_F16) LN_BY_C(f16_t);
  else return MTBT_EINVAL;
#undef LN_BY_C
#undef LN_LAUNCH
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_layernorm_backward_nhwc(const void* x, const void* dy, const float* w, float eps, void* dx, void* xhat, int64_t pixels, int C,
                                            int dtype, int accumulate, void* stream) {
  if (!x || !dy || !w || !dx || pixels <= 0 || C <= 0 || C % 8 || C > 2048) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || (xhat && !aligned16(xhat))) return MTBT_EALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int CH8 = C / 8;
  const int LP = CH8 <= 16 ? 16 : (CH8 <= 32 ? 32 : 64);
  const long blocks = (pixels + 4 * (64 / LP) - 1) / (4 * (64 / LP));
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
#define LNB_LAUNCH(T, MAXV, LPV) \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, MAXV, LPV>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)x, (const T*)dy, w, eps, (T*)dx, (T*)xhat, \
                     (long)pixels, C, accumulate)
#define LNB_BY_C(T) \
  do { if (LP == 16) LNB_LAUNCH(T, 1, 16); else if (LP == 32) LNB_LAUNCH(T, 1, 32); else if (CH8 <= 64) LNB_LAUNCH(T, 1, 64); \
       else if (CH8 <= 128) LNB_LAUNCH(T, 2, 64); else LNB_LAUNCH(T, 4, 64); } while (0)
  if (dtype == MTBT_F32) LNB_BY_C(float);
  else if (dtype == MTBT_BF16) LNB_BY_C(bf16_t);
  else return MTBT_EINVAL;
#undef LNB_BY_C
#undef LNB_LAUNCH
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_bifpn_fuse(const mtbt_fuse_args* a, void* stream) {
  if (!a || !a->y || a->n_in < 1 || a->n_in > 3 || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || a->C % 8) return MTBT_EINVAL;
  FuseP p;
  for (int i = 0; i < 3; ++i) {
    p.x[i] = i < a->n_in ? a->x[i] : nullptr;
    p.wgt[i] = a->wgt[i];
    p.resample[i] = a->resample[i];
    if (i < a->n_in) {
      if (!a->x[i] || a->resample[i] < 0 || a->resample[i] > 4) return MTBT_EINVAL;
      if ((a->resample[i] == 1 || a->resample[i] == 3) && ((a->H & 1) || (a->W & 1))) return MTBT_EINVAL;
      if (!aligned16(a->x[i])) return MTBT_EALIGN;
    }
  }
  if (!aligned16(a->y)) return MTBT_EALIGN;
  p.wgt_dev = a->wgt_dev;
  p.n_in = a->n_in; p.y = a->y; p.N = a->N; p.H = a->H; p.W = a->W; p.C = a->C; p.bug = a->add_weight_bug;
  const long total = (long)a->N * a->H * a->W * (a->C / 8);
  if (total > 0xffffffffL) return MTBT_EINVAL;          // (32-bit piece indices in the kernel)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == MTBT_F32) hipLaunchKernelGGL((fuse_kernel<float>), dim3(grid_for(total, 256)), dim3(256), 0, s, p);
  else if (a->dtype == MTBT_BF16) hipLaunchKernelGGL((fuse_kernel<bf16_t>), dim3(grid_for(total, 256)), dim3(256), 0, s, p);
  else if (a->dtype == MTBT_F16) hipLaunchKernelGGL((fuse_kernel<f16_t>), dim3(grid_for(total, 256)), dim3(256), 0, s, p);
  else return MTBT_EINVAL;
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_gap_fc(const void* x, const float* w, const float* b, float* y, int N, int HW, int C, int nout,
                           int dtype, void* stream) {
  if (!x || !w || !y || N <= 0 || HW <= 0 || C <= 0 || C % 8 || C > 2048 || nout <= 0) return MTBT_EINVAL;
  if (!aligned16(x)) return MTBT_EALIGN;
  const int CH8 = C / 8;
  int G = 256 / CH8;
  if (G < 1) G = 1;
  int threads = ((G * CH8 + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  // kernel derives G = blockDim / CH8; keep that equal to the G used for the LDS size
  G = threads / CH8;
  const size_t lds = (size_t)(G + 1) * C * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MTBT_F32) hipLaunchKernelGGL((gap_fc_kernel<float>), dim3(N), dim3(threads), lds, s, (const float*)x, w, b, y, HW, C, nout);
  else if (dtype == MTBT_BF16) hipLaunchKernelGGL((gap_fc_kernel<bf16_t>), dim3(N), dim3(threads), lds, s, (const bf16_t*)x, w, b, y, HW, C, nout);
  else if (dtype == MTBT_F16) hipLaunchKernelGGL((gap_fc_kernel<f16_t>), dim3(N), dim3(threads), lds, s, (const f16_t*)x, w, b, y, HW, C, nout);
  else return MTBT_EINVAL;
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_cast(const void* src, void* dst, int64_t n, int sd, int dd, void* stream) {
  if (!src || !dst || n <= 0) return MTBT_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned g = grid_for(n, 256);
  if (sd == MTBT_F32 && dd == MTBT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, (long)n);
  else if (sd == MTBT_BF16 && dd == MTBT_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, (long)n);
  else if (sd == MTBT_F32 && dd == MTBT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, (long)n);
  else if (sd == MTBT_BF16 && dd == MTBT_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, (long)n);
  else if (sd == MTBT_F32 && dd == MTBT_F16) hipLaunchKernelGGL((cast_kernel<float, f16_t>), dim3(g), dim3(256), 0, s, (const float*)src, (f16_t*)dst, (long)n);
  else if (sd == MTBT_F16 && dd == MTBT_F32) hipLaunchKernelGGL((cast_kernel<f16_t, float>), dim3(g), dim3(256), 0, s, (const f16_t*)src, (float*)dst, (long)n);
  else return MTBT_EINVAL;
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int64_t mtbt_layernorm_backward_params_workspace_bytes(int64_t pixels, int C) {
  if (pixels <= 0 || C <= 0) return 0;
  const int CH8 = C / 8;
  const int LP = CH8 <= 16 ? 16 : (CH8 <= 32 ? 32 : 64);
  const int64_t per = (int64_t)lnb_iters(pixels, 64 / LP) * (64 / LP);
  const int64_t waves = (pixels + per - 1) / per;
  return ((waves + 3) / 4) * 2 * (int64_t)C * (int64_t)sizeof(float);
}

// LayerNorm backward over the channels of every pixel plus the parameter gradients in the same pass:
//   dx (+)= rstd (g - mean_c g - xhat mean_c(g xhat)), g = dy gamma;   dgamma (+)= sum_p dy xhat;   dbeta (+)= sum_p dy
extern "C" int mtbt_layernorm_backward_params_nhwc(const void* x, const void* dy, const float* w, float eps, void* dx, int64_t pixels, int C, int dtype,
                                                   int accumulate_dx, float* dgamma, float* dbeta, int accumulate_params, void* workspace,
                                                   int64_t workspace_bytes, void* stream) {
  if (!x || !dy || !w || !dx || !dgamma || !dbeta || !workspace || pixels <= 0 || C <= 0 || C % 8 || C > 2048) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !aligned16(w) || !aligned16(workspace)) return MTBT_EALIGN;
  if (workspace_bytes < mtbt_layernorm_backward_params_workspace_bytes(pixels, C)) return MTBT_EWORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int CH8 = C / 8;
  const int LP = CH8 <= 16 ? 16 : (CH8 <= 32 ? 32 : 64);
  const int iters = lnb_iters(pixels, 64 / LP);
  const long waves = (pixels + (long)iters * (64 / LP) - 1) / ((long)iters * (64 / LP));
  const long blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  float* partial = reinterpret_cast<float*>(workspace);
  const int lds = 4 * 2 * C * (int)sizeof(float);
#define LNP_LAUNCH(T, MAXV, LPV)                                                                                                          \
  do {                                                                                                                                    \
    if (int rc = mtbt_allow_lds(layernorm_bwd_params_kernel<T, MAXV, LPV>, lds)) return rc;                                                \
    hipLaunchKernelGGL((layernorm_bwd_params_kernel<T, MAXV, LPV>), dim3((unsigned)blocks), dim3(256), lds, s, (const T*)x, (const T*)dy, w, eps, \
                       (T*)dx, (long)pixels, C, accumulate_dx, partial, iters);                                                            \
  } while (0)
#define LNP_BY_C(T) \
  do { if (LP == 16) LNP_LAUNCH(T, 1, 16); else if (LP == 32) LNP_LAUNCH(T, 1, 32); else if (CH8 <= 64) LNP_LAUNCH(T, 1, 64); \
       else if (CH8 <= 128) LNP_LAUNCH(T, 2, 64); else LNP_LAUNCH(T, 4, 64); } while (0)
  if (dtype == MTBT_F32) LNP_BY_C(float);
  else if (dtype == MTBT_BF16) LNP_BY_C(bf16_t);
  else return MTBT_EINVAL;
#undef LNP_BY_C
#undef LNP_LAUNCH
  const int rows = (int)blocks;           // (waves past the last pixel contribute zeros)
  hipLaunchKernelGGL(channel_sum_final_pitch, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, partial, rows, 2 * C, 0, C, dbeta, accumulate_params);
  hipLaunchKernelGGL(channel_sum_final_pitch, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, partial, rows, 2 * C, C, C, dgamma, accumulate_params);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
