// Depthwise conv for TWO or THREE channel chunks (128 < C <= 384: ConvNeXt stages 1-2, the 256-channel head DWConvs),
// one tile per workgroup.  Same arithmetic as dwconv.hip; kept as its own kernel because this register allocation -- all
// taps of a chunk in registers, scoped to the chunk, chunk loop fully unrolled, no state carried across tiles --
// compiles without spills at 3 x 32 accumulators and measured faster in the network than the persistent kernel's
// multi-chunk path (stage 2: 36 vs 48 us per layer, stage 1: 75 vs 81, head 3x3 at 80x80: 47 vs 60).
//
// Depthwise KS x KS convolution (stride 1, pad KS/2), NHWC, fp32 arithmetic on the VALU.
//
// Depthwise has no cross-channel reduction, so there is no MFMA shape for it; the roof is the packed
// fp32 FMA rate (v_pk_fma_f32).  What the first versions of this kernel ran into instead was the
// vector-memory ISSUE rate: a 4-byte-per-lane global load costs the CU's address unit as much as a
// 16-byte one, and a 7x7 window needs ~7 input vectors per output pixel.  So:
//
//   * a workgroup owns a TH x TW output tile and walks the channels in chunks of CC = 128;
//   * per chunk the (TH+KS-1) x (TW+KS-1) input halo tile and the chunk's KS*KS taps are staged in LDS
//     with 16-byte global loads (the only global reads), 256 B (bf16) per pixel;
//   * wave w owns the 2 x 8 output sub-tile w; lane l owns channel pair (2l, 2l+1) of the chunk: every
//     LDS read is a conflict-free 4-byte (bf16x2) / 8-byte (f32x2) access, every FMA a packed pair;
//     a sub-tile row of 8+KS-1 inputs is read once and feeds both output rows and all KS horizontal taps;
//   * all chunks' accumulators stay in registers; the LayerNorm statistics (ConvNeXt conv_dw + norm)
//     are per-thread partial sums + one LDS transpose-reduce per wave (a wave holds ALL channels of its
//     16 pixels), two-pass mean/variance, then the normalised pairs are stored straight from registers.
#include "common.h"
#include "conv_dma.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <typename T> struct Pair;
template <> struct Pair<float> {
  static __device__ __forceinline__ f32x2 ld(const void* p) { return *reinterpret_cast<const f32x2*>(p); }
  static __device__ __forceinline__ void st(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
};
template <> struct Pair<bf16_t> {
  static __device__ __forceinline__ f32x2 ld(const void* p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  }
  static __device__ __forceinline__ void st(bf16_t* p, f32x2 v) {
    *reinterpret_cast<uint32_t*>(p) = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
  }
};

// Sum each of 16 per-lane values over the 64 lanes of a wave and give every lane all 16 totals.
// LDS transpose instead of 96 ds_bpermute: red[p][lane] (16 conflict-free b32 writes), lane L then sums
// a quarter row (4 x b128) of pixel L/4, two quad-DPP adds finish the row, one b32 write per pixel and
// four broadcast b128 reads return the totals.  `red` = this wave's private 4 KiB + 64 B region.
__device__ __forceinline__ void wave_sum16(float (&v)[16], float* red, int lane) {
#pragma unroll
  for (int p = 0; p < 16; ++p) red[p * 64 + lane] = v[p];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // cross-lane hand-off inside the wave: order writes before reads
  const float4* row = reinterpret_cast<const float4*>(red + (lane >> 2) * 64 + (lane & 3) * 16);
  const float4 a = row[0], b = row[1], c = row[2], d = row[3];
  float t = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, true));  // quad xor 1
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, true));  // quad xor 2
  float* tot = red + 16 * 64;
  if ((lane & 3) == 0) tot[lane >> 2] = t;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float4* tv = reinterpret_cast<const float4*>(tot);
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) {
    const float4 r = tv[p4];
    v[p4 * 4 + 0] = r.x; v[p4 * 4 + 1] = r.y; v[p4 * 4 + 2] = r.z; v[p4 * 4 + 3] = r.w;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // totals read before the region is written again
}

constexpr int CC = 128;  // channels per chunk = 64 lanes x 2

// MAXCH = ceil(C / 128) chunks held in registers.
template <typename T, int KS, bool LN, int TH, int TW, int MAXCH>
__global__ __launch_bounds__((TH / 2) * (TW / 8) * 64, 2) void dwconv_kernel(
    const T* __restrict__ x, const T* __restrict__ w /* [KS*KS][C] */, const float* __restrict__ bias,
    const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, const float* __restrict__ scale,
    const float* __restrict__ shift, int act, T* __restrict__ y, int N, int H, int W, int C) {
  constexpr int PAD = KS / 2, XB = 8, YB = 2, SPAN = XB + KS - 1, ROWS = YB + KS - 1;
  constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
  constexpr int ES = (int)sizeof(T), PIXB = CC * ES;  // bytes per staged pixel
  constexpr int NT = (TH / 2) * (TW / 8) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                         // [IH*IW][CC] T

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int bt = blockIdx.x;
  const int tx = bt % tiles_x, ty = (bt / tiles_x) % tiles_y, n = bt / (tiles_x * tiles_y);
  const int ty0 = ty * TH, tx0 = tx * TW;
  const int sy = wave / (TW / 8), sx = wave % (TW / 8);  // sub-tile of this wave
  const int nchunks = (C + CC - 1) / CC;
  const T* xn = x + (long)n * H * W * C;
  constexpr int PARTS = PIXB / 16, NIT = (IH * IW * PARTS + NT - 1) / NT;
  const srd_t xsrd = make_srd(xn);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  f32x2 acc[MAXCH][YB][XB];
#pragma unroll
  for (int k = 0; k < MAXCH; ++k)
#pragma unroll
    for (int a = 0; a < YB; ++a)
#pragma unroll
      for (int i = 0; i < XB; ++i) acc[k][a][i] = f32x2{0.f, 0.f};

#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    if (k >= nchunks) break;
    const int cb = k * CC;                       // chunk base channel
    const int cc = min(CC, C - cb);              // channels in this chunk (multiple of 8)
    const int parts = cc * ES / 16;              // 16-byte pieces per pixel
    const bool active = lane * 2 < cc;
    // this lane's KS*KS taps of the chunk live in REGISTERS (fp32 pairs), loaded straight from global (L2-resident,
    // one 4/8-byte coalesced load per tap) while the tile is being staged: the FMA loop then reads only inputs from LDS
    f32x2 wr[KS * KS];
    if (active) {
#pragma unroll
      for (int t = 0; t < KS * KS; ++t) wr[t] = Pair<T>::ld(w + (long)t * C + cb + lane * 2);
    }
    if (k > 0) __syncthreads();                  // previous chunk's readers are done
    // Halo tile global -> LDS by LDS-DMA (conv_dma.h): every wave issues all its 1 KiB pieces back to back and waits
    // ONCE.  Pieces outside the image / past the chunk's channels read zeros.
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pc = it * NT + tid;
      const int pix = pc / PARTS, part = pc % PARTS;
      const int r = pix / IW, c = pix - r * IW;
      const int iy = ty0 + r - PAD, ix = tx0 + c - PAD;
      const bool ok = pix < IH * IW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && part < parts;
      const unsigned vo = ok ? (unsigned)((iy * W + ix) * C * ES + part * 16) : 0x80000000u;
      lds_dma16(xsrd, vo, cb * ES, lds0 + it * NT * 16 + wave_u * 1024);
    }
    wait_vm<0>();
    __syncthreads();
    if (active) {
      const char* lp = tile + ((sy * YB) * IW + sx * XB) * PIXB + lane * 2 * ES;
      // fully unrolled: tap registers need compile-time indices.  One input row (SPAN pairs) feeds YB output rows.
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        f32x2 in[SPAN];
#pragma unroll
        for (int j = 0; j < SPAN; ++j) in[j] = Pair<T>::ld(lp + (r * IW + j) * PIXB);
#pragma unroll
        for (int a = 0; a < YB; ++a) {
          const int ky = r - a;
          if (ky >= 0 && ky < KS) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
              for (int i = 0; i < XB; ++i) acc[k][a][i] = fma2(in[i + kx], wr[ky * KS + kx], acc[k][a][i]);
          }
        }
      }
    }
  }

  const int oy0 = ty0 + sy * YB, ox0 = tx0 + sx * XB;
  if constexpr (LN) {
    // + bias, per-pixel statistics over all C channels (held by this wave), normalise, store
    __syncthreads();  // every wave is done with the staged tile: its LDS is reused for the reductions
    float* red = reinterpret_cast<float*>(smem) + wave * (16 * 64 + 16);
    float s[YB * XB];
#pragma unroll
    for (int pq = 0; pq < YB * XB; ++pq) s[pq] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int c0 = k * CC + lane * 2;
      if (k < nchunks && c0 < C) {
        const f32x2 bv = *reinterpret_cast<const f32x2*>(bias + c0);
#pragma unroll
        for (int a = 0; a < YB; ++a)
#pragma unroll
          for (int i = 0; i < XB; ++i) {
            acc[k][a][i] += bv;
            s[a * XB + i] += acc[k][a][i].x + acc[k][a][i].y;
          }
      }
    }
    wave_sum16(s, red, lane);
    const float invC = 1.0f / C;
    float q[YB * XB];
#pragma unroll
    for (int pq = 0; pq < YB * XB; ++pq) { s[pq] *= invC; q[pq] = 0.f; }
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      if (k < nchunks && k * CC + lane * 2 < C) {
#pragma unroll
        for (int a = 0; a < YB; ++a)
#pragma unroll
          for (int i = 0; i < XB; ++i) {
            const float dx = acc[k][a][i].x - s[a * XB + i], dy = acc[k][a][i].y - s[a * XB + i];
            q[a * XB + i] += dx * dx + dy * dy;
          }
      }
    }
    wave_sum16(q, red, lane);
#pragma unroll
    for (int pq = 0; pq < YB * XB; ++pq) q[pq] = rsqrtf(q[pq] * invC + eps);
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int c0 = k * CC + lane * 2;
      if (k < nchunks && c0 < C) {
        const f32x2 gw = *reinterpret_cast<const f32x2*>(lnw + c0), gb = *reinterpret_cast<const f32x2*>(lnb + c0);
#pragma unroll
        for (int a = 0; a < YB; ++a) {
          if (oy0 + a >= H) continue;
#pragma unroll
          for (int i = 0; i < XB; ++i) {
            if (ox0 + i >= W) continue;
            const float m = s[a * XB + i], rs = q[a * XB + i];
            Pair<T>::st(y + (((long)n * H + oy0 + a) * W + ox0 + i) * C + c0, (acc[k][a][i] - m) * rs * gw + gb);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int c0 = k * CC + lane * 2;
      if (k < nchunks && c0 < C) {
        const f32x2 sc = *reinterpret_cast<const f32x2*>(scale + c0), sh = *reinterpret_cast<const f32x2*>(shift + c0);
#pragma unroll
        for (int a = 0; a < YB; ++a) {
          if (oy0 + a >= H) continue;
#pragma unroll
          for (int i = 0; i < XB; ++i) {
            if (ox0 + i >= W) continue;
            const f32x2 v = fma2(acc[k][a][i], sc, sh);
            Pair<T>::st(y + (((long)n * H + oy0 + a) * W + ox0 + i) * C + c0, f32x2{act_apply(v.x, act), act_apply(v.y, act)});
          }
        }
      }
    }
  }
}

template <typename T, int KS, bool LN, int TH, int TW, int MAXCH>
int launch_dw(const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps,
              const float* scale, const float* shift, int act, void* y, int N, int H, int W, int C, hipStream_t s) {
  constexpr int NT = (TH / 2) * (TW / 8) * 64;
  constexpr int PARTS = CC * (int)sizeof(T) / 16;
  constexpr int lds_tile = (((TH + KS - 1) * (TW + KS - 1) * PARTS + NT - 1) / NT) * NT * 16, lds_red = (NT / 64) * (16 * 64 + 16) * 4;
  constexpr int lds = lds_tile > lds_red ? lds_tile : lds_red;
  static_assert(lds <= 160 * 1024, "LDS");
  const long blocks = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  auto kern = dwconv_kernel<T, KS, LN, TH, TW, MAXCH>;
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    attr_set = true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MTBT_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, s, (const T*)x, (const T*)w, bias, lnw, lnb, eps, scale, shift,
                     act, (T*)y, N, H, W, C);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

}  // namespace

// Two or three chunks (128 < C <= 384); called from mtbt_dwconv_nhwc in dwconv.hip, which has validated the arguments.
int mtbt_dw_chunked(const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps,
                    const float* scale, const float* shift, int act, void* y, int N, int H, int W, int C, int ksize, int dtype,
                    hipStream_t s) {
  const bool ln = lnw != nullptr;
  const int nch = (C + CC - 1) / CC;
#define DWC(T, KS, LNB, TW, M) return launch_dw<T, KS, LNB, 4, TW, M>(x, w, bias, lnw, lnb, eps, scale, shift, act, y, N, H, W, C, s)
#define DWC_M(T, KS, LNB, TW) do { if (nch == 2) DWC(T, KS, LNB, TW, 2); else DWC(T, KS, LNB, TW, 3); } while (0)
  if (dtype == MTBT_BF16) {
    if (ksize == 7) { if (ln) DWC_M(bf16_t, 7, true, 16); else DWC_M(bf16_t, 7, false, 16); }
    else { if (ln) DWC_M(bf16_t, 3, true, 16); else DWC_M(bf16_t, 3, false, 16); }
  } else {
    if (ksize == 7) { if (ln) DWC_M(float, 7, true, 8); else DWC_M(float, 7, false, 8); }
    else { if (ln) DWC_M(float, 3, true, 8); else DWC_M(float, 3, false, 8); }
  }
#undef DWC_M
#undef DWC
  return MTBT_EINVAL;
}
