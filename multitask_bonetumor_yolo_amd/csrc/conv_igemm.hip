// Implicit-GEMM convolution on gfx950 MFMA, NHWC activations, KRSC weights.
//
//   D[k][p] = sum_{r,s,c} W[k][r][s][c] * X[n(p), oy(p)*st - pad + r, ox(p)*st - pad + s, c]
//
// GEMM view: rows = output channels (MFMA "A" operand = weights), columns = output pixels
// (MFMA "B" operand = gathered input), reduction = (r,s,c) in KRSC order.  With this orientation the
// 16x16 accumulator of one MFMA holds, per lane, FOUR CONSECUTIVE OUTPUT CHANNELS of one pixel
// (row = 4*(lane>>4)+reg, col = lane&15), i.e. an 8-byte (bf16) / 16-byte (f32) contiguous piece of
// the NHWC output, so the epilogue stores vector pieces straight from the accumulator.
//
// Workgroup: 256 threads = 4 waves arranged WAVES_C x WAVES_P over a TC x TP (channels x pixels)
// tile.  Per K-step both operand tiles (TC and TP rows of BKB bytes) are staged global -> registers
// -> LDS with 16-byte accesses (the im2col gather needs a per-lane bounds predicate, so register
// staging, not LDS-DMA), double-buffered in LDS with ONE barrier per step: the global loads of step
// t+1 are issued before the MFMAs of step t and written to the other buffer after them.
// LDS rows are XOR-swizzled so that both the ds_write_b128 staging stores and the ds_read_b128
// fragment reads are bank-conflict-free (checked against the gfx950 lane-group rule).
//
// bf16: v_mfma_f32_16x16x32_bf16 (one 16-B chunk = 8 k per lane).  f32: v_mfma_f32_16x16x4_f32, the
// exact-fp32 parity path; a lane's 16-B chunk holds 4 consecutive k which feed 4 MFMAs (the k order
// inside the reduction is permuted identically for both operands).
#include "common.h"

namespace {

struct ConvP {
  const void* x;
  const void* w;
  void* y;
  const float* scale;
  const float* shift;
  const void* res;
  long xbs, ybs, rbs;
  int ldx, ldy, ldr;
  int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
  int act, out_mode, out_f32, vec_ok;
  long M;      // N*Ho*Wo
  int ctiles;  // ceil(K / TC)
};

template <int CPR> __device__ __forceinline__ int swz(int row) {
  if constexpr (CPR == 4) return (-(row >> 2)) & 3;   // 64-B rows
  else return (row >> 1) & 7;                         // 128-B rows
}

template <typename T, int TC, int TP, int WAVES_C, int WAVES_P, int BKB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int CPR = BKB / 16;              // chunks per LDS row
  constexpr int BKE = BKB / (int)sizeof(T);  // reduction elements per step
  constexpr int WC = TC / WAVES_C, WP = TP / WAVES_P;
  constexpr int FC = WC / 16, FP = WP / 16;
  constexpr int WCH = (TC * CPR + 255) / 256, XCH = (TP * CPR + 255) / 256;
  constexpr int KSUB = BKB / 64;             // 16-B chunk groups (of 4) per row
  constexpr int BUF = (TC + TP) * BKB;
  static_assert(WAVES_C * WAVES_P == 4, "4 waves");
  static_assert(WC % 16 == 0 && WP % 16 == 0, "wave tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int ctile = blockIdx.x % p.ctiles;
  const long ptile = blockIdx.x / p.ctiles;
  const int cbase = ctile * TC;
  const long pbase = ptile * TP;
  const int HoWo = p.Ho * p.Wo;
  const int Kdim = p.R * p.S * p.C;

  // ---- per-thread staging descriptors (fixed over the whole reduction) ----
  const T* xptr[XCH];
  int xiy[XCH], xix[XCH], xdst[XCH];
  bool xok[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const int c = tid + i * 256;
    const int row = c / CPR, q = c % CPR;
    const long pix = pbase + row;
    xok[i] = (c < TP * CPR) && (pix < p.M);
    const long pp = xok[i] ? pix : 0;
    const int n = (int)(pp / HoWo);
    const int rem = (int)(pp - (long)n * HoWo);
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    xiy[i] = oy * p.stride - p.pad;
    xix[i] = ox * p.stride - p.pad;
    xptr[i] = reinterpret_cast<const T*>(p.x) + (long)n * p.xbs + ((long)xiy[i] * p.W + xix[i]) * p.ldx + q * EPC;
    xdst[i] = TC * BKB + row * BKB + ((q ^ swz<CPR>(row)) << 4);
  }
  const T* wptr[WCH];
  int wdst[WCH];
  bool wok[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int c = tid + i * 256;
    const int row = c / CPR, q = c % CPR;
    wok[i] = (c < TC * CPR) && (cbase + row < p.K);
    wptr[i] = reinterpret_cast<const T*>(p.w) + (long)(wok[i] ? cbase + row : 0) * Kdim + q * EPC;
    wdst[i] = row * BKB + ((q ^ swz<CPR>(row)) << 4);
  }

  // ---- wave / lane geometry ----
  const int wave = tid >> 6, lane = tid & 63;
  const int wc = wave / WAVES_P, wp = wave % WAVES_P;
  const int lr = lane & 15, lq = lane >> 4;
  int aoff[FC][KSUB], boff[FP][KSUB];
#pragma unroll
  for (int f = 0; f < FC; ++f) {
    const int row = wc * WC + f * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) aoff[f][ks] = row * BKB + (((ks * 4 + lq) ^ swz<CPR>(row)) << 4);
  }
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int row = wp * WP + f * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) boff[f][ks] = TC * BKB + row * BKB + (((ks * 4 + lq) ^ swz<CPR>(row)) << 4);
  }

  f32x4 acc[FC][FP];
#pragma unroll
  for (int i = 0; i < FC; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // reduction walk state (wave-uniform)
  int kr = 0, ks_ = 0, kc = 0;  // filter row, filter col, channel offset of the step being LOADED
  long kw = 0;                  // linear k offset of that step in KRSC order
  const int nsteps = Kdim / BKE;

  uint4 xreg[XCH], wreg[WCH];
  auto load_step = [&]() {
    const long xoff = ((long)kr * p.W + ks_) * p.ldx + kc;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const bool ok = xok[i] && (unsigned)(xiy[i] + kr) < (unsigned)p.H && (unsigned)(xix[i] + ks_) < (unsigned)p.W;
      xreg[i] = ok ? *reinterpret_cast<const uint4*>(xptr[i] + xoff) : uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i)
      wreg[i] = wok[i] ? *reinterpret_cast<const uint4*>(wptr[i] + kw) : uint4{0u, 0u, 0u, 0u};
    kw += BKE;
    kc += BKE;
    if (kc == p.C) { kc = 0; if (++ks_ == p.S) { ks_ = 0; ++kr; } }
  };
  auto store_step = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < XCH; ++i)
      if (XCH * 256 == TP * CPR || tid + i * 256 < TP * CPR) *reinterpret_cast<uint4*>(buf + xdst[i]) = xreg[i];
#pragma unroll
    for (int i = 0; i < WCH; ++i)
      if (WCH * 256 == TC * CPR || tid + i * 256 < TC * CPR) *reinterpret_cast<uint4*>(buf + wdst[i]) = wreg[i];
  };

  load_step();
  store_step(smem);
  __syncthreads();

  for (int t = 0; t < nsteps; ++t) {
    char* cur = smem + (t & 1) * BUF;
    const bool more = (t + 1 < nsteps);
    if (more) load_step();
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      uint4 a[FC], b[FP];
#pragma unroll
      for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const uint4*>(cur + aoff[f][ks]);
#pragma unroll
      for (int f = 0; f < FP; ++f) b[f] = *reinterpret_cast<const uint4*>(cur + boff[f][ks]);
#pragma unroll
      for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                               __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].x), __uint_as_float(b[j].x), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].y), __uint_as_float(b[j].y), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].z), __uint_as_float(b[j].z), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].w), __uint_as_float(b[j].w), acc[i][j], 0, 0, 0);
          }
        }
    }
    if (more) store_step(smem + ((t + 1) & 1) * BUF);
    __syncthreads();
  }

  // ---- epilogue: affine, activation, residual, store (4 consecutive channels per lane per tile) ----
  const int Cq = p.K >> 2;  // ConvT: channels per (dy,dx) quadrant
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    const long pix = pbase + wp * WP + j * 16 + lr;
    if (pix >= p.M) continue;
    const int n = (int)(pix / HoWo);
    const int rem = (int)(pix - (long)n * HoWo);
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
#pragma unroll
    for (int i = 0; i < FC; ++i) {
      const int ch = cbase + wc * WC + i * 16 + lq * 4;
      if (ch >= p.K) continue;
      long yoff, roff;
      int chout = ch;
      if (p.out_mode == MTBT_OUT_CONVT2X2) {
        const int quad = ch / Cq;
        chout = ch - quad * Cq;
        const long opix = (long)(2 * oy + (quad >> 1)) * (2 * p.Wo) + (2 * ox + (quad & 1));
        yoff = (long)n * p.ybs + opix * p.ldy + chout;
        roff = (long)n * p.rbs + opix * p.ldr + chout;
      } else {
        yoff = (long)n * p.ybs + (long)rem * p.ldy + ch;
        roff = (long)n * p.rbs + (long)rem * p.ldr + ch;
      }
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.vec_ok) {
        if (p.scale) { const float4 s = *reinterpret_cast<const float4*>(p.scale + ch); v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w; }
        if (p.shift) { const float4 s = *reinterpret_cast<const float4*>(p.shift + ch); v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
        if (p.res) {
          if constexpr (sizeof(T) == 2) {
            const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.res) + roff);
            v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
            v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
          } else {
            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + roff);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
          }
        }
        if (p.out_f32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + yoff) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.y) + yoff) = o;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ch + e >= p.K) break;
          float u = v[e];
          if (p.scale) u *= p.scale[ch + e];
          if (p.shift) u += p.shift[ch + e];
          u = act_apply(u, p.act);
          if (p.res) u += ld_elem<T>(reinterpret_cast<const T*>(p.res) + roff + e);
          if (p.out_f32) reinterpret_cast<float*>(p.y)[yoff + e] = u;
          else reinterpret_cast<bf16_t*>(p.y)[yoff + e] = f2bf(u);
        }
      }
    }
  }
}

template <typename T, int TC, int TP, int WAVES_C, int WAVES_P, int BKB>
int launch(const ConvP& p, hipStream_t stream) {
  ConvP q = p;
  q.ctiles = (p.K + TC - 1) / TC;
  const long ptiles = (p.M + TP - 1) / TP;
  const long blocks = ptiles * q.ctiles;
  if (blocks <= 0 || blocks > 0x7fffffffL) return MTBT_EINVAL;
  constexpr int lds = 2 * (TC + TP) * BKB;
  hipLaunchKernelGGL((conv_igemm_kernel<T, TC, TP, WAVES_C, WAVES_P, BKB>), dim3((unsigned)blocks), dim3(256), lds, stream, q);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

template <typename T, int BKB>
int dispatch_tile(const ConvP& p, int TC, int TP, hipStream_t s) {
  if (TP == 128) {
    if (TC == 128) return launch<T, 128, 128, 2, 2, BKB>(p, s);
    if (TC == 96) return launch<T, 96, 128, 2, 2, BKB>(p, s);
    if (TC == 64) return launch<T, 64, 128, 1, 4, BKB>(p, s);
    if (TC == 32) return launch<T, 32, 128, 1, 4, BKB>(p, s);
  } else if (TP == 64) {
    if (TC == 128) return launch<T, 128, 64, 4, 1, BKB>(p, s);
    if (TC == 96) return launch<T, 96, 64, 2, 2, BKB>(p, s);
    if (TC == 64) return launch<T, 64, 64, 2, 2, BKB>(p, s);
    if (TC == 32) return launch<T, 32, 64, 1, 4, BKB>(p, s);
  }
  return MTBT_EINVAL;
}

// Tile heuristic: channel tile with the least padding waste (ties -> larger), pixel tile 128 unless
// that leaves the 256 CUs under-filled.
void pick_tile(int K, long M, int* TC, int* TP) {
  const int cands[4] = {128, 96, 64, 32};
  long best_waste = -1;
  int best = 128;
  for (int c : cands) {
    const long padded = (long)((K + c - 1) / c) * c;
    if (best_waste < 0 || padded < best_waste) { best_waste = padded; best = c; }
  }
  *TC = best;
  const long ct = (K + best - 1) / best;
  *TP = (((M + 127) / 128) * ct >= 512) ? 128 : 64;
}

}  // namespace

extern "C" int mtbt_conv2d_nhwc(const mtbt_conv_args* a, void* stream) {
  if (!a || !a->x || !a->w || !a->y) return MTBT_EINVAL;
  if (a->dtype != MTBT_F32 && a->dtype != MTBT_BF16) return MTBT_EINVAL;
  if (a->out_dtype != a->dtype && a->out_dtype != MTBT_F32) return MTBT_EINVAL;
  if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || a->K <= 0 || a->R <= 0 || a->S <= 0 || a->stride <= 0 || a->pad < 0)
    return MTBT_EINVAL;
  const int es = a->dtype == MTBT_F32 ? 4 : 2;
  const int epc = 16 / es;
  if (a->C % (64 / es) != 0) return MTBT_EINVAL;
  if (a->Ho != (a->H + 2 * a->pad - a->R) / a->stride + 1 || a->Wo != (a->W + 2 * a->pad - a->S) / a->stride + 1) return MTBT_EINVAL;
  if (a->out_mode != MTBT_OUT_NHWC && a->out_mode != MTBT_OUT_CONVT2X2) return MTBT_EINVAL;
  if (a->out_mode == MTBT_OUT_CONVT2X2 && (a->K % 4 != 0)) return MTBT_EINVAL;
  if (a->act < 0 || a->act > MTBT_ACT_GELU) return MTBT_EINVAL;
  if (!aligned16(a->x) || !aligned16(a->w) || a->x_pixel_stride % epc != 0 || a->x_batch_stride % epc != 0) return MTBT_EALIGN;
  if (a->x_pixel_stride < a->C) return MTBT_EINVAL;

  ConvP p;
  p.x = a->x; p.w = a->w; p.y = a->y; p.scale = a->scale; p.shift = a->shift; p.res = a->res;
  p.xbs = a->x_batch_stride; p.ybs = a->y_batch_stride; p.rbs = a->res_batch_stride;
  p.ldx = a->x_pixel_stride; p.ldy = a->y_pixel_stride; p.ldr = a->res_pixel_stride;
  p.N = a->N; p.H = a->H; p.W = a->W; p.C = a->C; p.K = a->K; p.R = a->R; p.S = a->S;
  p.stride = a->stride; p.pad = a->pad; p.Ho = a->Ho; p.Wo = a->Wo;
  p.act = a->act; p.out_mode = a->out_mode; p.out_f32 = (a->out_dtype == MTBT_F32);
  p.M = (long)a->N * a->Ho * a->Wo;
  p.ctiles = 0;
  // vector epilogue: 4 consecutive output channels per lane must be one aligned access
  const int oes = p.out_f32 ? 4 : 2;
  const int kq = a->out_mode == MTBT_OUT_CONVT2X2 ? a->K / 4 : a->K;
  bool vec = (kq % 4 == 0) && (a->y_pixel_stride % 4 == 0) && (a->y_batch_stride % 4 == 0) &&
             ((reinterpret_cast<uintptr_t>(a->y) % (4 * oes)) == 0);
  if (a->scale) vec = vec && aligned16(a->scale);
  if (a->shift) vec = vec && aligned16(a->shift);
  if (a->res) vec = vec && (a->res_pixel_stride % 4 == 0) && (a->res_batch_stride % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(a->res) % (4 * es)) == 0);
  p.vec_ok = vec ? 1 : 0;

  int TC, TP;
  if (a->tile_hint) { TC = a->tile_hint >> 16; TP = a->tile_hint & 0xffff; }
  else pick_tile(a->K, p.M, &TC, &TP);
  const bool wide = (a->C % (128 / es) == 0);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == MTBT_F32) return wide ? dispatch_tile<float, 128>(p, TC, TP, s) : dispatch_tile<float, 64>(p, TC, TP, s);
  return wide ? dispatch_tile<bf16_t, 128>(p, TC, TP, s) : dispatch_tile<bf16_t, 64>(p, TC, TP, s);
}
