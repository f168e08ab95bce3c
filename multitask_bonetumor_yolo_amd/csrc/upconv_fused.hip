// ConvTranspose2d(2, stride 2, bias) followed by Conv 3x3 (pad 1) + per-channel shift + activation as ONE direct convolution on the
// LOW-resolution map -- ultralytics Proto: `upsample` -> `cv2` (main_model.py:326-328 [ultralytics Proto]).
//
// Both operators are linear and nothing sits between them, so for an output pixel of parity (a, b) = (Y & 1, X & 1) at
// (Y, X) = (2i + a, 2j + b) the nine taps of the 3x3 conv land on only 2 x 2 SOURCE pixels, rows i + a - 1 .. i + a, columns
// j + b - 1 .. j + b: the pair is a 2x2-tap conv per parity with host-composed weights
//     Wc[q = 2a + b][k][rho][sigma][ci] = sum over the (dy, dx) that map to source offset (rho, sigma) of
//                                         sum_cm W3[k][cm][dy + 1][dx + 1] * WT[ci][cm][(a + dy) & 1][(b + dx) & 1]
// -- 4 x (2 x 2) = 16 tap GEMMs of C x K instead of (1 ConvT + 9 conv) = 10 per HIGH-resolution pixel, i.e. 4/10 of the MACs, and the
// upsampled tensor (210 MB at batch 16 x 640^2, written and read straight back: the largest HBM item of the step) never exists.
// The transposed conv's bias reaches an output through every tap that lies INSIDE the upsampled map (the 3x3 conv pads with zeros, not
// with the bias), so its contribution depends on the pixel's border class: shift[rc * 3 + cc][k], rc / cc = 0 first row / column,
// 2 last, 1 interior (nine vectors, composed on the host together with the BatchNorm fold).
//
// Kernel = the row-reuse direct kernel of conv3x3_direct.inc with a tap LIST: GEMM rows = "virtual" output channels q * K + k, a
// workgroup owns 128 of them (one parity) x a 16 x 16 source tile; the 18 x 18 source halo of a 32-channel slab is LDS-resident and
// serves the parity's 2 x 2 taps by shifted addresses; waves 2 (channel halves) x 2 (row halves); per slab and filter column sigma the
// wave reads its 9 input-row fragments once for both filter rows.  Output pixel (2(ty0 + y) + a, 2(tx0 + x) + b): 16-byte stores.
#include "common.h"
#include "conv_dma.h"

namespace {

struct UpP {
  const void* x; const void* w; void* y; const float* shift;
  long xbs, ybs;
  int ldx, ldy;
  int N, H, W, C, K, act;
  int ctiles, ptiles_per_xcd;
};

template <typename T, int ACT>
__global__ __launch_bounds__(256, 2) void upconv_fused_kernel(const UpP p) {
  constexpr int TC = 128;
  constexpr int ES = (int)sizeof(T), EPC = 16 / ES, BKB = 64, BKE = BKB / ES;
  constexpr int HWID = 18, NHP = HWID * HWID;
  constexpr int FC = TC / 32, FP = 8, WCH = TC / 2;
  constexpr int XDMA = (NHP * 4 + 255) / 256;                  // halo DMA instructions per wave (6)
  constexpr int XBYTES = XDMA * 256 * 16;
  constexpr int WTAP = TC * BKB, WGRP = 2 * WTAP;              // one tap tile / the two filter rows of one filter column
  constexpr int WDMA = WGRP / 16 / 256;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int PITCH = WCH * 4 + 16, C8 = WCH / 8, ITER = (16 * C8) / 64;
  static_assert(WDMA * 256 * 16 == WGRP && (16 * C8) % 64 == 0, "whole wave-instructions");
  static_assert(4 * 16 * PITCH <= XBYTES + 2 * WGRP, "epilogue slabs below the shift table");
  typedef typename half_of<T>::type HT;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* aff = reinterpret_cast<float*>(smem + XBYTES + 2 * WGRP);   // [9][TC] class shifts of this workgroup's channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = wave >> 1;
  const int lr = lane & 15, lq = lane >> 4;

  const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int ctile = slot % p.ctiles;
  const int ptile = xcd * p.ptiles_per_xcd + slot / p.ctiles;
  const int tiles_x = p.W >> 4, tpi = tiles_x * (p.H >> 4);
  if (ptile >= p.N * tpi) return;
  const int n = ptile / tpi, trem = ptile - n * tpi;
  const int ty0 = (trem / tiles_x) << 4, tx0 = (trem % tiles_x) << 4;
  const int vbase = ctile * TC;                 // first virtual channel q * K + k of this workgroup (K % TC == 0: one parity)
  const int q = vbase / p.K, cbase = vbase - q * p.K;
  const int pa = q >> 1, pb = q & 1;
  const int Kdim = 4 * p.C;
  for (int c = tid; c < 9 * TC; c += 256) aff[c] = p.shift[(c / TC) * p.K + cbase + (c % TC)];

  const srd_t xsrd = make_srd(reinterpret_cast<const T*>(p.x) + (long)n * p.xbs);
  const srd_t wsrd = make_srd(reinterpret_cast<const T*>(p.w) + (long)vbase * Kdim);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  unsigned xvoff[XDMA];
#pragma unroll
  for (int i = 0; i < XDMA; ++i) {
    const int c = i * 256 + tid;
    const int hp = c >> 2, sl = c & 3;
    const int hy = hp / HWID, hx = hp - hy * HWID;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const bool ok = hp < NHP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    xvoff[i] = ok ? (unsigned)((((long)iy * p.W + ix) * p.ldx + (sl ^ ((hp >> 2) & 3)) * EPC) * ES) : OOB;
  }
  unsigned wvoff[WDMA];   // piece c of a filter column: filter row rho = c / (TC*4), weight row, 16-byte slot
#pragma unroll
  for (int i = 0; i < WDMA; ++i) {
    const int c = i * 256 + tid;
    const int rho = c / (TC * 4), rem = c - rho * (TC * 4);
    const int row = rem >> 2, sl = rem & 3;
    wvoff[i] = (unsigned)(((long)row * Kdim + (long)rho * 2 * p.C + (sl ^ ((row >> 2) & 3)) * EPC) * ES);
  }
  // halo row of output source row y with filter row rho: y + pa + rho; halo column of x with filter column sigma: x + pb + sigma
  const int hp0 = (8 * wr + pa) * HWID + lr + pb;                                  // + rr * 18 + sigma
  const int arow = wc * WCH + lr;                                                  // + f * 16
  const int aoff = XBYTES + arow * BKB + ((lq ^ ((arow >> 2) & 3)) << 4);          // + buf * WGRP + rho * WTAP + f * 16 * BKB

  f32x4 acc[FC][FP];
#pragma unroll
  for (int i = 0; i < FC; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nslabs = p.C / BKE, nsteps = 2 * nslabs;
  auto stage_w = [&](int g) {   // weights of column step g (slab g / 2, filter column g % 2) -> buffer g & 1
    const int cc = g >> 1, sg = g & 1;
    const int wsoff = (sg * p.C) * ES + cc * BKB;
#pragma unroll
    for (int i = 0; i < WDMA; ++i) lds_dma16(wsrd, wvoff[i], wsoff, lds0 + XBYTES + (g & 1) * WGRP + (i * 256 + wave * 64) * 16);
  };
  auto stage_x = [&](int cc) {
#pragma unroll
    for (int i = 0; i < XDMA; ++i) lds_dma16(xsrd, xvoff[i], cc * BKB, lds0 + (i * 256 + wave * 64) * 16);
  };
  stage_x(0);
  stage_w(0);
#pragma unroll 1
  for (int g0 = 0; g0 < nsteps; g0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // the slab's two filter columns: buffer parity and sigma are compile-time
      const int g = g0 + u;
      wait_vm<0>();               // my pieces of this step's weights (and, at sigma == 0, of the slab's halo) have landed
      lds_barrier();              // everyone's have; everyone is done with the previous step's weight buffer
      if (g + 1 < nsteps) stage_w(g + 1);
      uint4 b[FP + 1];
#pragma unroll
      for (int rr = 0; rr < FP + 1; ++rr) {
        const int hp = hp0 + rr * HWID + u;
        b[rr] = *reinterpret_cast<const uint4*>(smem + hp * BKB + ((lq ^ ((hp >> 2) & 3)) << 4));
      }
      if (u == 1 && g + 1 < nsteps) {   // the halo is free once every wave holds this column's fragments: request the next slab
        lds_barrier();
        stage_x((g >> 1) + 1);
      }
      const char* wb = smem + aoff + u * WGRP;
#pragma unroll
      for (int rho = 0; rho < 2; ++rho) {
        uint4 a[FC];
#pragma unroll
        for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const uint4*>(wb + rho * WTAP + f * 16 * BKB);
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
          for (int j = 0; j < FP; ++j) {
            if constexpr (sizeof(T) == 2) {
              acc[i][j] = mfma_16x16x32<T>(a[i], b[j + rho], acc[i][j]);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].x), __uint_as_float(b[j + rho].x), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].y), __uint_as_float(b[j + rho].y), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].z), __uint_as_float(b[j + rho].z), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].w), __uint_as_float(b[j + rho].w), acc[i][j], 0, 0, 0);
            }
          }
      }
    }
  }
  __syncthreads();  // LDS below the shift table is free for the epilogue slabs

  // ---- epilogue: per source row (slab) j, accumulators + class shift -> activation -> [16 px][WCH ch] fp32 slab -> row pass, 16-byte stores
  char* slab = smem + wave * (16 * PITCH);
  const int ccls = (pb == 0 && tx0 + lr == 0) ? 0 : ((pb == 1 && tx0 + lr == p.W - 1) ? 2 : 1);   // this lane's pixel column
  int so[ITER];
  long yo[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / C8, c8 = idx - row * C8;
    so[it] = row * PITCH + c8 * 32;
    yo[it] = (long)n * p.ybs + ((long)(2 * (ty0 + 8 * wr) + pa) * (2 * p.W) + 2 * (tx0 + row) + pb) * p.ldy + cbase + wc * WCH + c8 * 8;
  }
  const long ystep = (long)4 * p.W * p.ldy;      // one source row = two output rows
#pragma clang loop unroll(full)
  for (int j = 0; j < FP; ++j) {
    const int sy = ty0 + 8 * wr + j;
    const int rcls = (pa == 0 && sy == 0) ? 0 : ((pa == 1 && sy == p.H - 1) ? 2 : 1);
    const float* sh = aff + (rcls * 3 + ccls) * TC + wc * WCH + lq * 4;
#pragma unroll
    for (int i = 0; i < FC; ++i) {
      const float4 h4 = *reinterpret_cast<const float4*>(sh + i * 16);
      float4 v;
      v.x = act_apply(acc[i][j][0] + h4.x, ACT); v.y = act_apply(acc[i][j][1] + h4.y, ACT);
      v.z = act_apply(acc[i][j][2] + h4.z, ACT); v.w = act_apply(acc[i][j][3] + h4.w, ACT);
      *reinterpret_cast<float4*>(slab + lr * PITCH + (i * 16 + lq * 4) * 4) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local hand-off through LDS
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const float4 lo = *reinterpret_cast<const float4*>(slab + so[it]);
      const float4 hi = *reinterpret_cast<const float4*>(slab + so[it] + 16);
      const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      if constexpr (sizeof(T) == 4) st8<float>(reinterpret_cast<float*>(p.y) + yo[it] + j * ystep, v);
      else st8<HT>(reinterpret_cast<HT*>(p.y) + yo[it] + j * ystep, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab fully read before the next pass rewrites it
  }
}

template <typename T, int ACT>
int launch_upconv(const UpP& p, hipStream_t stream) {
  UpP qd = p;
  qd.ctiles = 4 * p.K / 128;
  const long ptiles = (long)p.N * (p.H >> 4) * (p.W >> 4);
  qd.ptiles_per_xcd = (int)((ptiles + 7) / 8);
  const long blocks = (long)qd.ptiles_per_xcd * 8 * qd.ctiles;
  if (blocks <= 0 || blocks > 0x7fffffffL) return MTBT_EINVAL;
  constexpr int XBYTES = ((324 * 4 + 255) / 256) * 256 * 16;
  constexpr int lds = XBYTES + 2 * 2 * 128 * 64 + 9 * 128 * 4;
  if (int rc = mtbt_allow_lds(upconv_fused_kernel<T, ACT>, lds)) return rc;
  hipLaunchKernelGGL((upconv_fused_kernel<T, ACT>), dim3((unsigned)blocks), dim3(256), lds, stream, qd);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

template <typename T>
int launch_upconv_act(const UpP& p, hipStream_t s) {
  switch (p.act) {
    case MTBT_ACT_NONE: return launch_upconv<T, MTBT_ACT_NONE>(p, s);
    case MTBT_ACT_SILU: return launch_upconv<T, MTBT_ACT_SILU>(p, s);
    default: return MTBT_EINVAL;
  }
}

}  // namespace

extern "C" int mtbt_convt2x2_conv3x3_nhwc(const mtbt_upconv_args* a, void* stream) {
  if (!a || !a->x || !a->w || !a->y || !a->shift) return MTBT_EINVAL;
  if (a->dtype != MTBT_F32 && a->dtype != MTBT_BF16 && a->dtype != MTBT_F16) return MTBT_EINVAL;
  if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || a->K <= 0) return MTBT_EINVAL;
  const int es = a->dtype == MTBT_F32 ? 4 : 2, epc = 16 / es;
  if (a->H % 16 || a->W % 16 || a->C % (64 / es) || a->K % 128) return MTBT_EINVAL;
  if (a->x_pixel_stride < a->C || a->y_pixel_stride < a->K) return MTBT_EINVAL;
  if (!aligned16(a->x) || !aligned16(a->w) || !aligned16(a->y) || !aligned16(a->shift) || a->x_pixel_stride % epc || a->x_batch_stride % epc ||
      a->y_pixel_stride % epc || a->y_batch_stride % epc)
    return MTBT_EALIGN;
  // LDS-DMA addressing: 32-bit byte offsets below 2 GiB relative to (image, weight tile row 0)
  if ((long)a->H * a->W * a->x_pixel_stride * es >= 0x7fff0000L || (long)128 * 4 * a->C * es >= 0x7fff0000L) return MTBT_EINVAL;
  UpP p;
  p.x = a->x; p.w = a->w; p.y = a->y; p.shift = a->shift;
  p.xbs = a->x_batch_stride; p.ybs = a->y_batch_stride; p.ldx = a->x_pixel_stride; p.ldy = a->y_pixel_stride;
  p.N = a->N; p.H = a->H; p.W = a->W; p.C = a->C; p.K = a->K; p.act = a->act;
  p.ctiles = p.ptiles_per_xcd = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == MTBT_F32) return launch_upconv_act<float>(p, s);
  if (a->dtype == MTBT_F16) return launch_upconv_act<f16_t>(p, s);
  return launch_upconv_act<bf16_t>(p, s);
}
