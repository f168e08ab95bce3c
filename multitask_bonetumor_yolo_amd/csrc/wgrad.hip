// Weight gradient of a k x k convolution of any stride, bf16 operands, fp32 result:
//     dW[k][r][s][c] = sum over output pixels p = (n, y, x) of  dY[p][k] * X[n][y * stride + r - pad][x * stride + s - pad][c]
// The reduction runs over the NHWC-SLOW index, so both MFMA operands arrive transposed with respect to how the forward
// kernels read them.  gfx950's ds_read_b64_tr_b16 delivers a 4-row x 16-column block of 16-bit elements column-major to a
// 16-lane group, which is exactly one half of a 16x16x32 operand fragment, so the tiles are staged in their natural layout
// ([pixel][channel] rows, 16-byte global loads) and transposed by the read.
//
//   workgroup   one (128 output channels) x (128 input channels) tile of one filter tap, over one slice of the pixel range;
//               waves 2 x 2, each 64 x 64 channels (16 accumulator fragments)
//   step        64 pixels: dY tile [64][128] and the tap-shifted X tile [64][128] (zeros outside the image) through registers
//               into LDS (32-byte units XOR-permuted so that the transposed reads are conflict-free), per 32 pixels 16
//               transposed reads and 16 MFMAs per wave
//   reduction   fp32 partial tiles per pixel slice, summed in slice order by a second kernel (deterministic; no atomics)
// Register double-buffering: the global loads of step t+1 are issued before the MFMAs of step t; one LDS buffer.  Its place in the plan and what comes next: DESIGN.md §7.
#include "common.h"
#include "rowreduce.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifndef MTBT_WGRAD_XCD_SLICES
#define MTBT_WGRAD_XCD_SLICES 0
#endif
#ifndef MTBT_WGRAD_WIDE
#define MTBT_WGRAD_WIDE 1
#endif
constexpr int TK = 128, TCH = 128, TPX = 64, LPT = TPX / 16;   // LPT: 16-byte loads per thread and tile in one step

struct WgP {
  const bf16_t* x; const bf16_t* dy; float* partial;
  float* bpartial;    // optional [nsplit][K]: per-slice sums of dy over the pixels (the bias gradient), or NULL
  int N, H, W, C, K, R, S, pad, stride, Ho, Wo;
  long x_bs, dy_bs; int ldx, ldy;
  long P, per;        // pixels, pixels per slice (multiple of TPX)
  int nsplit, ktiles, ctiles;
  int xcd_slices;     // slices grouped by XCD (see the kernel)
  int flat;           // 1 x 1 / stride 1 / pad 0 with dense batch strides: operand rows are indexed by the pixel number
};

__device__ __forceinline__ s16x4 tr_read(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// LDS image of a [pixel][128 channels] tile: 256-byte rows, the eight 32-byte units of a row XOR-permuted by
// (row & 3) | ((row >> 3) & 1) << 2.  One transposed read of a 32-lane half touches rows r0 .. r0+3 and r0+8 .. r0+11 of one
// unit column: with the permutation their eight 32-byte windows cover all 64 banks exactly once.
__device__ __forceinline__ int swz_unit(int row, int unit) { return unit ^ ((row & 3) | (((row >> 3) & 1) << 2)); }
__device__ __forceinline__ int tile_off(int row, int col) {   // element offset of (row, col); col % 4 == 0 stays inside its unit
  return row * 128 + (swz_unit(row, col >> 4) << 4) + (col & 15);
}

// GELU (gelu_poly of common.h, evaluated on packed pairs) of the 8 bf16 values of a 16-byte piece
typedef float f32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gelu2_bf16(uint32_t u) {
  const f32x2w x = f32x2w{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  const f32x2w xc = f32x2w{__builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f)};
  const f32x2w s = xc * xc;
  auto k = [](float c) { return f32x2w{c, c}; };
  f32x2w q = __builtin_elementwise_fma(k(2.1609857e-08f), s, k(-1.5335673e-06f));
  q = __builtin_elementwise_fma(q, s, k(4.6542096e-05f));
  q = __builtin_elementwise_fma(q, s, k(-7.9887325e-04f));
  q = __builtin_elementwise_fma(q, s, k(8.6900834e-03f));
  q = __builtin_elementwise_fma(q, s, k(-6.4366050e-02f));
  q = __builtin_elementwise_fma(q, s, k(3.9770728e-01f));
  const f32x2w g = x * __builtin_elementwise_fma(xc, q, k(0.5f));
  return (uint32_t)f2bf(g.x) | ((uint32_t)f2bf(g.y) << 16);
}
__device__ __forceinline__ uint4 gelu8_bf16(uint4 v) { return uint4{gelu2_bf16(v.x), gelu2_bf16(v.y), gelu2_bf16(v.z), gelu2_bf16(v.w)}; }

// NK x NC: the tile is (128 NK output channels) x (128 NC input channels), each of the 2 x 2 waves 64 NK x 64 NC.  (1, 1) is the base
// form.  The GEMM-shaped gradients of the ConvNeXt MLPs (K, C = 384 .. 3072 at 12 800 .. 204 800 pixels) are bound by operand traffic, not
// MFMA time: a 128 x 128 tile does 64 FLOP per byte it stages and ran at 360 TFLOP/s = 5.6 TB/s out of L2 / Infinity Cache; (2, 1) / (1, 2)
// stage 3/4 of the bytes for twice the MFMAs (85 FLOP/B) at the price of 128 accumulator registers.  The wide side is laid out as two
// 128-column panels of the base LDS image.
// XACT: the X operand is a PRE-activation and GELU (the bf16 forward's polynomial form) is applied while it is staged -- the fc2 weight gradient of a
// ConvNeXt block whose fused forward never wrote the activated hidden tensor (mtbt_convnext_mlp_fused_train).
// WK: waves along the output channels (2: 256 threads; 4: 512 threads, the 256 x 256 tile of the widest GEMM-shaped layers -- each wave still 64 x 128)
template <bool BIAS, int NK, int NC, bool XACT = false, int WK = 2, bool FLAT = false>   // FLAT: operand rows indexed by the pixel number (below); BIAS: also sum dY over the pixels (compiled out of the plain kernel: its registers and branch cost ~8 % there)
__global__ __launch_bounds__(128 * WK, 2) void wgrad_kernel(const WgP p) {
  constexpr int TKt = TK * NK, TCt = TCH * NC, FA = 8 * NK / WK, FB = 4 * NC;
  constexpr int RP = 8 * WK, LPT = TPX / RP;            // tile rows staged per pass, passes per step (shadows the file-scope LPT of the 256-thread form)
  static_assert(FA % 4 == 0 && !(BIAS && WK != 2), "wave tile");
  __shared__ __attribute__((aligned(16))) bf16_t sdy[TPX * TKt];
  __shared__ __attribute__((aligned(16))) bf16_t sx[TPX * TCt];
  // Workgroup -> (pixel slice, tap, tile).  xcd_slices (MTBT_WGRAD_XCD_SLICES, OFF): all tiles of one slice on ONE XCD (consecutive
  // workgroup ids go round the 8 XCDs) so that the slice reaches that XCD's L2 once instead of once per tile from the fabric -- the family
  // fetches 2.8x its algorithmic bytes (round-3 PMC; the stage-2 MLP gradients, 36 tiles per slice, 4.8x).  MEASURED SLOWER: the MLP weight
  // gradients went 168 -> 220 us (stage 2) and 210 -> 305 us (stage 0), everything else unchanged -- 3 .. 36 workgroups that start together
  // and walk the same rows in lockstep queue on the same L2 channels, where the spread mapping is served by eight L2s and the Infinity
  // Cache in parallel.  The re-reads are cheap; the fix for these layers is a wider tile (fewer, larger operand passes), not placement.
  const int taps = p.R * p.S;
  int b = blockIdx.x, split;
  if (p.xcd_slices) {
    const int T = p.ctiles * p.ktiles * taps, xcd = b & 7, j = b >> 3;
    split = (j / T) * 8 + xcd;
    b = j % T;
    if (split >= p.nsplit) return;                       // (padding of the slice count to a multiple of 8: whole workgroups leave)
  } else {
    split = b / (p.ctiles * p.ktiles * taps);
    b -= split * (p.ctiles * p.ktiles * taps);
  }
  const int ct = b % p.ctiles; b /= p.ctiles;
  const int kt = b % p.ktiles; b /= p.ktiles;
  const int tap = b;
  const int r = tap / p.S, s = tap - r * p.S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WK, wc = wave / WK;           // wave tile: 16 FA output channels x 64 NC input channels
  const int row = tid >> 4, chunk = tid & 15;         // staging: tile row (pixel of the pass), 16-byte chunk (8 channels) of a 128-column panel
  const int HW = p.Ho * p.Wo;                       // pixels are those of dY
  const long p0 = (long)split * p.per, p1 = min(p.P, p0 + p.per);
  const int kch = kt * TKt + chunk * 8, cch = ct * TCt + chunk * 8;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;

  f32x4 acc[FA][FB];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient for free: the workgroups of (first input-channel tile, first tap) also multiply their dY fragments with a
  // fragment of ONES -- D[k][j] = sum_p dY[p][k] in every column j (4 extra MFMAs per 16, in 1 / (ctiles * taps) of the workgroups)
  const bool do_bias = BIAS && p.bpartial != nullptr && ct == 0 && tap == 0 && wc == 0;
  f32x4 accb[FA];
#pragma unroll
  for (int i = 0; i < FA; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0

  uint4 vdy[NK][LPT], vx[NC][LPT];
  auto fetch = [&](long pb) {   // this thread's 16-byte pieces of the step starting at pixel pb (zeros past the slice / outside the image)
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const long pix = pb + i * RP + row;
#pragma unroll
      for (int h = 0; h < NK; ++h) vdy[h][i] = uint4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int h = 0; h < NC; ++h) vx[h][i] = uint4{0u, 0u, 0u, 0u};
      if (pix < p1) {
        if constexpr (FLAT) {   // 1 x 1, stride 1, dense batch strides: pixel index = row of both operands (no divisions: the general path costs
                        // two 32-bit divisions per piece -- and cost two 64-BIT ones, ~100 instructions each, until round 3)
          const bf16_t* dyp = p.dy + pix * p.ldy;
          const bf16_t* xp = p.x + pix * p.ldx;
#pragma unroll
          for (int h = 0; h < NK; ++h)
            if (kch + h * 128 < p.K) vdy[h][i] = *reinterpret_cast<const uint4*>(dyp + kch + h * 128);
#pragma unroll
          for (int h = 0; h < NC; ++h)
            if (cch + h * 128 < p.C) vx[h][i] = *reinterpret_cast<const uint4*>(xp + cch + h * 128);
          continue;
        }
        if constexpr (FLAT) continue;                            // (unreachable: keeps the general addressing out of the FLAT instances)
        const unsigned upix = (unsigned)pix;                     // (P < 2^31: checked on the host)
        const int n = (int)(upix / (unsigned)HW), rem = (int)(upix - (unsigned)n * (unsigned)HW);
        const int y = (int)((unsigned)rem / (unsigned)p.Wo), xx = rem - y * p.Wo;
        const bf16_t* dyp = p.dy + (long)n * p.dy_bs + (long)rem * p.ldy;
#pragma unroll
        for (int h = 0; h < NK; ++h)
          if (kch + h * 128 < p.K) vdy[h][i] = *reinterpret_cast<const uint4*>(dyp + kch + h * 128);
        const int iy = y * p.stride + r - p.pad, ix = xx * p.stride + s - p.pad;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
          const bf16_t* xp = p.x + (long)n * p.x_bs + ((long)iy * p.W + ix) * p.ldx;
#pragma unroll
          for (int h = 0; h < NC; ++h)
            if (cch + h * 128 < p.C) vx[h][i] = *reinterpret_cast<const uint4*>(xp + cch + h * 128);
        }
      }
    }
  };
  if (p0 < p1) fetch(p0);
  for (long pb = p0; pb < p1; pb += TPX) {
    __syncthreads();                                   // the previous step's fragment reads are done
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int tr = i * RP + row;
#pragma unroll
      for (int h = 0; h < NK; ++h) *reinterpret_cast<uint4*>(sdy + h * (TPX * 128) + tile_off(tr, chunk * 8)) = vdy[h][i];
#pragma unroll
      for (int h = 0; h < NC; ++h) {
        if constexpr (XACT) vx[h][i] = gelu8_bf16(vx[h][i]);
        *reinterpret_cast<uint4*>(sx + h * (TPX * 128) + tile_off(tr, chunk * 8)) = vx[h][i];
      }
    }
    __syncthreads();
    if (pb + TPX < p1) fetch(pb + TPX);                // the next step's global loads are in flight during this step's MFMAs
    // every lane takes part in the transposed reads (EXEC must be full): the branches above are uniform or closed by now
#pragma unroll
    for (int i = 0; i < TPX / 32; ++i) {
      const int r_lo = i * 32 + 8 * g + q, r_hi = r_lo + 4;     // operand k index 8g .. 8g+3 and 8g+4 .. 8g+7 of this 32-pixel group
      s16x8 B[FB];
#pragma unroll
      for (int f = 0; f < FB; ++f) {
        const int cb = 64 * NC * wc + 16 * f + 4 * pp;          // column of the wave's c range: panel cb / 128
        const bf16_t* base = sx + (cb >> 7) * (TPX * 128);
        B[f] = __builtin_shufflevector(tr_read(base + tile_off(r_lo, cb & 127)), tr_read(base + tile_off(r_hi, cb & 127)), 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int ha = 0; ha < FA / 4; ++ha) {                     // the A fragments four at a time (16 registers live)
        s16x8 A[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int ca = 16 * FA * wk + 16 * (ha * 4 + f) + 4 * pp;
          const bf16_t* base = sdy + (ca >> 7) * (TPX * 128);
          A[f] = __builtin_shufflevector(tr_read(base + tile_off(r_lo, ca & 127)), tr_read(base + tile_off(r_hi, ca & 127)), 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int fa = 0; fa < 4; ++fa)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb)
            acc[ha * 4 + fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[fa]), __builtin_bit_cast(bf16x8, B[fb]), acc[ha * 4 + fa][fb], 0, 0, 0);
        if (BIAS && do_bias) {   // (wave-uniform)
#pragma unroll
          for (int fa = 0; fa < 4; ++fa)
            accb[ha * 4 + fa] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[fa]), __builtin_bit_cast(bf16x8, ones), accb[ha * 4 + fa], 0, 0, 0);
        }
      }
    }
  }
  if (BIAS && do_bias && (lane & 15) == 0) {
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kt * TKt + 16 * FA * wk + 16 * fa + 4 * (lane >> 4) + e;
        if (k < p.K) p.bpartial[(long)split * p.K + k] = accb[fa][e];
      }
  }
  // lane: output channel kt * TKt + 64 NK wk + 16 fa + 4 (lane / 16) + e, input channel ct * TCt + 64 NC wc + 16 fb + lane % 16
  const long RSC = (long)taps * p.C;
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const int c = ct * TCt + 64 * NC * wc + 16 * fb + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kt * TKt + 16 * FA * wk + 16 * fa + 4 * (lane >> 4) + e;
        if (k < p.K && c < p.C) p.partial[((long)split * p.K + k) * RSC + (long)tap * p.C + c] = acc[fa][fb][e];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, long n, int nsplit, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? dw[i] : 0.f;
#pragma unroll 4
  for (int k = 0; k < nsplit; ++k) s += partial[(long)k * n + i];
  dw[i] = s;
}

// small tensors with many slices (a serial walk of the slices by n / 256 workgroups is latency-bound): workgroup = 64 elements x 4 slice
// groups, added in a fixed order through LDS
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ partial, float* __restrict__ dw, long n, int nsplit, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n) {
#pragma unroll 4
    for (int k = g; k < nsplit; k += 4) s += partial[(long)k * n + i];
  }
  red[g][lane] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

static inline void launch_wgrad_reduce(const float* partial, float* dw, long n, int nsplit, int accumulate, hipStream_t st) {
  if (nsplit >= 8 && n <= (1L << 20))
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, partial, dw, n, nsplit, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, dw, n, nsplit, accumulate);
}

// ---- 3x3 / stride 1 / pad 1 from an LDS HALO TILE.  The generic kernel above gives every filter tap its own workgroups, so a 3x3 layer
// streams dY and X through L2 nine times (Proto cv2 at batch 32: 15 GB of tile reads for 0.84 GB of tensors -- L2-bandwidth-bound at
// ~410 TFLOP/s).  Here a workgroup owns a 64 x 64 (k x c) channel tile for ALL nine taps: per 8 x 8-pixel spatial tile it stages the dY
// tile [64 px][64 k] and the 10 x 10 input halo [100 px][64 c] once, reads the dY fragments once and the tap-shifted X fragments from the
// halo (a tap is a row offset), 72 MFMAs per wave per tile into 9 x (32 x 32) accumulators (144 registers).  LDS rows keep the 256-byte
// pitch / 32-byte-unit XOR image of the generic kernel (half of each row unused); the halo's key uses the halo LINE parity where the
// generic image uses bit 3 of the row, so that the 4 + 4 rows of a transposed read still hit eight distinct units. ----
struct Wg3P {
  const bf16_t* x; const bf16_t* dy; float* partial;
  int N, H, W, C, K;
  long x_bs, dy_bs; int ldx, ldy;
  int ktiles, ctiles, nsplit;
  long ntiles, per;
  int xcd_slices;
};

__device__ __forceinline__ int halo_off(int hp, int col) {   // element offset of halo pixel hp (10 per line), channel col (col % 4 == 0)
  const int key = (hp & 3) | (((hp / 10) & 1) << 2);
  return hp * 128 + (((col >> 4) ^ key) << 4) + (col & 15);
}

__global__ __launch_bounds__(512, 2) void wgrad3x3_halo_kernel(const Wg3P p) {
  __shared__ __attribute__((aligned(16))) bf16_t sdy[64 * 128];
  __shared__ __attribute__((aligned(16))) bf16_t sx[100 * 128];
  int b = blockIdx.x, split;
  if (p.xcd_slices) {                                    // the channel tiles of one slice on one XCD (see wgrad_kernel)
    const int T = p.ctiles * p.ktiles, xcd = b & 7, j = b >> 3;
    split = (j / T) * 8 + xcd;
    b = j % T;
    if (split >= p.nsplit) return;
  } else {
    split = b / (p.ctiles * p.ktiles);
    b -= split * (p.ctiles * p.ktiles);
  }
  const int ct = b % p.ctiles;
  const int kt = b / p.ctiles;
  // EIGHT waves: the 2 x 2 wave tiles of the 64 x 64 channel tile, twice -- waves 0-3 accumulate taps 0..4, waves 4-7 taps 5..8.  With all
  // nine taps per wave (144 accumulator registers + addressing) the kernel ran ONE wave per SIMD, four per CU, and sat at ~10 % MFMA
  // utilisation waiting for its own tile loads; forcing two waves per SIMD on that form spilled into the MFMA loop (1.7 -> 2.9 ms on Proto
  // cv2).  Split by taps a wave holds 80 accumulators, eight waves fit a CU, and the staging is spread over 512 threads.
  const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6;
  const int wave = w8 & 3, tg = w8 >> 2;
  const int wk = wave & 1, wc = wave >> 1;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tiles_x = p.W >> 3, tiles_y = p.H >> 3;
  constexpr int NT = 5;                                  // taps of a wave: tap0 .. tap0 + nt - 1, tap0 = 5 tg, nt = 5 or 4
  const int tap0 = tg * NT;
  f32x4 acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long t0 = (long)split * p.per, t1 = min(p.ntiles, t0 + p.per);
  uint4 vdy[1], vx[2];
  auto fetch = [&](long tl) {   // this thread's 16-byte pieces of spatial tile tl (zeros outside the image / past K, C)
    const unsigned utl = (unsigned)tl;                  // (ntiles < 2^31: 32-bit divisions; the 64-bit ones were ~300 instructions per tile)
    const int n = (int)(utl / (unsigned)(tiles_x * tiles_y)), trem = (int)(utl - (unsigned)n * (unsigned)(tiles_x * tiles_y));
    const int ty = (int)((unsigned)trem / (unsigned)tiles_x), tx = trem - ty * tiles_x;
#pragma unroll
    for (int u = 0; u < 1; ++u) {                      // dY tile: 64 pixels x 8 pieces of 8 channels
      const int it = u * 512 + tid, pix = it >> 3, part = it & 7;
      const int oy = ty * 8 + (pix >> 3), ox = tx * 8 + (pix & 7), kch = kt * 64 + part * 8;
      vdy[u] = uint4{0u, 0u, 0u, 0u};
      if (kch < p.K) vdy[u] = *reinterpret_cast<const uint4*>(p.dy + (long)n * p.dy_bs + ((long)oy * p.W + ox) * p.ldy + kch);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {                      // input halo: 100 pixels x 8 pieces
      const int it = u * 512 + tid;
      vx[u] = uint4{0u, 0u, 0u, 0u};
      if (it < 800) {
        const int hp = it >> 3, part = it & 7;
        const int hy = hp / 10, hx = hp - hy * 10;
        const int iy = ty * 8 - 1 + hy, ix = tx * 8 - 1 + hx, cch = ct * 64 + part * 8;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && cch < p.C)
          vx[u] = *reinterpret_cast<const uint4*>(p.x + (long)n * p.x_bs + ((long)iy * p.W + ix) * p.ldx + cch);
      }
    }
  };
  if (t0 < t1) fetch(t0);
  for (long tl = t0; tl < t1; ++tl) {
    __syncthreads();                                   // the previous tile's fragment reads are done
#pragma unroll
    for (int u = 0; u < 1; ++u) {
      const int it = u * 512 + tid;
      *reinterpret_cast<uint4*>(sdy + tile_off(it >> 3, (it & 7) * 8)) = vdy[u];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = u * 512 + tid;
      if (it < 800) *reinterpret_cast<uint4*>(sx + halo_off(it >> 3, (it & 7) * 8)) = vx[u];
    }
    __syncthreads();
    if (tl + 1 < t1) fetch(tl + 1);                    // the next tile's global loads are in flight during this tile's 72 MFMAs
    // every lane takes part in the transposed reads (EXEC full): the branches above are closed
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j_lo = i * 32 + 8 * g + q, j_hi = j_lo + 4;             // pixels (operand k index) of this 32-pixel group
      s16x8 A[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int ca = 32 * wk + 16 * f + 4 * pp;
        A[f] = __builtin_shufflevector(tr_read(sdy + tile_off(j_lo, ca)), tr_read(sdy + tile_off(j_hi, ca)), 0, 1, 2, 3, 4, 5, 6, 7);
      }
      const int h_lo = (j_lo >> 3) * 10 + (j_lo & 7);                     // halo pixel of tap (0, 0); tap (r, s) adds r * 10 + s
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = tap0 + tt;                                          // (wave-uniform)
        if (t >= 9) break;
        const int hl = h_lo + (t / 3) * 10 + (t % 3);
        s16x8 B[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int cb = 32 * wc + 16 * f + 4 * pp;
          B[f] = __builtin_shufflevector(tr_read(sx + halo_off(hl, cb)), tr_read(sx + halo_off(hl + 4, cb)), 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int fa = 0; fa < 2; ++fa)
#pragma unroll
          for (int fb = 0; fb < 2; ++fb)
            acc[tt][fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[fa]), __builtin_bit_cast(bf16x8, B[fb]), acc[tt][fa][fb], 0, 0, 0);
      }
    }
  }
  const long RSC = 9L * p.C;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int fa = 0; fa < 2; ++fa)
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        const int t = tap0 + tt;
        if (t >= 9) continue;
        const int c = ct * 64 + 32 * wc + 16 * fb + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kt * 64 + 32 * wk + 16 * fa + 4 * (lane >> 4) + e;
          if (k < p.K && c < p.C) p.partial[((long)split * p.K + k) * RSC + (long)t * p.C + c] = acc[tt][fa][fb][e];
        }
      }
}

// ---- fp32 operands (the parity mode): the same tap / slice decomposition on the VALU.  64 x 64-channel tile, 16 pixels per step staged
// as they lie ([pixel][channel] fp32 rows), thread (ty, tx) owns a 4 x 4 block of the tile.  Not a throughput kernel: it exists so that
// the whole backward pass can be checked against fp32 autograd. ----
struct WgPF {
  const float* x; const float* dy; float* partial;
  float* bpartial;
  int N, H, W, C, K, R, S, pad, stride, Ho, Wo;
  long x_bs, dy_bs; int ldx, ldy;
  long P, per;
  int nsplit, ktiles, ctiles;
};

__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgPF p) {
  __shared__ __attribute__((aligned(16))) float sdy[16 * 64];
  __shared__ __attribute__((aligned(16))) float sx[16 * 64];
  int b = blockIdx.x;
  const int ct = b % p.ctiles; b /= p.ctiles;
  const int kt = b % p.ktiles; b /= p.ktiles;
  const int taps = p.R * p.S;
  const int tap = b % taps, split = b / taps;
  const int r = tap / p.S, s = tap - r * p.S;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int row = tid >> 4, q = tid & 15;        // staging: pixel of the step, 4-channel piece
  const int HW = p.Ho * p.Wo;
  const long p0 = (long)split * p.per, p1 = min(p.P, p0 + p.per);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const bool do_bias = p.bpartial != nullptr && ct == 0 && tap == 0 && tx == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (long pb = p0; pb < p1; pb += 16) {
    const long pix = pb + row;
    float4 vd = make_float4(0.f, 0.f, 0.f, 0.f), vx = vd;
    if (pix < p1) {
      const int n = (int)(pix / HW), rem = (int)(pix - (long)n * HW);
      const int y = rem / p.Wo, xx = rem - y * p.Wo;
      const int kch = kt * 64 + q * 4, cch = ct * 64 + q * 4;
      if (kch < p.K) vd = *reinterpret_cast<const float4*>(p.dy + (long)n * p.dy_bs + (long)rem * p.ldy + kch);
      const int iy = y * p.stride + r - p.pad, ix = xx * p.stride + s - p.pad;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && cch < p.C)
        vx = *reinterpret_cast<const float4*>(p.x + (long)n * p.x_bs + ((long)iy * p.W + ix) * p.ldx + cch);
    }
    __syncthreads();
    *reinterpret_cast<float4*>(sdy + row * 64 + q * 4) = vd;
    *reinterpret_cast<float4*>(sx + row * 64 + q * 4) = vx;
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) {
      const float4 a = *reinterpret_cast<const float4*>(sdy + pp * 64 + ty * 4);
      const float4 c = *reinterpret_cast<const float4*>(sx + pp * 64 + tx * 4);
      const float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], cv[j], acc[i][j]);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum[i] += av[i];
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kt * 64 + ty * 4 + i;
      if (k < p.K) p.bpartial[(long)split * p.K + k] = bsum[i];
    }
  }
  const long RSC = (long)taps * p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kt * 64 + ty * 4 + i, c = ct * 64 + tx * 4 + j;
      if (k < p.K && c < p.C) p.partial[((long)split * p.K + k) * RSC + (long)tap * p.C + c] = acc[i][j];
    }
}

// ---- ConvNeXt stem weight gradient: dW[k][c*16 + ky*4 + kx] = sum_p d[p][k] * img[n][c][4 oy + ky][4 ox + kx]  (the 4x4 / stride-4
// patchify conv on the caller's NCHW fp32 image, main_model.py:21-26 [timm stem_0]).  Persistent workgroups: 64 pixels per step staged in
// LDS (48-float patches, K gradients), thread (kg, c) accumulates K/4 rows of column c in registers; per-workgroup partials. ----
template <typename T>
__global__ __launch_bounds__(192) void stem_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ d, int N, int H, int W, int K,
                                                         float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* patch = reinterpret_cast<float*>(smem);   // [64][48]
  float* dk = patch + 64 * 48;                     // [64][K]
  const int Ho = H / 4, Wo = W / 4;
  const long total = (long)N * Ho * Wo;
  const int tid = threadIdx.x, cg = tid % 12, kg = tid / 12;   // thread = an 8 (rows of dW) x 4 (columns) register block
  const bool own = kg * 8 < K;
  float acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // Staging through registers, one step ahead (round 3: the first version loaded and stored piece by piece, the rows of d ELEMENT by element --
  // 32 dependent trips of 2-byte loads per step: 528 us for a launch that moves 315 MB): a thread's four image pieces and its 16-byte
  // pieces of d for step t + 1 are requested before the FMAs of step t.
  constexpr int EPC = 16 / (int)sizeof(T), ND = (64 * (128 / EPC) + 191) / 192;
  const int dparts = K / EPC;                              // 16-byte pieces per row of d (K % 8 == 0)
  float4 vi[4];
  uint4 vd[ND];
  auto fetch = [&](long base) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = tid + u * 192;
      const int pix = it % 64, cky = it / 64;
      const long gp = base + pix;
      vi[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gp < total) {
        const unsigned ugp = (unsigned)gp;                 // (total < 2^31: 32-bit divisions)
        const int n = (int)(ugp / (unsigned)(Ho * Wo));
        const int rem = (int)(ugp - (unsigned)n * (unsigned)(Ho * Wo));
        const int oy = rem / Wo, ox = rem - oy * Wo;
        vi[u] = *reinterpret_cast<const float4*>(img + (((long)n * 3 + (cky >> 2)) * H + (oy * 4 + (cky & 3))) * W + ox * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int it = tid + u * 192;
      const int pix = it / dparts, part = it - pix * dparts;
      vd[u] = uint4{0u, 0u, 0u, 0u};
      if (it < 64 * dparts && base + pix < total) vd[u] = *reinterpret_cast<const uint4*>(d + (base + pix) * K + part * EPC);
    }
  };
  const long bstep = (long)gridDim.x * 64;
  if ((long)blockIdx.x * 64 < total) fetch((long)blockIdx.x * 64);
  for (long base = (long)blockIdx.x * 64; base < total; base += bstep) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = tid + u * 192;
      *reinterpret_cast<float4*>(patch + (it % 64) * 48 + (it / 64) * 4) = vi[u];
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int it = tid + u * 192;
      if (it < 64 * dparts) {
        float f[EPC];
        if constexpr (sizeof(T) == 4) { f[0] = __uint_as_float(vd[u].x); f[1] = __uint_as_float(vd[u].y); f[2] = __uint_as_float(vd[u].z); f[3] = __uint_as_float(vd[u].w); }
        else {
          const uint32_t w4[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(w4[e] << 16); f[2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u); }
        }
        float* dst = dk + (it / dparts) * K + (it % dparts) * EPC;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) *reinterpret_cast<float4*>(dst + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
      }
    }
    __syncthreads();
    if (base + bstep < total) fetch(base + bstep);
    if (own) {
      for (int pix = 0; pix < 64; ++pix) {   // 3 x 16-byte LDS reads for 32 FMAs
        const float4 pv = *reinterpret_cast<const float4*>(patch + pix * 48 + cg * 4);
        const float4 d0 = *reinterpret_cast<const float4*>(dk + pix * K + kg * 8), d1 = *reinterpret_cast<const float4*>(dk + pix * K + kg * 8 + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, pp[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], pp[j], acc[i][j]);
      }
    }
  }
  if (own) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<float4*>(partial + (long)blockIdx.x * K * 48 + (long)(kg * 8 + i) * 48 + cg * 4) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
  }
}

// the tile of the generic kernel for a shape: wide on the side that is a multiple of 256 channels when the other side spans more than one
// 128-channel tile too (only then is there an operand pass to save)
// (not with the fused bias gradient: its extra accumulators spill at two waves per SIMD)
static inline void pick_wide(int K, int C, bool allow, int* nk, int* nc) {
  *nk = *nc = 1;
#if MTBT_WGRAD_WIDE
  if (allow && K > 128 && C > 128) {
    if (K % 256 == 0 && C % 256 == 0 && (long)K * C >= 384L * 1024) { *nk = 2; *nc = 2; }    // 256 x 256, 512 threads: the stage-2 / stage-3 MLP gradients
    else if (K % 256 == 0 && K >= C) *nk = 2;
    else if (C % 256 == 0) *nc = 2;
    else if (K % 256 == 0) *nk = 2;
  }
#endif
}

int pick_split(int K, int C, int taps, long P, bool allow_wide) {
  // Slices are the only parallelism beyond the (few) output tiles, but every slice writes and re-reads a full fp32 copy of dW:
  // aim at ~6 workgroups per CU while a slice keeps >= 24 steps; accept shorter slices (>= 8 steps) only to reach 2 per CU.
  // (Measured on five of the network's shapes, tools/wgrad_probe.py; within ~15 % of the best split found for each.)
  int nk, nc;
  pick_wide(K, C, allow_wide, &nk, &nc);
  const long tiles = (long)((K + TK * nk - 1) / (TK * nk)) * ((C + TCH * nc - 1) / (TCH * nc)) * taps;
  auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
#ifndef MTBT_WGRAD_TARGET
#define MTBT_WGRAD_TARGET 512    // workgroups aimed at (batch-32 shapes, tools/wgrad_ab.py: 1536 -> 1024 was 12 % less time: fewer fp32 partial copies of dW; 512 once the
                                  // weight gradients run BESIDE other launches on the plan's lanes: -0.6 % per step, 256 the same)
#endif
  long ns = cdiv(MTBT_WGRAD_TARGET, tiles);
  ns = ns < cdiv(P, 24 * TPX) ? ns : cdiv(P, 24 * TPX);
  long floor_ns = cdiv(512, tiles);
  floor_ns = floor_ns < cdiv(P, 8 * TPX) ? floor_ns : cdiv(P, 8 * TPX);
  if (ns < floor_ns) ns = floor_ns;
  return (int)(ns < 1 ? 1 : ns);
}

}  // namespace

extern "C" int64_t mtbt_conv_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0) return 0;
  // an upper bound for every stride: slices never outnumber those of the stride-1 case (fewer dY pixels)
  const int a = pick_split(K, C, R * S, (long)N * H * W, true), b = pick_split(K, C, R * S, (long)N * H * W, false);
  return (int64_t)(a > b ? a : b) * ((int64_t)K * R * S * C + K) * (int64_t)sizeof(float);
}

static int wgrad_entry(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int K, int R, int S, int pad, int stride,
                       int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride, int dtype,
                       int accumulate, void* workspace, int64_t workspace_bytes, void* stream, int x_act = MTBT_ACT_NONE) {
  if (x_act != MTBT_ACT_NONE && (x_act != MTBT_ACT_GELU_POLY || dtype != MTBT_BF16 || dbias || R != 1 || S != 1)) return MTBT_EINVAL;
  if (!x || !dy || !dw || !workspace || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0) return MTBT_EINVAL;
  if (dtype != MTBT_BF16 && dtype != MTBT_F32) return MTBT_EINVAL;
  const int epc = dtype == MTBT_BF16 ? 8 : 4;   // elements per 16-byte piece
  if (C % epc || K % epc || x_pixel_stride % epc || dy_pixel_stride % epc || x_batch_stride % epc || dy_batch_stride % epc) return MTBT_EINVAL;
  if (stride < 1 || pad < 0 || H + 2 * pad < R || W + 2 * pad < S) return MTBT_EINVAL;
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(workspace)) return MTBT_EALIGN;
  const long P = (long)N * Ho * Wo;
  const int nsplit = pick_split(K, C, R * S, P, dbias == nullptr);
  // the slices actually launched decide the workspace (a padding > (R-1)/2 makes Ho*Wo exceed H*W, beyond the documented bound)
  if (workspace_bytes < (int64_t)nsplit * ((int64_t)K * R * S * C + K) * (int64_t)sizeof(float)) return MTBT_EWORKSPACE;
  float* bpartial = dbias ? reinterpret_cast<float*>(workspace) + (int64_t)nsplit * K * R * S * C : nullptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)K * R * S * C;
  if (dtype == MTBT_F32) {
    WgPF p;
    p.x = reinterpret_cast<const float*>(x); p.dy = reinterpret_cast<const float*>(dy); p.partial = reinterpret_cast<float*>(workspace);
    p.bpartial = bpartial;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.pad = pad; p.stride = stride; p.Ho = Ho; p.Wo = Wo;
    p.x_bs = x_batch_stride; p.dy_bs = dy_batch_stride; p.ldx = x_pixel_stride; p.ldy = dy_pixel_stride;
    p.P = P; p.nsplit = nsplit;
    p.per = ((p.P + p.nsplit - 1) / p.nsplit + 15) / 16 * 16;
    p.ktiles = (K + 63) / 64; p.ctiles = (C + 63) / 64;
    const long blocks = (long)p.nsplit * R * S * p.ktiles * p.ctiles;
    if (blocks > 0x7fffffffL) return MTBT_EINVAL;
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    launch_wgrad_reduce(p.partial, dw, n, p.nsplit, accumulate, st);
    if (dbias) launch_wgrad_reduce(bpartial, dbias, (long)K, p.nsplit, accumulate, st);
    MTBT_LAUNCH_CHECK();
    return MTBT_OK;
  }
  if (!dbias && R == 3 && S == 3 && stride == 1 && pad == 1 && H % 8 == 0 && W % 8 == 0) {   // halo-tile kernel: all nine taps per workgroup
    Wg3P q;
    q.x = reinterpret_cast<const bf16_t*>(x); q.dy = reinterpret_cast<const bf16_t*>(dy); q.partial = reinterpret_cast<float*>(workspace);
    q.N = N; q.H = H; q.W = W; q.C = C; q.K = K;
    q.x_bs = x_batch_stride; q.dy_bs = dy_batch_stride; q.ldx = x_pixel_stride; q.ldy = dy_pixel_stride;
    q.ktiles = (K + 63) / 64; q.ctiles = (C + 63) / 64;
    q.ntiles = (long)N * (H / 8) * (W / 8);
    const long base = (long)q.ktiles * q.ctiles;
    long ns = (768 + base - 1) / base;                                     // ~3 workgroups per CU
    if (ns > q.ntiles / 4) ns = q.ntiles / 4;                              // a slice keeps >= 4 spatial tiles
    const long fit = workspace_bytes / ((int64_t)K * 9 * C * (int64_t)sizeof(float));
    if (ns > fit) ns = fit;
    if (ns < 1) ns = 1;
    q.nsplit = (int)ns;
    q.per = (q.ntiles + ns - 1) / ns;
    q.xcd_slices = MTBT_WGRAD_XCD_SLICES && ns >= 8;
    const long blocks3 = base * (q.xcd_slices ? (ns + 7) / 8 * 8 : ns);
    if (blocks3 > 0x7fffffffL) return MTBT_EINVAL;
    hipLaunchKernelGGL(wgrad3x3_halo_kernel, dim3((unsigned)blocks3), dim3(512), 0, st, q);
    launch_wgrad_reduce(q.partial, dw, n, q.nsplit, accumulate, st);
    MTBT_LAUNCH_CHECK();
    return MTBT_OK;
  }
  WgP p;
  p.x = reinterpret_cast<const bf16_t*>(x); p.dy = reinterpret_cast<const bf16_t*>(dy); p.partial = reinterpret_cast<float*>(workspace);
  p.bpartial = bpartial;
  p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.pad = pad; p.stride = stride; p.Ho = Ho; p.Wo = Wo;
  p.x_bs = x_batch_stride; p.dy_bs = dy_batch_stride; p.ldx = x_pixel_stride; p.ldy = dy_pixel_stride;
  p.P = P;
  p.nsplit = nsplit;
  p.per = ((p.P + p.nsplit - 1) / p.nsplit + TPX - 1) / TPX * TPX;
  int nk, nc;
  pick_wide(K, C, dbias == nullptr, &nk, &nc);
  p.ktiles = (K + TK * nk - 1) / (TK * nk); p.ctiles = (C + TCH * nc - 1) / (TCH * nc);
  p.xcd_slices = MTBT_WGRAD_XCD_SLICES && p.nsplit >= 8;
  p.flat = (R == 1 && S == 1 && stride == 1 && pad == 0 && x_batch_stride == (int64_t)H * W * x_pixel_stride &&
            dy_batch_stride == (int64_t)Ho * Wo * dy_pixel_stride) ? 1 : 0;
  if (P > 0x7fffffffL) return MTBT_EINVAL;
  const long blocks = (long)(p.xcd_slices ? (p.nsplit + 7) / 8 * 8 : p.nsplit) * R * S * p.ktiles * p.ctiles;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
#define WGK(B_, NK_, NC_, X_, WK_)                                                                                            \
  do {                                                                                                                        \
    if (p.flat) hipLaunchKernelGGL((wgrad_kernel<B_, NK_, NC_, X_, WK_, true>), dim3((unsigned)blocks), dim3(128 * WK_), 0, st, p);  \
    else hipLaunchKernelGGL((wgrad_kernel<B_, NK_, NC_, X_, WK_, false>), dim3((unsigned)blocks), dim3(128 * WK_), 0, st, p);        \
  } while (0)
#define WGN(B_, NK_, NC_, X_, WK_) hipLaunchKernelGGL((wgrad_kernel<B_, NK_, NC_, X_, WK_, false>), dim3((unsigned)blocks), dim3(128 * WK_), 0, st, p)
#define WG(B_, X_)                                                  \
  do {                                                              \
    if (nk == 2 && nc == 2) WGK(false, 2, 2, X_, 4);                \
    else if (nk == 2) WGK(B_, 2, 1, X_, 2);                         \
    else if (nc == 2) WGN(B_, 1, 2, X_, 2);   /* (the flat form of this shape spills 56 registers) */ \
    else WGK(B_, 1, 1, X_, 2);                                      \
  } while (0)
  if (x_act == MTBT_ACT_GELU_POLY) WG(false, true);
  else if (dbias) WGK(true, 1, 1, false, 2);
  else WG(false, false);
#undef WG
#undef WGN
#undef WGK
  launch_wgrad_reduce(p.partial, dw, n, p.nsplit, accumulate, st);
  if (dbias) launch_wgrad_reduce(bpartial, dbias, (long)K, p.nsplit, accumulate, st);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_conv_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int pad, int stride,
                               int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride, int dtype,
                               int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  return wgrad_entry(x, dy, dw, nullptr, N, H, W, C, K, R, S, pad, stride, x_batch_stride, x_pixel_stride, dy_batch_stride, dy_pixel_stride, dtype,
                     accumulate, workspace, workspace_bytes, stream);
}

// The same with an activation applied to x while it is staged: x_act = MTBT_ACT_GELU_POLY (bf16, 1 x 1) -- x is the fc1 PRE-activation kept by
// mtbt_convnext_mlp_fused_train and dw the fc2 weight gradient  dW2[k][c] = sum_p dy[p][k] * gelu(x[p][c]).
extern "C" int mtbt_conv_wgrad_xact(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int pad, int stride,
                                    int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride, int dtype,
                                    int x_act, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  return wgrad_entry(x, dy, dw, nullptr, N, H, W, C, K, R, S, pad, stride, x_batch_stride, x_pixel_stride, dy_batch_stride, dy_pixel_stride, dtype,
                     accumulate, workspace, workspace_bytes, stream, x_act);
}

// The same plus the BIAS gradient dbias[k] (+)= sum_p dy[p][k] from the dY fragments the kernel holds anyway (no extra pass over dy).
extern "C" int mtbt_conv_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int K, int R, int S, int pad,
                                    int stride, int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride,
                                    int dtype, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!dbias) return MTBT_EINVAL;
  return wgrad_entry(x, dy, dw, dbias, N, H, W, C, K, R, S, pad, stride, x_batch_stride, x_pixel_stride, dy_batch_stride, dy_pixel_stride, dtype,
                     accumulate, workspace, workspace_bytes, stream);
}

extern "C" int64_t mtbt_stem_wgrad_workspace_bytes(int K) { return K <= 0 ? 0 : (int64_t)512 * K * 48 * (int64_t)sizeof(float); }

// Weight gradient of the ConvNeXt stem conv (4x4, stride 4, 3 -> K channels) on the caller's NCHW fp32 image: dw [K][48] (torch's
// [K,3,4,4] flattened) (+)= sum over the N*(H/4)*(W/4) output pixels of d[p][k] * patch(p); d dense [pixels][K] in `dtype`.  K % 8 == 0, K <= 128.
extern "C" int mtbt_stem_wgrad(const float* x, const void* d, float* dw, int N, int H, int W, int K, int dtype, int accumulate, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  if (!x || !d || !dw || !workspace || N <= 0 || H <= 0 || W <= 0 || H % 4 || W % 4 || K <= 0 || K % 8 || K > 128) return MTBT_EINVAL;
  if (dtype != MTBT_F32 && dtype != MTBT_BF16) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(d) || !aligned16(workspace)) return MTBT_EALIGN;
  if (workspace_bytes < mtbt_stem_wgrad_workspace_bytes(K)) return MTBT_EWORKSPACE;
  const long total = (long)N * (H / 4) * (W / 4);
  long blocks = (total + 63) / 64;
  if (blocks > 512) blocks = 512;
  const size_t lds = (size_t)64 * (48 + K) * sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
#define SW(T) hipLaunchKernelGGL((stem_wgrad_kernel<T>), dim3((unsigned)blocks), dim3(192), lds, st, x, (const T*)d, N, H, W, K, partial)
  if (dtype == MTBT_F32) SW(float); else SW(bf16_t);
#undef SW
  const int n = K * 48;
  hipLaunchKernelGGL(channel_sum_final, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, partial, (int)blocks, n, dw, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
