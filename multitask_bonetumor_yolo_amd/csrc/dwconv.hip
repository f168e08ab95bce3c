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

// A channel pair as a 2-vector: `fma2` on it is ONE v_pk_fma_f32 (left as separate .x/.y fmaf calls the SLP
// vectoriser pairs values across pixels instead and pays ~0.6 shuffle moves per packed FMA).
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

template <> struct Pair<f16_t> {
  static __device__ __forceinline__ f32x2 ld(const void* p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    return f32x2{h_lo(u), h_hi(u)};
  }
  static __device__ __forceinline__ void st(f16_t* p, f32x2 v) { *reinterpret_cast<uint32_t*>(p) = pk_h2(v.x, v.y); }
};

constexpr int CC = 128;  // channels per chunk = 64 lanes x 2


// MAXCH = ceil(C / 128) chunks held in registers.  Workgroups are PERSISTENT: each walks a strided list of tiles
// (XCD-contiguous ranges, so neighbouring tiles' halos meet in one L2); with a single chunk (C <= 128) the taps stay
// in registers across tiles.
template <typename T, int KS, bool LN, int TH, int TW, int MAXCH, int XB, int OCC = 2, bool REGT = (MAXCH == 1)>
__global__ __launch_bounds__((TH / 2) * (TW / XB) * 64, OCC) void dwconv_kernel(
    const T* __restrict__ x, const T* __restrict__ w /* [KS*KS][C] */, const float* __restrict__ bias,
    const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, const float* __restrict__ scale,
    const float* __restrict__ shift, int act, T* __restrict__ y, T* __restrict__ raw, const T* res, int N, int H, int W,
    int C, int dbg) {
  constexpr int PAD = KS / 2, YB = 2, SPAN = XB + KS - 1, ROWS = YB + KS - 1;
  constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
  constexpr int ES = (int)sizeof(T), PIXB = CC * ES;  // bytes per staged pixel
  constexpr int NW = (TH / 2) * (TW / XB);            // waves
  // staging geometry: one LDS-DMA wave-instruction = 1 KiB = PXI whole pixels of ONE halo row
  constexpr int PARTS = PIXB / 16, PXI = 64 / PARTS, IWP = ((IW + PXI - 1) / PXI) * PXI, SEGS = IWP / PXI;
  constexpr int NDMA = IH * SEGS, DPW = (NDMA + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                  // [IH][IWP][CC] T
  constexpr int TILEB = IH * IWP * PIXB;
  // LayerNorm scratch: its own region when C <= 128 (the next tile's DMA is already landing in `tile` during the
  // epilogue); with several chunks it reuses the tile (LDS would otherwise not fit two workgroups per CU)
  constexpr int NTI_ = (KS * KS + (64 / (PIXB / 16)) - 1) / (64 / (PIXB / 16));
  // single chunk: the scratch sits behind the tile (and behind the taps when those live in LDS): the next tile's DMA lands during the epilogue
  constexpr int REDOFF = MAXCH == 1 ? TILEB + (REGT ? 0 : NTI_ * 1024) : 0;
  // several chunks: the chunk's KS*KS taps ride along with the halo tile ([tap][CC] T right behind it, PXI taps per
  // DMA wave-instruction) -- read row by row from L2 instead, each filter row waited ~1 us for its taps
  constexpr int NTI = (KS * KS + PXI - 1) / PXI;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int total = N * tiles_y * tiles_x;
  const int sy = wave / (TW / XB), sx = wave % (TW / XB);  // sub-tile of this wave
  const int nchunks = (C + CC - 1) / CC;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned rowbytes = (unsigned)(W * C * ES);
  const unsigned vlane = (unsigned)((lane / PARTS) * C * ES + (lane % PARTS) * 16);  // lane's piece inside a DMA segment

  // tile walk: workgroups b, b+8, ... share an XCD; each XCD owns a contiguous range of tiles
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (total + 7) >> 3, step = gridDim.x >> 3;
  const int t_end = min(total, (xcd + 1) * per_xcd);

  // Taps (fp32 pairs of this lane's channel pair).  One chunk (C <= 128): all KS*KS stay in registers across the tiles.
  // More chunks: registers are needed for the accumulators of every chunk, so the taps are fetched row by row (KS at a
  // time, straight from L2) inside the unrolled row loop -- only two filter rows are live at once.
  constexpr bool REGTAPS = REGT;        // all KS*KS taps in registers (single-chunk default); else they ride with the halo DMA into LDS
  f32x2 wr[REGTAPS ? KS * KS : 1];
  if (REGT && lane * 2 < C) {
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wr[REGTAPS ? t : 0] = Pair<T>::ld(w + (unsigned)(t * C + lane * 2));
  }

  // per-channel epilogue vectors of this lane's channel pairs: loaded once (a load inside the epilogue would expose a
  // full memory round trip per tile)
  f32x2 e0[MAXCH], e1[MAXCH], e2[MAXCH];  // LN: bias, ln weight, ln bias ; else: scale, shift, -
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int c0 = k * CC + lane * 2;
    e0[k] = e1[k] = e2[k] = f32x2{0.f, 0.f};
    if (c0 < C) {
      if constexpr (LN) {
        e0[k] = *reinterpret_cast<const f32x2*>(bias + c0);
        e1[k] = *reinterpret_cast<const f32x2*>(lnw + c0);
        e2[k] = *reinterpret_cast<const f32x2*>(lnb + c0);
      } else {
        e0[k] = *reinterpret_cast<const f32x2*>(scale + c0);
        e1[k] = *reinterpret_cast<const f32x2*>(shift + c0);
      }
    }
  }

  // Halo tile global -> LDS by LDS-DMA (conv_dma.h), addresses on the SCALAR
  // unit: one buffer descriptor per halo row (base = that image row, num_records = its bytes); rows above / below the
  // image use an empty descriptor and columns outside it an out-of-range offset: both land as zeros.  Per instruction
  // the vector unit only adds one scalar to the lane's constant piece offset and tests its column.  Every wave issues its
  // DPW pieces back to back; the caller waits once (s_waitcnt vmcnt(0) + barrier).
  auto stage = [&](int tl_, int cb, bool with_taps) {
    const int tx_ = tl_ % tiles_x, ty_ = (tl_ / tiles_x) % tiles_y, n_ = tl_ / (tiles_x * tiles_y);
    const char* xn = reinterpret_cast<const char*>(x + (long)n_ * H * W * C);
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
      const int j = d * NW + wave_u;
      if (NDMA % NW != 0 && j >= NDMA) break;
      const int row = j / SEGS, seg = j - row * SEGS;
      const int iy = ty_ * TH + row - PAD;
      const bool rowok = (unsigned)iy < (unsigned)H;
      srd_t srd = make_srd(xn + (long)(rowok ? iy : 0) * rowbytes);
      srd.z = __builtin_amdgcn_readfirstlane(rowok ? rowbytes : 0u);
      srd.w = __builtin_amdgcn_readfirstlane(srd.w);
      const int ix0 = tx_ * TW - PAD + seg * PXI;                // first pixel of this segment (may be < 0)
      const unsigned vo = (unsigned)(ix0 + lane / PARTS) < (unsigned)W ? vlane + (unsigned)((ix0 * C + cb) * ES) : 0x80000000u;
      if (!(dbg & 2)) lds_dma16(srd, vo, 0, __builtin_amdgcn_readfirstlane(lds0 + (row * IWP + seg * PXI) * PIXB));
    }
    if (!REGT && with_taps) {
      srd_t wsrd = make_srd(w);
      wsrd.z = __builtin_amdgcn_readfirstlane((unsigned)(KS * KS * C * ES));   // taps past the last read as zeros
#pragma unroll
      for (int d = 0; d < (NTI + NW - 1) / NW; ++d) {
        const int j = d * NW + wave_u;
        if (NTI % NW != 0 && j >= NTI) break;
        lds_dma16(wsrd, vlane + (unsigned)(cb * ES), j * PXI * C * ES, __builtin_amdgcn_readfirstlane(lds0 + TILEB + j * 1024));
      }
    }
  };
  bool first = true;

  for (int tl = xcd * per_xcd + slot; tl < t_end; tl += step) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int ty0 = ty * TH, tx0 = tx * TW;

    f32x2 acc[MAXCH][YB][XB];
#pragma unroll
    for (int k = 0; k < MAXCH; ++k)
#pragma unroll
      for (int a = 0; a < YB; ++a)
#pragma unroll
        for (int i = 0; i < XB; ++i) acc[k][a][i] = f32x2{0.f, 0.f};

    auto chunk = [&](int k) {
      const int cb = k * CC;                       // chunk base channel
      const int cc = min(CC, C - cb);              // channels in this chunk (multiple of 8)
      const bool active = lane * 2 < cc;
      if (MAXCH > 1 || first) {                    // (one chunk: every later tile was requested during the previous epilogue)
        // Restaging barrier = lds_barrier(), NOT a plain __syncthreads(): the DMA below lands through the vector-memory path,
        // which is not ordered with the LDS queue, so every wave's ds_reads of the previous chunk must have RETURNED
        // (lgkmcnt(0)) before any wave restages.  A workgroup-scope __syncthreads() does not wait for outstanding LDS
        // reads, and the compiler may park the FMAs that consume them behind the barrier: the tail of an in-flight read --
        // lanes 48..63, the last 16-lane pass -- then picks up bytes of the NEXT chunk.  That is the failure the removed
        // two / three-chunk kernel showed next to MFMA kernels (LDS port contention widens the window); see DESIGN.md 4.
        lds_barrier();
        stage(tl, cb, MAXCH > 1 || first);   // (single chunk: the taps never change, staged once)
      }
      wait_vm<0>();
      __syncthreads();
      if (active && !(dbg & 1)) {
        const char* lp = tile + ((sy * YB) * IWP + sx * XB) * PIXB + lane * 2 * ES;
        const T* wb = w + cb;
        const unsigned lane2 = (unsigned)lane * 2;
        // Fully unrolled (tap registers need compile-time indices) and software-pipelined by hand: row r+1's inputs (LDS)
        // and, with several chunks, filter row r+1's taps (L2) are requested before row r's FMAs; a scheduling barrier
        // per row keeps the compiler from hoisting ALL rows' loads to the top (which spills).  One input row (SPAN
        // pairs) feeds YB output rows.
        f32x2 wrow[3][KS], in[2][SPAN];
        auto taps = [&](int ky) {
          // scalar base + one shared lane offset.  The empty asm makes the row's base opaque HERE: otherwise the addresses
          // of all 49 x chunks taps are loop-invariant, get hoisted out of the tile loop and spill.
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) wrow[ky % 3][kx] = Pair<T>::ld(smem + TILEB + (ky * KS + kx) * PIXB + lane2 * ES);
        };
        auto inputs = [&](int r) {
#pragma unroll
          for (int j = 0; j < SPAN; ++j) in[r & 1][j] = Pair<T>::ld(lp + (r * IWP + j) * PIXB);
        };
        if (!REGTAPS) taps(0);
        inputs(0);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          if (!REGTAPS && r + 1 < KS) taps(r + 1);
          if (r + 1 < ROWS) inputs(r + 1);
#pragma unroll
          for (int a = 0; a < YB; ++a) {
            const int ky = r - a;
            if (ky >= 0 && ky < KS) {
#pragma unroll
              for (int kx = 0; kx < KS; ++kx) {
                const f32x2 wv = REGTAPS ? wr[REGTAPS ? ky * KS + kx : 0] : wrow[ky % 3][kx];
#pragma unroll
                for (int i = 0; i < XB; ++i) acc[k][a][i] = fma2(in[r & 1][i + kx], wv, acc[k][a][i]);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
#pragma clang loop unroll(full)
    for (int k = 0; k < MAXCH; ++k)
      if (k < nchunks) chunk(k);     // (uniform; a `break` would keep the loop rolled and acc[k] in scratch)
    first = false;
    const int oy0 = ty0 + sy * YB, ox0 = tx0 + sx * XB;
    T* yt = y + (((long)n * H + oy0) * W + ox0) * C;     // wave-uniform base; per-lane offsets below stay 32-bit
    const unsigned rowel = (unsigned)(W * C);
    if (dbg & 4) { if (acc[0][0][0].x == 123.f) st_elem<T>(y, 0.f); if (MAXCH == 1) { __syncthreads(); if (tl + step < t_end) stage(tl + step, 0, false); } continue; }
    if (MAXCH == 1) {   // one chunk: request the next tile now, it lands while this tile's epilogue runs
      lds_barrier();    // every wave's reads of the staged tile have returned (see the restaging barrier in chunk())
      if (tl + step < t_end) stage(tl + step, 0, false);
    }
    if constexpr (LN) {
      // + bias, per-pixel statistics over all C channels (held by this wave), normalise, store
      if (MAXCH > 1) lds_barrier();    // every wave is done with the staged tile: its LDS is reused for the reductions
      float* red = reinterpret_cast<float*>(smem + REDOFF) + wave * (16 * 64 + 16);
      float s[16];  // wave_sum16 reduces 16 values; sub-tiles with fewer pixels leave the rest zero
#pragma unroll
      for (int pq = 0; pq < 16; ++pq) s[pq] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int c0 = k * CC + lane * 2;
        if (k < nchunks && c0 < C) {
          const f32x2 bv = e0[k];
#pragma unroll
          for (int a = 0; a < YB; ++a)
#pragma unroll
            for (int i = 0; i < XB; ++i) {
              acc[k][a][i] += bv;
              s[a * XB + i] += acc[k][a][i].x + acc[k][a][i].y;
            }
        }
      }
      if (raw) {   // training forward: keep the LayerNorm input (conv + bias) for the LayerNorm backward
        T* rt = raw + (((long)n * H + oy0) * W + ox0) * C;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
          const int c0 = k * CC + lane * 2;
          if (k < nchunks && c0 < C) {
#pragma unroll
            for (int a = 0; a < YB; ++a) {
              if (oy0 + a >= H) continue;
#pragma unroll
              for (int i = 0; i < XB; ++i) {
                if (ox0 + i >= W) continue;
                Pair<T>::st(rt + (a * rowel + (unsigned)(i * C + c0)), acc[k][a][i]);
              }
            }
          }
        }
      }
      wave_sum16(s, red, lane);
      const float invC = 1.0f / C;
      float q[16];
#pragma unroll
      for (int pq = 0; pq < 16; ++pq) { s[pq] *= invC; q[pq] = 0.f; }
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        if (k < nchunks && k * CC + lane * 2 < C) {
#pragma unroll
          for (int a = 0; a < YB; ++a)
#pragma unroll
            for (int i = 0; i < XB; ++i) {
              const f32x2 d = acc[k][a][i] - s[a * XB + i];
              acc[k][a][i] = d;                       // keep the centred value: the normalisation below reuses it
              q[a * XB + i] += d.x * d.x + d.y * d.y;
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
          const f32x2 gw = e1[k], gb = e2[k];
#pragma unroll
          for (int a = 0; a < YB; ++a) {
            if (oy0 + a >= H) continue;
#pragma unroll
            for (int i = 0; i < XB; ++i) {
              if (ox0 + i >= W) continue;
              Pair<T>::st(yt + (a * rowel + (unsigned)(i * C + c0)), fma2(acc[k][a][i] * q[a * XB + i], gw, gb));
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int c0 = k * CC + lane * 2;
        if (k < nchunks && c0 < C) {
          const f32x2 sc = e0[k], sh = e1[k];
#pragma unroll
          for (int a = 0; a < YB; ++a) {
            if (oy0 + a >= H) continue;
#pragma unroll
            for (int i = 0; i < XB; ++i) {
              if (ox0 + i >= W) continue;
              const f32x2 v = fma2(acc[k][a][i], sc, sh);
              f32x2 o = f32x2{act_apply(v.x, act), act_apply(v.y, act)};
              if (res) o += Pair<T>::ld(res + (((long)n * H + oy0) * W + ox0) * C + (a * rowel + (unsigned)(i * C + c0)));  // may alias y
              Pair<T>::st(yt + (a * rowel + (unsigned)(i * C + c0)), o);
            }
          }
        }
      }
    }
  }
}

template <typename T, int KS, bool LN, int TH, int TW, int MAXCH, int XB = 8, int OCC = 2, bool REGT = (MAXCH == 1)>
int launch_dw(const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps,
              const float* scale, const float* shift, int act, void* y, void* raw, const void* res, int N, int H, int W, int C, hipStream_t s) {
  constexpr int NT = (TH / 2) * (TW / XB) * 64;
  constexpr int PARTS = CC * (int)sizeof(T) / 16, PXI = 64 / PARTS, IWP = ((TW + KS - 1 + PXI - 1) / PXI) * PXI;
  constexpr int lds_tile = (TH + KS - 1) * IWP * CC * (int)sizeof(T), lds_red = LN ? (NT / 64) * (16 * 64 + 16) * 4 : 0;
  constexpr int lds_taps = REGT ? 0 : ((KS * KS + PXI - 1) / PXI) * 1024;
  constexpr int lds = MAXCH == 1 ? lds_tile + lds_taps + lds_red : (lds_tile + lds_taps > lds_red ? lds_tile + lds_taps : lds_red);
  static_assert(lds <= 160 * 1024, "LDS");
  const long tiles = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  if (tiles > 0x7fffffffL) return MTBT_EINVAL;
  // persistent workgroups: as many as stay resident (LDS-limited, at most 4 per CU), a multiple of the 8 XCDs
  constexpr int cap = OCC > 4 ? OCC : 4;
  const long resident = 256L * (160 * 1024 / lds > cap ? cap : 160 * 1024 / lds);
  long blocks = tiles < resident ? tiles : resident;
  blocks = (blocks + 7) / 8 * 8;
  auto kern = dwconv_kernel<T, KS, LN, TH, TW, MAXCH, XB, OCC, REGT>;
  if (int rc = mtbt_allow_lds(kern, lds)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, s, (const T*)x, (const T*)w, bias, lnw, lnb, eps, scale, shift,
                     act, (T*)y, (T*)raw, (const T*)res, N, H, W, C, 0 /* ablation bits: development builds only */);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

template <typename T, int KS, bool LN, int TH, int TW>
int dispatch_chunks(const void* x, const void* w, const float* bias, const float* lnw, const float* lnb, float eps,
                    const float* scale, const float* shift, int act, void* y, void* raw, const void* res, int N, int H, int W, int C,
                    hipStream_t s) {
  // (An earlier separate kernel for two / three chunks -- taps of a chunk in registers, one tile per workgroup -- was
  // 10-25 % faster on those layers but produced wrong LayerNorm outputs in lanes 48-63 when a CU was shared with an MFMA
  // kernel (tools/pair_stress.py; cause not found in its ISA) and was removed: every shape runs this kernel.)
  const int nch = (C + CC - 1) / CC;
  if (nch <= 1) return launch_dw<T, KS, LN, TH, TW, 1>(x, w, bias, lnw, lnb, eps, scale, shift, act, y, raw, res, N, H, W, C, s);
  if (nch <= 2) return launch_dw<T, KS, LN, TH, TW, 2>(x, w, bias, lnw, lnb, eps, scale, shift, act, y, raw, res, N, H, W, C, s);
  if (nch <= 3) return launch_dw<T, KS, LN, TH, TW, 3>(x, w, bias, lnw, lnb, eps, scale, shift, act, y, raw, res, N, H, W, C, s);
  if (nch <= 6) return launch_dw<T, KS, LN, TH, TW / 2, 6, 4>(x, w, bias, lnw, lnb, eps, scale, shift, act, y, raw, res, N, H, W, C, s);
  return MTBT_EINVAL;
}

}  // namespace

// w: [k*k][C] in the activation dtype (bf16 taps in bf16 mode, like every other conv's weights).
static int dwconv_entry(const void* x, const void* w, const float* bias, const float* ln_w, const float* ln_b, float ln_eps, const float* scale,
                        const float* shift, int act, void* y, void* raw, const void* res, int N, int H, int W, int C, int ksize, int dtype,
                        void* stream) {
  if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C > 768) return MTBT_EINVAL;
  if (ksize != 3 && ksize != 7) return MTBT_EINVAL;
  const bool ln = ln_w != nullptr;
  if (ln && (!ln_b || !bias || res)) return MTBT_EINVAL;
  if (!ln && (!scale || !shift || raw)) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(w) || (raw && !aligned16(raw)) || (res && !aligned16(res))) return MTBT_EALIGN;
  if ((long)(W + 64) * C * 4 >= 0x7fff0000L || (long)H * W * C >= 0x7fff0000L) return MTBT_EINVAL;  // 32-bit offsets in a row / an image
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DW_ARGS x, w, bias, ln_w, ln_b, ln_eps, scale, shift, act, y, raw, res, N, H, W, C, s
  if (dtype == MTBT_BF16) {
    if (ksize == 7) return ln ? dispatch_chunks<bf16_t, 7, true, 4, 16>(DW_ARGS) : dispatch_chunks<bf16_t, 7, false, 4, 16>(DW_ARGS);
    return ln ? dispatch_chunks<bf16_t, 3, true, 4, 16>(DW_ARGS) : dispatch_chunks<bf16_t, 3, false, 4, 16>(DW_ARGS);
  } else if (dtype == MTBT_F16) {
    if (ksize == 7) return ln ? dispatch_chunks<f16_t, 7, true, 4, 16>(DW_ARGS) : dispatch_chunks<f16_t, 7, false, 4, 16>(DW_ARGS);
    return ln ? dispatch_chunks<f16_t, 3, true, 4, 16>(DW_ARGS) : dispatch_chunks<f16_t, 3, false, 4, 16>(DW_ARGS);
  } else if (dtype == MTBT_F32) {
    if (ksize == 7) return ln ? dispatch_chunks<float, 7, true, 4, 8>(DW_ARGS) : dispatch_chunks<float, 7, false, 4, 8>(DW_ARGS);
    return ln ? dispatch_chunks<float, 3, true, 4, 8>(DW_ARGS) : dispatch_chunks<float, 3, false, 4, 8>(DW_ARGS);
  }
#undef DW_ARGS
  return MTBT_EINVAL;
}

extern "C" int mtbt_dwconv_nhwc(const void* x, const void* w, const float* bias, const float* ln_w, const float* ln_b,
                                float ln_eps, const float* scale, const float* shift, int act, void* y, int N, int H,
                                int W, int C, int ksize, int dtype, void* stream) {
  return dwconv_entry(x, w, bias, ln_w, ln_b, ln_eps, scale, shift, act, y, nullptr, nullptr, N, H, W, C, ksize, dtype, stream);
}

// Training variants: `raw` (LayerNorm form only) also receives the LayerNorm INPUT conv + bias (what the LayerNorm backward needs);
// `res` (scale / shift form only) is added after the activation and may alias y (gradient accumulation of the depthwise dgrad).
extern "C" int mtbt_dwconv_nhwc_train(const void* x, const void* w, const float* bias, const float* ln_w, const float* ln_b,
                                      float ln_eps, const float* scale, const float* shift, int act, void* y, void* raw, const void* res,
                                      int N, int H, int W, int C, int ksize, int dtype, void* stream) {
  return dwconv_entry(x, w, bias, ln_w, ln_b, ln_eps, scale, shift, act, y, raw, res, N, H, W, C, ksize, dtype, stream);
}
