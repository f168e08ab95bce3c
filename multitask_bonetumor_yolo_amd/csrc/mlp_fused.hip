// Fused ConvNeXt MLP for the wide-map stages (d = 96, 192), bf16:
//
//     y[p][:] = res[p][:] + W2' . GELU(W1 . t[p][:] + b1) + b2'          (timm Mlp fc1 -> GELU -> fc2, layer-scale folded)
//
// Unfused, the 4d-wide hidden tensor makes two HBM round trips (16 x 160^2 x 384 bf16 = 315 MB in stage 0) and both
// 1x1 convs sit on the HBM roof (77 FLOP/B).  Here a workgroup keeps its pixels' hidden activations ON CHIP:
//
//   * a wave owns FP x 16 pixels; their fc1 inputs t[p][0..d) live in REGISTERS for the whole kernel as MFMA B fragments
//     (loaded straight from global: a lane's 16-byte piece of a pixel row IS its fragment);
//   * the hidden dimension is walked in chunks of 32: GEMM1 (rows = 32 hidden units of W1, k = d) gives, per lane, 4 + 4
//     consecutive hidden units of one pixel -- after bias + GELU + bf16 rounding these 8 values ARE the B fragment of
//     GEMM2 (k = 32 hidden units) if fc2's weight columns are stored in the matching order (packed once on the host:
//     slot 8g+j <- hidden 4g+j for j < 4, 16+4g+(j-4) otherwise).  No LDS round trip, no transposition;
//   * GEMM2 (rows = d output channels) accumulates y in registers over all chunks;
//   * only the weights stream: per chunk W1_j [32][d] and W2'_j [d][32] go global -> LDS by LDS-DMA (three LDS stages,
//     counted vmcnt, conv_dma.h) as 64-byte-row slabs with the conv kernel's swizzle; per wave and chunk that is
//     6-12 ds_read_b128 for 48 MFMAs;
//   * the residual initialises the output accumulators; epilogue = + b2', bf16, 8-byte stores from the accumulators.
//
// HBM traffic per pixel: read t, read res, write y (3 x 2d bytes) instead of 2d + 8d + 8d + 2d + 2d.
#include <cstdlib>

#include "common.h"
#include "conv_params.h"
#include "conv_dma.h"
#include "conv_epilogue.h"

namespace {

// two floats -> packed bf16 pair, one v_cvt_pk_bf16_f32 (round to nearest even)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// gelu_poly (common.h) on a pair: the clamp has no packed form, the 9 multiplies / FMAs do (v_pk_mul_f32 / v_pk_fma_f32) -- the GELU of
// the 4d hidden units is ~11 vector operations per element as scalar FMAs
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t x) {
  const f32x2_t xc = f32x2_t{__builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f)};
  const f32x2_t s = xc * xc;
  auto k = [](float c) { return f32x2_t{c, c}; };
  f32x2_t p = __builtin_elementwise_fma(k(2.1609857e-08f), s, k(-1.5335673e-06f));
  p = __builtin_elementwise_fma(p, s, k(4.6542096e-05f));
  p = __builtin_elementwise_fma(p, s, k(-7.9887325e-04f));
  p = __builtin_elementwise_fma(p, s, k(8.6900834e-03f));
  p = __builtin_elementwise_fma(p, s, k(-6.4366050e-02f));
  p = __builtin_elementwise_fma(p, s, k(3.9770728e-01f));
  return x * __builtin_elementwise_fma(xc, p, k(0.5f));
}

// 64-byte-row swizzle of conv_igemm.inc (CPR = 4)
__device__ __forceinline__ int swz4(int row) { return (-(row >> 2)) & 3; }

// development ablation bits of the resident kernel: compiled in only with -DMTBT_MLP_ABLATION (tools/probes/mlp_variants.hip)
#ifdef MTBT_MLP_ABLATION
#define MLP_ABL(p, bit) ((p).dbg & (bit))
#else
#define MLP_ABL(p, bit) 0
#endif

#ifndef MTBT_MLP_RESIDENT
#define MTBT_MLP_RESIDENT 0     // 1: d = 96 calls of >= 64 K pixels take the weight-resident persistent kernel (measured slower so far)
#endif

struct MlpP {
  const bf16_t* t;     // [M][D] fc1 input (LayerNorm output)
  const bf16_t* w1;    // [4D][D]
  const float* b1;     // [4D]
  const bf16_t* w2p;   // [D][4D], hidden order permuted per 32-chunk (see top)
  const bf16_t* res;   // [M][D] residual (may be NULL)
  bf16_t* hpre;        // HP kernels (training): [M][4D] fc1 PRE-activation, written in natural hidden order -- see mtbt_convnext_mlp_fused_train
  int M;
  int dbg;             // development ablation bits (MTBT_MLP_DEBUG): 1 = no GELU, 2 = no GEMM2, 4 = no GEMM1, 8 = no weight DMA
  ConvP ep;            // epilogue view: shift = b2', res, y, K = D
};

// HT = bf16_t or f16_t: the MFMA flavour and the fp32 <-> 16-bit conversions (everything else is byte-identical)
template <typename HT> __device__ __forceinline__ uint32_t pk2(float a, float b);
template <> __device__ __forceinline__ uint32_t pk2<bf16_t>(float a, float b) { return pk_bf16(a, b); }
template <> __device__ __forceinline__ uint32_t pk2<f16_t>(float a, float b) { return pk_h2(a, b); }
// Output-channel order of GEMM2.  An accumulator fragment gives a lane 4 consecutive rows (4 lq .. 4 lq + 3): stored as they are, 8 bytes of a pixel
// row per lane -- 24 dwordx2 stores (and as many residual loads) per lane at d = 384.  The STAGED row order of W2' is therefore permuted (a free
// change of the DMA source offsets): LDS row 16 F + 4 q + e holds output channel 32 (F / 2) + 8 q + 4 (F % 2) + e, so that a lane's rows of the
// fragment pair (2 m, 2 m + 1) are the 8 CONSECUTIVE channels 32 m + 8 q .. + 7: one 16-byte store / residual load per pair, 64 contiguous bytes
// of a pixel row per four lanes.  The caller's w2p / b2 / y stay in natural channel order.
__device__ __forceinline__ int w2_row_channel(int r) {
  const int F = r >> 4, q = (r >> 2) & 3, e = r & 3;
  return 32 * (F >> 1) + 8 * q + 4 * (F & 1) + e;
}
template <typename HT> __device__ __forceinline__ f32x4 unpack4(uint2 r);
template <> __device__ __forceinline__ f32x4 unpack4<bf16_t>(uint2 r) {
  return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ f32x4 unpack4<f16_t>(uint2 r) { return f32x4{h_lo(r.x), h_hi(r.x), h_lo(r.y), h_hi(r.y)}; }

// PIPE (wide d, few waves per SIMD): inside a chunk the wave's two pixel blocks run as a two-stage pipeline -- GEMM1(block 0), then GEMM1(block 1)
// NEXT TO the GELU of block 0 (independent: the matrix pipe and the VALU overlap, hipcc interleaves them when they are in one
// scheduling region), the GELU of block 1, GEMM2 of both.  The unpipelined order is GEMM1 -> GELU -> GEMM2 with nothing beside the
// GELU's ~130 VALU instructions, and at one wave per SIMD (d = 384) no other wave fills that hole (ISA of round 2: 96 MFMAs, then 260
// VALU, then 96 MFMAs per chunk).  The chunk's W1 fragments stay in registers for both blocks (one LDS read per two MFMAs as before).
template <int D, int FP, int WPS, typename HT, bool PIPE = false, bool HP = false>
__global__ __launch_bounds__(256, WPS) void mlp_fused_kernel(const MlpP p) {
  constexpr int KS1 = D / 32;            // k-steps of GEMM1
  constexpr int FC = D / 16;             // output-channel fragments of GEMM2
  constexpr int NCH = 4 * D / 32;        // hidden chunks
  constexpr int PW = FP * 16, P = 4 * PW;  // pixels per wave / workgroup
  constexpr int W1B = 32 * D * 2, W2B = D * 64;  // bytes of one chunk's weight tiles
  constexpr int STAGE = W1B + W2B;
  constexpr int NDMA = STAGE / 1024, DPW = NDMA / 4;  // LDS-DMA instructions per stage, per wave
  static_assert(NDMA % 4 == 0 && W1B % 1024 == 0, "whole wave-instructions");
  constexpr int NBUF = 3;                // LDS stages: two chunks' weights in flight behind the one being multiplied
  static_assert(NCH % NBUF == 0, "chunk loop is unrolled by the stage count");
  constexpr int AFF_OFF = NBUF * STAGE;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* aff = reinterpret_cast<float*>(smem + AFF_OFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int pbase = blockIdx.x * P + wave * PW;
  stage_affine<D>(p.ep, aff, 0, tid);
  // fc1 bias: LDS copy (a compiler-managed global load inside the chunk loop would make hipcc wait vmcnt(0) at its use and
  // drain the weight prefetch that is in flight)
  float* b1s = aff + 2 * D;
  for (int c = tid; c < 4 * D; c += 256) b1s[c] = p.b1[c];

  // ---- weight staging: constant per-lane source offsets, scalar chunk offsets ----
  // stage layout: W1_j as KS1 slabs [32 rows][64 B], then W2'_j as [D rows][64 B]; piece c of a slab -> row c/4,
  // slot c%4 fed with source chunk (c%4) ^ swz4(row) (swizzle on the source side, LDS image lane-linear).
  const srd_t w1srd = make_srd(p.w1), w2srd = make_srd(p.w2p);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned voff[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int inst = i * 4 + wave;                       // wave-instruction index inside the stage (scalar)
    const int c = inst * 64 + lane;                      // 16-byte piece index inside the stage
    if (inst * 1024 < W1B) {                             // (W1B is a multiple of 1 KiB: an instruction never straddles)
      const int slab = c / 128, r = (c % 128) / 4, q = (c % 4) ^ swz4(r);
      voff[i] = (unsigned)(r * D * 2 + slab * 64 + q * 16);          // + chunk * 32 rows * D * 2 (scalar)
    } else {
      const int c2 = c - W1B / 16;
      const int r = c2 / 4, q = (c2 % 4) ^ swz4(r);
      voff[i] = (unsigned)(w2_row_channel(r) * 4 * D * 2 + q * 16);  // + chunk * 64 (scalar); row order: w2_row_channel
    }
  }
  auto stage = [&](int j, int buf) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int inst = i * 4 + wave;
      const unsigned dst = lds0 + buf * STAGE + inst * 1024;
      if (inst * 1024 < W1B) lds_dma16(w1srd, voff[i], j * 32 * D * 2, dst);
      else lds_dma16(w2srd, voff[i], j * 64, dst);
    }
  };

  // ---- this wave's fc1 inputs as B fragments, straight from global (zeros past M) ----
  uint4 tf[FP][KS1];
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int pix = pbase + f * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
      tf[f][ks] = (pix < p.M && !(p.dbg & 64)) ? *reinterpret_cast<const uint4*>(p.t + (long)pix * D + ks * 32 + lq * 8) : uint4{0u, 0u, 0u, 0u};
  }

  // The output accumulators START as the residual: a lane's 4 channels of a pixel are 8 contiguous bytes of the residual
  // row, requested here (their latency hides under the whole chunk loop) -- the epilogue then has no global load in its
  // LDS-slab chain (with one it was 80 of the kernel's 166 us at d = 96: four dependent load round trips per workgroup).
  f32x4 acc2[FC][FP];
  static_assert(FC % 2 == 0, "fragment pairs");
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int pix = pbase + f * 16 + lr;
#pragma unroll
    for (int m = 0; m < FC / 2; ++m) {        // (w2_row_channel: the pair's rows of this lane = channels 32 m + 8 lq .. + 7)
      uint4 r = uint4{0u, 0u, 0u, 0u};
      if (p.res && pix < p.M) r = *reinterpret_cast<const uint4*>(p.res + (long)pix * D + m * 32 + lq * 8);
      acc2[2 * m][f] = unpack4<HT>(uint2{r.x, r.y});
      acc2[2 * m + 1][f] = unpack4<HT>(uint2{r.z, r.w});
    }
  }

  // fragment read offsets inside a stage
  int a1off[KS1], a2off;
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) a1off[ks] = ks * 2048 + lr * 64 + ((lq ^ swz4(lr)) << 4);   // + blk * 16 rows * 64
  a2off = W1B + lr * 64 + ((lq ^ swz4(lr)) << 4);                                              // + i * 16 rows * 64

  unsigned hpoff[HP ? FP : 1];            // HP: byte offset of (pixel, hidden 8 lq) in hpre (address = scalar chunk base + this; M * 8 D < 2^32)
  if constexpr (HP) {
#pragma unroll
    for (int f = 0; f < FP; ++f) hpoff[f] = (unsigned)(pbase + f * 16 + lr) * (unsigned)(8 * D) + (unsigned)(lq * 16);
  }
  stage(0, 0);
  stage(1, 1);
#pragma unroll 1
  for (int j = 0; j < NCH; j += NBUF) {
#pragma unroll
    for (int u = 0; u < NBUF; ++u) {  // unrolled by the LDS stages: buffer offsets are immediates
      const int jj = j + u;
      // this wave's pieces of chunk jj have landed (the next chunk's may still be in flight; the tf loads are older)
      // (HP: the previous chunk's FP pre-activation stores are younger than that DMA and count in vmcnt too)
      if (jj + 1 < NCH) wait_vm<DPW + (HP ? FP : 0)>(); else wait_vm<0>();
      lds_barrier();                  // everyone's have, and everyone is done reading the buffer restaged next
      if (jj + 2 < NCH && !(p.dbg & 8)) stage(jj + 2, (u + 2) % NBUF);
      const char* st = smem + u * STAGE;
      // bias of this chunk's 32 hidden units: rows 4 lq .. +3 of block a and of block b
      const float4 ba = *reinterpret_cast<const float4*>(b1s + jj * 32 + lq * 4);
      const float4 bb = *reinterpret_cast<const float4*>(b1s + jj * 32 + 16 + lq * 4);
      if constexpr (PIPE) {
        static_assert(FP == 2, "two pixel blocks per wave");
        // a lane's 4 hidden units of one 16-row block -> bias + GELU -> two packed pairs (half of GEMM2's B fragment)
        uint4 hp0, hp1;                 // HP: the packed PRE-activations of the lane's 4 + 4 hidden units (one 16-byte store per pixel block)
        auto gelu_half = [&](const f32x4& hv, const float4& bv, uint32_t& lo, uint32_t& hi, uint32_t& plo, uint32_t& phi) {
          f32x2_t a01 = f32x2_t{hv[0] + bv.x, hv[1] + bv.y}, a23 = f32x2_t{hv[2] + bv.z, hv[3] + bv.w};
          if constexpr (HP) { plo = pk2<HT>(a01.x, a01.y); phi = pk2<HT>(a23.x, a23.y); }
          a01 = gelu_poly2(a01); a23 = gelu_poly2(a23);
          lo = pk2<HT>(a01.x, a01.y); hi = pk2<HT>(a23.x, a23.y);
        };
        uint4 hb0, hb1;
        f32x4 h0a = f32x4{0.f, 0.f, 0.f, 0.f}, h0b = h0a, h1a = h0a, h1b = h0a;
        {   // region 1: hidden rows 0..15 of the chunk ("a" block of W1) for both pixel blocks
          uint4 wa[KS1];
#pragma unroll
          for (int ks = 0; ks < KS1; ++ks) wa[ks] = *reinterpret_cast<const uint4*>(st + a1off[ks]);
#pragma unroll
          for (int ks = 0; ks < KS1; ++ks) {
            h0a = mfma_16x16x32<HT>(wa[ks], tf[0][ks], h0a);
            h1a = mfma_16x16x32<HT>(wa[ks], tf[1][ks], h1a);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, KS1, 0);      // all twelve fragment reads in flight, then the MFMAs as they land
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * KS1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the "a" accumulators become visible HERE: without this the GELU below -- pure arithmetic, free to float in the DAG -- is emitted
        //  above the scheduling barrier, behind the last MFMAs of region 1, and nothing is left to put beside region 2's MFMAs)
#pragma unroll
        for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(h0a[e])); asm volatile("" : "+v"(h1a[e])); }
        {   // region 2: hidden rows 16..31 ("b" block) beside the GELU of the "a" rows: one MFMA, then up to three VALU instructions
          uint4 wb[KS1];
#pragma unroll
          for (int ks = 0; ks < KS1; ++ks) wb[ks] = *reinterpret_cast<const uint4*>(st + a1off[ks] + 1024);
#pragma unroll
          for (int ks = 0; ks < KS1; ++ks) {
            h0b = mfma_16x16x32<HT>(wb[ks], tf[0][ks], h0b);
            h1b = mfma_16x16x32<HT>(wb[ks], tf[1][ks], h1b);
          }
          gelu_half(h0a, ba, hb0.x, hb0.y, hp0.x, hp0.y);
          gelu_half(h1a, ba, hb1.x, hb1.y, hp1.x, hp1.y);
          // schedule of the region: all fragment reads first, a few GELU instructions while they fly, then one MFMA and up to three VALU
          // instructions, 2 * KS1 times (the GELU rides in the issue slots the MFMAs leave: ~66 VALU for 24 MFMAs)
          __builtin_amdgcn_sched_group_barrier(0x100, KS1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
#pragma unroll
          for (int g = 0; g < 2 * KS1; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // region 3: GELU of the "b" rows (nothing to put beside it inside this chunk), then GEMM2 of both blocks: one weight fragment, two
        // MFMAs, the fragments requested PD steps ahead (left to itself hipcc reads a fragment, waits, issues its two MFMAs)
        gelu_half(h0b, bb, hb0.z, hb0.w, hp0.z, hp0.w);
        gelu_half(h1b, bb, hb1.z, hb1.w, hp1.z, hp1.w);
        if constexpr (HP) {
          if (pbase + lr < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.hpre) + (size_t)(jj * 64) + (size_t)hpoff[0]) = hp0;
          if (pbase + 16 + lr < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.hpre) + (size_t)(jj * 64) + (size_t)hpoff[1]) = hp1;
        }
        constexpr int PD = 4;
        uint4 w2q[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) w2q[i] = *reinterpret_cast<const uint4*>(st + a2off + i * 1024);
#pragma unroll
        for (int i = 0; i < FC; ++i) {
          const uint4 w2 = w2q[i % PD];
          if (i + PD < FC) w2q[i % PD] = *reinterpret_cast<const uint4*>(st + a2off + (i + PD) * 1024);
          acc2[i][0] = mfma_16x16x32<HT>(w2, hb0, acc2[i][0]);
          acc2[i][1] = mfma_16x16x32<HT>(w2, hb1, acc2[i][1]);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);         // the first PD fragments requested before the GELU, so they land under it
        __builtin_amdgcn_sched_group_barrier(0x006, 400, 0);        // the GELU of the "b" rows
#pragma unroll
        for (int i = 0; i < FC; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else {
      // GEMM1: hidden[32][PW] = W1_j . t
      f32x4 h[2][FP];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int f = 0; f < FP; ++f) h[b][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!(p.dbg & 4))
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const uint4 wa = *reinterpret_cast<const uint4*>(st + a1off[ks]);
        const uint4 wb = *reinterpret_cast<const uint4*>(st + a1off[ks] + 1024);
#pragma unroll
        for (int f = 0; f < FP; ++f) {
          h[0][f] = mfma_16x16x32<HT>(wa, tf[f][ks], h[0][f]);
          h[1][f] = mfma_16x16x32<HT>(wb, tf[f][ks], h[1][f]);
        }
      }
      // bias + GELU + bf16: the lane's 4 + 4 hidden units of pixel lr are GEMM2's B fragment (see top)
      uint4 hb[FP];
#pragma unroll
      for (int f = 0; f < FP; ++f) {
        f32x2_t a01 = f32x2_t{h[0][f][0] + ba.x, h[0][f][1] + ba.y}, a23 = f32x2_t{h[0][f][2] + ba.z, h[0][f][3] + ba.w};
        f32x2_t c01 = f32x2_t{h[1][f][0] + bb.x, h[1][f][1] + bb.y}, c23 = f32x2_t{h[1][f][2] + bb.z, h[1][f][3] + bb.w};
        if constexpr (HP) {             // the pre-activation of the lane's 8 CONSECUTIVE hidden units (training weight order): one 16-byte store
          if (pbase + f * 16 + lr < p.M)
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.hpre) + (size_t)(jj * 64) + (size_t)hpoff[f]) =
                uint4{pk2<HT>(a01.x, a01.y), pk2<HT>(a23.x, a23.y), pk2<HT>(c01.x, c01.y), pk2<HT>(c23.x, c23.y)};
        }
        if (!(p.dbg & 1)) { a01 = gelu_poly2(a01); a23 = gelu_poly2(a23); c01 = gelu_poly2(c01); c23 = gelu_poly2(c23); }   // packed fp32 pairs
        hb[f].x = pk2<HT>(a01.x, a01.y); hb[f].y = pk2<HT>(a23.x, a23.y);
        hb[f].z = pk2<HT>(c01.x, c01.y); hb[f].w = pk2<HT>(c23.x, c23.y);
      }
      // GEMM2: y[D][PW] += W2'_j . hidden
      if (!(p.dbg & 2))
#pragma unroll
      for (int i = 0; i < FC; ++i) {
        const uint4 w2 = *reinterpret_cast<const uint4*>(st + a2off + i * 1024);
#pragma unroll
        for (int f = 0; f < FP; ++f)
          acc2[i][f] = mfma_16x16x32<HT>(w2, hb[f], acc2[i][f]);
      }
      }
    }
  }
  // Epilogue: + b2' and 16-byte stores STRAIGHT from the accumulators (w2_row_channel: a lane owns 8 consecutive channels of a pixel per
  // fragment pair; the four lq groups of a pixel write 64 contiguous bytes, the wave's FC / 2 passes complete the 2d-byte row).
  // Measured against the conv kernels' LDS-slab epilogue (16-byte stores after a transposition through LDS): 23 us
  // instead of 57 us of the kernel's time at d = 96 -- the slab's serialized LDS round trips cost more than the wider
  // stores save.
  if (p.dbg & 32) { if (acc2[0][0][0] == 123.f) reinterpret_cast<bf16_t*>(p.ep.y)[0] = 0; return; }
  bf16_t* yb = reinterpret_cast<bf16_t*>(p.ep.y);
#pragma unroll
  for (int m = 0; m < FC / 2; ++m) {
    const float4 s0 = *reinterpret_cast<const float4*>(aff + D + m * 32 + lq * 8), s1 = *reinterpret_cast<const float4*>(aff + D + m * 32 + lq * 8 + 4);
#pragma unroll
    for (int f = 0; f < FP; ++f) {
      const int pix = pbase + f * 16 + lr;
      if (pix >= p.M) continue;
      u32x4 o;
      o.x = pk2<HT>(acc2[2 * m][f][0] + s0.x, acc2[2 * m][f][1] + s0.y);
      o.y = pk2<HT>(acc2[2 * m][f][2] + s0.z, acc2[2 * m][f][3] + s0.w);
      o.z = pk2<HT>(acc2[2 * m + 1][f][0] + s1.x, acc2[2 * m + 1][f][1] + s1.y);
      o.w = pk2<HT>(acc2[2 * m + 1][f][2] + s1.z, acc2[2 * m + 1][f][3] + s1.w);
      *reinterpret_cast<u32x4*>(yb + (long)pix * D + m * 32 + lq * 8) = o;
    }
  }
}

template <int D, int FP, int WPS, typename HT, bool PIPE = false, bool HP = false>
int launch_mlp(const MlpP& p, hipStream_t s) {
  constexpr int P = 4 * FP * 16;
  constexpr int STAGE = 32 * D * 2 + D * 64;
  constexpr int lds = 3 * STAGE + 2 * D * 4 + 4 * D * 4;
  static_assert(lds <= 160 * 1024, "LDS");
  const long blocks = ((long)p.M + P - 1) / P;
  if (blocks <= 0 || blocks > 0x7fffffffL) return MTBT_EINVAL;
  if (int rc = mtbt_allow_lds(mlp_fused_kernel<D, FP, WPS, HT, PIPE, HP>, lds)) return rc;
  hipLaunchKernelGGL((mlp_fused_kernel<D, FP, WPS, HT, PIPE, HP>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Wide d (384): the PAIR form.  In the kernel above a wave carries its 32 pixels alone: 192 accumulator + 96 input registers leave room for
// ONE wave per SIMD, and at one wave per SIMD nothing covers an LDS fragment read, a GELU or the wait at a barrier -- its MFMA phases run
// at ~58 % of the matrix pipe's rate (ISA of round 3: every fragment read is waited for one or two MFMAs later) and skeleton, weight DMA,
// GELU and MFMA time simply add up (127 us per call where the MFMAs alone need 35).  Here TWO waves share a 32-pixel block:
//   * GEMM1: wave hh of the pair computes hidden rows 16 hh .. 16 hh + 15 of the chunk (half of the chunk's MFMAs);
//   * GELU on its 4 + 4 values, packed: these are the .xy (hh = 0) or .zw (hh = 1) words of GEMM2's B fragment FOR THE SAME LANE of both
//     waves, so the exchange is 8 bytes per lane and pixel block through LDS, published by the chunk's barrier;
//   * GEMM2: wave hh accumulates output channels hh * d/2 .. (hh + 1) * d/2 - 1 (half the accumulators: 96 registers).
// 8 waves per workgroup = TWO per SIMD (<= 256 registers each), and inside a wave the loop is skewed by one chunk -- iteration j runs
// GEMM1(j), then GEMM2(j - 1) beside the GELU of chunk j -- so a wave's own VALU work also has MFMAs next to it.  A stage of the weight
// stream holds W1(j) and W2'(j - 1); two stages.  Same arithmetic, same order of accumulation per output as the kernel above.
template <int D, typename HT, bool ORD = false, bool HP = false>
__global__ __launch_bounds__(512, 2) void mlp_pair_kernel(const MlpP p) {
  constexpr int FP = 2;
  constexpr int KS1 = D / 32, FCH = D / 32, NCH = 4 * D / 32;      // FCH: output-channel fragments of ONE wave (half of d / 16)
  constexpr int PW = FP * 16, P = 4 * PW;
  constexpr int W1B = 32 * D * 2, W2B = D * 64, STAGE = W1B + W2B;
  constexpr int NDMA = STAGE / 1024, DPW = NDMA / 8;
  static_assert(NDMA % 8 == 0 && W1B % 1024 == 0, "whole wave-instructions per wave");
  constexpr int XCH_OFF = 2 * STAGE, XCH_BYTES = 2 * 4 * 2 * FP * 64 * 8;                    // [parity][pair][half][block][lane] x 8 B
  constexpr int AFF_OFF = XCH_OFF + XCH_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* aff = reinterpret_cast<float*>(smem + AFF_OFF);           // [D] b2'
  float* b1s = aff + D;                                            // [4D] b1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 1, hh = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int pbase = blockIdx.x * P + g * PW;
  for (int c = tid; c < D; c += 512) aff[c] = p.ep.shift[c];
  for (int c = tid; c < 4 * D; c += 512) b1s[c] = p.b1[c];

  const srd_t w1srd = make_srd(p.w1), w2srd = make_srd(p.w2p);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned voff[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int inst = i * 8 + wave;
    const int c = inst * 64 + lane;
    if (inst * 1024 < W1B) {
      const int slab = c / 128, r = (c % 128) / 4, sl = (c % 4) ^ swz4(r);
      voff[i] = (unsigned)(r * D * 2 + slab * 64 + sl * 16);
    } else {
      const int c2 = c - W1B / 16;
      const int r = c2 / 4, sl = (c2 % 4) ^ swz4(r);
      voff[i] = (unsigned)(w2_row_channel(r) * 4 * D * 2 + sl * 16);
    }
  }
  auto stage = [&](int j, int buf) {     // stage j = W1 of chunk j (j < NCH) and W2' of chunk j - 1 (j >= 1)
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int inst = i * 8 + wave;
      const unsigned dst = lds0 + buf * STAGE + inst * 1024;
      if (inst * 1024 < W1B) { if (j < NCH) lds_dma16(w1srd, voff[i], j * 32 * D * 2, dst); }
      else { if (j >= 1) lds_dma16(w2srd, voff[i], (j - 1) * 64, dst); }
    }
  };

  uint4 tf[FP][KS1];
  f32x4 acc2[FCH][FP];
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int pix = pbase + f * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
      tf[f][ks] = (pix < p.M && !MLP_ABL(p, 64)) ? *reinterpret_cast<const uint4*>(p.t + (long)pix * D + ks * 32 + lq * 8) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int m = 0; m < FCH / 2; ++m) {       // (w2_row_channel: this lane's rows of the fragment pair = 8 consecutive channels)
      uint4 r = uint4{0u, 0u, 0u, 0u};
      if (p.res && pix < p.M) r = *reinterpret_cast<const uint4*>(p.res + (long)pix * D + (hh * FCH / 2 + m) * 32 + lq * 8);
      acc2[2 * m][f] = unpack4<HT>(uint2{r.x, r.y});
      acc2[2 * m + 1][f] = unpack4<HT>(uint2{r.z, r.w});
    }
  }
  int a1off[KS1];
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) a1off[ks] = ks * 2048 + hh * 1024 + lr * 64 + ((lq ^ swz4(lr)) << 4);      // W1 rows 16 hh + lr
  const int a2off = W1B + (hh * FCH * 16 + lr) * 64 + ((lq ^ swz4(lr)) << 4);                                   // W2' rows hh * d/2 + i * 16 + lr
  uint2* const xch = reinterpret_cast<uint2*>(smem + XCH_OFF);
  const int xmine = ((g * 2 + hh) * FP) * 64 + lane, xpeer = ((g * 2 + (1 - hh)) * FP) * 64 + lane;            // + parity * 4 * 2 * FP * 64 + f * 64

  // GEMM2 of one chunk over this wave's half of the output channels
  auto gemm2 = [&](const char* st, const uint4 (&hb)[FP]) {
    if (MLP_ABL(p, 2)) return;
#pragma unroll
    for (int i = 0; i < FCH; ++i) {
      const uint4 w2 = *reinterpret_cast<const uint4*>(st + a2off + i * 1024);
#pragma unroll
      for (int f = 0; f < FP; ++f) acc2[i][f] = mfma_16x16x32<HT>(w2, hb[f], acc2[i][f]);
    }
  };
  // Waves w and w + 4 share a SIMD and, started together behind the same barrier, would run the same phase at the same time: both in
  // MFMAs (the pipe is shared anyway), then both in GELU / exchange / barrier wait (the pipe idles).  The second half of the workgroup
  // therefore runs the iteration in the other order -- GEMM2(j - 1), GEMM1(j), GELU -- so that one wave's VALU tail has the other's
  // MFMAs beside it.  Same operations on the same data: the order inside an iteration is free (everything crosses iterations at the barrier).
  const bool ord = ORD && (wave & 4);
  uint2 mine[FP];
#pragma unroll
  for (int f = 0; f < FP; ++f) mine[f] = uint2{0u, 0u};
  unsigned hpoff[HP ? FP : 1];            // HP: byte offset of (pixel, hidden 8 lq + 4 hh) in hpre; the host keeps M * 8 D below 2^32
  if constexpr (HP) {
#pragma unroll
    for (int f = 0; f < FP; ++f) hpoff[f] = (unsigned)(pbase + f * 16 + lr) * (unsigned)(8 * D) + (unsigned)(lq * 16 + hh * 8);
  }
  stage(0, 0);
#pragma unroll 1
  for (int j0 = 0; j0 <= NCH; j0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {      // unrolled by the two LDS stages / exchange parities: offsets are immediates
      const int j = j0 + u;
      if (j > NCH) break;
      wait_vm<0>();                    // my pieces of stage j have landed
      lds_barrier();                   // everyone's have; the halves written in iteration j - 1 are visible; the other stage is free
      if (j + 1 <= NCH && !MLP_ABL(p, 8)) stage(j + 1, (u + 1) & 1);
      const char* st = smem + u * STAGE;
      uint4 hb[FP];
      if (j >= 1) {                    // GEMM2's B fragment of chunk j - 1: .xy from the pair's wave 0, .zw from its wave 1
#pragma unroll
        for (int f = 0; f < FP; ++f) {
          const uint2 peer = xch[((u + 1) & 1) * (4 * 2 * FP * 64) + xpeer + f * 64];
          hb[f] = hh == 0 ? uint4{mine[f].x, mine[f].y, peer.x, peer.y} : uint4{peer.x, peer.y, mine[f].x, mine[f].y};
        }
      }
      f32x4 h[FP];
      const bool g2_first = ord && j >= 1;      // see `ord` above
      if (g2_first) gemm2(st, hb);
      if (j < NCH) {                   // GEMM1: this wave's 16 hidden rows of chunk j
#pragma unroll
        for (int f = 0; f < FP; ++f) h[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!MLP_ABL(p, 4))
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const uint4 w = *reinterpret_cast<const uint4*>(st + a1off[ks]);
#pragma unroll
          for (int f = 0; f < FP; ++f) h[f] = mfma_16x16x32<HT>(w, tf[f][ks], h[f]);
        }
      }
      if (j >= 1 && !g2_first) gemm2(st, hb);      // GEMM2 of chunk j - 1 beside the GELU below (independent of it)
      if (j < NCH) {                   // ... beside the bias + GELU + packing of chunk j's hidden units (independent of GEMM2)
        const float4 bv = *reinterpret_cast<const float4*>(b1s + j * 32 + hh * 16 + lq * 4);
#pragma unroll
        for (int f = 0; f < FP; ++f) {
          f32x2_t a01 = f32x2_t{h[f][0] + bv.x, h[f][1] + bv.y}, a23 = f32x2_t{h[f][2] + bv.z, h[f][3] + bv.w};
          if constexpr (HP) {           // this wave's 4 hidden units 8 lq + 4 hh .. + 3 of the chunk (training weight order), pre-activation
            // (address = scalar chunk base + one 32-bit lane offset per pixel block: the kernel has no registers to spare for 64-bit pointers)
            if (pbase + f * 16 + lr < p.M)
              *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.hpre) + (size_t)(j * 64) + (size_t)hpoff[f]) = uint2{pk2<HT>(a01.x, a01.y), pk2<HT>(a23.x, a23.y)};
          }
          if (!MLP_ABL(p, 1)) { a01 = gelu_poly2(a01); a23 = gelu_poly2(a23); }
          mine[f] = uint2{pk2<HT>(a01.x, a01.y), pk2<HT>(a23.x, a23.y)};
          xch[u * (4 * 2 * FP * 64) + xmine + f * 64] = mine[f];
        }
      }
    }
  }
  HT* const yb = reinterpret_cast<HT*>(p.ep.y);
  static_assert(FCH % 2 == 0, "fragment pairs");
#pragma unroll
  for (int m = 0; m < FCH / 2; ++m) {
    const int cb = (hh * FCH / 2 + m) * 32 + lq * 8;
    const float4 s0 = *reinterpret_cast<const float4*>(aff + cb), s1 = *reinterpret_cast<const float4*>(aff + cb + 4);
#pragma unroll
    for (int f = 0; f < FP; ++f) {
      const int pix = pbase + f * 16 + lr;
      if (pix >= p.M || MLP_ABL(p, 32)) continue;
      u32x4 o;
      o.x = pk2<HT>(acc2[2 * m][f][0] + s0.x, acc2[2 * m][f][1] + s0.y);
      o.y = pk2<HT>(acc2[2 * m][f][2] + s0.z, acc2[2 * m][f][3] + s0.w);
      o.z = pk2<HT>(acc2[2 * m + 1][f][0] + s1.x, acc2[2 * m + 1][f][1] + s1.y);
      o.w = pk2<HT>(acc2[2 * m + 1][f][2] + s1.z, acc2[2 * m + 1][f][3] + s1.w);
      *reinterpret_cast<u32x4*>(yb + (long)pix * D + cb) = o;
    }
  }
}

template <int D, typename HT, bool ORD = false, bool HP = false>
int launch_mlp_pair(const MlpP& p, hipStream_t s) {
  constexpr int P = 4 * 2 * 16;
  constexpr int STAGE = 32 * D * 2 + D * 64;
  constexpr int lds = 2 * STAGE + 2 * 4 * 2 * 2 * 64 * 8 + 5 * D * 4;
  static_assert(lds <= 160 * 1024, "LDS");
  const long blocks = ((long)p.M + P - 1) / P;
  if (blocks <= 0 || blocks > 0x7fffffffL) return MTBT_EINVAL;
  if (int rc = mtbt_allow_lds(mlp_pair_kernel<D, HT, ORD, HP>, lds)) return rc;
  hipLaunchKernelGGL((mlp_pair_kernel<D, HT, ORD, HP>), dim3((unsigned)blocks), dim3(512), lds, s, p);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// d = 96: the WEIGHT-RESIDENT form.  W1 and W2' are 4 * 96 * 96 * 2 B * 2 = 147 KB: they fit the CU's 160 KB of LDS, so ONE persistent
// workgroup per CU stages them once (all twelve chunks, the stage layout above) and its 16 waves (4 per SIMD) then run FREE: every wave
// walks its own list of 32-pixel blocks -- load the block's fc1 inputs and residual, the twelve-chunk loop on the resident weights,
// store -- with no workgroup barrier, no DMA pipeline and no lock step between waves, so one wave's global-load latency and GELU sit
// under the other waves' MFMAs.  (The streaming kernel above re-streams the weights for every 128 pixels: 3200 workgroups x 147 KB =
// 470 MB of L2 -> LDS traffic per call, and a barrier per chunk keeps a workgroup's waves in phase: measured, its load / DMA / GELU / MFMA
// phases ADD UP -- 45 + 8 + 24 + 25 = 95 us per stage-0 block -- where the HBM floor of the call is 47 us.)
// Blocks: the full rounds are dealt wave by wave; the remainder goes one block per CU first (then a second wave of the CU, which sits on
// another SIMD), so every SIMD of the chip gets the same number of blocks give or take one.
template <int D, typename HT, int NW, bool PREF = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void mlp_resident_kernel(const MlpP p) {
  constexpr int FP = 2;
  constexpr int KS1 = D / 32, FC = D / 16, NCH = 4 * D / 32;
  constexpr int W1B = 32 * D * 2, W2B = D * 64, STAGE = W1B + W2B;
  static_assert(NCH * STAGE + 6 * D * 4 <= 160 * 1024, "the weights must fit the LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* aff = reinterpret_cast<float*>(smem + NCH * STAGE);   // [D] b2' | [4D] b1
  float* b1s = aff + D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  for (int c = tid; c < D; c += NW * 64) aff[c] = p.ep.shift[c];
  for (int c = tid; c < 4 * D; c += NW * 64) b1s[c] = p.b1[c];
  // ---- all weights -> LDS, once: chunk j at j * STAGE in the streaming kernel's stage layout (LDS-DMA, swizzle on the source side) ----
  {
    const srd_t w1srd = make_srd(p.w1), w2srd = make_srd(p.w2p);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    constexpr int NDMA = STAGE / 1024;            // wave-instructions per chunk
    for (int q = wave; q < NCH * NDMA; q += NW) {
      const int j = q / NDMA, inst = q - j * NDMA;
      const int c = inst * 64 + lane;
      unsigned vo;
      if (inst * 1024 < W1B) {
        const int slab = c / 128, r = (c % 128) / 4, sl = (c % 4) ^ swz4(r);
        vo = (unsigned)(r * D * 2 + slab * 64 + sl * 16);
        lds_dma16(w1srd, vo, j * 32 * D * 2, __builtin_amdgcn_readfirstlane(lds0 + j * STAGE + inst * 1024));
      } else {
        const int c2 = c - W1B / 16;
        const int r = c2 / 4, sl = (c2 % 4) ^ swz4(r);
        vo = (unsigned)(r * 4 * D * 2 + sl * 16);
        lds_dma16(w2srd, vo, j * 64, __builtin_amdgcn_readfirstlane(lds0 + j * STAGE + inst * 1024));
      }
    }
    wait_vm<0>();
  }
  __syncthreads();   // the only barrier: from here on the LDS is read-only

  int a1off[KS1], a2off;
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) a1off[ks] = ks * 2048 + lr * 64 + ((lq ^ swz4(lr)) << 4);
  a2off = W1B + lr * 64 + ((lq ^ swz4(lr)) << 4);

  // (32-bit indices: the host checks M * D < 2^31)
  const int nblk = (p.M + 31) / 32;
  const int G = gridDim.x, per_round = G * NW;
  const int rounds = nblk / per_round, rem = nblk - rounds * per_round;
  const int mine = rounds + ((wave * G + (int)blockIdx.x) < rem ? 1 : 0);
  HT* const yb = reinterpret_cast<HT*>(p.ep.y);
  auto block_of = [&](int it) { return it < rounds ? it * per_round + (int)blockIdx.x * NW + wave : rounds * per_round + wave * G + (int)blockIdx.x; };
  auto fetch = [&](int blk, uint4 (&tf_)[FP][KS1], uint2 (&rr_)[FC][FP]) {
#pragma unroll
    for (int f = 0; f < FP; ++f) {
      const int pix = blk * 32 + f * 16 + lr;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        tf_[f][ks] = (pix < p.M && !MLP_ABL(p, 64)) ? *reinterpret_cast<const uint4*>(p.t + (unsigned)(pix * D + ks * 32 + lq * 8)) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < FC; ++i)
        rr_[i][f] = (p.res && pix < p.M) ? *reinterpret_cast<const uint2*>(p.res + (unsigned)(pix * D + i * 16 + lq * 4)) : uint2{0u, 0u};
    }
  };
  uint4 tf[FP][KS1], tfn[FP][KS1];
  uint2 rr[FC][FP], rrn[FC][FP];
  if (PREF && mine > 0) fetch(block_of(0), tf, rr);
#pragma unroll 1
  for (int it = 0; it < mine; ++it) {
    const int blk = block_of(it);
    const int pbase = blk * 32;
    // PREF: the NEXT block's inputs are requested before this block's chunk loop, so the loads of every wave are in flight under its own
    // MFMAs and GELU whatever the other waves do.  (Without it the waves of a SIMD fall into lock step -- all wait for memory together,
    // then all compute together -- and the memory and compute phases of the call ADD UP: tools/mlp_variants.py ablation.)
    if (PREF) { if (it + 1 < mine) fetch(block_of(it + 1), tfn, rrn); }
    else fetch(blk, tf, rr);
    f32x4 acc2[FC][FP];
#pragma unroll
    for (int f = 0; f < FP; ++f)
#pragma unroll
      for (int i = 0; i < FC; ++i) acc2[i][f] = unpack4<HT>(rr[i][f]);
#pragma unroll 1
    for (int jj = 0; jj < NCH; ++jj) {
      const char* st = smem + jj * STAGE;
      const float4 ba = *reinterpret_cast<const float4*>(b1s + jj * 32 + lq * 4);
      const float4 bb = *reinterpret_cast<const float4*>(b1s + jj * 32 + 16 + lq * 4);
      f32x4 h[2][FP];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int f = 0; f < FP; ++f) h[b][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!MLP_ABL(p, 4))
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const uint4 wa = *reinterpret_cast<const uint4*>(st + a1off[ks]);
        const uint4 wb = *reinterpret_cast<const uint4*>(st + a1off[ks] + 1024);
#pragma unroll
        for (int f = 0; f < FP; ++f) {
          h[0][f] = mfma_16x16x32<HT>(wa, tf[f][ks], h[0][f]);
          h[1][f] = mfma_16x16x32<HT>(wb, tf[f][ks], h[1][f]);
        }
      }
      uint4 hb[FP];
#pragma unroll
      for (int f = 0; f < FP; ++f) {
        f32x2_t a01 = f32x2_t{h[0][f][0] + ba.x, h[0][f][1] + ba.y}, a23 = f32x2_t{h[0][f][2] + ba.z, h[0][f][3] + ba.w};
        f32x2_t c01 = f32x2_t{h[1][f][0] + bb.x, h[1][f][1] + bb.y}, c23 = f32x2_t{h[1][f][2] + bb.z, h[1][f][3] + bb.w};
        if (!MLP_ABL(p, 1)) { a01 = gelu_poly2(a01); a23 = gelu_poly2(a23); c01 = gelu_poly2(c01); c23 = gelu_poly2(c23); }
        hb[f].x = pk2<HT>(a01.x, a01.y); hb[f].y = pk2<HT>(a23.x, a23.y);
        hb[f].z = pk2<HT>(c01.x, c01.y); hb[f].w = pk2<HT>(c23.x, c23.y);
      }
      if (!MLP_ABL(p, 2))
#pragma unroll
      for (int i = 0; i < FC; ++i) {
        const uint4 w2 = *reinterpret_cast<const uint4*>(st + a2off + i * 1024);
#pragma unroll
        for (int f = 0; f < FP; ++f) acc2[i][f] = mfma_16x16x32<HT>(w2, hb[f], acc2[i][f]);
      }
    }
#pragma unroll
    for (int i = 0; i < FC; ++i) {
      const float4 sh = *reinterpret_cast<const float4*>(aff + i * 16 + lq * 4);
#pragma unroll
      for (int f = 0; f < FP; ++f) {
        const int pix = pbase + f * 16 + lr;
        if (pix >= p.M || MLP_ABL(p, 32)) continue;
        uint2 o;
        o.x = pk2<HT>(acc2[i][f][0] + sh.x, acc2[i][f][1] + sh.y);
        o.y = pk2<HT>(acc2[i][f][2] + sh.z, acc2[i][f][3] + sh.w);
        *reinterpret_cast<uint2*>(yb + (unsigned)(pix * D + i * 16 + lq * 4)) = o;
      }
    }
    if (PREF) {
#pragma unroll
      for (int f = 0; f < FP; ++f) {
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) tf[f][ks] = tfn[f][ks];
#pragma unroll
        for (int i = 0; i < FC; ++i) rr[i][f] = rrn[i][f];
      }
    }
  }
}

template <int D, typename HT, int NW = 16, bool PREF = false>
int launch_mlp_resident(const MlpP& p, hipStream_t s) {
  constexpr int lds = (4 * D / 32) * (32 * D * 2 + D * 64) + 5 * D * 4;
  const long nblk = ((long)p.M + 31) / 32;
  long blocks = (nblk + NW - 1) / NW;
  if (blocks > 256) blocks = 256;             // one persistent workgroup per CU
  if (blocks <= 0) return MTBT_EINVAL;
  if (int rc = mtbt_allow_lds(mlp_resident_kernel<D, HT, NW, PREF>, lds)) return rc;
  hipLaunchKernelGGL((mlp_resident_kernel<D, HT, NW, PREF>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, p);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

}  // namespace

// t, res, y: dense [M][D] bf16 (y may not alias t; it may alias res only if equal pixel-for-pixel, which the in-place
// residual stream does not need).  w1 [4D][D] bf16, b1 [4D] f32, w2p [D][4D] bf16 with the per-32 hidden permutation,
// b2 [D] f32.  D in {96, 192}.
static int mlp_entry(const void* t, const void* res, const void* w1, const float* b1, const void* w2p, const float* b2, void* y, int64_t M, int D,
                     int dtype, void* stream, void* hpre = nullptr) {
  if (!t || !w1 || !b1 || !w2p || !b2 || !y || M <= 0 || M > 0x7fffff00L) return MTBT_EINVAL;
  if (hpre && (dtype != MTBT_BF16 || !aligned16(hpre) || M * 8 * (int64_t)D >= 0xffff0000L)) return MTBT_EINVAL;
  if (dtype != MTBT_BF16 && dtype != MTBT_F16) return MTBT_EINVAL;
  if (D != 96 && D != 192 && D != 384) return MTBT_EINVAL;
  if (!aligned16(t) || !aligned16(w1) || !aligned16(w2p) || !aligned16(y) || !aligned16(b1) || (res && !aligned16(res))) return MTBT_EALIGN;
  if ((long)4 * D * D * 2 >= 0x7fff0000L) return MTBT_EINVAL;
  MlpP p;
  p.t = reinterpret_cast<const bf16_t*>(t); p.w1 = reinterpret_cast<const bf16_t*>(w1); p.b1 = b1;
  p.w2p = reinterpret_cast<const bf16_t*>(w2p); p.M = (int)M;
  p.dbg = 0;   // ablation bits: development builds only (the library reads no environment variables)
  p.hpre = reinterpret_cast<bf16_t*>(hpre);
  ConvP& e = p.ep;
  e = ConvP{};
  p.res = (p.dbg & 16) ? nullptr : reinterpret_cast<const bf16_t*>(res);
  e.y = y; e.res = nullptr;   // the residual enters through the accumulators, not the epilogue
  e.scale = nullptr; e.shift = b2; e.K = D; e.ldy = D; e.ldr = D; e.act = MTBT_ACT_NONE;
  e.out_mode = MTBT_OUT_NHWC; e.out_f32 = 0; e.vec_ok = 1; e.M = (int)M;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // pixels per wave / waves per SIMD, measured (tools/mlp_probe.py): d = 96: 2 x 16 pixels at 4 waves per SIMD 100 us,
  // 3 x 16 at 3: 116, 4 x 16 at 2: 139 -- the kernel's skeleton (input / residual loads, stores) is latency-bound, so
  // residency beats the larger register tile; d = 192: 2 x 16 at 2: 110 us, 1 x 16 at 4: 125 (LDS-read bound).
  // d = 384 (stage 2): three 48 KiB weight stages fill the LDS, so ONE workgroup per CU (one wave per SIMD, up to 512 registers): the
  // wave keeps 2 x 16 pixels' inputs (96 registers) and their 384-channel outputs (192) in registers; per hidden chunk it reads 48
  // weight fragments for 96 MFMAs -- LDS port 50 % busy, the fill path 32 B/clk.
  // d = 96: the weight-resident persistent form once every CU has at least ~4 blocks of 32 pixels per wave-slot to run (its fixed cost is
  // the one-off 147 KB weight stage per CU); small calls (tests, tiny maps) keep the streaming kernel
  if (D == 96 && M >= 64 * 1024 && M * 96 < 0x7fffffffL && MTBT_MLP_RESIDENT)
    return dtype == MTBT_F16 ? launch_mlp_resident<96, f16_t, 8, true>(p, s) : launch_mlp_resident<96, bf16_t, 8, true>(p, s);
  if (hpre) {   // training forward (bf16): the same kernels with the pre-activation store compiled in
    if (D == 384) return MTBT_EINVAL;   // (the pair kernel has no registers for the store: 208 spills -- stage 2 trains through the two GEMMs)
    if (D == 192) return launch_mlp<192, 2, 2, bf16_t, true, true>(p, s);
    return launch_mlp<96, 2, 4, bf16_t, false, true>(p, s);
  }
  // round 3 (tools/mlp_variants.py, bit-identical outputs): d = 384 in the pair form 96 us against 122; d = 192 with the pipelined chunk
  // 95 against 101 (the pair form there: 131 -- at d = 192 two waves per SIMD already fit without it and the exchange is pure overhead)
  if (D == 384) return dtype == MTBT_F16 ? launch_mlp_pair<384, f16_t>(p, s) : launch_mlp_pair<384, bf16_t>(p, s);
  if (D == 192) return dtype == MTBT_F16 ? launch_mlp<192, 2, 2, f16_t, true>(p, s) : launch_mlp<192, 2, 2, bf16_t, true>(p, s);
  return dtype == MTBT_F16 ? launch_mlp<96, 2, 4, f16_t>(p, s) : launch_mlp<96, 2, 4, bf16_t>(p, s);
}

extern "C" int mtbt_convnext_mlp_fused(const void* t, const void* res, const void* w1, const float* b1, const void* w2p,
                                       const float* b2, void* y, int64_t M, int D, void* stream) {
  return mlp_entry(t, res, w1, b1, w2p, b2, y, M, D, MTBT_BF16, stream);
}

// the same with the storage / MFMA type given: MTBT_BF16 or MTBT_F16 (BASELINE configs[4])
extern "C" int mtbt_convnext_mlp_fused_dt(const void* t, const void* res, const void* w1, const float* b1, const void* w2p,
                                          const float* b2, void* y, int64_t M, int D, int dtype, void* stream) {
  return mlp_entry(t, res, w1, b1, w2p, b2, y, M, D, dtype, stream);
}

// Training forward of the ConvNeXt Mlp (bf16): y as above, plus the fc1 PRE-activation hpre [M][4D] (bf16, natural hidden order) that the
// backward reads (GELU' in the fc2 input gradient; GELU re-applied while the fc2 weight gradient stages it) -- the activated hidden tensor
// itself is never written.  Weight order of THIS entry: w1 rows (and b1) permuted per 32-row chunk -- staged row 16 blk + 4 q + e holds
// hidden unit 8 q + 4 blk + e -- and w2 [D][4D] in natural column order (layer scale folded into its rows): with that order a lane's
// 4 + 4 GEMM1 accumulators are 8 CONSECUTIVE hidden units, i.e. GEMM2's B fragment in natural order and one 16-byte store of hpre.
extern "C" int mtbt_convnext_mlp_fused_train(const void* t, const void* res, const void* w1_perm, const float* b1_perm, const void* w2, const float* b2,
                                             void* y, void* hpre, int64_t M, int D, void* stream) {
  if (!hpre) return MTBT_EINVAL;
  return mlp_entry(t, res, w1_perm, b1_perm, w2, b2, y, M, D, MTBT_BF16, stream, hpre);
}
