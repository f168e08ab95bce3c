// Prototype x coefficient mask assembly, x4 upsampling fast path (the model's case: protos at S/4).
//
//   low[k][y][x] = sum_c coeff[k][c] * protos[y][x][c] (+ bias)      -- MFMA, exact fp32 (v_mfma_f32_16x16x4_f32)
//   up = bilinear x4 (align_corners=False);  mask = sigmoid(up) > 0.5
//
// The general kernel (postprocess.hip) re-reads the 18x18x32 prototype halo patch for every box: at 100
// boxes x 100 tiles x 16 images that is ~6.5 GB of L2 traffic, and its per-pixel index arithmetic + exp
// made it VALU-bound at 0.6 TB/s of output.  Here a workgroup owns one output tile of one image (128 x 32, or 64 x 64 for odd widths):
//   1. the prototype halo patch (tile / 4 + halo low-res pixels x 32 channels, coordinates CLAMPED to the image so the
//      edge rule of torch's bilinear falls out of the uniform interior formula) is staged in LDS once;
//   2. boxes are processed 16 at a time: coefficients [16 x 32] x patch [32 x 324] on the fp32 MFMA
//      (rows = boxes, columns = low-res pixels, 8 K-steps of 4 channels) -> low-res masks in LDS;
//   3. each thread upsamples 16 consecutive output pixels of one row per box with compile-time x4 weights
//      (frac = .625 .875 .125 .375 ...), horizontal-then-vertical like torch, and stores 16 mask bytes
//      (one 16-B store) and/or 16 logits.
// sigmoid(v) > 0.5 in fp32 is exactly v > 2^-24 (1 + exp(-v) rounds to 2 below that), so no exp is evaluated.
#include "common.h"

namespace {

struct MaskX4P {
  const float* protos;
  const float* coeff;
  long cbs, cks, ccs;
  const int* gather;
  const int* counts;
  float bias;
  int N, K, hp, wp, Hout, Wout;
  float* logits;
  unsigned char* masks;
  int tiles_x, tiles_y;
  int dbg;   // development ablation bits (tools/probes/mask_variants.hip; compiled in only with -DMTBT_MASK_ABLATION)
};
#ifdef MTBT_MASK_ABLATION
#define MASK_ABL(p, bit) ((p).dbg & (bit))
#else
#define MASK_ABL(p, bit) 0
#endif

constexpr int NM = 32;          // prototype channels
constexpr int PPITCH = NM + 1;  // patch row pitch (floats): conflict-free column reads

// LW x LH low-res pixels per tile = a 4 LW x 4 LH output tile.  16 x 16 (64 x 64 outputs) is the general shape; 32 x 8 (128 x 32 outputs)
// when the width allows: a mask row of the tile is then a whole 128-byte line -- with 64-byte rows two workgroups each wrote HALF of every
// line of the uint8 masks (2.2 TB/s of output at 640x640; the kernel is nothing but that write).
// MASKS_ONLY: the post-process form (no logits output) as its own instantiation.  As a run-time branch next to the general path the compiler
// merged the two paths' mask stores into a shared tail and split this path's 16-byte store into a dword + a dwordx3 (ISA of round 3): every
// 64-byte line of the largest write of the step was then written by two instructions in pieces of 4 and 12 bytes.
template <int LW, int LH, bool MASKS_ONLY>
__global__ __launch_bounds__(256) void mask_x4_kernel(const MaskX4P p) {
  constexpr int PW = LW + 2, PH = LH + 2;          // patch = tile + halo
  constexpr int NPX = PW * PH;
  constexpr int NPXP = (NPX + 15) / 16 * 16;       // padded to whole MFMA column groups
  constexpr int LPITCH = NPXP + 4;
  constexpr int SEGS = LW / 4;                     // 16-pixel output segments per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* patch = reinterpret_cast<float*>(smem);       // [NPXP][PPITCH]
  float* coef = patch + NPXP * PPITCH;                 // [16][PPITCH]
  float* low = coef + 16 * PPITCH;                     // [16][LPITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y;
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int lx_base = tx * LW - 1, ly_base = ty * LH - 1;
  const int cnt = p.counts ? min(p.counts[n], p.K) : p.K;

  // 1. stage the clamped halo patch
  const float* pr = p.protos + (long)n * p.hp * p.wp * NM;
  for (int i = tid; i < NPXP * (NM / 4); i += 256) {
    const int px = i >> 3, c4 = i & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (px < NPX && !MASK_ABL(p, 8)) {
      const int py = px / PW, pxx = px - py * PW;
      const int ly = min(max(ly_base + py, 0), p.hp - 1), lx = min(max(lx_base + pxx, 0), p.wp - 1);
      v = *reinterpret_cast<const float4*>(pr + ((long)ly * p.wp + lx) * NM + c4 * 4);
    }
    float* d = patch + px * PPITCH + c4 * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }

  const int row = tid / SEGS, seg = tid % SEGS;  // upsample role: output row of the tile (4 LH rows x SEGS segments = 256 threads), 16-pixel segment
  const int oy = ty * (4 * LH) + row, ox0 = tx * (4 * LW) + seg * 16;
  // vertical taps of this thread's row (x4): low row index inside the patch and weight
  const int iy = (row + 2) >> 2;
  const float wy1 = ((row + 2) & 3) * 0.25f + 0.125f, wy0 = 1.0f - wy1;

  // coefficients of a group of 16 boxes: gather index -> coefficient, two DEPENDENT global loads.  Fetched one group ahead into registers
  // (two values per thread): with the loads at the top of each group every one of the ~7 groups of an image paid both latencies in front of
  // its first barrier (round 3).
  float cpre[2];
  auto fetch_coef = [&](int g) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * 256, b = i >> 5, c = i & 31;
      float v = 0.f;
      if (g + b < cnt) {
        const long kk = p.gather ? p.gather[(long)n * p.K + g + b] : (g + b);
        v = p.coeff[(long)n * p.cbs + kk * p.cks + c * p.ccs];
      }
      cpre[u] = v;
    }
  };
  if (cnt > 0) fetch_coef(0);
  for (int g0 = 0; g0 < cnt; g0 += 16) {
    // 2a. coefficients of boxes g0..g0+15 (zeros past cnt)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * 256;
      coef[(i >> 5) * PPITCH + (i & 31)] = cpre[u];
    }
    __syncthreads();
    if (g0 + 16 < cnt) fetch_coef(g0 + 16);            // in flight during this group's MFMAs and upsampling
    // 2b. low[16][NPXP] = coef[16][32] x patch^T on the fp32 MFMA; wave w takes column groups w, w+4, ...
    {
      float a[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) a[ks] = coef[(lane & 15) * PPITCH + ks * 4 + (lane >> 4)];
      for (int pg = wave; pg < NPXP / 16 && !MASK_ABL(p, 4); pg += 4) {
        f32x4 acc = f32x4{p.bias, p.bias, p.bias, p.bias};
        const float* bp = patch + (pg * 16 + (lane & 15)) * PPITCH + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bp[ks * 4], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) low[(4 * (lane >> 4) + r) * LPITCH + pg * 16 + (lane & 15)] = acc[r];
      }
    }
    __syncthreads();
    // 2c. x4 bilinear + threshold, 16 pixels per thread per box
    if (oy < p.Hout && !MASK_ABL(p, 2)) {
      const int nb = min(16, cnt - g0);
      for (int b = 0; b < nb; ++b) {
        const float* r0 = low + b * LPITCH + iy * PW + seg * 4;
        const float* r1 = r0 + PW;
        float t0[6], t1[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) { t0[c] = r0[c]; t1[c] = r1[c]; }
        const long obase_m = (((long)n * p.K + g0 + b) * p.Hout + oy) * p.Wout + ox0;
        if constexpr (MASKS_ONLY) {
          // Masks only (the post-process path: 16 x 100 x 640^2 bytes per step, the largest single write of the step).  The kernel is
          // VALU-bound, not HBM-bound (~140 vector instructions per 16 stored bytes in the logits form): so the vertical taps go FIRST --
          // 6 columns x 2 + 16 pixels x 2 multiply-adds instead of 16 x 6 -- and the threshold + byte packing is one FMA and one
          // v_cvt_pk_u8_f32 per pixel: (v - 2^-24) * 2^100 saturates to 255 above the threshold and to 0 at or below it.  The value
          // differs from torch's horizontal-first association in the last bits only: the mask can differ where |logit| ~ 1e-7.
          // Vertical taps first (6 columns), then per PIXEL PAIR two packed FMAs that produce the thresholding value directly:
          //   d = (wx0 2^100) v[ix] + (wx1 2^100) v[ix + 1] - 2^76      (an even pixel and its right neighbour share ix)
          // d saturates to 255 above the threshold 2^-24 and to 0 at or below it in v_cvt_pk_u8_f32: 2 vector operations per pixel instead of 4.
          typedef float f2 __attribute__((ext_vector_type(2)));
          float v[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) v[c] = wy0 * t0[c] + wy1 * t1[c];
          unsigned w[4] = {0u, 0u, 0u, 0u};
          if (MASK_ABL(p, 16)) {   // ablation: the scalar form (one multiply-add chain and one conversion per pixel)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int ix = (e + 2) >> 2;
              const float wx1 = ((e + 2) & 3) * 0.25f + 0.125f, wx0 = 1.0f - wx1;
              const float val = wx0 * v[ix] + wx1 * v[ix + 1];
              w[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(val, 0x1p100f, -0x1p76f), e & 3, w[e >> 2]);
            }
          } else
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const int ix = (e + 2) >> 2;                                   // compile-time; the same for e and e + 1
            const float a1 = ((e + 2) & 3) * 0.25f + 0.125f, b1 = ((e + 3) & 3) * 0.25f + 0.125f;
            const f2 W1 = f2{a1 * 0x1p100f, b1 * 0x1p100f}, W0 = f2{(1.0f - a1) * 0x1p100f, (1.0f - b1) * 0x1p100f};
            f2 d = __builtin_elementwise_fma(W0, f2{v[ix], v[ix]}, f2{-0x1p76f, -0x1p76f});
            d = __builtin_elementwise_fma(W1, f2{v[ix + 1], v[ix + 1]}, d);
            w[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(d.x, e & 3, w[e >> 2]);
            w[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(d.y, (e + 1) & 3, w[e >> 2]);
          }
          if (!MASK_ABL(p, 1) || (w[0] ^ w[1] ^ w[2] ^ w[3]) == 0x12345679u) *reinterpret_cast<uint4*>(p.masks + obase_m) = uint4{w[0] & 0x01010101u, w[1] & 0x01010101u, w[2] & 0x01010101u, w[3] & 0x01010101u};
          continue;
        }
        float o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ix = (e + 2) >> 2;                                   // compile-time
          const float wx1 = ((e + 2) & 3) * 0.25f + 0.125f, wx0 = 1.0f - wx1;
          o[e] = wy0 * (wx0 * t0[ix] + wx1 * t0[ix + 1]) + wy1 * (wx0 * t1[ix] + wx1 * t1[ix + 1]);
        }
        const long obase = (((long)n * p.K + g0 + b) * p.Hout + oy) * p.Wout + ox0;
        if (p.logits) {
#pragma unroll
          for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(p.logits + obase + e) = make_float4(o[e], o[e + 1], o[e + 2], o[e + 3]);
        }
        if (p.masks) {
          unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int e = 0; e < 16; ++e) w[e >> 2] |= (o[e] > 5.9604645e-8f ? 1u : 0u) << ((e & 3) * 8);
          *reinterpret_cast<uint4*>(p.masks + obase) = uint4{w[0], w[1], w[2], w[3]};
        }
      }
    }
    __syncthreads();
  }
  // 3. padded slots k >= cnt: zeros
  if (oy < p.Hout) {
    for (int k = cnt; k < p.K; ++k) {
      const long obase = (((long)n * p.K + k) * p.Hout + oy) * p.Wout + ox0;
      if (p.logits) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(p.logits + obase + e) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (p.masks) *reinterpret_cast<uint4*>(p.masks + obase) = uint4{0u, 0u, 0u, 0u};
    }
  }
}

}  // namespace

// Returns MTBT_OK if the fast path applied, 1 if the shape is not the x4 case (caller falls back).
template <int LW, int LH>
static int launch_mask_x4(MaskX4P& p, const mtbt_mask_args* a, hipStream_t stream) {
  static_assert(LW * LH == 256 && (LW / 4) * (4 * LH) == 256, "256 threads: one per (output row, 16-pixel segment)");
  constexpr int NPXP = ((LW + 2) * (LH + 2) + 15) / 16 * 16;
  p.tiles_x = a->Wout / (4 * LW);
  p.tiles_y = (a->Hout + 4 * LH - 1) / (4 * LH);
  const size_t lds = (size_t)(NPXP * PPITCH + 16 * PPITCH + 16 * (NPXP + 4)) * sizeof(float);
  if (!a->logits && a->masks) {
    if (int rc = mtbt_allow_lds(mask_x4_kernel<LW, LH, true>, (int)lds)) return rc;
    hipLaunchKernelGGL((mask_x4_kernel<LW, LH, true>), dim3(p.tiles_x * p.tiles_y, a->N), dim3(256), lds, stream, p);
  } else {
    if (int rc = mtbt_allow_lds(mask_x4_kernel<LW, LH, false>, (int)lds)) return rc;
    hipLaunchKernelGGL((mask_x4_kernel<LW, LH, false>), dim3(p.tiles_x * p.tiles_y, a->N), dim3(256), lds, stream, p);
  }
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

int mtbt_mask_x4_try(const mtbt_mask_args* a, hipStream_t stream) {
  if (a->nm != NM || a->Hout != 4 * a->hp || a->Wout != 4 * a->wp || a->Wout % 64 || a->Hout % 4) return 1;
  MaskX4P p;
  p.protos = a->protos; p.coeff = a->coeff; p.cbs = a->coeff_batch_stride; p.cks = a->coeff_k_stride; p.ccs = a->coeff_c_stride;
  p.gather = a->gather_idx; p.counts = a->counts; p.bias = a->bias;
  p.N = a->N; p.K = a->K; p.hp = a->hp; p.wp = a->wp; p.Hout = a->Hout; p.Wout = a->Wout;
  p.logits = a->logits; p.masks = a->masks;
  p.dbg = 0;
  return a->Wout % 128 == 0 ? launch_mask_x4<32, 8>(p, a, stream) : launch_mask_x4<16, 16>(p, a, stream);
}
