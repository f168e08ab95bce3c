// BiFPN node as ONE kernel: weighted sum of the (resampled) inputs -> DepthwiseConvBlock (k = 1 depthwise scale folded into the pointwise
// weight, folded BatchNorm, ELU)   --   main_model.py:198-243 (`BiFPNUnit.forward`: w * p4 + w * up2(p5) ...) + :62-102 (`DepthwiseConvBlock`).
//
//   y[p][k] = act( sum_c W'[k][c] * s[p][c] + shift[k] ),     s[p][c] = sum_i wgt_i * resample_i(x_i)[p][c]
//
// As two launches (mtbt_bifpn_fuse, then the 1x1 conv) the fused map s makes an HBM round trip (2 x 52 MB at P3, batch 16) and both
// kernels sit far below either roof (fusion 3.6 TB/s, the 256 -> 256 GEMM 0.08 of the MFMA peak / 0.18 of HBM).  Here the fusion arithmetic
// IS the staging of the GEMM's B operand: a workgroup owns 64 consecutive pixels, its 256 threads compute s for them (16-byte pieces, fp32,
// the stand-alone kernel's association order, rounded to the 16-bit storage type exactly as that kernel stores it -- so the result is
// equal to the two-launch form up to the accumulation order) straight into an XOR-swizzled [64 px][C] LDS image, then the four waves (one
// quarter of the output channels each, all 64 pixels) run the K x C GEMM with their weight fragments already in registers and the conv
// kernels' slab epilogue stores y.  HBM traffic per pixel: the inputs once + y once.
#include "common.h"
#include "conv_dma.h"
#include "conv_epilogue.h"
#include "conv_params.h"
#include "fuse_fetch.h"

namespace {

struct NodeP {
  FuseP f;          // inputs, weights, modes, N / H / W / C of the OUTPUT map (f.y unused)
  const void* w;    // [K][C] 16-bit
  ConvP ep;         // epilogue view: y, ldy, shift, act, K
};

// NIN / M0..M2: the node's input count and resampling modes as compile-time constants (the top-down nodes are identity + bilinear x2, the
// output nodes identity + identity + 2x2 mean): the fusion phase is then straight-line code whose 16-byte loads the compiler issues
// four pieces ahead.  (With the modes behind run-time branches every piece waited for its own loads: 8 dependent round trips per
// workgroup, 105 us for the P3 node where the two separate launches take 90.)
// Weights: K = C <= 256, so the whole reduction is 4 or 8 steps of 32 channels.  A wave owns KT / 4 output channels x all 64 pixels and
// loads ITS weight fragments for EVERY step straight from global memory into registers (a lane's 16-byte piece of a weight row IS its
// MFMA A fragment; the 128 KB matrix is L2-resident, each row is read by exactly one wave of the workgroup) right behind the fusion
// loads: they land while the fused map is computed, and the GEMM is then 32 .. 128 MFMAs per wave with nothing to wait for -- no weight
// tiles in LDS, no barrier per reduction step.  (First version: 32-channel weight slabs through LDS-DMA, one barrier per step: 82 us.)
template <typename T, int KT, int NIN, int M0, int M1, int M2>
__global__ __launch_bounds__(256, 2) void node_gemm_kernel(const NodeP p) {
  constexpr int TP = 64;                          // pixels per workgroup
  constexpr int ES = 2;                           // 16-bit storage
  constexpr int C = KT, rowb = C * ES;            // (square nodes only: C == K)
  constexpr int WCH = KT / 4, FC = WCH / 16, FP = 4, NST = C / 32;
  constexpr int BBYTES = TP * C * ES;
  constexpr int PITCH = WCH * 4 + 16;
  static_assert(4 * 16 * PITCH <= BBYTES, "epilogue slabs inside the B image");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bimg = smem;                              // [TP][C] T, 16-byte slot c8 of pixel q at q * C * 2 + ((c8 ^ (q & 15)) << 4) (low four slot bits)
  float* aff = reinterpret_cast<float*>(smem + BBYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const long M = (long)p.f.N * p.f.H * p.f.W;
  const long pix0 = (long)blockIdx.x * TP;
  stage_affine<KT>(p.ep, aff, 0, tid);

  // ---- B operand = the fused map of this workgroup's pixels ----
  constexpr int CH8 = C >> 3, PCS = TP * CH8 / 256;
  const float w0 = p.f.wgt_dev ? p.f.wgt_dev[0] : p.f.wgt[0], w1 = p.f.wgt_dev ? p.f.wgt_dev[1] : p.f.wgt[1];
  const float w2 = NIN > 2 ? (p.f.wgt_dev ? p.f.wgt_dev[2] : p.f.wgt[2]) : 0.f;
#pragma unroll 4
  for (int k = 0; k < PCS; ++k) {
    const int idx = tid + k * 256;
    const int q = idx / CH8, c8 = idx - q * CH8;
    const long pix = pix0 + q;
    const bool ok = pix < M;
    const long pc = ok ? pix : M - 1;             // (ragged last tile: fetch a valid pixel, store zeros)
    int x, y;                                     // (32-bit index arithmetic: M < 2^31 -- as `long` these were 64-bit divisions, ~100 instructions each,
    long ny, nn;                                  //  sixteen per thread and workgroup)
    divmod_u32(pc, p.f.W, ny, x);
    divmod_u32(ny, p.f.H, nn, y);
    const int n = (int)nn;
    float t0[8], t1[8], t2[8], acc[8];
    fuse_fetch<T>(reinterpret_cast<const T*>(p.f.x[0]), M0, n, y, x, p.f.H, p.f.W, C, c8 * 8, t0);
    fuse_fetch<T>(reinterpret_cast<const T*>(p.f.x[1]), M1, n, y, x, p.f.H, p.f.W, C, c8 * 8, t1);
    if constexpr (NIN > 2) fuse_fetch<T>(reinterpret_cast<const T*>(p.f.x[2]), M2, n, y, x, p.f.H, p.f.W, C, c8 * 8, t2);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = w0 * t0[e];
      v = v + w1 * t1[e];
      if constexpr (NIN > 2) v = v + w2 * t2[e];
      acc[e] = ok ? v : 0.f;
    }
    st8<T>(reinterpret_cast<T*>(bimg + q * rowb + ((((c8 & 15) ^ (q & 15)) | (c8 & ~15)) << 4)), acc);
  }

  // ---- this wave's weight fragments for the whole reduction, 16-byte piece (step g, quarter lq).  Fragment row f * 16 + lr holds weight row
  //      wave * WCH + epi_row_channel(f * 16 + lr) (conv_epilogue.h PERM): a lane's accumulators of a fragment pair are then 8 consecutive output
  //      channels and the epilogue stores 16-byte pieces straight from them, no slab ----
  static_assert(FC % 2 == 0, "fragment pairs (PERM epilogue)");
  uint4 a[NST][FC];
  {
    const T* wbase = reinterpret_cast<const T*>(p.w) + (long)(wave * WCH) * C + lq * 8;
#pragma unroll
    for (int g = 0; g < NST; ++g)
#pragma unroll
      for (int f = 0; f < FC; ++f) a[g][f] = *reinterpret_cast<const uint4*>(wbase + (long)epi_row_channel(f * 16 + lr) * C + g * 32);
  }
  __syncthreads();   // the whole B image is written

  f32x4 acc[FC][FP];
#pragma unroll
  for (int i = 0; i < FC; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NST; ++g) {
    uint4 b[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) {
      const int q = j * 16 + lr;
      const int slot = g * 4 + lq;
      b[j] = *reinterpret_cast<const uint4*>(bimg + q * rowb + ((((slot & 15) ^ (q & 15)) | (slot & ~15)) << 4));
    }
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
      for (int j = 0; j < FP; ++j) acc[i][j] = mfma_16x16x32<T>(a[g][i], b[j], acc[i][j]);
  }
  __syncthreads();   // every wave is done with the B image: its LDS is reused for the epilogue slabs

  const EpiSeq seq{pix0, 16, M, 0L, 0L};
  conv_epilogue<T, KT, FC, FP, false, 0, false, true>(p.ep, acc, smem + wave * (16 * PITCH), aff, 0, wave * WCH, lane,
                                      [&](int j, int row, int ch, long& yoff, long& roff) -> bool {
    const long pix = pix0 + j * 16 + row;
    yoff = pix * p.ep.ldy + ch;
    roff = 0;
    return pix < M;
  }, seq, true);
}

template <typename T, int KT, int NIN, int M0, int M1, int M2>
int launch_node_t(const NodeP& p, hipStream_t s) {
  const long M = (long)p.f.N * p.f.H * p.f.W;
  const long blocks = (M + 63) / 64;
  if (blocks <= 0 || blocks > 0x7fffffffL || M > 0x7fffffffL) return MTBT_EINVAL;      // (32-bit pixel indices in the kernel)
  constexpr int lds = 64 * KT * 2 + 2 * KT * 4;
  if (int rc = mtbt_allow_lds(node_gemm_kernel<T, KT, NIN, M0, M1, M2>, lds)) return rc;
  hipLaunchKernelGGL((node_gemm_kernel<T, KT, NIN, M0, M1, M2>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

// the two node shapes of BiFPNUnit.forward (main_model.py:198-243): identity + bilinear x2 (top-down), identity + identity + 2x2 mean (output)
template <typename T, int KT>
int launch_node(const NodeP& p, hipStream_t s) {
  if (p.f.n_in == 2 && p.f.resample[0] == 0 && p.f.resample[1] == 1) return launch_node_t<T, KT, 2, 0, 1, 0>(p, s);
  if (p.f.n_in == 3 && p.f.resample[0] == 0 && p.f.resample[1] == 0 && p.f.resample[2] == 2) return launch_node_t<T, KT, 3, 0, 0, 2>(p, s);
  return MTBT_EINVAL;
}

}  // namespace

extern "C" int mtbt_bifpn_node_nhwc(const mtbt_node_args* a, void* stream) {
  if (!a || !a->w || !a->y || !a->shift) return MTBT_EINVAL;
  const mtbt_fuse_args& f = a->fuse;
  if (f.n_in < 1 || f.n_in > 3 || f.N <= 0 || f.H <= 0 || f.W <= 0 || f.add_weight_bug) return MTBT_EINVAL;
  if (f.dtype != MTBT_BF16 && f.dtype != MTBT_F16) return MTBT_EINVAL;       // (fp32 parity mode: mtbt_bifpn_fuse + mtbt_conv2d_nhwc)
  if ((f.C != 128 && f.C != 256) || a->K != f.C) return MTBT_EINVAL;   // square nodes; the B image's slot swizzle spans 16 slots = 128 channels
  if (a->act < 0 || a->act > MTBT_ACT_GELU_POLY) return MTBT_EINVAL;
  if (a->y_pixel_stride < a->K || a->y_pixel_stride % 8) return MTBT_EINVAL;
  if (!aligned16(a->w) || !aligned16(a->y)) return MTBT_EALIGN;
  NodeP p;
  for (int i = 0; i < 3; ++i) {
    p.f.x[i] = i < f.n_in ? f.x[i] : nullptr;
    p.f.wgt[i] = f.wgt[i];
    p.f.resample[i] = f.resample[i];
    if (i < f.n_in) {
      if (!f.x[i] || f.resample[i] < 0 || f.resample[i] > 4) return MTBT_EINVAL;
      if ((f.resample[i] == 1 || f.resample[i] == 3) && ((f.H & 1) || (f.W & 1))) return MTBT_EINVAL;
      if (!aligned16(f.x[i])) return MTBT_EALIGN;
    }
  }
  p.f.wgt_dev = f.wgt_dev;
  p.f.n_in = f.n_in; p.f.y = nullptr; p.f.N = f.N; p.f.H = f.H; p.f.W = f.W; p.f.C = f.C; p.f.bug = 0;
  p.w = a->w;
  ConvP& e = p.ep;
  e = ConvP{};
  e.y = a->y; e.shift = a->shift; e.scale = nullptr; e.res = nullptr; e.y2 = nullptr;
  e.ldy = a->y_pixel_stride; e.ldr = 0; e.K = a->K; e.act = a->act; e.out_mode = MTBT_OUT_NHWC; e.out_f32 = 0; e.vec_ok = 1;
  e.cs_part = nullptr;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (f.dtype == MTBT_F16) return a->K == 256 ? launch_node<f16_t, 256>(p, s) : launch_node<f16_t, 128>(p, s);
  return a->K == 256 ? launch_node<bf16_t, 256>(p, s) : launch_node<bf16_t, 128>(p, s);
}
