// C entry point of the implicit-GEMM convolution: argument validation, tile / pipeline-depth heuristics and
// dispatch to the per-dtype translation units (conv_igemm_bf16.hip, conv_igemm_f32.hip; kernel in conv_igemm.inc).
#include "common.h"
#include "conv_params.h"
#include "rowreduce.h"

// per (storage type, K-step width) translation units; two LDS stages (deeper pipelines never paid: residency beats prefetch depth -- re-checked in
// round 3 inside a dependent launch chain with 4 / 6 / 8 stages on 64x64, 128x64, 64x32 and 128x32 tiles: level on the 20x20 maps, 1.5 - 2.5x
// slower wherever the grid needs the residency; profiles/r03_chain_probe_deep_pipelines.txt)
#define MTBT_DECL(dt)                                                                      \
  int mtbt_conv_dispatch_##dt##_wide(const ConvP& p, int TC, int TP, hipStream_t s);       \
  int mtbt_conv_dispatch_##dt##_narrow(const ConvP& p, int TC, int TP, hipStream_t s);     \
  int mtbt_conv3x3_direct_##dt(const ConvP& p, int TC, hipStream_t s);                     \
  static int mtbt_conv_dispatch_##dt(const ConvP& p, int TC, int TP, int wide, int nbuf, hipStream_t s) { \
    (void)nbuf;                                                                            \
    return wide ? mtbt_conv_dispatch_##dt##_wide(p, TC, TP, s) : mtbt_conv_dispatch_##dt##_narrow(p, TC, TP, s); \
  }
MTBT_DECL(bf16)
MTBT_DECL(f32)
MTBT_DECL(f16)
#undef MTBT_DECL
// pw_stream.hip: the heads' output 1x1 convolutions (few input channels, <= 64 outputs, fp32 strided store) without LDS
bool mtbt_pw_stream_applies(const ConvP& p, int dtype, bool any_width);
int mtbt_pw_stream(const ConvP& p, int dtype, hipStream_t s);

// Tile heuristics, from the sweep in tools/conv_tune.py on the shapes of the 640x640 batch-16 forward
// (numbers in DESIGN.md):
//   * channel tile = the divisor of K among 128 / 96 (else the smallest tile that covers K);
//   * 1x1 convolutions (streaming GEMMs, HBM-bound): a SMALL 128x64 / 96x128 / 64x128 tile with 64-byte K-steps
//     -> 32 KiB of LDS, three or more workgroups per CU hide the load / epilogue latency of the short K loop;
//   * k x k convolutions (MFMA-bound): 128-pixel tiles with 128-byte K-steps while that still gives >= 2
//     workgroups per CU, else 64x64 (small pyramid levels, 64-channel head convs).
static void pick_tile(int pol, int K, long M, int taps, int C, int es, int* TC, int* TP, int* narrow, bool no96 = false, int act = 0) {
  // pol (mtbt_conv_args.policy, a development A/B knob the host passes per call): bit0 = small 1x1 tiles, bit1 = 64x64 for small k x k
  int tc;
  if (K % 128 == 0) tc = 128;
  else if (K % 96 == 0 && !no96) tc = 96;
  else if (K > 96) tc = (no96 && K % 64 == 0) ? 64 : 128;
  else if (K > 64) tc = no96 ? 128 : 96;
  else if (K > 32) tc = 64;
  else tc = 32;
  const long ct = (K + tc - 1) / tc;
  if (taps == 1 && (pol & 1) && !no96 && K % 96 == 0 && tc == 128 && ((M + 127) / 128) * (K / 96) >= 3000) {
    // large 1x1 GEMMs whose width is a multiple of both: 96-channel tiles (batch-32 sweep, profiles/r02_j_conv_tune_1x1_batch32.txt: fc1 of
    // stages 0-2 3-9 % ahead of 128x128; level at batch 16)
    *TC = 96; *TP = 128; *narrow = (C * es <= 1536) ? 1 : 0;
    return;
  }
  if (taps == 1 && (pol & 1)) {
    // Round 3, re-measured IN A DEPENDENT CHAIN (tools/chain_tune.py, profiles/r03_chain_tune.txt: 32 launches in one captured graph, operands
    // out of the Infinity Cache as behind a producer).  The round-2 rules came from an eager same-buffer loop that is host-bound below ~15 us
    // and L2-warm, and had it backwards on the small maps:
    //   * 128-byte K-steps wherever the row allows (they never lose; 64-byte steps only for rows that are not a multiple of 128 bytes);
    //   * what decides the tile is the ROUND structure on the 512 workgroup slots (256 CUs x 2): 300 - 512 workgroups of the largest tile
    //     that gives that many -- one full round -- beat twice as many half tiles (256 -> 256 @40x40: 128x128 11.3 us, 128x64 14.3;
    //     512 -> 256 @40x40: 15.8 against 21.5) and 513 - 1023 is the worst place to be (a second, mostly empty round: 384 -> 384 @40x40);
    //   * 64-channel outputs (the heads' box convs): 64-pixel tiles at every level; narrower ones run on the streaming kernel (pw_stream.hip).
    const bool can_wide = (C * es) % 128 == 0;
    *narrow = can_wide ? 0 : 1;
    if (tc == 128) {
      const long g = ((M + 127) / 128) * ct, g2 = ((M + 63) / 64) * ct, g4 = ((M + 63) / 64) * ((K + 63) / 64);
      *TC = 128;
      if (g >= 1024) *TP = (act == MTBT_ACT_ELU && g < 4096) ? 64 : 128;      // (the ELU epilogue is long: BiFPN pointwise @80x80 49.9 against 58.7 us)
      else if (g >= 300 && g <= 512) *TP = 128;
      else if (g > 512 || (g2 >= 300 && g2 <= 512)) *TP = 64;
      else if (K % 64 == 0 && g4 <= 1024) { *TC = 64; *TP = 64; }
      else *TP = 64;
      return;
    }
    *TC = tc;
    if (tc <= 64) { *TP = 64; return; }
    *TP = 128;
    *narrow = (C * es <= 1536 || !can_wide) ? 1 : 0;
    return;
  }
  *narrow = 0;
  if (!(pol & 2) || taps == 1) {  // round-1 baseline: largest fitting channel tile, 128 pixels unless the grid is tiny
    *TC = tc;
    *TP = (((M + 127) / 128) * ct >= 512) ? 128 : 64;
    return;
  }
  if (K <= 64) { *TC = K > 32 ? 64 : 32; *TP = 64; return; }
  // (round-2 sweep, profiles/r02_f_conv_tune_3x3.txt: one 128-pixel tile per CU is already enough -- c2f_p4.m 3x3 192->192 @40: 40.3 -> 31.6 us)
  const long gk = ((M + 127) / 128) * ct;
  if (gk > 512 && gk < 1024 && tc == 128) { *TC = tc; *TP = 64; return; }   // (chain sweep: a second, mostly empty round of 128x128 tiles -- stage-2 downsample 38.1 -> 33.9 us)
  if (gk >= 256) { *TC = tc; *TP = 128; return; }
  if (((M + 63) / 64) * ct >= 256) { *TC = tc; *TP = 64; return; }
  *TC = (tc == 96) ? 96 : 64;
  *TP = 64;
  if (no96 && *TC == 96) *TC = 64;
}

// LDS stages: as deep as fits 64 KiB (two workgroups per CU stay resident), at least 2, no deeper than the K loop.
static int pick_nbuf(int TC, int TP, int BKB, int nsteps) {
  const int cpr = BKB / 16;
  const int tcs = ((TC * cpr + 255) / 256) * 256 / cpr;
  const int bufsz = (tcs + TP) * BKB;
  int n = 64 * 1024 / bufsz;
  if (n > 2) n = 2;  // deeper pipelines never paid in the sweep: residency (workgroups per CU) beats prefetch depth
  if (n > nsteps) n = nsteps;
  if (n < 2) n = 2;
  return n;
}

extern "C" int64_t mtbt_conv_colsum_workspace_bytes(int64_t pixels, int K, int with_squares) {
  if (pixels <= 0 || K <= 0) return 0;
  return (pixels / 64 + 1) * 4 * (int64_t)K * (with_squares ? 2 : 1) * (int64_t)sizeof(float);   // <= 4 partial rows per 64 pixels
}

// second level of the column sums: `rows` partial rows -> colsum (one wave per channel, fixed order)
static int colsum_finish(const mtbt_conv_args* a, const ConvP& p, long rows, hipStream_t s) {
  if (rows <= 0 || rows > 0x7fffffffL) return MTBT_EINVAL;
  int pitch = p.cs_pitch;
  colsum_prereduce(p.cs_part, rows, pitch, 0, p.cs_pitch, s);       // (tall matrices: in place, see rowreduce.h)
  hipLaunchKernelGGL(channel_sum_final_pitch, dim3((unsigned)((a->K + 3) / 4)), dim3(256), 0, s, p.cs_part, (int)rows, pitch, 0, a->K, a->colsum, a->colsum_accumulate);
  if (a->colsum_sq)
    hipLaunchKernelGGL(channel_sum_final_pitch, dim3((unsigned)((a->K + 3) / 4)), dim3(256), 0, s, p.cs_part, (int)rows, pitch, a->K, a->K, a->colsum + a->K, a->colsum_accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

// `layout` non-null: validate and choose the kernel as a launch would, report the column-sum partial layout, launch nothing.
static int conv_impl(const mtbt_conv_args* a, void* stream, int64_t* layout /* [6]: rows, pitch, kernel kind, TC, TP, 128-byte K-steps */) {
  if (!a || !a->x || !a->w || !a->y) return MTBT_EINVAL;
  if (a->dtype != MTBT_F32 && a->dtype != MTBT_BF16 && a->dtype != MTBT_F16) return MTBT_EINVAL;
  if (a->out_dtype != a->dtype && a->out_dtype != MTBT_F32) return MTBT_EINVAL;
  if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || a->K <= 0 || a->R <= 0 || a->S <= 0 || a->stride <= 0 || a->pad < 0)
    return MTBT_EINVAL;
  const int es = a->dtype == MTBT_F32 ? 4 : 2;
  const int epc = 16 / es;
  if (a->C % (64 / es) != 0) return MTBT_EINVAL;
  if (a->Ho != (a->H + 2 * a->pad - a->R) / a->stride + 1 || a->Wo != (a->W + 2 * a->pad - a->S) / a->stride + 1) return MTBT_EINVAL;
  if (a->out_mode != MTBT_OUT_NHWC && a->out_mode != MTBT_OUT_CONVT2X2) return MTBT_EINVAL;
  if (a->out_mode == MTBT_OUT_CONVT2X2 && (a->K % 4 != 0)) return MTBT_EINVAL;
  if (a->act < 0 || a->act > MTBT_ACT_DGELU_POLY) return MTBT_EINVAL;
  if (a->act >= MTBT_ACT_DSILU && !a->res) return MTBT_EINVAL;  // the derivative epilogues read the pre-activation through res
  if (a->y2 && (a->out_mode != MTBT_OUT_NHWC || !aligned16(a->y2))) return MTBT_EINVAL;
  if (!aligned16(a->x) || !aligned16(a->w) || a->x_pixel_stride % epc != 0 || a->x_batch_stride % epc != 0) return MTBT_EALIGN;
  if (a->x_pixel_stride < a->C) return MTBT_EINVAL;

  ConvP p;
  p.x = a->x; p.w = a->w; p.y = a->y; p.scale = a->scale; p.shift = a->shift; p.res = a->res; p.y2 = a->y2;
  p.xbs = a->x_batch_stride; p.ybs = a->y_batch_stride; p.rbs = a->res_batch_stride;
  p.ldx = a->x_pixel_stride; p.ldy = a->y_pixel_stride; p.ldr = a->res_pixel_stride;
  p.N = a->N; p.H = a->H; p.W = a->W; p.C = a->C; p.K = a->K; p.R = a->R; p.S = a->S;
  p.stride = a->stride; p.pad = a->pad; p.Ho = a->Ho; p.Wo = a->Wo;
  p.act = a->act; p.out_mode = a->out_mode; p.out_f32 = (a->out_dtype == MTBT_F32);
  if ((long)a->N * a->Ho * a->Wo > 0x7fffff00L) return MTBT_EINVAL;
  p.M = a->N * a->Ho * a->Wo;
  p.y_linear = (a->out_mode == MTBT_OUT_NHWC) && (a->y_batch_stride == (int64_t)a->Ho * a->Wo * a->y_pixel_stride) &&
               (!a->res || a->res_batch_stride == (int64_t)a->Ho * a->Wo * a->res_pixel_stride);
  p.ctiles = 0;
  p.ptiles_per_xcd = 0;
  p.debug = a->debug;
  // column sums (header): partial rows in the caller's workspace, second level after the conv
  p.cs_part = nullptr; p.cs_shift = a->colsum_shift; p.cs_sq = a->colsum_sq ? 1 : 0; p.cs_pitch = a->K * (a->colsum_sq ? 2 : 1);
  const bool want_cs = a->colsum_ws != nullptr;      // partial rows (and, with colsum, the finished sums)
  if (a->colsum && !want_cs) return MTBT_EINVAL;
  if (want_cs) {
    if (a->out_mode != MTBT_OUT_NHWC || !aligned16(a->colsum_ws)) return MTBT_EINVAL;
    p.cs_part = reinterpret_cast<float*>(a->colsum_ws);
  }
  const int pol = (a->policy & 0x100) ? (a->policy & 0xff) : 7;   // 0 = the default policy
  // LDS-DMA addressing: 32-bit byte offsets below 2 GiB relative to (first image of a tile, weight tile row 0)
  if ((double)a->R * a->S > 31) return MTBT_EINVAL;
  if (((long)(128 / (a->Ho * a->Wo) + 2) * a->x_batch_stride + 2L * ((long)a->pad * a->W + a->pad) * a->x_pixel_stride) * es >= 0x7fff0000L) return MTBT_EINVAL;
  if ((long)128 * a->R * a->S * a->C * es >= 0x7fff0000L || (long)a->R * a->S * a->C * es >= 0x7fff0000L) return MTBT_EINVAL;
  // vector epilogue: a lane's 8 consecutive output channels must be whole aligned 16-byte accesses
  const int oes = p.out_f32 ? 4 : 2;
  const int kq = a->out_mode == MTBT_OUT_CONVT2X2 ? a->K / 4 : a->K;
  const int ovec = 16 / oes;  // elements per 16-byte store
  bool vec = (kq % 8 == 0 || a->out_mode == MTBT_OUT_NHWC) && (a->y_pixel_stride % ovec == 0) &&
             (a->y_batch_stride % ovec == 0) && aligned16(a->y);
  if (a->out_mode == MTBT_OUT_CONVT2X2) vec = vec && (kq % 8 == 0);
  if (a->res) vec = vec && (a->res_pixel_stride % epc == 0) && (a->res_batch_stride % epc == 0) && aligned16(a->res);
  p.vec_ok = vec ? 1 : 0;

  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // narrow 1x1 convolutions into an fp32 map (the heads' output convs): streaming kernel, same arithmetic (a tile hint keeps the call on
  // the implicit-GEMM kernel: tests, A/B)
  if (!a->tile_hint && (pol & 64) == 0 && (long)a->K * a->C * es < 0x7fff0000L && mtbt_pw_stream_applies(p, a->dtype, (pol & 128) != 0)) {
    if (layout) { layout[0] = layout[1] = 0; layout[2] = 2; layout[3] = a->K; layout[4] = 128; layout[5] = 0; return MTBT_OK; }   // (no column sums on this path)
    return mtbt_pw_stream(p, a->dtype, s);
  }
  // 3x3 / stride 1 / pad 1 on 16-aligned maps: direct convolution with an LDS-resident halo tile (conv3x3_direct.inc).
  // tile_hint bit 26 (or MTBT_CONV_POLICY bit 2 cleared) keeps such a conv on the implicit-GEMM kernel (tests, A/B).
  {
    if ((pol & 4) && !((a->tile_hint >> 26) & 1) && !a->y2 && a->act < MTBT_ACT_DSILU && a->R == 3 && a->S == 3 && a->stride == 1 && a->pad == 1 && a->H % 16 == 0 &&
        a->W % 16 == 0 && a->out_mode == MTBT_OUT_NHWC && a->C % (128 / es) == 0 && a->K > 32 &&
        (long)a->H * a->W * a->x_pixel_stride * es < 0x7fff0000L && (long)128 * 9 * a->C * es < 0x7fff0000L) {
      int tc = (a->K >= 96 && !(pol & 32)) ? 128 : 64;
      // row-reuse variant (conv3x3_rr_kernel): the default (round 1: only for 64-channel tiles; after the epilogue split it is also
      // 9 % faster on the 128-channel C2f convs and 3 % on Proto -- MTBT_CONV_POLICY A/B, 7.20 -> 7.15 ms per step); policy bit 4
      // selects the first formulation, hint bit 25 / policy bit 3 force this one
      if ((pol & 8) || ((a->tile_hint >> 25) & 1) || !(pol & 16)) tc |= 0x1000;
      const long rows = (long)a->N * (a->H >> 4) * (a->W >> 4) * ((tc & 0x1000) ? 2 : 4);   // partial rows per 16x16 tile: see conv3x3_direct.inc
      if (layout) { layout[0] = rows; layout[1] = p.cs_pitch; layout[2] = 1; layout[3] = tc & 0xfff; layout[4] = 256; layout[5] = (tc & 0x1000) ? 0 : 1; return MTBT_OK; }
      if (want_cs && a->colsum_ws_bytes < rows * p.cs_pitch * (int64_t)sizeof(float)) return MTBT_EWORKSPACE;
      const int rc = a->dtype == MTBT_F32 ? mtbt_conv3x3_direct_f32(p, tc, s) : (a->dtype == MTBT_F16 ? mtbt_conv3x3_direct_f16(p, tc, s) : mtbt_conv3x3_direct_bf16(p, tc, s));
      if (rc != MTBT_OK || !a->colsum) return rc;
      return colsum_finish(a, p, rows, s);
    }
  }
  int TC, TP, nbuf = 0;
  if (a->tile_hint) { nbuf = (a->tile_hint >> 28) & 7; TC = (a->tile_hint >> 16) & 0x1ff; TP = a->tile_hint & 0xffff; }
  int narrow = (a->tile_hint >> 27) & 1;  // hint bit 27: force 64-byte K-steps
  if (!a->tile_hint || !TC || !TP) pick_tile(pol, a->K, p.M, a->R * a->S, a->C, es, &TC, &TP, &narrow, want_cs, a->act);
  if (want_cs && TC == 96) return MTBT_EINVAL;   // (a wave's 48 / 96 channels are not a power-of-two number of 8-channel pieces)
  const int wide = (a->C % (128 / es) == 0 && !narrow) ? 1 : 0;
  if (nbuf < 2 || nbuf > 4) nbuf = pick_nbuf(TC, TP, wide ? 128 : 64, a->R * a->S * a->C / ((wide ? 128 : 64) / es));
  const int waves_p = (TC == 128 || (TC == 96 && TP == 64)) ? 2 : 4;    // wave layouts of conv_igemm.inc's dispatch_tile
  const long rows = (((long)p.M + TP - 1) / TP) * waves_p;
  if (layout) { layout[0] = rows; layout[1] = p.cs_pitch; layout[2] = 0; layout[3] = TC; layout[4] = TP; layout[5] = wide; return MTBT_OK; }
  if (want_cs && a->colsum_ws_bytes < rows * p.cs_pitch * (int64_t)sizeof(float)) return MTBT_EWORKSPACE;
  const int rc = a->dtype == MTBT_F32 ? mtbt_conv_dispatch_f32(p, TC, TP, wide, nbuf, s)
                                      : (a->dtype == MTBT_F16 ? mtbt_conv_dispatch_f16(p, TC, TP, wide, nbuf, s) : mtbt_conv_dispatch_bf16(p, TC, TP, wide, nbuf, s));
  if (rc != MTBT_OK || !a->colsum) return rc;
  return colsum_finish(a, p, rows, s);
}

extern "C" int mtbt_conv2d_nhwc(const mtbt_conv_args* a, void* stream) { return conv_impl(a, stream, nullptr); }

// The partial rows a call with these arguments writes into colsum_ws: rows x pitch floats, row r = [sum (K) | sum of squares (K, with
// colsum_sq)] of one (pixel tile, wave row); rows that cover no pixel hold zeros.  For a consumer that reduces them itself
// (mtbt_bn_forward_partials_nhwc) instead of asking for the finished sums.
extern "C" int mtbt_conv_colsum_layout(const mtbt_conv_args* a, int64_t* rows, int32_t* pitch) {
  if (!rows || !pitch || !a || !a->colsum_ws) return MTBT_EINVAL;
  int64_t lay[6] = {0, 0, 0, 0, 0, 0};
  const int rc = conv_impl(a, nullptr, lay);
  if (rc != MTBT_OK) return rc;
  *rows = lay[0]; *pitch = (int32_t)lay[1];
  return MTBT_OK;
}

// Which kernel and tile a call with these arguments runs (nothing is launched, no pointer is dereferenced): choice[0] = 0 implicit GEMM,
// 1 direct 3x3 with an LDS-resident halo, 2 streaming head conv; [1] channel tile; [2] pixel tile (256 = the 16 x 16 halo tile; the
// streaming kernel: 128 pixels per workgroup); [3] = 128-byte K-steps (implicit GEMM) / first formulation (direct).  For tests of the tile
// rules (tests/test_cpu_host_logic.py) and for tools.
extern "C" int mtbt_conv_kernel_choice(const mtbt_conv_args* a, int32_t* choice) {
  if (!a || !choice) return MTBT_EINVAL;
  int64_t lay[6] = {0, 0, 0, 0, 0, 0};
  const int rc = conv_impl(a, nullptr, lay);
  if (rc != MTBT_OK) return rc;
  for (int i = 0; i < 4; ++i) choice[i] = (int32_t)lay[2 + i];
  return MTBT_OK;
}
