// C entry point of the implicit-GEMM convolution: argument validation, tile / pipeline-depth heuristics and
// dispatch to the per-dtype translation units (conv_igemm_bf16.hip, conv_igemm_f32.hip; kernel in conv_igemm.inc).
#include "common.h"
#include "conv_params.h"

int mtbt_conv_dispatch_bf16(const ConvP& p, int TC, int TP, int wide, int nbuf, hipStream_t s);
int mtbt_conv_dispatch_f32(const ConvP& p, int TC, int TP, int wide, int nbuf, hipStream_t s);

// Tile heuristic: channel tile with the least padding waste (ties -> larger), pixel tile 128 unless
// that leaves the 256 CUs under-filled.
static void pick_tile(int K, long M, int* TC, int* TP) {
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

// LDS stages: as deep as fits 64 KiB (two workgroups per CU stay resident), at least 2, no deeper than the K loop.
static int pick_nbuf(int TC, int TP, int BKB, int nsteps) {
  const int cpr = BKB / 16;
  const int tcs = ((TC * cpr + 255) / 256) * 256 / cpr;
  const int bufsz = (tcs + TP) * BKB;
  int n = 64 * 1024 / bufsz;
  if (n > 4) n = 4;
  if (n > nsteps) n = nsteps;
  if (n < 2) n = 2;
  return n;
}

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
  if ((long)a->N * a->Ho * a->Wo > 0x7fffff00L) return MTBT_EINVAL;
  p.M = a->N * a->Ho * a->Wo;
  p.y_linear = (a->out_mode == MTBT_OUT_NHWC) && (a->y_batch_stride == (int64_t)a->Ho * a->Wo * a->y_pixel_stride) &&
               (!a->res || a->res_batch_stride == (int64_t)a->Ho * a->Wo * a->res_pixel_stride);
  p.ctiles = 0;
  p.ptiles_per_xcd = 0;
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

  int TC, TP, nbuf = 0;
  if (a->tile_hint) { nbuf = (a->tile_hint >> 28) & 7; TC = (a->tile_hint >> 16) & 0xfff; TP = a->tile_hint & 0xffff; }
  if (!a->tile_hint || !TC || !TP) pick_tile(a->K, p.M, &TC, &TP);
  const int wide = (a->C % (128 / es) == 0) ? 1 : 0;
  if (nbuf < 2 || nbuf > 4) nbuf = pick_nbuf(TC, TP, wide ? 128 : 64, a->R * a->S * a->C / ((wide ? 128 : 64) / es));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == MTBT_F32) return mtbt_conv_dispatch_f32(p, TC, TP, wide, nbuf, s);
  return mtbt_conv_dispatch_bf16(p, TC, TP, wide, nbuf, s);
}
