// Launch parameters of the implicit-GEMM conv kernel (host <-> device).
#pragma once

struct ConvP {
  const void* x;
  const void* w;
  void* y;
  const float* scale;
  const float* shift;
  const void* res;
  void* y2;  // optional pre-activation output, addressed like y
  long xbs, ybs, rbs;
  int ldx, ldy, ldr;
  int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
  int act, out_mode, out_f32, vec_ok;
  int y_linear;  // output (and residual) batch stride == Ho*Wo*pixel stride: offset = pix*ld, no division
  int M;       // N*Ho*Wo (< 2^31)
  int ctiles;  // ceil(K / TC)
  int ptiles_per_xcd;  // ceil(ceil(M / TP) / 8)
  // optional per-channel column sums of the STORED output (bias gradients, BatchNorm batch statistics): partial rows [rows][cs_pitch]
  // of sum (v - cs_shift[k]) in [0, K) and, with cs_sq, of its square in [K, 2K); one row per (pixel tile, wave row), see conv_epilogue.h
  float* cs_part;
  const float* cs_shift;
  int cs_sq, cs_pitch;
  int debug;  // development ablation bits (MTBT_CONV_DEBUG): 1 = no DMA in the K loop, 2 = no fragment reads / MFMAs
};

// Development ablation bits (MTBT_CONV_DEBUG) are compiled in only with -DMTBT_CONV_ABLATION: run-time tests
// inside the K loop split it into dozens of basic blocks and keep the scheduler from batching the fragment reads.
#ifdef MTBT_CONV_ABLATION
#define MTBT_ABL(p, bit) ((p).debug & (bit))
#else
#define MTBT_ABL(p, bit) 0
#endif
