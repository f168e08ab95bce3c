// bf16 instantiations of the implicit-GEMM conv (see conv_igemm.inc).
#include "conv_igemm.inc"
#include "conv3x3_direct.inc"

int mtbt_conv_dispatch_bf16(const ConvP& p, int TC, int TP, int wide, int nbuf, hipStream_t s) {
  (void)nbuf;   // two LDS stages (deeper pipelines never paid: residency beats prefetch depth)
  return wide ? dispatch_tile<bf16_t, 128, 2>(p, TC, TP, s) : dispatch_tile<bf16_t, 64, 2>(p, TC, TP, s);
}

int mtbt_conv3x3_direct_bf16(const ConvP& p, int TC, hipStream_t s) {
  if (TC == (128 | 0x1000)) return launch_direct3x3_rr<bf16_t, 128>(p, s);
  if (TC == (64 | 0x1000)) return launch_direct3x3_rr<bf16_t, 64>(p, s);
  if (TC == 128) return launch_direct3x3<bf16_t, 128>(p, s);
  if (TC == 64) return launch_direct3x3<bf16_t, 64>(p, s);
  return MTBT_EINVAL;
}
