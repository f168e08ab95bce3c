// Development probe: the x4 mask-assembly kernel with its phases removable (1 = no stores, 2 = no upsample / threshold phase, 4 = no MFMA phase,
// 8 = no prototype-patch loads), timed by tools/mask_variants.py.  Includes the product source.
#define MTBT_MASK_ABLATION 1
#include "../../multitask_bonetumor_yolo_amd/csrc/mask_mfma.hip"

extern "C" int mask_variant(int dbg, const float* protos, const float* coeff, long cbs, long cks, long ccs, const int* gather, const int* counts, int N, int K,
                            int hp, int wp, unsigned char* masks, void* stream) {
  MaskX4P p;
  p.protos = protos; p.coeff = coeff; p.cbs = cbs; p.cks = cks; p.ccs = ccs; p.gather = gather; p.counts = counts; p.bias = 0.f;
  p.N = N; p.K = K; p.hp = hp; p.wp = wp; p.Hout = 4 * hp; p.Wout = 4 * wp; p.logits = nullptr; p.masks = masks; p.dbg = dbg;
  mtbt_mask_args a{};
  a.N = N; a.Hout = 4 * hp; a.Wout = 4 * wp; a.masks = masks;
  return launch_mask_x4<32, 8>(p, &a, reinterpret_cast<hipStream_t>(stream));
}
