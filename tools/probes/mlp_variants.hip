// Development probe: the fused ConvNeXt MLP with its ablation bits exposed (1 = no GELU, 2 = no GEMM2, 4 = no GEMM1, 8 = no weight DMA after
// the first two stages, 32 = no output stores, 64 = no input loads), timed by tools/mlp_variants.py.  Includes the product source.
#define MTBT_MLP_ABLATION 1
#include "../../multitask_bonetumor_yolo_amd/csrc/mlp_fused.hip"

extern "C" int mlp_variant(int dbg, const void* t, const void* res, const void* w1, const float* b1, const void* w2p, const float* b2, void* y, long M, int D,
                           void* stream) {
  MlpP p;
  p.t = reinterpret_cast<const bf16_t*>(t); p.w1 = reinterpret_cast<const bf16_t*>(w1); p.b1 = b1;
  p.w2p = reinterpret_cast<const bf16_t*>(w2p); p.M = (int)M;
  p.dbg = dbg;
  ConvP& e = p.ep;
  e = ConvP{};
  p.res = (dbg & 16) ? nullptr : reinterpret_cast<const bf16_t*>(res);
  e.y = y; e.res = nullptr;
  e.scale = nullptr; e.shift = b2; e.K = D; e.ldy = D; e.ldr = D; e.act = MTBT_ACT_NONE;
  e.out_mode = MTBT_OUT_NHWC; e.out_f32 = 0; e.vec_ok = 1; e.M = (int)M;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (D == 96 && (dbg & 0x700) == 0x400) return launch_mlp_resident<96, bf16_t, 8, true>(p, s);
  if (D == 96 && (dbg & 0x700) == 0x500) return launch_mlp_resident<96, bf16_t, 12, true>(p, s);
  if (D == 96 && (dbg & 0x300) == 0x100) return launch_mlp_resident<96, bf16_t, 16>(p, s);
  if (D == 96 && (dbg & 0x300) == 0x200) return launch_mlp_resident<96, bf16_t, 12>(p, s);
  if (D == 96 && (dbg & 0x300) == 0x300) return launch_mlp_resident<96, bf16_t, 8>(p, s);
  if (D == 96) return launch_mlp<96, 2, 4, bf16_t>(p, s);
  if (D == 384 && (dbg & 0x2000)) return launch_mlp_pair<384, bf16_t, true>(p, s);
  if (D == 384 && (dbg & 0x1000)) return launch_mlp_pair<384, bf16_t>(p, s);
  if (D == 192 && (dbg & 0x1000)) return launch_mlp_pair<192, bf16_t>(p, s);
  if (D == 192 && (dbg & 0x800)) return launch_mlp<192, 2, 2, bf16_t, true>(p, s);
  if (D == 384 && (dbg & 0x800)) return launch_mlp<384, 2, 1, bf16_t, true>(p, s);
  if (D == 192) return launch_mlp<192, 2, 2, bf16_t>(p, s);
  if (D == 384) return launch_mlp<384, 2, 1, bf16_t>(p, s);
  return -100;
}
