// Shared epilogue of the MFMA conv kernels.
//
// Accumulator layout (16x16 MFMA, rows = channels): lane (lr = lane&15, lq = lane>>4) holds, per fragment
// (i, j), channels 16 i + 4 lq .. +3 of pixel 16 j + lr.  Per 16-pixel slab j:
//   (1) affine in fp32 and a float4 store into the wave's [16 px][WCH ch] fp32 LDS slab
//       (row pitch +16 B: conflict-free b128 writes); scale/shift come from a per-workgroup LDS copy;
//   (2) lanes re-read the slab as rows: 8 consecutive channels of one pixel per lane (two ds_read_b128),
//       optionally store them as the pre-activation (y2, training forward), apply the activation, add the residual (one
//       16-byte load) -- or, in the backward modes MTBT_ACT_D*, multiply by act'(residual) -- and issue ONE 16-byte (bf16) /
//       two 16-byte (f32) coalesced stores.
// The caller guarantees the main loop is over (barrier) before the slabs are written.
// (Measured alternative: 8-byte stores straight from the accumulators, no LDS pass -- 3.5 % SLOWER over the network's
// convs in round 1 and again 1.3 % slower per step in round 2 after the epilogue split, although the same idea wins in
// mlp_fused.hip where the slab form needed four passes of 12-piece rows.)
#pragma once
#include "common.h"
#include "conv_params.h"

namespace {

// Stage this workgroup's TC (scale, shift) pairs in LDS: aff[0..TC) = scale, aff[TC..2TC) = shift.
template <int TC>
__device__ __forceinline__ void stage_affine(const ConvP& p, float* aff, int cbase, int tid) {
  for (int c = tid; c < TC; c += 256) {
    const bool ok = cbase + c < p.K;
    aff[c] = (ok && p.scale) ? p.scale[cbase + c] : 1.f;
    aff[TC + c] = (ok && p.shift) ? p.shift[cbase + c] : 0.f;
  }
}

// ---- per-channel column sums of the stored output, accumulated in the row pass ----
// In the row pass a lane holds 8 consecutive channels (piece c8 = lane % C8, constant over the slabs when C8 divides 64) of one pixel per
// (slab, iteration): the sums of ITS pixels stay in 8 (+ 8) registers; at the end the lanes that share c8 (lane bits log2(C8) .. 5) are added
// by xor-shuffles and lanes 0 .. C8-1 write the wave's partial row.  The values summed are the ones STORED (rounded to the output type) minus
// an optional per-channel shift (BatchNorm: the running mean, which keeps sum / sum-of-squares well conditioned).  Deterministic.
template <typename OUT> __device__ __forceinline__ float stored_value(float v);
template <> __device__ __forceinline__ float stored_value<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored_value<bf16_t>(float v) { return __uint_as_float((unsigned)f2bf(v) << 16); }
template <> __device__ __forceinline__ float stored_value<f16_t>(float v) { return h2f(f2h(v)); }

// PERM: the kernel staged its weight tile in the PERMUTED row order below (a free change of the DMA source offsets): inside a wave's channel
// block, LDS row 16 F + 4 q + e holds channel 32 (F / 2) + 8 q + 4 (F % 2) + e.  A lane's accumulator rows (4 q .. 4 q + 3 of every fragment)
// of the fragment PAIR (2 m, 2 m + 1) are then the 8 CONSECUTIVE channels 32 m + 8 q .. + 7 -- one 16-byte piece of the NHWC output row, stored
// straight from the accumulators (conv_epilogue_direct) instead of through the LDS slab transposition.  The slab bodies write at epi_cpos.
__device__ __forceinline__ int epi_row_channel(int r) {
  const int F = r >> 4, q = (r >> 2) & 3, e = r & 3;
  return 32 * (F >> 1) + 8 * q + 4 * (F & 1) + e;
}
template <bool PERM> __device__ __forceinline__ int epi_cpos(int i, int lq) { return PERM ? 32 * (i >> 1) + 8 * lq + 4 * (i & 1) : i * 16 + lq * 4; }

template <int C8>
__device__ __forceinline__ void colsum_flush(const ConvP& p, float (&cs)[8], float (&cq)[8], long srow, int ch, bool ch_ok, int lane) {
  static_assert((C8 & (C8 - 1)) == 0 && C8 <= 64, "lanes sharing a channel piece differ in whole lane bits");
#pragma unroll
  for (int o = C8; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs[e] += __shfl_xor(cs[e], o, 64); cq[e] += __shfl_xor(cq[e], o, 64); }
  }
  if (lane < C8 && ch_ok) {
    float* row = p.cs_part + srow * p.cs_pitch + ch;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (ch + e < p.K) { row[e] = cs[e]; if (p.cs_sq) row[p.K + e] = cq[e]; }
    }
  }
}

// AddrFn: bool operator()(int j, int row, long& pixel_offset_y, long& pixel_offset_res) -- offsets in elements of the
// slab pixel (without the channel), false if the pixel is outside the output.
// TRAIN = false compiles the training epilogues (y2, MTBT_ACT_D*) out: the direct 3x3 kernels never need them (they are used by the 1x1
// GEMMs fc1 / fc2-dgrad only) and lost 13 % with the extra code present (tools/conv_ab.py, same box, same run).
//
// ACT / VEC: the activation code and "every 8-channel piece is one aligned 16-byte access" as COMPILE-TIME constants of the body; the
// dispatcher below switches on them ONCE per call.  With the activation switch inside the fully unrolled per-element loops (and the
// ragged-tail code next to every vector store) the epilogue of the 128x64 tile was ~12 000 instructions with 1 160 scalar branches for a
// main loop of 16 MFMAs: tens of KiB of code streamed through the instruction cache per tile and a taken branch every few instructions.
// ACT = -1 is the run-time form (ragged outputs: the nc-channel class conv, the 66-wide detect map).
template <typename T, int TC, int FC, int FP, bool TRAIN, int ACT, bool VEC, bool PERM = false, typename AddrFn>
__device__ __forceinline__ void conv_epilogue_body(const ConvP& p, f32x4 (&acc)[FC][FP], char* slab, const float* aff, int cbase,
                                                   int chl0 /* first channel-in-tile of this wave */, int lane, AddrFn addr, long srow) {
  constexpr int WCH = FC * 16;
  constexpr int PITCH = WCH * 4 + 16;
  constexpr int C8 = WCH / 8;
  constexpr int ITER = (16 * C8 + 63) / 64;
  typedef typename half_of<T>::type HT;   // 16-bit output element (bf16 / fp16) when the output is not fp32
  const int lr = lane & 15, lq = lane >> 4;
  const bool has_scale = p.scale != nullptr;
  const int act = ACT >= 0 ? ACT : p.act;
  // training epilogues keep the PRE-activation in the slab and finish in the row pass: y2 (second output) / MTBT_ACT_D* (multiply by act'(res))
  const bool deriv = TRAIN && act >= MTBT_ACT_DSILU;
  const bool late_act = TRAIN && (p.y2 != nullptr || deriv);
  // column sums (see colsum_flush): this lane's channel piece is the same in every slab / iteration when C8 is a power of two
  constexpr bool CS_OK = (C8 & (C8 - 1)) == 0;
  const bool csum = CS_OK && p.cs_part != nullptr && srow >= 0;
  const int cs_ch = cbase + chl0 + (lane % C8) * 8;
  float cs[8], cq[8], csh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs[e] = 0.f; cq[e] = 0.f; csh[e] = (csum && p.cs_shift && cs_ch + e < p.K) ? p.cs_shift[cs_ch + e] : 0.f; }
#pragma clang loop unroll(full)  // must unroll: a runtime j would put the whole accumulator array in scratch
  for (int j = 0; j < FP; ++j) {
#pragma unroll
    for (int i = 0; i < FC; ++i) {
      const int cl = chl0 + epi_cpos<PERM>(i, lq);
      const float4 sh = *reinterpret_cast<const float4*>(aff + TC + cl);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (has_scale) {
        const float4 sc = *reinterpret_cast<const float4*>(aff + cl);
        v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
      }
      v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w;
      if (!late_act) {   // the usual case: activation here, on the accumulators (wave-uniform branch)
        v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
      }
      *reinterpret_cast<float4*>(slab + lr * PITCH + epi_cpos<PERM>(i, lq) * 4) = v;
    }
    // wave-local hand-off through LDS (other lanes' data): a compiler barrier is REQUIRED -- the float4 row reads
    // below are a different type from the stores above and would otherwise be hoisted over them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = it * 64 + lane;
      if ((16 * C8) % 64 != 0 && idx >= 16 * C8) break;
      const int row = idx / C8, c8 = idx - row * C8;
      const int ch = cbase + chl0 + c8 * 8;
      long yoff, roff;
      if (ch >= p.K || !addr(j, row, ch, yoff, roff)) continue;
      const float4 lo = *reinterpret_cast<const float4*>(slab + row * PITCH + c8 * 32);
      const float4 hi = *reinterpret_cast<const float4*>(slab + row * PITCH + c8 * 32 + 16);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      if (VEC || (p.vec_ok && ch + 8 <= p.K)) {
        if (TRAIN && p.y2) {   // training forward: keep the pre-activation next to the activated output
          if (p.out_f32) st8<float>(reinterpret_cast<float*>(p.y2) + yoff, v);
          else st8<HT>(reinterpret_cast<HT*>(p.y2) + yoff, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], act);
        }
        if (p.res) {
          float r[8];
          ld8<T>(reinterpret_cast<const T*>(p.res) + roff, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = deriv ? v[e] * act_grad(r[e], act) : v[e] + r[e];
        }
        if (p.out_f32) st8<float>(reinterpret_cast<float*>(p.y) + yoff, v);
        else st8<HT>(reinterpret_cast<HT*>(p.y) + yoff, v);
        if (csum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = (p.out_f32 ? v[e] : stored_value<HT>(v[e])) - csh[e]; cs[e] += d; cq[e] += d * d; }
        }
      } else if (!VEC) {
        // unaligned / ragged channel tail (e.g. the nc-channel class conv, the 66-wide detect map)
        const int lim = (p.out_mode == MTBT_OUT_CONVT2X2) ? (ch / (p.K >> 2) + 1) * (p.K >> 2) : p.K;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (ch + e >= lim) break;
          float u = v[e];
          if (TRAIN && p.y2) {
            if (p.out_f32) reinterpret_cast<float*>(p.y2)[yoff + e] = u;
            else st_elem<HT>(reinterpret_cast<HT*>(p.y2) + yoff + e, u);
          }
          if (TRAIN && p.y2) u = act_apply(u, act);
          if (p.res) {
            const float r = ld_elem<T>(reinterpret_cast<const T*>(p.res) + roff + e);
            u = deriv ? u * act_grad(r, act) : u + r;
          }
          if (p.out_f32) reinterpret_cast<float*>(p.y)[yoff + e] = u;
          else st_elem<HT>(reinterpret_cast<HT*>(p.y) + yoff + e, u);
          if (csum) { const float d = (p.out_f32 ? u : stored_value<HT>(u)) - csh[e]; cs[e] += d; cq[e] += d * d; }   // (e: compile-time after unrolling)
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slab fully read before the next pass rewrites it
  }
  if constexpr (CS_OK) {
    if (csum) colsum_flush<C8>(p, cs, cq, srow, cs_ch, cs_ch < p.K && (lane % C8) < C8, lane);
  }
}

// The FAST body: 16-bit NHWC output whose pixels are an arithmetic sequence per slab (pixel(j, row) = pix0 + j * jstep + row, element offset
// bias + pixel * pitch + channel), every piece one aligned 16-byte access, activation a compile-time constant.  Everything that does not
// depend on the slab j -- the lane's rows / channel pieces of the row pass, their offsets and channel bound, the scale / shift vectors -- is
// computed once; per slab what is left is FC x (4 FMA + activation + one b128 LDS write) and ITER x (two b128 LDS reads, the pack, one
// 16-byte store).  For the short-K 1x1 GEMMs (K = 96 .. 384: 3 .. 12 K-steps of 16 MFMAs) the general body above executed more
// instructions than the main loop.
// MODE 0: y = act(affine) (+ res);  1: y2 = affine (pre-activation), y = act(affine) (+ res);  2: y = affine * act'(res)  (ACT = MTBT_ACT_D*)
struct EpiSeq { long pix0; int jstep; long npix; long ybias, rbias; };

// RESPF: the residual / pre-activation pieces of the WHOLE tile are requested before the first slab is written.  With the load next to each
// store (res may BE y: a gradient accumulating into its own buffer) every piece waits for its own round trip -- the compiler cannot move a
// load above the previous, possibly aliasing, store: FP x ITER dependent latencies per tile (round 3: found in the depthwise kernel first,
// where the same pattern cost 40 % of the input-gradient launches).  Reading everything first is safe under aliasing: a lane reads exactly the
// locations it writes later, and tiles are disjoint.  Implicit-GEMM kernels only (FP x ITER x 4 registers; the direct kernels have none to spare).
// RESPF = 2: ONE slab ahead instead of the whole tile (ITER x 4 registers): slab j + 1's pieces are requested before slab j is stored -- for the
// direct 3x3 kernels, which have no room for the whole tile's.
// DB: the wave owns TWO slab regions (slab, slab + 16 * PITCH): slab j + 1 is written while slab j is read back, one LDS round trip per slab
// instead of two (implicit-GEMM kernels: their LDS has the room).
template <typename T, int TC, int FC, int FP, int ACT, int MODE, bool SUMS = false, int RESPF = 0, bool DB = false, bool PERM = false>
__device__ __forceinline__ void conv_epilogue_fast(const ConvP& p, f32x4 (&acc)[FC][FP], char* slab, const float* aff, int cbase, int chl0, int lane,
                                                   const EpiSeq q, long srow = -1) {
  constexpr int WCH = FC * 16;
  constexpr int PITCH = WCH * 4 + 16;
  constexpr int C8 = WCH / 8;
  constexpr int ITER = (16 * C8 + 63) / 64;
  constexpr bool HOIST = FC <= 4;          // scale / shift of the lane's channels in registers across the slabs
  typedef typename half_of<T>::type HT;
  const int lr = lane & 15, lq = lane >> 4;
  float4 sc[HOIST ? FC : 1], sh[HOIST ? FC : 1];
  if (HOIST) {
#pragma unroll
    for (int i = 0; i < FC; ++i) {
      const int cl = chl0 + epi_cpos<PERM>(i, lq);
      sc[HOIST ? i : 0] = *reinterpret_cast<const float4*>(aff + cl);
      sh[HOIST ? i : 0] = *reinterpret_cast<const float4*>(aff + TC + cl);
    }
  }
  long yo[ITER], ro[ITER], px[ITER];
  int so[ITER];
  bool ok[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / C8, c8 = idx - row * C8;
    const int ch = cbase + chl0 + c8 * 8;
    ok[it] = ((16 * C8) % 64 == 0 || idx < 16 * C8) && ch < p.K;
    px[it] = q.pix0 + row;
    yo[it] = q.ybias + px[it] * p.ldy + ch;
    ro[it] = q.rbias + px[it] * p.ldr + ch;
    so[it] = row * PITCH + c8 * 32;
  }
  const long ystep = (long)q.jstep * p.ldy, rstep = (long)q.jstep * p.ldr;
  const bool has_res = p.res != nullptr;
  const int cs_ch = cbase + chl0 + (lane % C8) * 8;     // column sums (SUMS): see colsum_flush
  float cs[8], cq[8], csh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs[e] = 0.f; cq[e] = 0.f; csh[e] = (SUMS && p.cs_shift && cs_ch + e < p.K) ? p.cs_shift[cs_ch + e] : 0.f; }
  HT* const yp = reinterpret_cast<HT*>(p.y);
  HT* const y2p = reinterpret_cast<HT*>(p.y2);
  const T* const rp = reinterpret_cast<const T*>(p.res);
  uint4 rpf[RESPF == 1 ? FP : 1][RESPF ? ITER : 1];     // RESPF == 2: one slab's pieces (the NEXT slab's, see the row pass)
  uint4 rnx[RESPF == 2 ? ITER : 1];
  auto fetch_res = [&](int j, uint4 (&dst)[RESPF ? ITER : 1]) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      dst[RESPF ? it : 0] = uint4{0u, 0u, 0u, 0u};
      if (ok[it] && px[it] + (long)j * q.jstep < q.npix) dst[RESPF ? it : 0] = *reinterpret_cast<const uint4*>(rp + ro[it] + j * rstep);
    }
  };
  if constexpr (RESPF == 2) {
    if (MODE == 2 || has_res) fetch_res(0, rpf[0]);
  }
  if constexpr (RESPF == 1) {
    if (MODE == 2 || has_res) {
#pragma unroll
      for (int j = 0; j < FP; ++j)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          rpf[j][it] = uint4{0u, 0u, 0u, 0u};
          if (ok[it] && px[it] + (long)j * q.jstep < q.npix) rpf[j][it] = *reinterpret_cast<const uint4*>(rp + ro[it] + j * rstep);
        }
    }
  }
  constexpr int SLB = 16 * PITCH;
  auto write_slab = [&](int j, char* sb) {
#pragma unroll
    for (int i = 0; i < FC; ++i) {
      float4 s4, h4;
      if (HOIST) { s4 = sc[HOIST ? i : 0]; h4 = sh[HOIST ? i : 0]; }
      else {
        const int cl = chl0 + epi_cpos<PERM>(i, lq);
        s4 = *reinterpret_cast<const float4*>(aff + cl);
        h4 = *reinterpret_cast<const float4*>(aff + TC + cl);
      }
      float4 v;
      v.x = fmaf(acc[i][j][0], s4.x, h4.x); v.y = fmaf(acc[i][j][1], s4.y, h4.y);
      v.z = fmaf(acc[i][j][2], s4.z, h4.z); v.w = fmaf(acc[i][j][3], s4.w, h4.w);
      if (MODE == 0) { v.x = act_apply(v.x, ACT); v.y = act_apply(v.y, ACT); v.z = act_apply(v.z, ACT); v.w = act_apply(v.w, ACT); }
      *reinterpret_cast<float4*>(sb + lr * PITCH + epi_cpos<PERM>(i, lq) * 4) = v;
    }
  };
  char* const slab0 = slab;
  if constexpr (DB) write_slab(0, slab0);
#pragma clang loop unroll(full)
  for (int j = 0; j < FP; ++j) {
    slab = slab0 + (DB ? (j & 1) * SLB : 0);
    if constexpr (DB) {
      if (j + 1 < FP) {
        write_slab(j + 1, slab0 + ((j + 1) & 1) * SLB);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FC) : "memory");   // slab j is in LDS (its FC writes are older than the FC just issued)
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
      write_slab(j, slab);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local hand-off through LDS (see the general body)
    }
    if constexpr (RESPF == 2) {
      if ((MODE == 2 || has_res) && j + 1 < FP) fetch_res(j + 1, rnx);      // the next slab's residual: in flight across this slab's stores
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (!ok[it] || px[it] + (long)j * q.jstep >= q.npix) continue;
      const float4 lo = *reinterpret_cast<const float4*>(slab + so[it]);
      const float4 hi = *reinterpret_cast<const float4*>(slab + so[it] + 16);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const long yoff = yo[it] + j * ystep, roff = ro[it] + j * rstep;
      if (MODE == 1) {
        st8<HT>(y2p + yoff, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], ACT);
      }
      if (MODE == 2) {
        float r[8];
        if constexpr (RESPF != 0) ld8<T>(reinterpret_cast<const T*>(&rpf[RESPF == 1 ? j : 0][RESPF ? it : 0]), r); else ld8<T>(rp + roff, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= act_grad(r[e], ACT);
      } else if (has_res) {
        float r[8];
        if constexpr (RESPF != 0) ld8<T>(reinterpret_cast<const T*>(&rpf[RESPF == 1 ? j : 0][RESPF ? it : 0]), r); else ld8<T>(rp + roff, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
      st8<HT>(yp + yoff, v);
      if (SUMS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = stored_value<HT>(v[e]) - csh[e]; cs[e] += d; cq[e] += d * d; }
      }
    }
    if constexpr (RESPF == 2) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) rpf[0][it] = rnx[it];
    }
    // slab fully read before it is rewritten (DB: the NEXT write goes to the other region; this one is rewritten a pass later, by
    // which time these reads were consumed -- the barrier below only keeps the compiler from moving that write up)
    if constexpr (DB) asm volatile("" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if constexpr (SUMS && (C8 & (C8 - 1)) == 0) colsum_flush<C8>(p, cs, cq, srow, cs_ch, cs_ch < p.K, lane);
}

// The DIRECT body (PERM kernels, 16-bit NHWC output, MODE 0: y = act(affine) (+ res)): no LDS at all.  Per fragment pair m the lane's 4 + 4
// accumulators of a pixel are 8 consecutive channels (see epi_row_channel): scale / shift / activation in registers, one 16-byte residual
// load, one 16-byte store.  A wave's store instruction covers 16 pixels x 64 contiguous bytes; the pair index completes the rows.  The residual
// pieces of a pair are requested before its first store (the residual may BE the output: a lane reads exactly the pieces it writes).
template <typename T, int TC, int FC, int FP, int ACT>
__device__ __forceinline__ void conv_epilogue_direct(const ConvP& p, f32x4 (&acc)[FC][FP], const float* aff, int cbase, int chl0, int lane, const EpiSeq q) {
  static_assert(FC % 2 == 0 && sizeof(T) == 2, "fragment pairs, 16-bit storage");
  typedef typename half_of<T>::type HT;
  const int lr = lane & 15, lq = lane >> 4;
  HT* const yp = reinterpret_cast<HT*>(p.y);
  const T* const rp = reinterpret_cast<const T*>(p.res);
  const bool has_res = p.res != nullptr;
  const long px0 = q.pix0 + lr;
#pragma unroll
  for (int m = 0; m < FC / 2; ++m) {
    const int cl = chl0 + m * 32 + lq * 8;
    const int ch = cbase + cl;
    if (ch >= p.K) continue;                 // (K is a multiple of 8 on this path)
    const float4 s0 = *reinterpret_cast<const float4*>(aff + cl), s1 = *reinterpret_cast<const float4*>(aff + cl + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(aff + TC + cl), h1 = *reinterpret_cast<const float4*>(aff + TC + cl + 4);
    u32x4 rv[FP];
    if (has_res) {
#pragma unroll
      for (int j = 0; j < FP; ++j) {
        const long px = px0 + (long)j * q.jstep;
        rv[j] = u32x4{0u, 0u, 0u, 0u};
        if (px < q.npix) rv[j] = *reinterpret_cast<const u32x4*>(rp + q.rbias + px * p.ldr + ch);
      }
    }
#pragma unroll
    for (int j = 0; j < FP; ++j) {
      const long px = px0 + (long)j * q.jstep;
      if (px >= q.npix) continue;
      float v[8];
      v[0] = act_apply(fmaf(acc[2 * m][j][0], s0.x, h0.x), ACT); v[1] = act_apply(fmaf(acc[2 * m][j][1], s0.y, h0.y), ACT);
      v[2] = act_apply(fmaf(acc[2 * m][j][2], s0.z, h0.z), ACT); v[3] = act_apply(fmaf(acc[2 * m][j][3], s0.w, h0.w), ACT);
      v[4] = act_apply(fmaf(acc[2 * m + 1][j][0], s1.x, h1.x), ACT); v[5] = act_apply(fmaf(acc[2 * m + 1][j][1], s1.y, h1.y), ACT);
      v[6] = act_apply(fmaf(acc[2 * m + 1][j][2], s1.z, h1.z), ACT); v[7] = act_apply(fmaf(acc[2 * m + 1][j][3], s1.w, h1.w), ACT);
      if (has_res) {
        float r[8];
        ld8<T>(reinterpret_cast<const T*>(&rv[j]), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
      st8<HT>(yp + q.ybias + px * p.ldy + ch, v);
    }
  }
}

// Direct or slab?  Same-box library A/Bs (round 3): the direct body is 3 - 6 % faster per launch on the P3 / P4 shapes of the batch-16 forward (a
// few rounds of tiles: the slab's LDS round trips are exposed latency; step -1.0 ... -1.6 %), level on the 1280 x 1280 batch-64 forward and ~0.5 %
// slower on the batch-32 training step (many rounds, bandwidth-bound: the slab's whole-row stores beat 64-byte pieces).  A run-time switch on the
// pixel count compiled BOTH bodies into every kernel and lost the inference gain again (code size): the direct body is unconditional.

// Dispatcher: ONE switch per call.  `seq` non-null = the caller's output pixels form the arithmetic sequence the fast body wants.
template <typename T, int TC, int FC, int FP, bool TRAIN = true, int RESPF = (TRAIN ? 1 : 0), bool DB = false, bool PERM = false, typename AddrFn>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x4 (&acc)[FC][FP], char* slab, const float* aff, int cbase,
                                              int chl0 /* first channel-in-tile of this wave */, int lane, AddrFn addr, const EpiSeq seq, bool use_seq,
                                              long srow = -1 /* partial row of the column sums this wave writes (p.cs_part) */) {
  // (seq by VALUE: behind a conditional pointer the struct was materialised in scratch memory in the fp16 kernels)
#define MTBT_FAST(ACTV, MODEV)                                                                                                     \
  do {                                                                                                                            \
    if constexpr (PERM && (MODEV) == 0) conv_epilogue_direct<T, TC, FC, FP, ACTV>(p, acc, aff, cbase, chl0, lane, seq);           \
    else conv_epilogue_fast<T, TC, FC, FP, ACTV, MODEV, false, RESPF, DB, PERM>(p, acc, slab, aff, cbase, chl0, lane, seq);        \
  } while (0)
  // (fp32 storage always writes fp32: the fast bodies -- 16-bit outputs -- are not even compiled for it)
  const bool fast = sizeof(T) == 2 && use_seq && p.vec_ok && !(p.K & 7) && !p.out_f32 && p.out_mode == MTBT_OUT_NHWC;
  constexpr bool CS_OK = (((FC * 16) / 8) & ((FC * 16) / 8 - 1)) == 0;
  if constexpr (sizeof(T) == 2) {
  if (CS_OK && fast && p.cs_part && srow >= 0 && !(TRAIN && p.y2)) {   // column sums: the raw conv in front of a BatchNorm, fc2-dgrad * GELU' (d fc1 bias)
    if (p.act == MTBT_ACT_NONE) { conv_epilogue_fast<T, TC, FC, FP, MTBT_ACT_NONE, 0, true, RESPF, DB, PERM>(p, acc, slab, aff, cbase, chl0, lane, seq, srow); return; }
    // (16-bit storage: the polynomial derivative has the compiled body; MTBT_ACT_DGELU falls through to the general one)
    if (TRAIN && p.act == MTBT_ACT_DGELU_POLY) { conv_epilogue_fast<T, TC, FC, FP, MTBT_ACT_DGELU_POLY, 2, true, RESPF, DB, PERM>(p, acc, slab, aff, cbase, chl0, lane, seq, srow); return; }
  } else if (fast && !(TRAIN && p.y2)) {
    switch (p.act) {
      case MTBT_ACT_NONE: MTBT_FAST(MTBT_ACT_NONE, 0); return;
      case MTBT_ACT_SILU: MTBT_FAST(MTBT_ACT_SILU, 0); return;
      case MTBT_ACT_ELU: MTBT_FAST(MTBT_ACT_ELU, 0); return;
      case MTBT_ACT_GELU: MTBT_FAST(MTBT_ACT_GELU, 0); return;
      case MTBT_ACT_GELU_POLY: MTBT_FAST(MTBT_ACT_GELU_POLY, 0); return;
      case MTBT_ACT_DSILU: if (TRAIN) { MTBT_FAST(MTBT_ACT_DSILU, 2); return; } break;
      case MTBT_ACT_DELU: if (TRAIN) { MTBT_FAST(MTBT_ACT_DELU, 2); return; } break;
      case MTBT_ACT_DGELU_POLY: if (TRAIN) { MTBT_FAST(MTBT_ACT_DGELU_POLY, 2); return; } break;
      default: break;
    }
  } else if (TRAIN && fast) {   // second output = the pre-activation (training forward of fc1)
    switch (p.act) {
      case MTBT_ACT_NONE: MTBT_FAST(MTBT_ACT_NONE, 1); return;
      case MTBT_ACT_GELU: MTBT_FAST(MTBT_ACT_GELU, 1); return;
      case MTBT_ACT_GELU_POLY: MTBT_FAST(MTBT_ACT_GELU_POLY, 1); return;
      default: break;
    }
  }
  }
#undef MTBT_FAST
  // everything else (fp32 / ragged / transposed-conv outputs, rare activation + mode pairs): the general body, run-time activation
  conv_epilogue_body<T, TC, FC, FP, TRAIN, -1, false, PERM>(p, acc, slab, aff, cbase, chl0, lane, addr, srow);
}

}  // namespace
