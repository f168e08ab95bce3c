// Post-process kernels: box decode, per-image NMS, prototype x coefficient mask assembly.
// Integer / index results (NMS) are bit-exact restatements of the reference's CPU arithmetic:
// no FMA contraction, IEEE division, stable ordering.
#include "common.h"

// No FMA contraction in this file (built with -ffp-contract=off, see build.py): the NMS suppression test and the
// pairwise IoU must round exactly like the CPU reference (torchvision / torch eager evaluate every product and sum
// separately).  hipcc's default contracts in the backend, and the __fmul_rn / __fadd_rn spellings are plain operators.

namespace {

// ------------------------------------------------------------------------------------------------
// Decode: 4 lanes per anchor, lane `side` owns one of (l,t,r,b): softmax over reg_max bins and the
// expectation with arange; the quad exchanges the four distances by shuffles; the class part is
// split over the quad and reduced to (best score, first best label).
// ------------------------------------------------------------------------------------------------
struct DecodeP {
  const float* map[3];
  int h[3], w[3], ld[3], off[4];
  float stride[3];
  int n_levels, N, nc, reg_max, xywh, A;
  float* boxes;
  float* scores;
  float* best_score;
  int* best_label;
  float* preds_cat;
  int cat_stride;
};

__global__ void decode_kernel(const DecodeP p) {
  const long g = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int side = threadIdx.x & 3;
  const long total = (long)p.N * p.A;
  const bool live = g < total;
  const long gg = live ? g : 0;
  const int n = (int)(gg / p.A), a = (int)(gg - (long)n * p.A);
  int l = 0;
  if (p.n_levels > 1 && a >= p.off[1]) l = 1;
  if (p.n_levels > 2 && a >= p.off[2]) l = 2;
  const int cell = a - p.off[l];
  const int w = p.w[l], hw = p.h[l] * w;
  const int cy = cell / w, cx = cell - cy * w;
  const float* row = p.map[l] + ((long)n * hw + cell) * p.ld[l];

  // distribution of this lane's side: one pass, exponentials kept in registers (reg_max <= 16 fast path with
  // 8-byte loads: rows are 4*reg_max+nc floats, 8-byte aligned whenever the pixel stride is even)
  const float* d = row + side * p.reg_max;
  float dist = 0.f;
  if (p.reg_max == 16 && ((p.ld[l] & 1) == 0)) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i += 2) { const float2 t = *reinterpret_cast<const float2*>(d + i); v[i] = t.x; v[i + 1] = t.y; }
    float m = v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, v[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) dist += (v[i] / s) * (float)i;
  } else {
    float m = -INFINITY;
    for (int i = 0; i < p.reg_max; ++i) m = fmaxf(m, d[i]);
    float s = 0.f;
    for (int i = 0; i < p.reg_max; ++i) s += expf(d[i] - m);
    for (int i = 0; i < p.reg_max; ++i) dist += (expf(d[i] - m) / s) * (float)i;
  }

  const int qbase = (threadIdx.x & 63) & ~3;
  const float dl = __shfl(dist, qbase + 0, 64), dt = __shfl(dist, qbase + 1, 64);
  const float dr = __shfl(dist, qbase + 2, 64), db = __shfl(dist, qbase + 3, 64);
  const float ax = cx + 0.5f, ay = cy + 0.5f, st = p.stride[l];
  float out;
  if (!p.xywh) {  // running_main_v3.py:100-110,529: dist2bbox(ltrb*stride, anchor*stride)
    const float a_ = (side & 1) ? ay * st : ax * st;
    const float d_ = (side == 0 ? dl : side == 1 ? dt : side == 2 ? dr : db) * st;
    out = (side < 2) ? a_ - d_ : a_ + d_;
  } else {        // ultralytics dist2bbox(xywh=True) in grid units, then * stride
    const float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;
    out = (side == 0 ? (x1 + x2) / 2 : side == 1 ? (y1 + y2) / 2 : side == 2 ? x2 - x1 : y2 - y1) * st;
  }
  if (live) {
    if (p.boxes) p.boxes[g * 4 + side] = out;
    if (p.preds_cat) p.preds_cat[g * p.cat_stride + side] = out;
  }
  // classes
  const float* cl = row + 4 * p.reg_max;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = side; c < p.nc; c += 4) {
    const float sc = 1.0f / (1.0f + expf(-cl[c]));
    if (live) {
      if (p.scores) p.scores[g * p.nc + c] = sc;
      if (p.preds_cat) p.preds_cat[g * p.cat_stride + 4 + c] = sc;
    }
    if (sc > best) { best = sc; bi = c; }
  }
#pragma unroll
  for (int o = 1; o < 4; o <<= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (live && side == 0) {
    if (p.best_score) p.best_score[g] = best;
    if (p.best_label) p.best_label[g] = bi;
  }
}

// ------------------------------------------------------------------------------------------------
// NMS.  One 1024-thread workgroup per image.
//  1. stable compaction of { a : score[a] > conf_th } (each wave owns a contiguous anchor range: count, prefix over the
//     waves, ballot-prefix writes) -> cand list, plus a 4096-bin histogram of the candidates' scores
//  2. PRESELECTION: the greedy pass almost never looks past the first few hundred candidates, so only the candidates of
//     the top histogram bins (the fewest bins holding >= NMS_TARGET of them) are sorted first; every candidate left out has
//     a strictly smaller score than every one selected, so the sorted selection IS the head of the full order.  If the
//     greedy pass runs out of selected candidates before top_k boxes are kept, the whole list is sorted and the pass redone
//     (same result as sorting everything up front, which is what attempt 1 is).
//  3. bitonic sort of 64-bit keys (~orderable(score) << 32 | cand index): descending score,
//     ascending candidate index on ties == torch's stable descending sort
//  4. greedy pass by wave 0, 64 sorted candidates per step: each lane tests its candidate against
//     every box kept so far, the 64x64 in-chunk dependencies are bitmasks resolved by a scalar scan
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool iou_gt(const float4 bi, const float4 bj, float thr) {
  // torchvision nms_kernel.cpp arithmetic, i = kept (earlier) box, j = candidate
  const float iarea = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
  const float jarea = __fmul_rn(__fsub_rn(bj.z, bj.x), __fsub_rn(bj.w, bj.y));
  const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
  const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
  const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
  const float inter = __fmul_rn(w, h);
  const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(iarea, jarea), inter));
  return ovr > thr;
}

__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int NMS_KEPT_LDS = 1024;  // kept boxes cached in LDS; beyond that they are read back from out_boxes
constexpr int NMS_NT = 1024, NMS_NW = NMS_NT / 64;
constexpr int NMS_BINS = 4096;      // score histogram (aliases the kept-box cache: 16 KiB, used before the greedy pass)
constexpr int NMS_TARGET = 1024;    // preselect at least this many candidates ...
constexpr int NMS_SEL_MAX = 4096;   // ... and sort everything at once when the top bins hold more than this

// monotone non-decreasing in the score (any float; lo / scale = the candidates' own score range, so that scores crowded into a narrow
// band -- random-initialised heads: everything near 0.5 -- still spread over the bins): a candidate in a lower bin has a strictly smaller score
__device__ __forceinline__ int nms_bin(float v, float lo, float scale) {
  return (int)fminf(fmaxf((v - lo) * scale, 0.f), (float)(NMS_BINS - 1));
}

__global__ __launch_bounds__(NMS_NT) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ score,
                                                     const int* __restrict__ label, int A, float conf_th, float iou_th,
                                                     float clamp_max, int top_k, long long* __restrict__ keep_idx,
                                                     int* __restrict__ keep_anchor, float* __restrict__ out_boxes,
                                                     float* __restrict__ out_scores, long long* __restrict__ out_labels,
                                                     int* __restrict__ counts, int* __restrict__ n_cand, char* __restrict__ ws,
                                                     long ws_per_image, int P2, int keys_in_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int wave_tot[NMS_NW];
  __shared__ int s_cnt, s_bin, s_sel, s_redo, s_nkept;
  __shared__ unsigned s_lo, s_hi;        // orderable(min / max candidate score)
  __shared__ unsigned long long part_supp[NMS_NW * 64], part_dead[NMS_NW];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* wsi = ws + (long)n * ws_per_image;
  float4* sbox = reinterpret_cast<float4*>(wsi);                      // [A] sorted, clamped boxes
  int* cand_anchor = reinterpret_cast<int*>(wsi + (long)A * 16);     // [A]
  unsigned long long* gkeys = reinterpret_cast<unsigned long long*>(wsi + (long)A * 20 + ((16 - ((long)A * 20) % 16) % 16));
  float4* kept_lds = reinterpret_cast<float4*>(smem);                // [NMS_KEPT_LDS]
  int* hist = reinterpret_cast<int*>(smem);                          // [NMS_BINS] (before the greedy pass)
  unsigned long long* keys = keys_in_lds ? reinterpret_cast<unsigned long long*>(smem + NMS_KEPT_LDS * 16) : gkeys;
  const float* sc = score + (long)n * A;
  const unsigned long long below = (1ull << lane) - 1ull;

  // ---- 1. compaction (ascending anchor order) + score histogram ----
  for (int i = tid; i < NMS_BINS; i += NMS_NT) hist[i] = 0;
  if (tid == 0) { s_lo = 0xffffffffu; s_hi = 0u; }
  __syncthreads();
  const int per_wave = ((A + NMS_NW - 1) / NMS_NW + 63) / 64 * 64;
  const int a_lo = wave * per_wave, a_hi = min(A, a_lo + per_wave);
  int mine = 0;
  unsigned lo_u = 0xffffffffu, hi_u = 0u;
  for (int a0 = a_lo; a0 < a_hi; a0 += 64) {
    const int a = a0 + lane;
    const float v = a < a_hi ? sc[a] : 0.f;
    const bool f = a < a_hi && v > conf_th;
    mine += __popcll(__ballot(f));
    if (f) { const unsigned u = orderable(v); lo_u = min(lo_u, u); hi_u = max(hi_u, u); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo_u = min(lo_u, (unsigned)__shfl_xor((int)lo_u, o, 64)); hi_u = max(hi_u, (unsigned)__shfl_xor((int)hi_u, o, 64)); }
  if (lane == 0) { wave_tot[wave] = mine; atomicMin(&s_lo, lo_u); atomicMax(&s_hi, hi_u); }
  __syncthreads();
  int off = 0, M = 0;
  for (int k = 0; k < NMS_NW; ++k) { if (k < wave) off += wave_tot[k]; M += wave_tot[k]; }
  // the candidates' score range -> histogram bins (orderable() is its own inverse up to the sign handling below)
  auto unorder = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
  const float b_lo = M > 0 ? unorder(s_lo) : 0.f, b_hi = M > 0 ? unorder(s_hi) : 1.f;
  const float b_scale = b_hi > b_lo ? (float)NMS_BINS / (b_hi - b_lo) : 0.f;      // (all equal: one bin, the full sort below)
  for (int a0 = a_lo; a0 < a_hi; a0 += 64) {
    const int a = a0 + lane;
    const float v = a < a_hi ? sc[a] : 0.f;
    const bool f = a < a_hi && v > conf_th;
    const unsigned long long bal = __ballot(f);
    if (f) {
      cand_anchor[off + __popcll(bal & below)] = a;
      atomicAdd(&hist[nms_bin(v, b_lo, b_scale)], 1);
    }
    off += __popcll(bal);
  }
  __threadfence_block();
  __syncthreads();

  // ---- 2. the lowest histogram bin of the preselection: the largest b with  #{bin >= b} >= min(M, NMS_TARGET) ----
  if (tid == 0) { s_bin = 0; s_sel = M; }
  __syncthreads();
  if (M > NMS_TARGET) {
    const int h0 = hist[tid * 4], h1 = hist[tid * 4 + 1], h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
    int x = h0 + h1 + h2 + h3;                                      // -> inclusive suffix sum over the threads
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(x, o, 64); if (lane + o < 64) x += t; }
    __syncthreads();                                                // (wave_tot is reused)
    if (lane == 0) wave_tot[wave] = x;
    __syncthreads();
    for (int k = wave + 1; k < NMS_NW; ++k) x += wave_tot[k];
    // S(b) for this thread's four bins and the one above them
    const int s3 = x - h0 - h1 - h2, s2 = x - h0 - h1, s1 = x - h0, s0 = x, s4 = s3 - h3;
    const int S[5] = {s0, s1, s2, s3, s4};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (S[i] >= NMS_TARGET && S[i + 1] < NMS_TARGET) { s_bin = tid * 4 + i; s_sel = S[i]; }
  }
  __syncthreads();
  const int sel_bin = s_bin, sel_cnt = s_sel;
  const bool presel = M > NMS_TARGET && sel_cnt <= NMS_SEL_MAX && sel_cnt < M;
  __syncthreads();

  long long* ki = keep_idx + (long)n * top_k;
  int* ka = keep_anchor + (long)n * top_k;
  float4* ob = reinterpret_cast<float4*>(out_boxes) + (long)n * top_k;
  float* os = out_scores + (long)n * top_k;
  long long* ol = out_labels + (long)n * top_k;
  const float4* bx = reinterpret_cast<const float4*>(boxes) + (long)n * A;

  for (int attempt = presel ? 0 : 1; attempt < 2; ++attempt) {
    // ---- 3. keys of the selection (attempt 0) / of every candidate (attempt 1), bitonic sort ascending ----
    const int Ms = attempt == 0 ? sel_cnt : M;
    if (tid == 0) { s_cnt = 0; s_redo = 0; }
    __syncthreads();
    for (int p0 = wave * 64; p0 < M; p0 += NMS_NT) {
      const int pos = p0 + lane;
      const float v = pos < M ? sc[cand_anchor[pos]] : 0.f;
      if (attempt == 1) {
        if (pos < M) keys[pos] = ((unsigned long long)(~orderable(v)) << 32) | (unsigned)pos;
      } else {
        const bool f = pos < M && nms_bin(v, b_lo, b_scale) >= sel_bin;
        const unsigned long long bal = __ballot(f);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_cnt, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (f) keys[base + __popcll(bal & below)] = ((unsigned long long)(~orderable(v)) << 32) | (unsigned)pos;   // (any slot: sorted next)
      }
    }
    int P = 1;
    while (P < Ms) P <<= 1;
    for (int i = Ms + tid; i < P; i += NMS_NT) keys[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (P >> 1); t += NMS_NT) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool up = (lo & k) == 0;
          const unsigned long long x = keys[lo], y = keys[hi];
          if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
        }
        __syncthreads();
      }
    }

    // sorted, clamped boxes to workspace
    for (int i = tid; i < Ms; i += NMS_NT) {
      const int ci = (int)(keys[i] & 0xffffffffu);
      float4 b = bx[cand_anchor[ci]];
      b.x = fminf(fmaxf(b.x, 0.f), clamp_max); b.y = fminf(fmaxf(b.y, 0.f), clamp_max);
      b.z = fminf(fmaxf(b.z, 0.f), clamp_max); b.w = fminf(fmaxf(b.w, 0.f), clamp_max);
      sbox[i] = b;
    }
    __threadfence_block();
    __syncthreads();

    // ---- 4. greedy, 64 sorted candidates per step, all waves: every wave holds the chunk (lane = candidate); wave w tests it
    //         against kept boxes w, w + 16, ... and against chunk members 4w .. 4w + 3, the partial results meet in LDS and
    //         wave 0 resolves the in-chunk order (scalar scan over 64-bit masks) and writes the newly kept boxes ----
    if (tid == 0) s_nkept = 0;
    __syncthreads();
    for (int j0 = 0; j0 < Ms; j0 += 64) {
      const int nkept = s_nkept;
      if (nkept >= top_k) break;                 // (uniform)
      const int j = j0 + lane;
      const bool have = j < Ms;
      const float4 mine4 = have ? sbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      bool dead = false;
      for (int k = wave; k < nkept; k += NMS_NW) {
        float4 kb;
        if (k < NMS_KEPT_LDS) {
          kb = kept_lds[k];
        } else {   // beyond the LDS cache: written by wave 0 in an earlier step -- read past this CU's L1 (device-scope loads)
          const float* g = out_boxes + ((long)n * top_k + k) * 4;
          kb.x = __hip_atomic_load(g + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          kb.y = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          kb.z = __hip_atomic_load(g + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          kb.w = __hip_atomic_load(g + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (iou_gt(kb, mine4, iou_th)) dead = true;
      }
      unsigned long long supp_by = 0ull;         // bit i set: earlier chunk member i would suppress me
#pragma unroll
      for (int t = 0; t < 64 / NMS_NW; ++t) {
        const int i = wave * (64 / NMS_NW) + t;
        float4 o;
        o.x = __shfl(mine4.x, i, 64); o.y = __shfl(mine4.y, i, 64); o.z = __shfl(mine4.z, i, 64); o.w = __shfl(mine4.w, i, 64);
        if (i < lane && iou_gt(o, mine4, iou_th)) supp_by |= 1ull << i;
      }
      part_supp[wave * 64 + lane] = supp_by;
      const unsigned long long dead_mask = __ballot(dead);
      if (lane == 0) part_dead[wave] = dead_mask;
      __syncthreads();
      if (wave == 0) {
        unsigned long long dm = 0ull;
#pragma unroll
        for (int w = 0; w < NMS_NW; ++w) { supp_by |= part_supp[w * 64 + lane]; dm |= part_dead[w]; }   // (own part: OR-ed twice, harmless)
        const unsigned long long alive_mask = __ballot(have) & ~dm;
        const unsigned lo32 = (unsigned)supp_by, hi32 = (unsigned)(supp_by >> 32);
        unsigned long long keepmask = 0ull;
#pragma unroll
        for (int L = 0; L < 64; ++L) {
          const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi32, L) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)lo32, L);
          if (((alive_mask >> L) & 1ull) && (sb & keepmask) == 0ull) keepmask |= 1ull << L;
        }
        const bool kept = (keepmask >> lane) & 1ull;
        const int pos = nkept + __popcll(keepmask & below);
        if (kept && pos < top_k) {
          const unsigned long long key = keys[j];
          const int ci = (int)(key & 0xffffffffu);
          const int a = cand_anchor[ci];
          ki[pos] = ci;
          ka[pos] = a;
          ob[pos] = mine4;
          os[pos] = sc[a];
          ol[pos] = label ? (long long)label[(long)n * A + a] : 0ll;
          if (pos < NMS_KEPT_LDS) kept_lds[pos] = mine4;
        }
        if (lane == 0) s_nkept = min(top_k, nkept + (int)__popcll(keepmask));
      }
      __threadfence_block();
      __syncthreads();
    }
    if (wave == 0) {
      const int nkept = s_nkept;
      if (attempt == 0 && nkept < top_k) {       // the selection ran out: sort everything and start over
        if (lane == 0) s_redo = 1;
      } else {
        for (int k = nkept + lane; k < top_k; k += 64) {
          ki[k] = -1; ka[k] = -1; ob[k] = make_float4(0.f, 0.f, 0.f, 0.f); os[k] = 0.f; ol[k] = -1;
        }
        if (lane == 0) { counts[n] = nkept; if (n_cand) n_cand[n] = M; }
      }
    }
    __syncthreads();
    if (attempt == 0 && !s_redo) break;     // (uniform)
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Mask assembly / proto projector.  Block = (tile, k, n); 64x64 output tile, 256 threads.
//   low-res patch (+1 halo) = coeff . protos in fp32 -> LDS;  bilinear (align_corners=False) -> out.
// ------------------------------------------------------------------------------------------------
struct MaskP {
  const float* protos;
  const float* coeff;
  long cbs, cks, ccs;
  const int* gather;
  const int* counts;
  float bias;
  int N, K, nm, hp, wp, Hout, Wout;
  float* logits;
  unsigned char* masks;
  int tiles_x;
};

constexpr int MT = 64;  // output tile edge

__global__ __launch_bounds__(256) void mask_kernel(const MaskP p) {
  __shared__ float patch[(MT + 2) * (MT + 2)];
  __shared__ float cf[64];
  const int n = blockIdx.z, k = blockIdx.y, tid = threadIdx.x;
  const int ty0 = (blockIdx.x / p.tiles_x) * MT, tx0 = (blockIdx.x % p.tiles_x) * MT;
  if (p.counts && k >= p.counts[n]) {  // padded slot: defined (zero) output, no compute
    const int row_ = tid >> 2, seg_ = tid & 3, oy_ = ty0 + row_, ox_ = tx0 + seg_ * 16;
    if (oy_ >= p.Hout) return;
    const long ob_ = (((long)n * p.K + k) * p.Hout + oy_) * p.Wout + ox_;
    for (int e = 0; e < 16 && ox_ + e < p.Wout; ++e) {
      if (p.logits) p.logits[ob_ + e] = 0.f;
      if (p.masks) p.masks[ob_ + e] = 0;
    }
    return;
  }
  const float ry = (float)p.hp / (float)p.Hout, rx = (float)p.wp / (float)p.Wout;
  auto src = [](int d, float r) { float s = (d + 0.5f) * r - 0.5f; return s < 0.f ? 0.f : s; };
  const int ty1 = min(ty0 + MT, p.Hout) - 1, tx1 = min(tx0 + MT, p.Wout) - 1;
  const int ly0 = (int)src(ty0, ry), lx0 = (int)src(tx0, rx);
  const int ly1 = min((int)src(ty1, ry) + 1, p.hp - 1), lx1 = min((int)src(tx1, rx) + 1, p.wp - 1);
  const int ph = ly1 - ly0 + 1, pw = lx1 - lx0 + 1;  // <= MT+2 for r <= 1
  if (tid < p.nm) {
    const long kk = p.gather ? p.gather[(long)n * p.K + k] : k;
    cf[tid] = p.coeff[(long)n * p.cbs + kk * p.cks + tid * p.ccs];
  }
  __syncthreads();
  const float* pr = p.protos + (long)n * p.hp * p.wp * p.nm;
  for (int i = tid; i < ph * pw; i += 256) {
    const int py = i / pw, px = i - py * pw;
    const float* v = pr + ((long)(ly0 + py) * p.wp + (lx0 + px)) * p.nm;
    float acc = 0.f;
    for (int c = 0; c < p.nm; c += 4) {
      const float4 q = *reinterpret_cast<const float4*>(v + c);
      acc = fmaf(cf[c], q.x, acc); acc = fmaf(cf[c + 1], q.y, acc);
      acc = fmaf(cf[c + 2], q.z, acc); acc = fmaf(cf[c + 3], q.w, acc);
    }
    patch[py * pw + px] = acc + p.bias;
  }
  __syncthreads();
  const int row = tid >> 2, seg = tid & 3;  // 64 rows x 4 segments of 16 px
  const int oy = ty0 + row;
  if (oy >= p.Hout) return;
  const float sy = src(oy, ry);
  const int y0 = (int)sy, y1 = y0 + (y0 < p.hp - 1 ? 1 : 0);
  const float wy1 = sy - y0, wy0 = 1.f - wy1;
  const float* r0 = patch + (y0 - ly0) * pw;
  const float* r1 = patch + (y1 - ly0) * pw;
  float o[16];
  const int ox0 = tx0 + seg * 16;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int ox = ox0 + e;
    const float sx = src(ox < p.Wout ? ox : p.Wout - 1, rx);
    const int x0 = (int)sx, x1 = x0 + (x0 < p.wp - 1 ? 1 : 0);
    const float wx1 = sx - x0, wx0 = 1.f - wx1;
    o[e] = wy0 * (wx0 * r0[x0 - lx0] + wx1 * r0[x1 - lx0]) + wy1 * (wx0 * r1[x0 - lx0] + wx1 * r1[x1 - lx0]);
  }
  const long obase = (((long)n * p.K + k) * p.Hout + oy) * p.Wout + ox0;
  const bool full = ox0 + 16 <= p.Wout && (p.Wout % 16 == 0);
  if (p.logits) {
    if (full) {
#pragma unroll
      for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(p.logits + obase + e) = make_float4(o[e], o[e + 1], o[e + 2], o[e + 3]);
    } else {
      for (int e = 0; e < 16 && ox0 + e < p.Wout; ++e) p.logits[obase + e] = o[e];
    }
  }
  if (p.masks) {
    unsigned char mb[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) mb[e] = (1.0f / (1.0f + expf(-o[e])) > 0.5f) ? 1 : 0;
    if (full) {
      uint4 u;
      u.x = mb[0] | (mb[1] << 8) | (mb[2] << 16) | ((unsigned)mb[3] << 24);
      u.y = mb[4] | (mb[5] << 8) | (mb[6] << 16) | ((unsigned)mb[7] << 24);
      u.z = mb[8] | (mb[9] << 8) | (mb[10] << 16) | ((unsigned)mb[11] << 24);
      u.w = mb[12] | (mb[13] << 8) | (mb[14] << 16) | ((unsigned)mb[15] << 24);
      *reinterpret_cast<uint4*>(p.masks + obase) = u;
    } else {
      for (int e = 0; e < 16 && ox0 + e < p.Wout; ++e) p.masks[obase + e] = mb[e];
    }
  }
}

inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
inline long nms_ws_per_image(int A) {
  long b = (long)A * 20;
  b += (16 - b % 16) % 16;
  b += (long)pow2ceil(A) * 8;
  return (b + 255) / 256 * 256;
}

// pairwise IoU: thread = (row i, 4 consecutive columns j); boxes2 is small (ground-truth boxes) and stays in L1/L2
__global__ void bbox_iou_kernel(const float4* __restrict__ b1, int n, const float4* __restrict__ b2, int m, float eps,
                                float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int mg = (m + 3) >> 2;
  const long i = t / mg;
  const int j0 = (int)(t - i * mg) * 4;
  if (i >= n) return;
  const float4 a = b1[i];
  const float a1 = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
  for (int j = j0; j < j0 + 4 && j < m; ++j) {
    const float4 b = b2[j];
    const float a2 = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    const float iw = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
    const float ih = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
    const float inter = __fmul_rn(iw, ih);
    out[i * m + j] = __fdiv_rn(inter, __fadd_rn(__fsub_rn(__fadd_rn(a1, a2), inter), eps));
  }
}

}  // namespace

extern "C" int mtbt_bbox_iou_pairwise(const float* boxes1, int n, const float* boxes2, int m, float eps, float* out, void* stream) {
  if (n < 0 || m < 0) return MTBT_EINVAL;
  if (n == 0 || m == 0) return MTBT_OK;
  if (!boxes1 || !boxes2 || !out) return MTBT_EINVAL;
  if (!aligned16(boxes1) || !aligned16(boxes2)) return MTBT_EALIGN;
  const long total = (long)n * ((m + 3) / 4);
  const long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  hipLaunchKernelGGL(bbox_iou_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(boxes1), n, reinterpret_cast<const float4*>(boxes2), m, eps, out);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_decode_boxes(const mtbt_decode_args* a, void* stream) {
  if (!a || a->n_levels < 1 || a->n_levels > 3 || a->N <= 0 || a->nc <= 0 || a->reg_max <= 0 || a->reg_max > 64) return MTBT_EINVAL;
  DecodeP p;
  int A = 0;
  for (int l = 0; l < 3; ++l) {
    p.off[l] = A;
    if (l < a->n_levels) {
      if (!a->map[l] || a->h[l] <= 0 || a->w[l] <= 0 || a->map_pixel_stride[l] < 4 * a->reg_max + a->nc) return MTBT_EINVAL;
      p.map[l] = a->map[l]; p.h[l] = a->h[l]; p.w[l] = a->w[l]; p.ld[l] = a->map_pixel_stride[l]; p.stride[l] = a->stride[l];
      A += a->h[l] * a->w[l];
    } else { p.map[l] = nullptr; p.h[l] = p.w[l] = 1; p.ld[l] = 0; p.stride[l] = 0.f; }
  }
  p.off[3] = A;
  if (a->preds_cat && a->cat_stride < 4 + a->nc) return MTBT_EINVAL;
  p.n_levels = a->n_levels; p.N = a->N; p.nc = a->nc; p.reg_max = a->reg_max; p.xywh = a->xywh; p.A = A;
  p.boxes = a->boxes; p.scores = a->scores; p.best_score = a->best_score; p.best_label = a->best_label;
  p.preds_cat = a->preds_cat; p.cat_stride = a->cat_stride;
  const long total = (long)a->N * A;
  const long blocks = (total + 63) / 64;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int64_t mtbt_nms_workspace_bytes(int N, int A) {
  if (N <= 0 || A <= 0) return 0;
  return (int64_t)N * nms_ws_per_image(A);
}

extern "C" int mtbt_nms_batched(const float* boxes, const float* best_score, const int32_t* best_label, int N, int A,
                                float conf_th, float iou_th, float clamp_max, int top_k, int64_t* keep_idx,
                                int32_t* keep_anchor, float* out_boxes, float* out_scores, int64_t* out_labels,
                                int32_t* counts, int32_t* n_cand, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!boxes || !best_score || !keep_idx || !keep_anchor || !out_boxes || !out_scores || !out_labels || !counts || !workspace)
    return MTBT_EINVAL;
  if (N <= 0 || A <= 0 || top_k <= 0) return MTBT_EINVAL;
  if (!aligned16(boxes) || !aligned16(out_boxes) || !aligned16(workspace)) return MTBT_EALIGN;
  const long per = nms_ws_per_image(A);
  if (workspace_bytes < (int64_t)N * per) return MTBT_EWORKSPACE;
  const int P2 = pow2ceil(A);
  const int keys_in_lds = (long)P2 * 8 <= 128 * 1024 ? 1 : 0;
  const size_t lds = (size_t)NMS_KEPT_LDS * 16 + (keys_in_lds ? (size_t)P2 * 8 : 0);
  // one-time (per device) opt-in to the LARGEST dynamic LDS this kernel ever asks for: an idempotent driver attribute, not state
  if (int rc = mtbt_allow_lds(nms_kernel, NMS_KEPT_LDS * 16 + 128 * 1024)) return rc;
  hipLaunchKernelGGL(nms_kernel, dim3(N), dim3(NMS_NT), lds, reinterpret_cast<hipStream_t>(stream), boxes, best_score, best_label, A,
                     conf_th, iou_th, clamp_max, top_k, reinterpret_cast<long long*>(keep_idx), keep_anchor, out_boxes, out_scores,
                     reinterpret_cast<long long*>(out_labels), counts, n_cand, reinterpret_cast<char*>(workspace), per, P2, keys_in_lds);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

int mtbt_mask_x4_try(const mtbt_mask_args* a, hipStream_t stream);  // mask_mfma.hip: MFMA fast path for x4 upsampling

extern "C" int mtbt_mask_assemble(const mtbt_mask_args* a, void* stream) {
  if (!a || !a->protos || !a->coeff || (!a->logits && !a->masks)) return MTBT_EINVAL;
  if (a->N <= 0 || a->K <= 0 || a->nm <= 0 || a->nm > 64 || a->nm % 4 || a->hp <= 0 || a->wp <= 0) return MTBT_EINVAL;
  if (a->Hout < a->hp || a->Wout < a->wp) return MTBT_EINVAL;  // upsampling (or identity) only
  if (a->K > 65535 || a->N > 65535) return MTBT_EINVAL;
  if (!aligned16(a->protos)) return MTBT_EALIGN;
  if (a->logits && !aligned16(a->logits)) return MTBT_EALIGN;
  if (a->masks && !aligned16(a->masks)) return MTBT_EALIGN;
  {
    const int rc = mtbt_mask_x4_try(a, reinterpret_cast<hipStream_t>(stream));
    if (rc != 1) return rc;
  }
  MaskP p;
  p.protos = a->protos; p.coeff = a->coeff; p.cbs = a->coeff_batch_stride; p.cks = a->coeff_k_stride; p.ccs = a->coeff_c_stride;
  p.gather = a->gather_idx; p.counts = a->counts; p.bias = a->bias;
  p.N = a->N; p.K = a->K; p.nm = a->nm; p.hp = a->hp; p.wp = a->wp; p.Hout = a->Hout; p.Wout = a->Wout;
  p.logits = a->logits; p.masks = a->masks;
  p.tiles_x = (a->Wout + MT - 1) / MT;
  const int tiles_y = (a->Hout + MT - 1) / MT;
  hipLaunchKernelGGL(mask_kernel, dim3(p.tiles_x * tiles_y, a->K, a->N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_abi_version(void) { return MTBT_ABI_VERSION; }
extern "C" int mtbt_sizeof_args(int which) {
  switch (which) {
    case 0: return (int)sizeof(mtbt_conv_args);
    case 1: return (int)sizeof(mtbt_fuse_args);
    case 2: return (int)sizeof(mtbt_decode_args);
    case 3: return (int)sizeof(mtbt_mask_args);
    case 4: return (int)sizeof(mtbt_loss_args);
    case 5: return (int)sizeof(mtbt_prep_desc);
    case 6: return (int)sizeof(mtbt_raw_image);
    case 7: return (int)sizeof(mtbt_upconv_args);
    case 8: return (int)sizeof(mtbt_node_args);
    default: return -1;
  }
}
extern "C" const char* mtbt_target_arch(void) { return "gfx950"; }
