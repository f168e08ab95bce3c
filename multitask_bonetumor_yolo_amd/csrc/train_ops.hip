// HBM-bound kernels of the TRAINING step (configs[2]-[3]: forward in train mode, backward, parameter gradients), NHWC, 16-byte
// accesses, fp32 arithmetic, deterministic (fixed-order two-level reductions, no atomics):
//   mtbt_bn_backward_nhwc        activation derivative + BatchNorm backward (batch or running statistics) in two passes over (dy, x)
//   mtbt_weight_prep             ONE launch per step that re-lays out every master weight (fp32, any strides) into the kernels' packed
//                                compute-dtype layouts (forward KRSC, dgrad CRSK flipped, folded per-row / per-column scales)
//   mtbt_bifpn_norm_weights(_backward)   the BiFPN fusion weights ELU(w) / (sum_0 ELU(w) + eps) on the device (main_model.py:194-196)
//   mtbt_gap_fc_backward         AdaptiveAvgPool2d(1) + Linear backward (main_model.py:333-334, :364)
//   mtbt_copy_strided            channel-slice copy with dtype conversion and zero padding (fp32 loss gradients -> dense bf16 operands)
//   mtbt_scale_grad              parameter gradients of a weight that was folded with a per-row (ConvNeXt layer scale gamma) or per-column
//                                (DepthwiseConvBlock's k=1 depthwise scale) vector, from the raw GEMM weight gradient
//   mtbt_add_nhwc                dst += src (gradient accumulation where a producer kernel cannot accumulate itself)
#include "common.h"
#include "rowreduce.h"

namespace {

inline unsigned grid_cap(long work, int block, long cap = 8192) {
  long g = (work + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------------------------------------
// BatchNorm backward.  Forward (bn_train.hip): xhat = (x - mean) * rstd, u = xhat * gamma + beta, y = act(u).
//   du = dy * act'(u);  s1 = sum_p du;  s2 = sum_p du * xhat;  d beta = s1;  d gamma = s2
//   batch statistics:    dx = gamma * rstd * (du - s1 / M - xhat * s2 / M)
//   running statistics:  dx = gamma * rstd * du
// Pass 1 forms per-workgroup partial (s1, s2) rows; pass 2 sums them (one wave per column); pass 3 recomputes du and writes dx.
// dy may be a channel slice (row pitch dy_ld); x (the conv output the forward normalised) and dx are dense.
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void bn_du(const T* dy, long dy_off, const T* x, long x_off, const float* mean, const float* var, const float* gamma,
                                      const float* beta, float eps, int act, int c0, float (&du)[8], float (&xh)[8], float (&gr)[8]) {
  float a[8], b[8], mu[8], va[8], ga[8], be[8];
  ld8<T>(dy + dy_off, a);
  ld8<T>(x + x_off, b);
  ld8<float>(mean + c0, mu);
  ld8<float>(var + c0, va);
  ld8<float>(gamma + c0, ga);
  ld8<float>(beta + c0, be);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float rstd = rsqrtf(va[e] + eps);
    xh[e] = (b[e] - mu[e]) * rstd;
    du[e] = a[e] * act_grad(xh[e] * ga[e] + be[e], act);
    gr[e] = ga[e] * rstd;
  }
}

// The per-channel constants of a lane's 8-channel piece: in every kernel below the piece a thread works on is the same for all its
// rows (the workgroup size is a multiple of the pieces per row), so mean / rstd / gamma / beta are read ONCE per thread and the row loop
// holds only the two 16-byte loads and the arithmetic -- unrolled four rows deep so eight loads are in flight per lane.  (Round 3: with
// the four parameter vectors re-read and one row in flight per trip the statistics pass ran at 2.0 TB/s, SLOWER than the apply pass that
// also writes a tensor: 39 against 25 us on average over the 98 BatchNorms of the step.)
struct BnChan { float mu[8], rs[8], ga[8], be[8]; };
__device__ __forceinline__ BnChan bn_chan(const float* mean, const float* var, const float* gamma, const float* beta, float eps, int c0) {
  BnChan k;
  float va[8];
  ld8<float>(mean + c0, k.mu);
  ld8<float>(var + c0, va);
  ld8<float>(gamma + c0, k.ga);
  ld8<float>(beta + c0, k.be);
#pragma unroll
  for (int e = 0; e < 8; ++e) k.rs[e] = rsqrtf(va[e] + eps);
  return k;
}
// du = dy * act'(xhat * gamma + beta), xhat = (x - mean) * rstd  (the arithmetic of bn_du above with the constants hoisted)
template <int ACT>
__device__ __forceinline__ void bn_du_k(const float (&a)[8], const float (&b)[8], const BnChan& k, int act, float (&du)[8], float (&xh)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    xh[e] = (b[e] - k.mu[e]) * k.rs[e];
    du[e] = a[e] * act_grad(xh[e] * k.ga[e] + k.be[e], ACT < 0 ? act : ACT);
  }
}

// ACT: the activation as a compile-time constant (MTBT_ACT_NONE / SILU / ELU), or -1 = the run-time `act`
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_partial(const T* __restrict__ dy, int dy_ld, const T* __restrict__ x, long P, int C,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                                      float* __restrict__ partial /* [blocks][2C] */) {
  __shared__ float red[256 * 16];
  const int tid = threadIdx.x, chunks = C >> 3;
  const long p0 = (long)blockIdx.x * ROWS_PER_BLOCK, p1 = min(P, p0 + ROWS_PER_BLOCK);
  float* dst = partial + (long)blockIdx.x * 2 * C;
  if (chunks >= 256) {   // (C >= 2048: not used by this network; serial over the rows)
    for (int ch = tid; ch < chunks; ch += 256) {
      float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (long p = p0; p < p1; ++p) {
        float du[8], xh[8], gr[8];
        bn_du<T>(dy, p * dy_ld + ch * 8, x, p * C + ch * 8, mean, var, gamma, beta, eps, act, ch * 8, du, xh, gr);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] += du[k]; s2[k] += du[k] * xh[k]; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { dst[ch * 8 + k] = s1[k]; dst[C + ch * 8 + k] = s2[k]; }
    }
    return;
  }
  const int rpp = 256 / chunks, rg = tid / chunks, ch = tid - rg * chunks;
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rg < rpp) {
    const BnChan k = bn_chan(mean, var, gamma, beta, eps, ch * 8);
    const T* dyp = dy + ch * 8;
    const T* xp = x + ch * 8;
    long p = p0 + rg;
    for (; p + 3 * rpp < p1; p += 4 * rpp) {      // (rows are added in the same order as one at a time)
      float a[4][8], b[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) { ld8<T>(dyp + (p + u * rpp) * dy_ld, a[u]); ld8<T>(xp + (p + u * rpp) * C, b[u]); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float du[8], xh[8];
        bn_du_k<ACT>(a[u], b[u], k, act, du, xh);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += du[e]; s2[e] += du[e] * xh[e]; }
      }
    }
    for (; p < p1; p += rpp) {
      float a[8], b[8], du[8], xh[8];
      ld8<T>(dyp + p * dy_ld, a); ld8<T>(xp + p * C, b);
      bn_du_k<ACT>(a, b, k, act, du, xh);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += du[e]; s2[e] += du[e] * xh[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[(rg * chunks + ch) * 16 + e] = s1[e]; red[(rg * chunks + ch) * 16 + 8 + e] = s2[e]; }
  }
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t1 = 0.f, t2 = 0.f;
      for (int g = 0; g < rpp; ++g) { t1 += red[(g * chunks + ch) * 16 + k]; t2 += red[(g * chunks + ch) * 16 + 8 + k]; }
      dst[ch * 8 + k] = t1;
      dst[C + ch * 8 + k] = t2;
    }
  }
}

// one wave per column of the [blocks][2C] partials: sums[0..C) = s1 = d beta, sums[C..2C) = s2 = d gamma
__global__ __launch_bounds__(256) void bn_bwd_final(const float* __restrict__ partial, int blocks, int pitch, int C, float* __restrict__ sums,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= 2 * C) return;
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += partial[(long)b * pitch + c];
  s = wave_sum(s);
  if (lane == 0) {
    sums[c] = s;
    float* out = c < C ? (dbeta ? dbeta + c : nullptr) : (dgamma ? dgamma + (c - C) : nullptr);
    if (out) *out = accumulate ? *out + s : s;
  }
}

// FIXED: 256 % (C / 8) == 0 -- a thread's 8-channel piece is the same in every trip of the grid-stride loop (constants hoisted, four
// rows in flight); otherwise the general loop
template <typename T, int ACT, bool FIXED>
__global__ __launch_bounds__(256) void bn_bwd_apply(const T* __restrict__ dy, int dy_ld, const T* __restrict__ x, T* __restrict__ dx, long P, int C,
                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                                    const float* __restrict__ sums, int use_running) {
  const int chunks = C >> 3;
  const long n8 = P * chunks;
  const float invM = 1.0f / (float)P;
  if constexpr (FIXED) {
    const int ch = threadIdx.x % chunks;
    const int sh = 31 - __builtin_clz(chunks);           // chunks is a power of two here: pixel = piece >> sh (a 64-bit division per piece otherwise)
    const BnChan k = bn_chan(mean, var, gamma, beta, eps, ch * 8);
    float m1[8], m2[8], gr[8];
    if (use_running) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { m1[e] = 0.f; m2[e] = 0.f; }
    } else {
      ld8<float>(sums + ch * 8, m1);
      ld8<float>(sums + C + ch * 8, m2);
#pragma unroll
      for (int e = 0; e < 8; ++e) { m1[e] = m1[e] * invM; m2[e] = m2[e] * invM; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) gr[e] = k.ga[e] * k.rs[e];
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    auto one = [&](const float (&a)[8], const float (&b)[8], long idx) {
      float du[8], xh[8], o[8];
      bn_du_k<ACT>(a, b, k, act, du, xh);
      if (use_running) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gr[e] * du[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gr[e] * (du[e] - m1[e] - xh[e] * m2[e]);
      }
      st8<T>(dx + idx * 8, o);
    };
    for (; i + 3 * stride < n8; i += 4 * stride) {
      float a[4][8], b[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long idx = i + u * stride;
        ld8<T>(dy + (idx >> sh) * dy_ld + ch * 8, a[u]);
        ld8<T>(x + idx * 8, b[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(a[u], b[u], i + u * stride);
    }
    for (; i < n8; i += stride) {
      float a[8], b[8];
      ld8<T>(dy + (i >> sh) * dy_ld + ch * 8, a);
      ld8<T>(x + i * 8, b);
      one(a, b, i);
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    const long p = i / chunks;
    float du[8], xh[8], gr[8], o[8];
    bn_du<T>(dy, p * dy_ld + ch * 8, x, i * 8, mean, var, gamma, beta, eps, act, ch * 8, du, xh, gr);
    if (use_running) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = gr[k] * du[k];
    } else {
      float s1[8], s2[8];
      ld8<float>(sums + ch * 8, s1);
      ld8<float>(sums + C + ch * 8, s2);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = gr[k] * (du[k] - s1[k] * invM - xh[k] * (s2[k] * invM));
    }
    st8<T>(dx + i * 8, o);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight preparation: descriptor j turns a master tensor (fp32, arbitrary strides) into a dense row-major [d0][d1][d2][d3] tensor in
// the compute dtype:  dst[a][b][c][d] = src[ia*s0 + ib*s1 + ic*s2 + id*s3] * scale0[idx(dim0)] * scale1[idx(dim1)],
// where ix = flip ? extent-1-x : x.  Workgroup b serves descriptor j with block_start[j] <= b < block_start[j+1] (binary search).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int PREP_ELEMS = 2048;  // per workgroup

__global__ __launch_bounds__(256) void weight_prep_kernel(const mtbt_prep_desc* __restrict__ table, const int32_t* __restrict__ block_start, int n_desc) {
  int lo = 0, hi = n_desc - 1;
  const int b = blockIdx.x;
  while (lo < hi) {   // largest j with block_start[j] <= b
    const int mid = (lo + hi + 1) >> 1;
    if (block_start[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const mtbt_prep_desc d = table[lo];
  const long total = (long)d.dim[0] * d.dim[1] * d.dim[2] * d.dim[3];
  const long e0 = (long)(b - block_start[lo]) * PREP_ELEMS;
  for (int t = threadIdx.x; t < PREP_ELEMS; t += 256) {
    const long e = e0 + t;
    if (e >= total) break;
    long r = e;
    int idx[4];
    idx[3] = (int)(r % d.dim[3]); r /= d.dim[3];
    idx[2] = (int)(r % d.dim[2]); r /= d.dim[2];
    idx[1] = (int)(r % d.dim[1]); r /= d.dim[1];
    idx[0] = (int)r;
    long off = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) off += (long)(d.flip[q] ? d.dim[q] - 1 - idx[q] : idx[q]) * d.sstride[q];
    float v = (d.src_dim3 > 0 && idx[3] >= d.src_dim3) ? 0.f : d.src[off];   // zero padding of the last dimension
    if (d.scale0) v *= d.scale0[idx[d.scale0_dim]];
    if (d.scale1) v *= d.scale1[idx[d.scale1_dim]];
    if (d.dst_dtype == MTBT_F32) reinterpret_cast<float*>(d.dst)[e] = v;
    else reinterpret_cast<bf16_t*>(d.dst)[e] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// BiFPN fusion weights (main_model.py:194-196): e = ELU(w), out[j][i] = e[i][j] / (sum_i e[i][j] + eps), w [n][2]; out (and dout of the
// backward) is TRANSPOSED, [2][n]: the n weights of fusion node j are contiguous, as mtbt_bifpn_fuse's wgt_dev wants them.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

__global__ void bifpn_norm_kernel(const float* __restrict__ w, int n, float eps, float* __restrict__ out) {
  const int j = threadIdx.x;
  if (j >= 2) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += elu1(w[i * 2 + j]);
  for (int i = 0; i < n; ++i) out[j * n + i] = elu1(w[i * 2 + j]) / (s + eps);   // transposed: a node's n weights are contiguous
}

__global__ void bifpn_norm_bwd_kernel(const float* __restrict__ w, int n, float eps, const float* __restrict__ dout, float* __restrict__ dw,
                                      int accumulate) {
  const int j = threadIdx.x;
  if (j >= 2) return;
  float s = 0.f, dot = 0.f;
  for (int i = 0; i < n; ++i) { const float e = elu1(w[i * 2 + j]); s += e; dot += dout[j * n + i] * e; }
  const float inv = 1.f / (s + eps);
  for (int i = 0; i < n; ++i) {
    const float p = w[i * 2 + j];
    const float de = dout[j * n + i] * inv - dot * inv * inv;
    const float g = de * (p > 0.f ? 1.f : expf(p));
    dw[i * 2 + j] = accumulate ? dw[i * 2 + j] + g : g;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// GAP + Linear backward.  pool[n][c] = mean_p x[n][p][c];  logits = pool W^T + b.
//   kernel A (block per image): pool[n][:] (recomputed) into the workspace, dpool[c] = sum_j dl[n][j] W[j][c],
//                               dx[n][p][c] (+)= dpool[c] / HW
//   kernel B (thread per (j, c)): dW[j][c] (+)= sum_n dl[n][j] pool[n][c];  db[j] (+)= sum_n dl[n][j]
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void gap_fc_bwd_dx(const T* __restrict__ x, const float* __restrict__ dl, const float* __restrict__ w, T* __restrict__ dx,
                              float* __restrict__ pool, int HW, int C, int nout, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = reinterpret_cast<float*>(smem);  // [G][C]
  const int CH8 = C >> 3, G = blockDim.x / CH8;
  float* dpool = part + G * C;                   // [C]
  const int tid = threadIdx.x, n = blockIdx.x;
  const int chunk = tid % CH8, g = tid / CH8;
  if (g < G) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
    for (int p = g; p < HW; p += G) {
      ld8<T>(x + ((long)n * HW + p) * C + chunk * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[g * C + chunk * 8 + e] = s[e];
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < G; ++k) s += part[k * C + c];
    pool[(long)n * C + c] = s / HW;
    float d = 0.f;
    for (int j = 0; j < nout; ++j) d += dl[n * nout + j] * w[j * C + c];
    dpool[c] = d / HW;
  }
  __syncthreads();
  if (g < G) {
    float d[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = dpool[chunk * 8 + e];
    for (int p = g; p < HW; p += G) {
      T* dst = dx + ((long)n * HW + p) * C + chunk * 8;
      if (accumulate) {
        ld8<T>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += d[e];
        st8<T>(dst, o);
      } else {
        st8<T>(dst, d);
      }
    }
  }
}

__global__ void gap_fc_bwd_w(const float* __restrict__ dl, const float* __restrict__ pool, float* __restrict__ dw, float* __restrict__ db, int N, int C,
                             int nout, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nout * C) {
    const int j = i / C, c = i - j * C;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dl[n * nout + j] * pool[(long)n * C + c];
    dw[i] = accumulate ? dw[i] + s : s;
  }
  if (db && i < nout) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dl[n * nout + i];
    db[i] = accumulate ? db[i] + s : s;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Strided channel-slice copy with conversion: dst[n][p][c] = c < C ? src[n][p][c] : 0 for c < Cpad  (element-wise: the fp32 Detect
// gradient maps are 66 floats wide, their 64 / nc channel slices start at arbitrary 4-byte offsets).
// ------------------------------------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void copy_strided_kernel(const S* __restrict__ src, long sbs, int sld, D* __restrict__ dst, long dbs, int dld, int N, long P, int C, int Cpad) {
  const long total = (long)N * P * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c, pi;
    long np, n;
    divmod_u32(i, Cpad, np, c);
    divmod_u32(np, (int)P, n, pi);
    const long p = pi;
    const float v = c < C ? ld_elem<S>(src + n * sbs + p * sld + c) : 0.f;
    st_elem<D>(dst + n * dbs + p * dld + c, v);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// mode 0 (row scale: ConvNeXt fc2 folded with layer-scale gamma, y = x + gamma * (W h + b)):  given G[k][c] = sum_p dy[p][k] h[p][c]
//        and s[k] = sum_p dy[p][k]:   dW = gamma[k] G;  d gamma[k] = sum_c W[k][c] G[k][c] + b[k] s[k];  d b[k] = gamma[k] s[k]
// mode 1 (column scale: DepthwiseConvBlock, y = W (v * x)):  given G[k][c] = sum_p dy[p][k] x[p][c]:
//        dW = G[k][c] v[c];  d v[c] = sum_k G[k][c] W[k][c]
// W, G, dW: dense [K][C] fp32.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_grad_rows(const float* __restrict__ G, const float* __restrict__ W, const float* __restrict__ vec,
                                                       const float* __restrict__ bias, const float* __restrict__ s, float* __restrict__ dW,
                                                       float* __restrict__ dvec, float* __restrict__ dbias, int K, int C, int accumulate) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= K) return;
  const float g = vec[k];
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float gv = G[(long)k * C + c];
    dot += W[(long)k * C + c] * gv;
    dW[(long)k * C + c] = accumulate ? dW[(long)k * C + c] + g * gv : g * gv;
  }
  dot = wave_sum(dot);
  if (lane == 0) {
    const float sk = s ? s[k] : 0.f;
    const float dg = dot + (bias ? bias[k] * sk : 0.f);
    dvec[k] = accumulate ? dvec[k] + dg : dg;
    if (dbias) dbias[k] = accumulate ? dbias[k] + g * sk : g * sk;
  }
}

// workgroup = 64 columns x 4 row groups (lane = column: 256-byte row segments); the row groups are added in a fixed order through LDS
__global__ __launch_bounds__(256) void scale_grad_cols(const float* __restrict__ G, const float* __restrict__ W, const float* __restrict__ vec,
                                                       float* __restrict__ dW, float* __restrict__ dvec, int K, int C, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float dot = 0.f;
  if (c < C) {
    const float v = vec[c];
#pragma unroll 4
    for (int k = rg; k < K; k += 4) {
      const float gv = G[(long)k * C + c];
      dot += gv * W[(long)k * C + c];
      dW[(long)k * C + c] = accumulate ? dW[(long)k * C + c] + gv * v : gv * v;
    }
  }
  red[rg][lane] = dot;
  __syncthreads();
  if (rg == 0 && c < C) {
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    dvec[c] = accumulate ? dvec[c] + t : t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(T* __restrict__ dst, long dbs, int dld, const T* __restrict__ src, long sbs, int sld, int N, long P, int C) {
  const int chunks = C >> 3;
  const long total = (long)N * P * chunks;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int ch, pi;
    long np, n;
    divmod_u32(i, chunks, np, ch);
    divmod_u32(np, (int)P, n, pi);
    const long p = pi;
    float a[8], b[8];
    ld8<T>(dst + n * dbs + p * dld + ch * 8, a);
    ld8<T>(src + n * sbs + p * sld + ch * 8, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    st8<T>(dst + n * dbs + p * dld + ch * 8, a);
  }
}

}  // namespace

extern "C" int64_t mtbt_bn_backward_workspace_bytes(int64_t pixels, int C) {
  if (pixels <= 0 || C <= 0) return 0;
  return (((pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) * 2 * (int64_t)C + 2 * (int64_t)C) * (int64_t)sizeof(float);
}

extern "C" int mtbt_bn_backward_nhwc(const void* dy, int32_t dy_pixel_stride, const void* x, const float* stats, const float* gamma, const float* beta,
                                     float eps, int act, int use_running, void* dx, float* dgamma, float* dbeta, int accumulate, int64_t pixels, int C,
                                     int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!dy || !x || !stats || !gamma || !beta || !dx || !workspace || pixels <= 0 || C <= 0 || C % 8 || C > 2048) return MTBT_EINVAL;
  if (dy_pixel_stride < C || dy_pixel_stride % 8 || act < MTBT_ACT_NONE || act > MTBT_ACT_GELU_POLY) return MTBT_EINVAL;
  if (dtype != MTBT_F32 && dtype != MTBT_BF16) return MTBT_EINVAL;
  if (!aligned16(dy) || !aligned16(x) || !aligned16(dx) || !aligned16(stats) || !aligned16(gamma) || !aligned16(beta) || !aligned16(workspace)) return MTBT_EALIGN;
  if (workspace_bytes < mtbt_bn_backward_workspace_bytes(pixels, C)) return MTBT_EWORKSPACE;
  const long blocks = (pixels + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (blocks > 0x7fffffffL) return MTBT_EINVAL;
  float* partial = reinterpret_cast<float*>(workspace);
  float* sums = partial + blocks * 2 * C;
  const float* mean = stats;
  const float* var = stats + C;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // the apply pass: four 8-channel pieces per thread and trip, at most 2048 workgroups (8 per CU)
  const bool fixed = 256 % (C / 8) == 0;
  const unsigned ga = grid_cap(pixels * (C / 8), fixed ? 1024 : 256, 2048);
#define BNB2(T, A, F)                                                                                                                      \
  hipLaunchKernelGGL((bn_bwd_partial<T, A>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)dy, dy_pixel_stride, (const T*)x, (long)pixels, C, mean, var, \
                     gamma, beta, eps, act, partial);                                                                                      \
  { long rr = blocks; int pp = 2 * C; colsum_prereduce(partial, rr, pp, 0, 2 * C, s);                                                      \
    hipLaunchKernelGGL(bn_bwd_final, dim3((unsigned)((2 * C + 3) / 4)), dim3(256), 0, s, partial, (int)rr, pp, C, sums, dgamma, dbeta, accumulate); }  \
  hipLaunchKernelGGL((bn_bwd_apply<T, A, F>), dim3(ga), dim3(256), 0, s, (const T*)dy, dy_pixel_stride, (const T*)x, (T*)dx, (long)pixels, C, mean, var, gamma, \
                     beta, eps, act, sums, use_running);
#define BNB(T)                                                                        \
  if (!fixed) { BNB2(T, -1, false) }                                                  \
  else if (act == MTBT_ACT_SILU) { BNB2(T, MTBT_ACT_SILU, true) }                     \
  else if (act == MTBT_ACT_ELU) { BNB2(T, MTBT_ACT_ELU, true) }                       \
  else if (act == MTBT_ACT_NONE) { BNB2(T, MTBT_ACT_NONE, true) }                     \
  else { BNB2(T, -1, true) }
  if (dtype == MTBT_F32) { BNB(float) } else { BNB(bf16_t) }
#undef BNB
#undef BNB2
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_weight_prep_blocks(int64_t elements) { return elements <= 0 ? 0 : (int)((elements + PREP_ELEMS - 1) / PREP_ELEMS); }

extern "C" int mtbt_weight_prep(const mtbt_prep_desc* table_dev, const int32_t* block_start_dev, int n_desc, int total_blocks, void* stream) {
  if (!table_dev || !block_start_dev || n_desc <= 0 || total_blocks <= 0) return MTBT_EINVAL;
  hipLaunchKernelGGL(weight_prep_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table_dev, block_start_dev, n_desc);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_bifpn_norm_weights(const float* w, int n, float eps, float* out, void* stream) {
  if (!w || !out || n < 1 || n > 8) return MTBT_EINVAL;
  hipLaunchKernelGGL(bifpn_norm_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), w, n, eps, out);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_bifpn_norm_weights_backward(const float* w, int n, float eps, const float* dout, float* dw, int accumulate, void* stream) {
  if (!w || !dout || !dw || n < 1 || n > 8) return MTBT_EINVAL;
  hipLaunchKernelGGL(bifpn_norm_bwd_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), w, n, eps, dout, dw, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_gap_fc_backward(const void* x, const float* dlogits, const float* w, void* dx, int accumulate_dx, float* dw, float* db,
                                    int accumulate_dw, float* pool_ws /* [N][C] */, int N, int HW, int C, int nout, int dtype, void* stream) {
  if (!x || !dlogits || !w || !dx || !dw || !pool_ws || N <= 0 || HW <= 0 || C <= 0 || C % 8 || C > 2048 || nout <= 0) return MTBT_EINVAL;
  if (!aligned16(x) || !aligned16(dx)) return MTBT_EALIGN;
  const int CH8 = C / 8;
  int G = 256 / CH8;
  if (G < 1) G = 1;
  int threads = ((G * CH8 + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  G = threads / CH8;
  const size_t lds = (size_t)(G + 1) * C * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MTBT_F32)
    hipLaunchKernelGGL((gap_fc_bwd_dx<float>), dim3(N), dim3(threads), lds, s, (const float*)x, dlogits, w, (float*)dx, pool_ws, HW, C, nout, accumulate_dx);
  else if (dtype == MTBT_BF16)
    hipLaunchKernelGGL((gap_fc_bwd_dx<bf16_t>), dim3(N), dim3(threads), lds, s, (const bf16_t*)x, dlogits, w, (bf16_t*)dx, pool_ws, HW, C, nout, accumulate_dx);
  else return MTBT_EINVAL;
  hipLaunchKernelGGL(gap_fc_bwd_w, dim3((unsigned)((nout * C + 255) / 256)), dim3(256), 0, s, dlogits, pool_ws, dw, db, N, C, nout, accumulate_dw);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_copy_strided(const void* src, int src_dtype, int64_t src_batch_stride, int32_t src_pixel_stride, void* dst, int dst_dtype,
                                 int64_t dst_batch_stride, int32_t dst_pixel_stride, int N, int64_t pixels, int C, int C_pad, void* stream) {
  if (!src || !dst || N <= 0 || pixels <= 0 || C <= 0 || C_pad < C || src_pixel_stride < C || dst_pixel_stride < C_pad) return MTBT_EINVAL;
  if (pixels > 0x7fffffffL || (long)N * pixels * C_pad > 0xffffffffL) return MTBT_EINVAL;   // (32-bit element indices in the kernel)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned g = grid_cap((long)N * pixels * C_pad, 256);
#define CS(S, D) hipLaunchKernelGGL((copy_strided_kernel<S, D>), dim3(g), dim3(256), 0, s, (const S*)src, (long)src_batch_stride, src_pixel_stride, (D*)dst, \
                                    (long)dst_batch_stride, dst_pixel_stride, N, (long)pixels, C, C_pad)
  if (src_dtype == MTBT_F32 && dst_dtype == MTBT_F32) CS(float, float);
  else if (src_dtype == MTBT_F32 && dst_dtype == MTBT_BF16) CS(float, bf16_t);
  else if (src_dtype == MTBT_BF16 && dst_dtype == MTBT_F32) CS(bf16_t, float);
  else if (src_dtype == MTBT_BF16 && dst_dtype == MTBT_BF16) CS(bf16_t, bf16_t);
  else return MTBT_EINVAL;
#undef CS
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_scale_grad(int mode, const float* G, const float* W, const float* vec, const float* bias, const float* s, float* dW, float* dvec,
                               float* dbias, int K, int C, int accumulate, void* stream) {
  if (!G || !W || !vec || !dW || !dvec || K <= 0 || C <= 0 || (mode != 0 && mode != 1)) return MTBT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (mode == 0) {
    if ((bias || dbias) && !s) return MTBT_EINVAL;
    hipLaunchKernelGGL(scale_grad_rows, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, st, G, W, vec, bias, s, dW, dvec, dbias, K, C, accumulate);
  } else {
    hipLaunchKernelGGL(scale_grad_cols, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, G, W, vec, dW, dvec, K, C, accumulate);
  }
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_add_nhwc(void* dst, int64_t dst_batch_stride, int32_t dst_pixel_stride, const void* src, int64_t src_batch_stride,
                             int32_t src_pixel_stride, int N, int64_t pixels, int C, int dtype, void* stream) {
  if (!dst || !src || N <= 0 || pixels <= 0 || C <= 0 || C % 8 || dst_pixel_stride % 8 || src_pixel_stride % 8 || dst_batch_stride % 8 || src_batch_stride % 8)
    return MTBT_EINVAL;
  if (!aligned16(dst) || !aligned16(src)) return MTBT_EALIGN;
  if (pixels > 0x7fffffffL || (long)N * pixels * (C / 8) > 0xffffffffL) return MTBT_EINVAL;   // (32-bit piece indices in the kernel)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned g = grid_cap((long)N * pixels * (C / 8), 256);
  if (dtype == MTBT_F32)
    hipLaunchKernelGGL(add_kernel<float>, dim3(g), dim3(256), 0, s, (float*)dst, (long)dst_batch_stride, dst_pixel_stride, (const float*)src, (long)src_batch_stride, src_pixel_stride, N, (long)pixels, C);
  else if (dtype == MTBT_BF16)
    hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (bf16_t*)dst, (long)dst_batch_stride, dst_pixel_stride, (const bf16_t*)src, (long)src_batch_stride, src_pixel_stride, N, (long)pixels, C);
  else return MTBT_EINVAL;
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
