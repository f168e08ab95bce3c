// Backward of the resampling operators of the path (HBM-bound, deterministic):
//   mtbt_bifpn_fuse_backward   one input of a BiFPN fusion node y = sum_i w_i * resample_i(x_i) (main_model.py:211-240):
//                              dx_i (+)= w_i * resample_i^T(dy)   and   dw_i = <dy, resample_i(x_i)>
//   mtbt_projector_backward    the trainer's proto projector + bilinear resize (running_main_v3.py:251-255) behind the seg BCE:
//                              d_low = bilinear^T(d seg logits), d protos = w (x) d_low, d w = <d_low, protos>, d b = sum d_low
//   mtbt_resample_backward     one input of the oldest variant's WeightedAdd node y = sum_i (w_i + resample_i(x_i)) (src/model.py:27-37, 60-74):
//                              dx_i (+)= resample_i^T(dy) for identity, nearest x2 (F.interpolate) and 2x2 max pooling (F.max_pool2d)
//   mtbt_wadd_norm_weights(_backward)  that node's weights relu(w) / (sum relu(w) + eps) on the device, and d w from sum(dy)
// The transposes are written in GATHER form: an input pixel visits the output pixels whose forward footprint can contain it and
// re-evaluates the forward's own index / weight arithmetic, so forward and backward agree by construction (borders included).
#include "common.h"
#include "rowreduce.h"

namespace {

inline unsigned grid_cap(long work, int block, long cap = 8192) {
  long g = (work + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// forward coordinates of torch's bilinear resize, align_corners=False: src = max(0, (dst + .5) * scale - .5), i0 = floor, i1 = min(i0 + 1, last)
__device__ __forceinline__ void bil_coord(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float s = (dst + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - i0;
  l0 = 1.f - l1;
}
// weight with which input index `i` enters output index `o`
__device__ __forceinline__ float bil_weight(int o, int i, float scale, int in_size) {
  int i0, i1; float l0, l1;
  bil_coord(o, scale, in_size, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// dx kernel: thread = (input pixel, 8-channel chunk).  mode 0 identity, 1 bilinear x2 up (input Hs = H/2), 2 mean 2x2 down (input 2H x 2W).
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_dx(const T* __restrict__ dy, T* __restrict__ dx, const float* __restrict__ wgt, int mode, int N, int H, int W,
                                                   int C, int accumulate) {
  const int CH8 = C >> 3;
  const int Hi = mode == 1 ? H >> 1 : (mode == 2 ? H << 1 : H), Wi = mode == 1 ? W >> 1 : (mode == 2 ? W << 1 : W);
  const long total = (long)N * Hi * Wi * CH8;
  const float wv = *wgt;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int chunk, x, y;
    long pix, ny, nn;
    divmod_u32(idx, CH8, pix, chunk);
    divmod_u32(pix, Wi, ny, x);
    divmod_u32(ny, Hi, nn, y);
    const int n = (int)nn;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[8];
    const T* dyn = dy + (long)n * H * W * C + chunk * 8;
    if (mode == 0) {
      ld8<T>(dyn + ((long)y * W + x) * C, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = t[e];
    } else if (mode == 2) {
      ld8<T>(dyn + ((long)(y >> 1) * W + (x >> 1)) * C, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.25f * t[e];
    } else {
      for (int oy = 2 * y - 1; oy <= 2 * y + 2; ++oy) {
        if ((unsigned)oy >= (unsigned)H) continue;
        const float wy = bil_weight(oy, y, 0.5f, Hi);
        if (wy == 0.f) continue;
        for (int ox = 2 * x - 1; ox <= 2 * x + 2; ++ox) {
          if ((unsigned)ox >= (unsigned)W) continue;
          const float wx = bil_weight(ox, x, 0.5f, Wi);
          if (wx == 0.f) continue;
          ld8<T>(dyn + ((long)oy * W + ox) * C, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += (wy * wx) * t[e];
        }
      }
    }
    T* dst = dx + pix * C + chunk * 8;
    if (accumulate) {
      ld8<T>(dst, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = t[e] + wv * acc[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= wv;
    }
    st8<T>(dst, acc);
  }
}

// the forward fetch of pointwise.hip's fuse kernel (modes 0..2), for the weight gradient <dy, resample(x)>
template <typename T>
__device__ __forceinline__ void fetch_resampled(const T* src, int mode, int n, int y, int x, int H, int W, int C, int c0, float (&o)[8]) {
  if (mode == 0) {
    ld8<T>(src + (((long)n * H + y) * W + x) * C + c0, o);
  } else if (mode == 1) {
    const int Hs = H >> 1, Ws = W >> 1;
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bil_coord(y, 0.5f, Hs, y0, y1, ly0, ly1);
    bil_coord(x, 0.5f, Ws, x0, x1, lx0, lx1);
    float a[8], b[8], c[8], d[8];
    const T* base = src + (long)n * Hs * Ws * C + c0;
    ld8<T>(base + ((long)y0 * Ws + x0) * C, a);
    ld8<T>(base + ((long)y0 * Ws + x1) * C, b);
    ld8<T>(base + ((long)y1 * Ws + x0) * C, c);
    ld8<T>(base + ((long)y1 * Ws + x1) * C, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ly0 * (lx0 * a[e] + lx1 * b[e]) + ly1 * (lx0 * c[e] + lx1 * d[e]);
  } else {
    const int Ws = W << 1;
    float a[8], b[8], c[8], d[8];
    const T* base = src + (long)n * (H << 1) * Ws * C + c0;
    ld8<T>(base + ((long)(2 * y) * Ws + 2 * x) * C, a);
    ld8<T>(base + ((long)(2 * y) * Ws + 2 * x + 1) * C, b);
    ld8<T>(base + ((long)(2 * y + 1) * Ws + 2 * x) * C, c);
    ld8<T>(base + ((long)(2 * y + 1) * Ws + 2 * x + 1) * C, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.25f * ((a[e] + b[e]) + (c[e] + d[e]));
  }
}

// partial[block] = sum over this block's (output pixel, chunk) items of dy . resample(x): fixed item -> thread mapping, LDS tree
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_dot(const T* __restrict__ dy, const T* __restrict__ xin, int mode, int N, int H, int W, int C,
                                                    float* __restrict__ partial) {
  __shared__ float red[256];
  const int CH8 = C >> 3;
  const long total = (long)N * H * W * CH8;
  float s = 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int chunk, x, y;
    long pix, ny, nn;
    divmod_u32(idx, CH8, pix, chunk);
    divmod_u32(pix, W, ny, x);
    divmod_u32(ny, H, nn, y);
    const int n = (int)nn;
    float t[8], g[8];
    fetch_resampled<T>(xin, mode, n, y, x, H, W, C, chunk * 8, t);
    ld8<T>(dy + pix * C + chunk * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += t[e] * g[e];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void scalar_final(const float* __restrict__ partial, int n, float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = accumulate ? *out + red[0] : red[0];
}

// ---- projector backward ----
// d_low[n][y][x] = sum over output pixels (Y, X) of wY(Y, y) * wX(X, x) * dseg[n][Y][X]
__global__ __launch_bounds__(256) void proj_bwd_low(const float* __restrict__ dseg, float* __restrict__ dlow, int N, int hp, int wp, int Ho, int Wo) {
  const long total = (long)N * hp * wp;
  const float sy = (float)hp / (float)Ho, sx = (float)wp / (float)Wo;   // torch: scale = in / out (size given, no scale_factor)
  const int ry = (Ho + hp - 1) / hp, rx = (Wo + wp - 1) / wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % wp);
    const long ny = i / wp;
    const int y = (int)(ny % hp), n = (int)(ny / hp);
    const int Y0 = max(0, (y - 1) * ry - 1), Y1 = min(Ho - 1, (y + 2) * ry + 1);
    const int X0 = max(0, (x - 1) * rx - 1), X1 = min(Wo - 1, (x + 2) * rx + 1);
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
      const float wy = bil_weight(Y, y, sy, hp);
      if (wy == 0.f) continue;
      const float* row = dseg + ((long)n * Ho + Y) * Wo;
      float r = 0.f;
      for (int X = X0; X <= X1; ++X) {
        const float wx = bil_weight(X, x, sx, wp);
        if (wx != 0.f) r += wx * row[X];
      }
      acc += wy * r;
    }
    dlow[i] = acc;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void proj_bwd_protos(const float* __restrict__ dlow, const float* __restrict__ w, T* __restrict__ dprotos, long P, int nm,
                                                       int accumulate) {
  const long total = P * nm;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % nm);
    const float v = dlow[i / nm] * w[c];
    st_elem<T>(dprotos + i, accumulate ? ld_elem<T>(dprotos + i) + v : v);
  }
}

// partial[block][c] = sum_p dlow[p] * protos[p][c]  (c < nm),  partial[block][nm .. nm+8) = sum_p dlow[p] (bias; 8 copies)
__global__ __launch_bounds__(256) void proj_bwd_w_partial(const float* __restrict__ dlow, const float* __restrict__ protos, long P, int nm, float* __restrict__ partial) {
  const long p0 = (long)blockIdx.x * ROWS_PER_BLOCK, p1 = min(P, p0 + ROWS_PER_BLOCK);
  const int chunks = (nm >> 3) + 1;
  rows_reduce(p0, p1, chunks, partial + (long)blockIdx.x * chunks * 8, [&](long p, int ch, float (&v)[8]) {
    const float d = dlow[p];
    if (ch * 8 < nm) {
      ld8<float>(protos + p * nm + ch * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= d;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = d;
    }
    return true;
  });
}

__global__ void proj_bwd_w_out(const float* __restrict__ sums, int nm, float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const int c = threadIdx.x;
  if (c < nm) dw[c] = accumulate ? dw[c] + sums[c] : sums[c];
  if (c == nm && db) *db = accumulate ? *db + sums[nm] : sums[nm];
}

// src/model.py variant: thread = (input pixel, 8-channel chunk).  mode 0 identity; 3 nearest x2 up (input [N,H/2,W/2,C]: the transpose sums
// the 2x2 block of dy); 4 max pooling 2x2 / 2 (input [N,2H,2W,C]: dy goes to the window's FIRST maximum in row-major order, torch's choice
// -- `val > maxval` while scanning -- and every other position gets zero).
template <typename T>
__global__ __launch_bounds__(256) void resample_bwd_dx(const T* __restrict__ dy, const T* __restrict__ xin, T* __restrict__ dx, int mode, int N, int H, int W,
                                                       int C, int accumulate) {
  const int CH8 = C >> 3;
  const int Hi = mode == 3 ? H >> 1 : (mode == 4 ? H << 1 : H), Wi = mode == 3 ? W >> 1 : (mode == 4 ? W << 1 : W);
  const long total = (long)N * Hi * Wi * CH8;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int chunk, x, y;
    long pix, ny, nn;
    divmod_u32(idx, CH8, pix, chunk);
    divmod_u32(pix, Wi, ny, x);
    divmod_u32(ny, Hi, nn, y);
    const int n = (int)nn;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[8];
    const T* dyn = dy + (long)n * H * W * C + chunk * 8;
    if (mode == 0) {
      ld8<T>(dyn + ((long)y * W + x) * C, acc);
    } else if (mode == 3) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          ld8<T>(dyn + ((long)(2 * y + a) * W + (2 * x + b)) * C, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += t[e];
        }
    } else {
      const int wy = y & ~1, wx = x & ~1, me = (y & 1) * 2 + (x & 1);
      const T* xn = xin + (long)n * Hi * Wi * C + chunk * 8;
      float best[8];
      int arg[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ld8<T>(xn + ((long)(wy + (k >> 1)) * Wi + (wx + (k & 1))) * C, t);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k == 0 || t[e] > best[e] || t[e] != t[e]) { best[e] = t[e]; arg[e] = k; }
      }
      ld8<T>(dyn + ((long)(y >> 1) * W + (x >> 1)) * C, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = arg[e] == me ? t[e] : 0.f;
    }
    T* dst = dx + pix * C + chunk * 8;
    if (accumulate) {
      ld8<T>(dst, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += t[e];
    }
    st8<T>(dst, acc);
  }
}

// WeightedAdd of src/model.py:27-37: w~ = relu(w) / (sum relu(w) + eps); the node ADDS sum_i w~_i = s / (s + eps) to every element.
__global__ void wadd_norm_kernel(const float* __restrict__ w, int n, float eps, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += fmaxf(w[i], 0.f);
  for (int i = 0; i < n; ++i) out[i] = fmaxf(w[i], 0.f) / (s + eps);
}
// d w_j (+)= [w_j > 0] * eps / (s + eps)^2 * sum(dy): `colsum` = the per-channel sums of dy (mtbt_channel_sum), folded here in a fixed order
__global__ __launch_bounds__(64) void wadd_norm_bwd_kernel(const float* __restrict__ w, int n, float eps, const float* __restrict__ colsum, int C, float* __restrict__ dw,
                                                           int accumulate) {
  float t = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) t += colsum[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if (threadIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += fmaxf(w[i], 0.f);
  const float g = eps / ((s + eps) * (s + eps)) * t;
  for (int i = 0; i < n; ++i) {
    const float v = w[i] > 0.f ? g : 0.f;
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

}  // namespace

extern "C" int64_t mtbt_bifpn_fuse_backward_workspace_bytes(void) { return 1024 * (int64_t)sizeof(float); }

// One input of a fusion node.  dy [N,H,W,C] (the node's output gradient), x_in / dx: that input, [N,H,W,C] (mode 0), [N,H/2,W/2,C]
// (mode 1, bilinear x2 up) or [N,2H,2W,C] (mode 2, 2x2 mean); wgt: DEVICE scalar w_i; dwgt: device scalar (+)= <dy, resample(x_in)>
// (NULL to skip); dx NULL to skip.  All dense NHWC in `dtype`.
extern "C" int mtbt_bifpn_fuse_backward(const void* dy, const void* x_in, int mode, const float* wgt, void* dx, int accumulate_dx, float* dwgt,
                                        int accumulate_dwgt, int N, int H, int W, int C, int dtype, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  if (!dy || !wgt || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || mode < 0 || mode > 2) return MTBT_EINVAL;
  if (mode == 1 && ((H & 1) || (W & 1))) return MTBT_EINVAL;
  if (dtype != MTBT_F32 && dtype != MTBT_BF16) return MTBT_EINVAL;
  if (!aligned16(dy) || (dx && !aligned16(dx)) || (x_in && !aligned16(x_in))) return MTBT_EALIGN;
  if (dwgt && (!x_in || !workspace || workspace_bytes < mtbt_bifpn_fuse_backward_workspace_bytes())) return MTBT_EWORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int Hi = mode == 1 ? H / 2 : (mode == 2 ? H * 2 : H), Wi = mode == 1 ? W / 2 : (mode == 2 ? W * 2 : W);
  if ((long)N * (Hi > H ? Hi : H) * (Wi > W ? Wi : W) * (C / 8) > 0xffffffffL) return MTBT_EINVAL;   // (32-bit piece indices in the kernels)
  if (dx) {
    const unsigned g = grid_cap((long)N * Hi * Wi * (C / 8), 256);
    if (dtype == MTBT_F32) hipLaunchKernelGGL(fuse_bwd_dx<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (float*)dx, wgt, mode, N, H, W, C, accumulate_dx);
    else hipLaunchKernelGGL(fuse_bwd_dx<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, wgt, mode, N, H, W, C, accumulate_dx);
  }
  if (dwgt) {
    float* partial = reinterpret_cast<float*>(workspace);
    const unsigned g = grid_cap((long)N * H * W * (C / 8), 256, 1024);
    if (dtype == MTBT_F32) hipLaunchKernelGGL(fuse_bwd_dot<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)x_in, mode, N, H, W, C, partial);
    else hipLaunchKernelGGL(fuse_bwd_dot<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x_in, mode, N, H, W, C, partial);
    hipLaunchKernelGGL(scalar_final, dim3(1), dim3(256), 0, s, partial, (int)g, dwgt, accumulate_dwgt);
  }
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int64_t mtbt_projector_backward_workspace_bytes(int N, int hp, int wp, int nm) {
  if (N <= 0 || hp <= 0 || wp <= 0 || nm <= 0) return 0;
  const int64_t P = (int64_t)N * hp * wp;
  const int64_t blocks = (P + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (P + blocks * (nm + 8) + (nm + 8)) * (int64_t)sizeof(float);
}

// dseg [N,Hout,Wout] f32 = d loss / d (bilinear-resized projector logits); protos [N,hp,wp,nm] f32 NHWC (the forward output);
// w [nm] = the projector's Conv2d(nm,1,1) weight.  d_protos [N,hp,wp,nm] in `dprotos_dtype` (+)=; dw [nm], db [1] fp32 (+)= (NULL to skip).
extern "C" int mtbt_projector_backward(const float* dseg, const float* protos, const float* w, void* d_protos, int dprotos_dtype, int accumulate_dprotos,
                                       float* dw, float* db, int accumulate_dw, int N, int hp, int wp, int nm, int Hout, int Wout, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
  if (!dseg || !w || !d_protos || !workspace || N <= 0 || hp <= 0 || wp <= 0 || nm <= 0 || nm % 8 || nm > 248 || Hout < hp || Wout < wp) return MTBT_EINVAL;
  if (dw && !protos) return MTBT_EINVAL;
  if (dprotos_dtype != MTBT_F32 && dprotos_dtype != MTBT_BF16) return MTBT_EINVAL;
  if (workspace_bytes < mtbt_projector_backward_workspace_bytes(N, hp, wp, nm)) return MTBT_EWORKSPACE;
  if (!aligned16(workspace) || (protos && !aligned16(protos))) return MTBT_EALIGN;
  const long P = (long)N * hp * wp;
  const long blocks = (P + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  float* dlow = reinterpret_cast<float*>(workspace);
  float* partial = dlow + (P + 3) / 4 * 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(proj_bwd_low, dim3(grid_cap(P, 256)), dim3(256), 0, s, dseg, dlow, N, hp, wp, Hout, Wout);
  if (dprotos_dtype == MTBT_F32)
    hipLaunchKernelGGL(proj_bwd_protos<float>, dim3(grid_cap(P * nm, 256)), dim3(256), 0, s, dlow, w, (float*)d_protos, P, nm, accumulate_dprotos);
  else
    hipLaunchKernelGGL(proj_bwd_protos<bf16_t>, dim3(grid_cap(P * nm, 256)), dim3(256), 0, s, dlow, w, (bf16_t*)d_protos, P, nm, accumulate_dprotos);
  if (dw) {
    const int cols = nm + 8;
    float* sums = partial + blocks * cols;
    hipLaunchKernelGGL(proj_bwd_w_partial, dim3((unsigned)blocks), dim3(256), 0, s, dlow, protos, P, nm, partial);
    hipLaunchKernelGGL(channel_sum_final, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, s, partial, (int)blocks, cols, sums, 0);
    hipLaunchKernelGGL(proj_bwd_w_out, dim3(1), dim3(256), 0, s, sums, nm, dw, db, accumulate_dw);
  }
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}


// One input of a WeightedAdd node of the src/model.py variant.  dy [N,H,W,C] (the node's output gradient); x_in (mode 4 only: the forward input,
// needed to find each window's maximum) / dx: [N,H,W,C] (mode 0), [N,H/2,W/2,C] (mode 3, nearest x2) or [N,2H,2W,C] (mode 4, max pooling 2x2).
extern "C" int mtbt_resample_backward(const void* dy, const void* x_in, int mode, void* dx, int accumulate, int N, int H, int W, int C, int dtype, void* stream) {
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (mode != 0 && mode != 3 && mode != 4)) return MTBT_EINVAL;
  if (mode == 3 && ((H & 1) || (W & 1))) return MTBT_EINVAL;
  if (mode == 4 && !x_in) return MTBT_EINVAL;
  if (dtype != MTBT_F32 && dtype != MTBT_BF16) return MTBT_EINVAL;
  if (!aligned16(dy) || !aligned16(dx) || (x_in && !aligned16(x_in))) return MTBT_EALIGN;
  const int Hi = mode == 3 ? H / 2 : (mode == 4 ? H * 2 : H), Wi = mode == 3 ? W / 2 : (mode == 4 ? W * 2 : W);
  if ((long)N * (Hi > H ? Hi : H) * (Wi > W ? Wi : W) * (C / 8) > 0xffffffffL) return MTBT_EINVAL;   // (32-bit piece indices in the kernel)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned g = grid_cap((long)N * Hi * Wi * (C / 8), 256);
  if (dtype == MTBT_F32) hipLaunchKernelGGL(resample_bwd_dx<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)x_in, (float*)dx, mode, N, H, W, C, accumulate);
  else hipLaunchKernelGGL(resample_bwd_dx<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x_in, (bf16_t*)dx, mode, N, H, W, C, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_wadd_norm_weights(const float* w, int n, float eps, float* out, void* stream) {
  if (!w || !out || n <= 0 || n > 8) return MTBT_EINVAL;
  hipLaunchKernelGGL(wadd_norm_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), w, n, eps, out);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}

extern "C" int mtbt_wadd_norm_weights_backward(const float* w, int n, float eps, const float* dy_colsum, int C, float* dw, int accumulate, void* stream) {
  if (!w || !dy_colsum || !dw || n <= 0 || n > 8 || C <= 0) return MTBT_EINVAL;
  hipLaunchKernelGGL(wadd_norm_bwd_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), w, n, eps, dy_colsum, C, dw, accumulate);
  MTBT_LAUNCH_CHECK();
  return MTBT_OK;
}
