// Shared device helpers for the gfx950 kernels of libmtbt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mtbt_hip.h"

typedef uint16_t bf16_t;  // storage type of a bf16 element
struct f16_t { uint16_t v; };  // storage type of an IEEE binary16 element (a distinct type: the kernels dispatch on it)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // a 16-byte access the optimiser keeps whole (HIP's uint4 / float4 are structs: see st8<float>)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round-to-nearest-even, NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// fp16 <-> fp32.  Stores SATURATE at +-65504 instead of overflowing to infinity (BASELINE configs[4]: "fp16 MFMA conv path"): an
// activation beyond the binary16 range would otherwise poison every later layer.  NaN stays NaN, by an explicit select: v_med3_f32
// with a NaN operand returns MIN3 of the operands and the IEEE minimum DROPS the NaN, so the clamp alone would store a NaN activation as
// -65504 and hide a divergence (tests/test_gpu_kernels.py::test_fp16_store_saturates_and_keeps_nan).
__device__ __forceinline__ float h2f(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.v); }
__device__ __forceinline__ uint16_t f2h_bits(float f) {
  const float c = __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f);
  const _Float16 h = (_Float16)(f != f ? f : c);
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ f16_t f2h(float f) { return f16_t{f2h_bits(f)}; }
__device__ __forceinline__ uint32_t pk_h2(float a, float b) {   // two floats -> packed half pair (one v_cvt_pkrtz-free RNE convert each)
  return (uint32_t)f2h_bits(a) | ((uint32_t)f2h_bits(b) << 16);
}
__device__ __forceinline__ float h_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float h_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }

// the 16-bit type a kernel templated on T stores when its output is not fp32 (T itself; bf16_t stands in for the dead T = float branch)
template <typename T> struct half_of { typedef T type; };
template <> struct half_of<float> { typedef bf16_t type; };

template <typename T> __device__ __forceinline__ float ld_elem(const T* p);
template <> __device__ __forceinline__ float ld_elem<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_elem<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <> __device__ __forceinline__ float ld_elem<f16_t>(const f16_t* p) { return h2f(*p); }

template <typename T> __device__ __forceinline__ void st_elem(T* p, float v);
template <> __device__ __forceinline__ void st_elem<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_elem<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

template <> __device__ __forceinline__ void st_elem<f16_t>(f16_t* p, float v) { *p = f2h(v); }

// 8 consecutive elements <-> 8 floats (two 16-B accesses for f32, one for bf16 / fp16)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld8<f16_t>(const f16_t* p, float (&v)[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0] = h_lo(u.x); v[1] = h_hi(u.x); v[2] = h_lo(u.y); v[3] = h_hi(u.y);
  v[4] = h_lo(u.z); v[5] = h_hi(u.z); v[6] = h_lo(u.w); v[7] = h_hi(u.w);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  // two <4 x float> stores as VECTORS (ext_vector_type): written as HIP float4 structs they reach the optimiser as eight scalar stores, and
  // next to a scalar tail path that writes the same addresses it regrouped them as 12 + 16 + 4 bytes -- a dwordx3, a MISALIGNED dwordx4 and a
  // dword per 32 bytes (ISA of round 3: the fp32 maps of the heads and of Proto cv3)
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}

template <> __device__ __forceinline__ void st8<f16_t>(f16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = pk_h2(v[0], v[1]); u.y = pk_h2(v[2], v[3]); u.z = pk_h2(v[4], v[5]); u.w = pk_h2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}

// one 16x16x32 MFMA on 8-element fragments held as raw 16-byte words: bf16 or fp16 by the storage type
template <typename T> __device__ __forceinline__ f32x4 mfma_16x16x32(uint4 a, uint4 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma_16x16x32<bf16_t>(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma_16x16x32<f16_t>(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma_16x16x32<float>(uint4, uint4, f32x4 c) { return c; }   // (never called: the f32 path uses 16x16x4)

// Epilogue activations.  These run on every conv output, so they use the hardware exp2 / rcp
// (v_exp_f32, v_rcp_f32: ~1 ulp) instead of libm: absolute error <= ~3e-7 of the libm value, three
// orders below the 1e-3 parity tolerance.
__device__ __forceinline__ float fast_exp(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): 1 exp2 + 1 rcp + 6 FMA
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = fast_rcp(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * fast_exp(-ax * ax);
  return copysignf(r, x);
}

// GELU(x) = x * Phi(x), Phi(x) - 1/2 = xc * P(xc^2) with xc = clamp(x, -4, 4): weighted least-squares fit on Chebyshev
// nodes (max |GELU error| 2.3e-4 over all x, relative 3.6e-5 for x > 4).  11 plain VALU operations, no transcendental:
// the erf form costs 2 half-rate + ~14 plain and made the GELU, not the MFMAs, the bound of the fused ConvNeXt MLP.
__device__ __forceinline__ float gelu_poly(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
  const float s = xc * xc;
  float p = fmaf(2.1609857e-08f, s, -1.5335673e-06f);
  p = fmaf(p, s, 4.6542096e-05f);
  p = fmaf(p, s, -7.9887325e-04f);
  p = fmaf(p, s, 8.6900834e-03f);
  p = fmaf(p, s, -6.4366050e-02f);
  p = fmaf(p, s, 3.9770728e-01f);
  return x * fmaf(xc, p, 0.5f);
}

// d/dx of gelu_poly: g(x) = x / 2 + x^2 P(x^2) on |x| < 4  ->  g'(x) = 1/2 + 2 x Q(x^2), Q(s) = P(s) + s P'(s) (coefficients (i + 1) c_i);
// outside, xc is constant and g is linear: g' = 1/2 + xc P(16).  The exact derivative of what the forward computed; 8 plain VALU operations.
__device__ __forceinline__ float gelu_poly_grad(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
  const float s = xc * xc;
  float q = fmaf(7.0f * 2.1609857e-08f, s, 6.0f * -1.5335673e-06f);
  q = fmaf(q, s, 5.0f * 4.6542096e-05f);
  q = fmaf(q, s, 4.0f * -7.9887325e-04f);
  q = fmaf(q, s, 3.0f * 8.6900834e-03f);
  q = fmaf(q, s, 2.0f * -6.4366050e-02f);
  q = fmaf(q, s, 3.9770728e-01f);
  constexpr float P16 = ((((((2.1609857e-08f * 16.f + -1.5335673e-06f) * 16.f + 4.6542096e-05f) * 16.f + -7.9887325e-04f) * 16.f + 8.6900834e-03f) * 16.f +
                          -6.4366050e-02f) * 16.f + 3.9770728e-01f);
  const float inside = fmaf(2.0f * xc, q, 0.5f), outside = fmaf(xc, P16, 0.5f);
  return fabsf(x) < 4.0f ? inside : outside;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case MTBT_ACT_SILU: return v * fast_rcp(1.0f + fast_exp(-v));
    case MTBT_ACT_ELU: return v > 0.0f ? v : (v > -0.03f ? expm1f(v) : fast_exp(v) - 1.0f);
    case MTBT_ACT_GELU: return 0.5f * v * (1.0f + fast_erf(v * 0.70710678118654752440f));
    case MTBT_ACT_GELU_POLY: return gelu_poly(v);
    default: return v;
  }
}

// Derivative of the epilogue activations at the PRE-activation z (backward pass): dz = dy * act_grad(z).  Codes 5..7 are the
// "multiply by act'(res)" epilogue modes of the conv kernel (MTBT_ACT_DSILU / DELU / DGELU).
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == MTBT_ACT_SILU || act == MTBT_ACT_DSILU) { const float s = fast_rcp(1.f + fast_exp(-z)); return s * (1.f + z * (1.f - s)); }
  if (act == MTBT_ACT_ELU || act == MTBT_ACT_DELU) return z > 0.f ? 1.f : fast_exp(z);
  if (act == MTBT_ACT_GELU_POLY || act == MTBT_ACT_DGELU_POLY) return gelu_poly_grad(z);
  if (act == MTBT_ACT_GELU || act == MTBT_ACT_DGELU)
    return 0.5f * (1.f + fast_erf(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * fast_exp(-0.5f * z * z);
  return 1.f;
}

// v = q * d + r for an element / piece index v that fits 32 bits (every tensor of this library: the entry points refuse more than 2^32 - 1
// pieces).  `long` quotients compile to 64-bit divisions of ~100 instructions each; four of them per 16-byte piece made index arithmetic the
// bound of several "HBM-bound" kernels (round 3).
__device__ __forceinline__ void divmod_u32(long v, int d, long& q, int& r) {
  const unsigned u = (unsigned)v, qq = u / (unsigned)d;
  q = (long)qq;
  r = (int)(u - qq * (unsigned)d);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// One-time (per kernel instantiation AND per device) opt-in to more than 64 KiB of dynamic LDS.  The flag array is racy
// only in the benign direction (two host threads may both set the same attribute).
template <typename K>
static inline int mtbt_allow_lds(K kern, int lds) {
  static bool done[64] = {};
  if (lds <= 64 * 1024) return MTBT_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MTBT_ELAUNCH;
  if (!done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MTBT_ELAUNCH;
    done[dev] = true;
  }
  return MTBT_OK;
}

#define MTBT_LAUNCH_CHECK()                          \
  do {                                               \
    if (hipGetLastError() != hipSuccess) return MTBT_ELAUNCH; \
  } while (0)
