// LDS-DMA helpers shared by the conv kernels: raw buffer descriptors in SGPRs, buffer_load ... lds from inline asm,
// counted vmcnt waits.  (Device code; include inside an anonymous namespace-free context.)
#pragma once
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned srd_t;  // raw buffer descriptor words (SGPR quad)

// Raw buffer descriptor over [base, base + 2 GiB): stride 0, DATA_FORMAT=32 (0x00020000), wave-uniform.
__device__ __forceinline__ srd_t make_srd(const void* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  srd_t d;
  d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  d.z = 0x80000000u;
  d.w = 0x00020000u;
  return d;
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B from (descriptor + soffset + per-lane voffset) to LDS bytes
// [lds_addr, lds_addr + 1024), lane-linear.  Out-of-range voffset -> zeros.  M0 carries the LDS base and is
// saved/restored (compiler-reserved); s_nop 4 covers freshly written scalar operands (guide 5.7).
// The compiler does NOT count this load: callers wait with their own s_waitcnt vmcnt(N).
__device__ __forceinline__ void lds_dma16(srd_t srd, unsigned voffset, int soffset, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(srd), "s"(lds_addr), "s"(soffset)
      : "memory");
}

// Workgroup barrier of the LDS-DMA pipelines.  Two things the bare `__builtin_amdgcn_s_barrier()` does not give:
//  * `s_waitcnt lgkmcnt(0)` first: the buffer restaged right after this barrier is the one this wave read LAST step;
//    the compiler may leave those ds_reads in flight across the barrier (it parks the MFMAs that consume them behind
//    it), and a pending read that loses the race against another wave's DMA write returns the NEW bytes (WAR);
//  * a compiler memory barrier: the builtin is IntrNoMem, so fragment reads of the NEXT step may be hoisted above it,
//    i.e. before the other waves' DMA pieces have landed (RAW).
// Both showed up as rare wrong tiles only when the CU was shared with another kernel (LDS port contention stretches
// the window); found with tools/pair_stress.py.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most k*G vector-memory operations (LDS-DMA pieces) of this wave are outstanding, k in [0, KMAX]
template <int G, int KMAX> __device__ __forceinline__ void wait_stages(int k) {
  if constexpr (KMAX <= 0) wait_vm<0>();
  else { if (k >= KMAX) wait_vm<KMAX * G>(); else wait_stages<G, KMAX - 1>(k); }
}

}  // namespace
