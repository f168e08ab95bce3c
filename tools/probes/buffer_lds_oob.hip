// Probe: what does a raw-buffer LDS-DMA load (buffer_load_dwordx4 ... lds) write to LDS for lanes whose offset is
// out of the descriptor's range?  (Needed to use range-checked DMA as the zero-fill of conv padding.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out, int soff) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // even lanes: in range (lane*16), odd lanes: far out of range
  unsigned voff = (lane & 1) ? 0x80000000u + lane * 16 : lane * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  for (int soff : {0, 1024}) {
    probe<<<1, 64>>>(d, 2048, o, soff);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("soffset=%d num_records=2048\n", soff);
    for (int l = 0; l < 8; ++l) printf("  lane %d: %08x %08x %08x %08x\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
    printf("  lane 62: %08x   lane 63: %08x\n", r[62*4], r[63*4]);
  }
  return 0;
}
