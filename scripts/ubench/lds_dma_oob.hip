// What does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer resource) write for a lane whose offset fails the range check?
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_oob lds_dma_oob.hip ; prints the LDS words of four lanes after the DMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* in, unsigned* out, int nrec) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  for (int i = threadIdx.x; i < 512; i += 64) reinterpret_cast<unsigned*>(sm)[i] = 0x7f7f7f7fu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, nrec, 0x00020000);
  // lanes 0..31 in range (nrec = 512 bytes), lanes 32..47 beyond num_records, lanes 48..63 pushed out with 0xfffffff0
  const unsigned voff = threadIdx.x < 48 ? threadIdx.x * 16u : 0xfffffff0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sm, 16, voff, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + 1024), 16, voff, 256, 0, 0);  // soffset 256: lanes 16.. out of range
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = reinterpret_cast<unsigned*>(sm)[i];
}
int main() {
  char* d; unsigned* o;
  hipMalloc(&d, 4096); hipMalloc(&o, 2048);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 512);
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  for (int l : {0, 15, 16, 31, 32, 47, 48, 63}) printf("lane %2d: dma0 %08x  dma1(soffset 256) %08x\n", l, r[l * 4], r[256 + l * 4]);
  return 0;
}
