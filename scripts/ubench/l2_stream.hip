// Micro-benchmark: how fast can ONE CU stream an L2-resident weight set (the latency-mode LightGlue FFN streams 1.0-1.15 MB per
// workgroup from L2 and gets ~30 B/clk/CU with 16-byte-per-lane buffer loads, 8 waves, 16-32 loads in flight per wave)?
// Every workgroup reads the same `bytes` region `reps` times; variants: waves per workgroup, loads in flight per wave (unroll),
// load form (0 = global_load_dwordx4 into registers, 1 = LDS-DMA global_load_lds_dwordx4), workgroups (38 = the one-pair launch, 256 = all CUs).
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int UNROLL, int FORM>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ w, int bytes, int reps, unsigned* sink, unsigned long long* clk) {
  extern __shared__ char smem[];
  const int nthr = blockDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    // wave-contiguous 1 KB units, unit u -> wave u % nwaves (like the FFN's per-wave fragment streams)
    const int nwaves = nthr >> 6, units = bytes >> 10;
    for (int u0 = wave; u0 < units; u0 += nwaves * UNROLL) {
      if (FORM == 0) {
        u4 v[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
          const int u = u0 + i * nwaves;
          v[i] = u < units ? *reinterpret_cast<const u4*>(w + (size_t)u * 1024 + lane * 16) : u4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
      } else {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
          const int u = u0 + i * nwaves;
          if (u < units) {
            const char* g = w + (size_t)u * 1024 + lane * 16;
            const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(smem + ((wave * UNROLL + i) & 63) * 1024));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds) : "memory");
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345u) sink[0] = acc;
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int UNROLL, int FORM>
static void run(const char* w, int bytes, int waves, int wgs, unsigned* sink, unsigned long long* clk) {
  const int reps = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<UNROLL, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<UNROLL, FORM>), dim3(wgs), dim3(waves * 64), 65536, 0, w, bytes, 2, sink, clk);  // warm
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<UNROLL, FORM>), dim3(wgs), dim3(waves * 64), 65536, 0, w, bytes, reps, sink, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(wgs);
  hipMemcpy(h.data(), clk, wgs * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (auto v : h) c += (double)v;
  c /= wgs;
  printf("form %d  waves %2d  in-flight/wave %2d  wgs %3d : %7.1f us per pass  %6.1f GB/s per CU  %5.1f B/clk per CU (clock %.2f GHz)\n", FORM, waves, UNROLL,
         wgs, ms * 1e3 / reps, (double)bytes * reps / (ms * 1e-3) / 1e9, (double)bytes * reps / c, c / (ms * 1e6));
}

int main() {
  const int bytes = 1152 * 1024;
  char* w; unsigned* sink; unsigned long long* clk;
  hipMalloc(&w, bytes); hipMemset(w, 1, bytes);
  hipMalloc(&sink, 64); hipMalloc(&clk, 4096 * 8);
  for (int wgs : {38, 256}) {
    for (int waves : {4, 8, 16}) {
      run<4, 0>(w, bytes, waves, wgs, sink, clk);
      run<8, 0>(w, bytes, waves, wgs, sink, clk);
      run<16, 0>(w, bytes, waves, wgs, sink, clk);
      run<32, 0>(w, bytes, waves, wgs, sink, clk);
      run<8, 1>(w, bytes, waves, wgs, sink, clk);
      run<16, 1>(w, bytes, waves, wgs, sink, clk);
    }
  }
  return 0;
}
