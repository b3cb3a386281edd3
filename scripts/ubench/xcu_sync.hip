// Micro-benchmark: what does an exchange between workgroups on DIFFERENT CUs cost inside one launch?
// (the latency-mode LightGlue FFN split over a cluster of four workgroups per 32-token tile needs three of them.)
// A cluster of C = 4 workgroups runs R rounds of: every member writes an 8 KB slice, the cluster synchronises through a counter in
// global memory, every member reads the three other slices and checks them.  Variants:
//   placement 0: cluster members on the SAME XCD (workgroup ids are dealt round-robin over the 8 XCDs: ids b, b + 8, b + 16, b + 24)
//   placement 1: consecutive ids = four different XCDs
//   mode 0: release / acquire at agent scope through the compiler's builtins (buffer_wbl2 sc1 / buffer_inv sc1 around the atomics)
//   mode 1: "light": stores, s_waitcnt vmcnt(0), relaxed atomic add; poll with a relaxed agent-scope load; data loads with sc1 (served by L2)
//   mode 2: sync only (no data) with mode 1's primitives
//   mode 3 / 4: per-member flag words written by stores and polled with one 16-byte load - sync only / with the data exchange
// Reports shader clocks per round (max over workgroups), the number of wrong values read, and the XCC_ID the workgroups saw.
// Spins are bounded: a cluster that never completes reports a timeout instead of hanging the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 -o xcu_sync xcu_sync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int C = 4, SLICE = 512;  // 16-byte units per slice (8 KB)

__device__ __forceinline__ u4 load_sc1(const u4* p) {
  u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(u4* buf, unsigned* ctr, int rounds, int placement, unsigned base, unsigned long long* clk, unsigned* err,
                                         unsigned* xcc) {
  const int b = blockIdx.x, tid = threadIdx.x;
  int tile, mem;
  if (placement == 0) { const int x = b & 7, q = b >> 3; tile = (q / C) * 8 + x; mem = q % C; }
  else { tile = b / C; mem = b % C; }
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (tid == 0) xcc[b] = id & 0xf;
  unsigned bad = 0, timeouts = 0;
  __shared__ int s_ok;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    u4* mine = buf + (((size_t)tile * C + mem) * 2 + (r & 1)) * SLICE;
    if (MODE != 2 && MODE != 3) mine[tid] = u4{(unsigned)r, (unsigned)mem, (unsigned)tid, (unsigned)(r * 131 + mem * 7 + tid)};
    const unsigned target = base + (unsigned)(r + 1) * C;
    if (MODE >= 3) {
      // every member owns one flag word of its cluster's 16-byte line: a store (no read-modify-write at L2), polled with ONE 16-byte load
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        unsigned* fl = ctr + 64 + tile * 4;
        const unsigned ep = base + (unsigned)(r + 1);
        asm volatile("global_store_dword %0, %1, off sc1" :: "v"(fl + mem), "v"(ep) : "memory");
        int spins = 0;
        for (;;) {
          const u4 f = load_sc1(reinterpret_cast<const u4*>(fl));
          if ((int)(f.x - ep) >= 0 && (int)(f.y - ep) >= 0 && (int)(f.z - ep) >= 0 && (int)(f.w - ep) >= 0) break;
          if (++spins >= (1 << 22)) break;
        }
        s_ok = spins < (1 << 22);
      }
      __syncthreads();
    } else if (MODE == 0) {
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(ctr + tile, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while ((int)(__hip_atomic_load(ctr + tile, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < (1 << 22)) {}
        s_ok = spins < (1 << 22);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have been acknowledged by L2
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while ((int)(__hip_atomic_load(ctr + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < (1 << 22)) {}
        s_ok = spins < (1 << 22);
      }
      __syncthreads();
    }
    if (!s_ok) { ++timeouts; break; }
    if (MODE != 2 && MODE != 3) {
#pragma unroll
      for (int o = 1; o < C; ++o) {
        const int m2 = (mem + o) % C;
        const u4* theirs = buf + (((size_t)tile * C + m2) * 2 + (r & 1)) * SLICE;
        const u4 v = MODE == 0 ? theirs[tid] : load_sc1(theirs + tid);  // (one round trip per partner slice here; the kernel would issue them together)
        bad += v.x != (unsigned)r || v.y != (unsigned)m2 || v.z != (unsigned)tid || v.w != (unsigned)(r * 131 + m2 * 7 + tid);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) clk[b] = t1 - t0;
  if (bad) atomicAdd(err, bad);
  if (timeouts && tid == 0) atomicAdd(err + 1, 1u);
}

int main() {
  const int tiles = 40, wgs = tiles * C, rounds = 64;
  u4* buf; unsigned *ctr, *err, *xcc; unsigned long long* clk;
  hipMalloc(&buf, (size_t)tiles * C * 2 * SLICE * 16);
  hipMalloc(&ctr, 64 * 4 + 64 * 16); hipMalloc(&err, 8); hipMalloc(&xcc, wgs * 4); hipMalloc(&clk, wgs * 8);
  std::vector<unsigned long long> h(wgs);
  std::vector<unsigned> hx(wgs);
  for (int placement = 0; placement < 2; ++placement)
    for (int mode = 0; mode < 5; ++mode) {
      hipMemset(ctr, 0, 64 * 4 + 64 * 16); hipMemset(err, 0, 8); hipMemset(buf, 0xff, (size_t)tiles * C * 2 * SLICE * 16);
      unsigned base = 0;
      for (int rep = 0; rep < 3; ++rep) {  // the counters keep counting across launches (base), as they would across FFN launches
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(512), 0, 0, buf, ctr, rounds, placement, base, clk, err, xcc);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(512), 0, 0, buf, ctr, rounds, placement, base, clk, err, xcc);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(512), 0, 0, buf, ctr, rounds, placement, base, clk, err, xcc);
        else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(512), 0, 0, buf, ctr, rounds, placement, base, clk, err, xcc);
        else hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(512), 0, 0, buf, ctr, rounds, placement, base, clk, err, xcc);
        base += mode >= 3 ? rounds : rounds * C;
      }
      hipDeviceSynchronize();
      unsigned he[2];
      hipMemcpy(h.data(), clk, wgs * 8, hipMemcpyDeviceToHost);
      hipMemcpy(he, err, 8, hipMemcpyDeviceToHost);
      hipMemcpy(hx.data(), xcc, wgs * 4, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; double avg = 0;
      for (auto v : h) { mx = v > mx ? v : mx; avg += (double)v; }
      int same = 0;  // clusters whose members all saw the same XCC_ID
      for (int t = 0; t < tiles; ++t) {
        unsigned ids[C]; int n = 0;
        for (int b = 0; b < wgs; ++b) {
          int tile, mem;
          if (placement == 0) { const int x = b & 7, q = b >> 3; tile = (q / C) * 8 + x; mem = q % C; } else { tile = b / C; mem = b % C; }
          if (tile == t && n < C) { ids[mem] = hx[b]; ++n; }
        }
        same += n == C && ids[0] == ids[1] && ids[1] == ids[2] && ids[2] == ids[3];
      }
      printf("placement %d (%s) mode %d (%s): %.0f clk per round (max over workgroups; mean %.0f), wrong values %u, timeouts %u, clusters on one XCD %d / %d\n",
             placement, placement ? "consecutive ids" : "ids b + 8 k", mode, mode == 0 ? "release/acquire builtins" : mode == 1 ? "light: vmcnt + relaxed atomics + sc1 loads" : mode == 2 ? "sync only (atomics)" : mode == 3 ? "sync only (flag stores, one 16-byte poll)" : "flag stores + data",
             (double)mx / rounds, avg / wgs / rounds, he[0], he[1], same, tiles);
    }
  return 0;
}
