// Probe (not shipped): sustained v_mfma_f32_32x32x16_f16 rate on this box with real (random) operands, so the
// conv kernels' MFMA fraction can be judged against what the chip sustains under its power budget.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void probe(const _Float16* a, const _Float16* b, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  h8 fa[4], fb[4];
  for (int i = 0; i < 4; ++i) {
    fa[i] = *reinterpret_cast<const h8*>(a + (i * 64 + lane) * 8);
    fb[i] = *reinterpret_cast<const h8*>(b + (i * 64 + lane) * 8);
  }
  f16x acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[(u + n) & 3], fb[u & 3], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  if (s == 12345.f) out[0] = s;
}
int main() {
  std::vector<_Float16> ha(2048), hb(2048);
  _Float16 *a, *b; float* o;
  hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 2048; ++i) { ha[i] = mode ? (_Float16)((rand() % 2001 - 1000) * 1e-3f) : (_Float16)0.f; hb[i] = mode ? (_Float16)((rand() % 2001 - 1000) * 1e-3f) : (_Float16)0.f; }
    hipMemcpy(a, ha.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 4096, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
      const int iters = 4000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<4>, dim3(256), dim3(threads), 0, 0, a, b, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * (threads / 64) * iters * 32.0 * 32768.0;
        if (rep) printf("%s operands, %d waves/SIMD: %.1f TFLOP/s (%.2f ms)\n", mode ? "random" : "zero", threads / 256, flops / ms / 1e9, ms);
      }
    }
  }
  return 0;
}
