// Micro-benchmark: issue cost (shader clocks per wave-instruction) of a few VALU instruction classes on gfx950, alone and next to
// a co-resident wave that streams v_mfma_f32_32x32x16_f16 on the same SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// block = 512 threads = 8 waves = 2 per SIMD.  waves 0..3: the measured VALU stream; waves 4..7: MFMA stream (if with_mfma) or idle.
template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int with_mfma, int iters) {
  const int wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (!with_mfma) return;
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    f16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters * 6; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[0] + c2[0] + c3[0] == 12345.f) out[1000] = 1;
    return;
  }
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // v_mul_f32, 8 independent chains
      REP8(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (KIND == 1) {  // v_exp_f32
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (KIND == 2) {  // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (KIND == 3) {  // v_cvt_pk_f16_f32
      REP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %1, %0, %1\n v_cvt_pk_f16_f32 %3, %2, %3\n v_cvt_pk_f16_f32 %5, %4, %5\n v_cvt_pk_f16_f32 %7, %6, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (KIND == 4) {  // v_rcp_f32
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (KIND == 5) {  // v_pk_fma_f16
      REP8(asm volatile("v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3\n v_pk_fma_f16 %4, %4, %4, %4\n v_pk_fma_f16 %5, %5, %5, %5\n v_pk_fma_f16 %6, %6, %6, %6\n v_pk_fma_f16 %7, %7, %7, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (KIND == 6) {  // v_exp_f16
      REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[0] + p2[0] + p3[0] == 12345.f) out[1001] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND>
static void run(const char* name, unsigned long long* d) {
  const int iters = 200, nblk = 256;
  for (int with = 0; with < 2; ++with) {
    hipMemset(d, 0, 2048 * 8);
    hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(512), 0, 0, d, with, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-18s %s: %.2f clocks per wave-instruction\n", name, with ? "next to an MFMA stream" : "alone                 ", s / h.size() / (iters * 64.0));
  }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 2048 * 8);
  run<0>("v_mul_f32", d);
  run<1>("v_exp_f32", d);
  run<4>("v_rcp_f32", d);
  run<2>("v_pk_mul_f32", d);
  run<3>("v_cvt_pk_f16_f32", d);
  run<5>("v_pk_fma_f16", d);
  run<6>("v_exp_f16", d);
  return 0;
}
