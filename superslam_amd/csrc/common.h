// Shared device/host helpers for libsuperslam_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace sship {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));   // MFMA A/B fragment: 8 f16 (4 VGPRs)
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f16x_t __attribute__((ext_vector_type(16)));   // 32x32 MFMA accumulator
typedef float f4x_t __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// Developer A/B switches.  The DEFAULT library reads exactly the environment variables documented in include/sship.h ("Environment":
// SUPERSLAM_HIP_DEVICE, SSHIP_RCCL_LIBRARY) and runs one kernel per layer; every other switch (kernel selection, phase traces) and every
// rejected kernel exists only in the developer build  `python -m superslam_amd.build --variant dev -DSSHIP_DEV_SWITCHES=1`
// (superslam_amd/lib/variants/dev.so, loaded explicitly - superslam_amd._lib.set_library_path - by tests/test_gpu_alt_paths.py and scripts/dev/*): there
// dev_env() is getenv(), here it is a constant null and the branches behind it fold away.
#ifndef SSHIP_DEV_SWITCHES
#define SSHIP_DEV_SWITCHES 0
#endif
#if SSHIP_DEV_SWITCHES
inline const char* dev_env(const char* name) { return ::getenv(name); }
#else
constexpr const char* dev_env(const char*) { return nullptr; }
#endif

// k order of the 64-input-channel 3x3 convolutions (conv_pp.hip CIN = 64, conv_fuse2.hip).  1 (round 6): (kx, k-step, ky) - the three tap rows of one
// (kx, k-step) read FOUR input-row fragments for their six (row, ky) pairs instead of six: one LDS fragment read in three is gone.  0: tap-major
// (ky, kx, k-step), rounds 1-5 (A/B builds: build.py --variant ... -DSSHIP_K_ROWSHARE=0).  Both files must agree: a frame extracted alone (two launches)
// and the same frame inside a throughput batch (fused kernel) are bit-identical only if every accumulator sees its products in the same order.
#ifndef SSHIP_K_ROWSHARE
#define SSHIP_K_ROWSHARE 1
#endif

// Thread-local last error + status plumbing (host side).
void set_error(const std::string& msg);
void log_msg(int level, const char* fmt, ...);

#define SSHIP_HIP_CHECK(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::sship::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
      return SSHIP_ERR_HIP;                                                                \
    }                                                                                      \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// MFMA 32x32x16 f16: D[32x32] += A[32x16] * B[16x32].
//   A fragment: lane l holds A[i = l & 31][k = 8*(l >> 5) + e], e = 0..7
//   B fragment: lane l holds B[k = 8*(l >> 5) + e][j = l & 31]
//   D: lane l, reg r holds D[row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][col = l & 31]
__device__ __forceinline__ f16x_t mfma32(h8_t a, h8_t b, f16x_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): ~14 VALU instead of ~40 for erff.  The GELU that uses
// it is rounded to fp16 (2^-11 relative) right after, so the result is the correctly rounded exact-erf GELU in all
// but ~1e-4 of the cases (and then off by one fp16 ulp).
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(fmaf(-p, e, 1.0f), x);
}

__device__ __forceinline__ h4_t to_h4(float a, float b, float c, float d) {
  h4_t v;
  v[0] = (_Float16)a; v[1] = (_Float16)b; v[2] = (_Float16)c; v[3] = (_Float16)d;
  return v;
}

}  // namespace sship
