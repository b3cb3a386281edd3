// Shared pieces of the fused LightGlue FFN kernels (lg_kernels.hip: k_lg_ffn / k_lg_ffn4, lg_ffn16.hip: k_lg_ffn16).
#pragma once
#include "igemm.h"
#include "kernels.h"

namespace sship {

constexpr int kFfnLd = 520;
typedef float f2_t __attribute__((ext_vector_type(2)));
// GELU (exact-erf form, nn.GELU()) on two values.  gelu(y) = y Phi(y) = y sigmoid(g(y)) with g = logit(Phi), an odd function:
// g(y) = y Q(y^2), Q a degree-4 polynomial fitted (minimax on the relative error over |y| <= 12, scripts/fit_gelu.py) to
//   max |gelu_approx - y Phi(y)| = 7.0e-6,  relative 4.4e-5 where |gelu| >= 0.05
// i.e. under a fifth of half an fp16 ulp of the result, which is rounded to fp16 right after (tests/test_lightglue_known_answers.py
// evaluates these very constants in fp32 against erf).  Q > 0 everywhere, so the form saturates correctly (y -> +inf: y,
// y -> -inf: -0).  Cost per pair of values: 8 packed fp32 ops + 2 v_exp + 2 v_rcp; the Abramowitz-Stegun erfc form it
// replaces (|err| 1.5e-7) took 16 packed ops + 4 transcendentals + 2 max, and the GELU phase is VALU-issue bound.
// The coefficients carry the factor -log2(e) so that sigmoid(g) = 1 / (1 + exp2(y q(y^2))).
constexpr float kGeluQ0 = -2.301893292e+00f, kGeluQ1 = -1.054467824e-01f, kGeluQ2 = 4.423903354e-04f, kGeluQ3 = 7.747288073e-05f,
                kGeluQ4 = -2.787147429e-06f;
__device__ __forceinline__ f2_t gelu2(f2_t y) {
  const f2_t s = y * y;
  f2_t q = s * kGeluQ4 + kGeluQ3;
  q = q * s + kGeluQ2;
  q = q * s + kGeluQ1;
  q = q * s + kGeluQ0;
  const f2_t t = y * q;
  const f2_t d = {1.0f + __builtin_amdgcn_exp2f(t[0]), 1.0f + __builtin_amdgcn_exp2f(t[1])};
  const f2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return y * r;
}
struct FfnTail {
  int ntiles;                               // token tiles of the launch (the kernel is persistent: tile = blockIdx.x + k gridDim.x)
  IgemmArgs proj;          // epilogue arguments of the fused projection (wpack/bias/outputs/rope/np/flags/cout/H)
  unsigned long long* trace;  // SSHIP_FFN_TRACE: [workgroup][wave][12] shader-clock stamps of the workgroup's 2nd tile
  int trace_it;               // ... or of tile iteration SSHIP_FFN_TRACE_IT (0 = first: latency mode has one tile per workgroup)
  const float* match_w;    // final block only: matchability weights [256] ...
  float match_b;
  float* logsig;           // ... -> logsigmoid(z) per token
  // latency mode (k_lg_ffn with a few workgroups): workgroups n_main .. gridDim.x - 1 do no FFN work - they pull the NEXT launch's packed
  // weights (up to three regions) into the L2 of the XCD they happen to run on, so that launch streams from L2 instead of HBM / MALL
  int n_main;              // 0: no prefetch workgroups (gridDim.x workgroups walk the tiles)
  const void* pf_ptr[3];
  int pf_bytes[3];
};

// lg_ffn16.hip: the 16-wave, one-workgroup-per-CU form of the fused block (throughput batches)
bool ffn16_applicable(int tokens, int next_mt, bool heads);
hipError_t launch_lg_ffn16(int tokens, int next_mt, bool heads, hipStream_t s, const _Float16* ctx, const _Float16* w0p, const float* b0,
                           const float* gamma, const float* beta, const _Float16* w3p, const float* b3, _Float16* x, FfnTail t);

}  // namespace sship
