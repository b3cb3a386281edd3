// Micro-benchmark: how MFMA and VALU work of the waves of ONE SIMD overlap on gfx950 (round 3).
// The LightGlue attention and FFN kernels run two waves per SIMD whose instruction streams mix v_mfma_f32_32x32x16_f16 with
// VALU / transcendental work; their key-loop time looks like the SUM of the VALU issue time and the MFMA time.  This measures,
// for W waves per SIMD (block = 256 W threads, one block per CU), shader clocks per loop iteration and per wave of
//   mode 0  MFMA only          : 16 MFMAs on 4 independent accumulators
//   mode 1  VALU only          : 128 independent v_fma_f32 (16 chains)
//   mode 2  block mix          : 16 MFMAs, then 128 VALU that do not depend on them
//   mode 3  fine mix           : (1 MFMA, 8 VALU) x 16, independent
//   mode 4  dependent mix      : attention-like - 2 chains of 4 MFMAs -> v_max3 over the result -> 16 v_exp + 8 cvt_pk ->
//                                the packed values are the B operand of 4 MFMAs (PV); x 2 "query tiles"
//   mode 5  mode 4 with the two query tiles' QK chains issued first (the kernel's variant 3 ordering)
// Build: hipcc --offload-arch=gfx950 -O3 -fno-honor-nans -Xclang -target-feature -Xclang -packed-fp32-ops -o mix_rates mix_rates.hip ; run: ./mix_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16v mfma(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int MODE>
__global__ __launch_bounds__((MODE >= 6 && MODE <= 8 ? 512 : 1024)) void k(unsigned long long* out, float* sink, int iters, const _Float16* kv, int share) {
  const int lane = threadIdx.x & 63;
  h8 a, b, a2, b2;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); a2[e] = (_Float16)(0.03f * (lane + 2 * e)); b2[e] = (_Float16)(0.015f * (lane - 3 * e)); }
  if (share < 0) {  // random operands in [-0.25, 0.25): every mantissa bit toggles, as with real descriptors (the constants above leave the datapath cold)
    share = -share;
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (_Float16)(((int)(h >> 8) & 0xffff) * (0.5f / 65536.f) - 0.25f); };
    for (int e = 0; e < 8; ++e) { a[e] = rnd(); b[e] = rnd(); a2[e] = rnd(); b2[e] = rnd(); }
  }
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = 0.001f * (lane + i);
  const float m = 0.999f, ad = 1e-4f;
  h8 fA[8], fB[8];
  for (int i = 0; i < 8; ++i) { fA[i] = a; fB[i] = b; }
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { c0 = mfma(a, b, c0); c1 = mfma(a, b, c1); c2 = mfma(a, b, c2); c3 = mfma(a, b, c3); }
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], m, ad);
    }
    if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if ((r & 3) == 0) c0 = mfma(a, b, c0); else if ((r & 3) == 1) c1 = mfma(a, b, c1); else if ((r & 3) == 2) c2 = mfma(a, b, c2); else c3 = mfma(a, b, c3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[(r & 1) * 8 + i] = fmaf(x[(r & 1) * 8 + i], m, ad);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE >= 6 && MODE <= 8) {
      // mode 6: mode 5 + the kernel's K / V^T fragment traffic: 8 x 1 KiB wave loads per iteration from an L2-resident 156 KB
      //         region per (block, wave pair), requested one iteration ahead into a second register set (two sets alternate);
      // mode 7: + lane^32 exchange of the tile maximum (v_permlane32_swap) and the `__any(max > r + 8)` test with the rescale
      //         branch behind it (never taken here: the cost is the VALU -> SALU round trip of the vote);
      // mode 8: + the ragged-tile test (scalar compare, not taken) and the first-tile flag.
      const f16v z = {0};
      // `share` workgroups of one XCD read the same 156 KB region (share = 5: the real kernel's query blocks of one (sequence, head);
      // share = 1: every workgroup its own region - nothing is shared beyond the waves of a workgroup)
      const _Float16* base = kv + (size_t)((((blockIdx.x >> 3) / share) * 8 + (blockIdx.x & 7)) % 1024) * (19 * 4096) + lane * 8;
      auto fetch = [&](h8 (&f)[8], int kt) {
        const _Float16* pk = base + (size_t)(kt % 19) * 4096;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const h8*>(pk + i * 512);
      };
      auto tile = [&](const h8 (&f)[8], int kt) {
        f16v s[2];
        s[0] = mfma(a, b, z); s[1] = mfma(a, b2, z);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { s[0] = mfma(f[ks], b, s[0]); s[1] = mfma(f[ks], b2, s[1]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (MODE >= 8 && kt * 32 + 32 > iters * 1000) {  // ragged tile: never
#pragma unroll
            for (int r = 0; r < 16; ++r) if (lane + r > 70) s[t][r] = -1e30f;
          }
          float tm = max3(s[t][0], s[t][1], s[t][2]);
          tm = max3(tm, s[t][3], s[t][4]);
#pragma unroll
          for (int r = 5; r < 15; r += 2) tm = max3(tm, s[t][r], s[t][r + 1]);
          if (MODE >= 7) {
            const unsigned u = __float_as_uint(tm);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            tm = max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), s[t][15]);
            const bool need = (MODE >= 8 && kt < 0) || tm > 1e30f;
            if (__any(need)) {
              const float d = tm - x[t];
              const float al = __builtin_amdgcn_exp2f(-d);
              x[2 + t] *= al;
#pragma unroll
              for (int r = 0; r < 16; ++r) { if (t == 0) { c0[r] *= al; c1[r] *= al; } else { c2[r] *= al; c3[r] *= al; } s[t][r] -= d; }
              x[t] = tm;
            }
          } else x[t] = fmaxf(x[t], tm);
          h8 p[2];
          float l0 = 0.f, l1 = 0.f;
          const h2 ones = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const h2 pp = {(_Float16)__builtin_amdgcn_exp2f(s[t][8 * kk + e]), (_Float16)__builtin_amdgcn_exp2f(s[t][8 * kk + e + 1])};
              p[kk][e] = pp[0]; p[kk][e + 1] = pp[1];
              if (kk == 0) l0 = __builtin_amdgcn_fdot2(pp, ones, l0, false); else l1 = __builtin_amdgcn_fdot2(pp, ones, l1, false);
            }
          x[4 + t] += l0 + l1;
          if (t == 0) { c0 = mfma(f[4], p[0], c0); c1 = mfma(f[5], p[0], c1); c0 = mfma(f[6], p[1], c0); c1 = mfma(f[7], p[1], c1); }
          else { c2 = mfma(f[4], p[0], c2); c3 = mfma(f[5], p[0], c3); c2 = mfma(f[6], p[1], c2); c3 = mfma(f[7], p[1], c3); }
        }
      };
      if (it == 0) fetch(fA, 0);
      fetch(fB, 2 * it + 1);
      __builtin_amdgcn_sched_barrier(0);
      tile(fA, 2 * it);
      fetch(fA, 2 * it + 2);
      __builtin_amdgcn_sched_barrier(0);
      tile(fB, 2 * it + 1);
    }
    if (MODE == 9 || MODE == 10) {
      // Winograd F(2x2, 3x3)-shaped mix per 8 MFMAs (one group of four positions x two M-tiles, one 16-channel k-step):
      //   input transform of the lane's tile column: 8 vector adds on 4 packed-fp16 registers = 32 v_pk_add_f16,
      //   mode 10 also the output transform of one finished position group per FOUR such k-steps: 7 fp32 adds per MFMA on average
      //   (224 v_add_f32 per 32 MFMAs), i.e. 56 per iteration here.
      // (instruction counts pinned with inline asm: the compiler folded a third of a C++ version away)
      unsigned* ua = reinterpret_cast<unsigned*>(&fA[0]);
      unsigned* ub = reinterpret_cast<unsigned*>(&fB[0]);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if ((r & 3) == 0) c0 = mfma(fA[r & 7], b, c0); else if ((r & 3) == 1) c1 = mfma(fA[(r + 1) & 7], b2, c1); else if ((r & 3) == 2) c2 = mfma(fB[r & 7], b, c2); else c3 = mfma(fB[(r + 1) & 7], b2, c3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(ua[(4 * r + i) & 31]) : "v"(ub[(4 * r + i + 5) & 31]));
        if (MODE == 10) {
#pragma unroll
          for (int i = 0; i < 7; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(7 * r + i) & 15]) : "v"(x[(7 * r + i + 3) & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 4 || MODE == 5) {
      const f16v z = {0};
      f16v s[2];
      auto qk = [&](int t) {
        s[t] = mfma(a, t ? b2 : b, z);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[t] = mfma(ks & 1 ? a2 : a, t ? b2 : b, s[t]);
      };
      auto softmax_pv = [&](int t) {
        float tm = max3(s[t][0], s[t][1], s[t][2]);
        tm = max3(tm, s[t][3], s[t][4]);
#pragma unroll
        for (int r = 5; r < 15; r += 2) tm = max3(tm, s[t][r], s[t][r + 1]);
        x[t] = fmaxf(x[t], tm);
        h8 p[2];
        float l0 = 0.f, l1 = 0.f;
        const h2 ones = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const h2 pp = {(_Float16)__builtin_amdgcn_exp2f(s[t][8 * kk + e]), (_Float16)__builtin_amdgcn_exp2f(s[t][8 * kk + e + 1])};
            p[kk][e] = pp[0]; p[kk][e + 1] = pp[1];
            if (kk == 0) l0 = __builtin_amdgcn_fdot2(pp, ones, l0, false); else l1 = __builtin_amdgcn_fdot2(pp, ones, l1, false);
          }
        x[2 + t] += l0 + l1;
        if (t == 0) { c0 = mfma(a, p[0], c0); c1 = mfma(a, p[0], c1); c0 = mfma(a, p[1], c0); c1 = mfma(a, p[1], c1); }
        else { c2 = mfma(a, p[0], c2); c3 = mfma(a, p[0], c3); c2 = mfma(a, p[1], c2); c3 = mfma(a, p[1], c3); }
      };
      if (MODE == 4) { qk(0); softmax_pv(0); qk(1); softmax_pv(1); }
      else { qk(0); qk(1); __builtin_amdgcn_sched_barrier(0); softmax_pv(0); softmax_pv(1); }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 16; ++i) acc += x[i];
  if (acc == 12345.678f) sink[0] = acc;
  if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

static _Float16* kvbuf = nullptr;
template <int MODE>
static void run(const char* name, unsigned long long* d, float* sink, int share = 5) {
  const int iters = MODE >= 6 && MODE <= 8 ? 1000 : 2000, nblk = 256;  // modes >= 6 run two key tiles per loop iteration: figures are per TWO tiles
  for (int wps = 1; wps <= (MODE >= 6 && MODE <= 8 ? 2 : 4); ++wps) {  // waves per SIMD (modes >= 6 need ~200 VGPRs: two at most)
    hipMemset(d, 0, nblk * 16 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256 * wps), 0, 0, d, sink, 10, kvbuf, share);  // warm
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256 * wps), 0, 0, d, sink, iters, kvbuf, share);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < nblk; ++b) for (int w = 0; w < 4 * wps; ++w) { s += (double)h[b * 16 + w]; ++n; }
    printf("%-22s %d wave(s)/SIMD: %8.1f ticks per iteration per wave   %8.2f ns per iteration per wave (wall)   %8.1f ns per iteration per SIMD\n", name, wps,
           s / n / iters, ms * 1e6 / iters, ms * 1e6 / iters / wps);
  }
}
int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 64);
  const size_t kvn = (size_t)1024 * 19 * 4096 + 32768;
  hipMalloc(&kvbuf, kvn * 2); hipMemset(kvbuf, 0x3c, kvn * 2);
  run<0>("0 mfma x16", d, sink);
  run<1>("1 valu x128", d, sink);
  run<2>("2 mfma x16 | valu x128", d, sink);
  run<3>("3 (mfma, valu x8) x16", d, sink);
  run<4>("4 attention-like", d, sink);
  run<5>("5 attention-like, QK first", d, sink);
  run<6>("6 = 5 + K/V loads (x2 tiles)", d, sink);
  run<7>("7 = 6 + vote / rescale (x2)", d, sink);
  run<8>("8 = 7 + ragged / first (x2)", d, sink);
  run<8>("8, no sharing across WGs", d, sink, 1);
  run<9>("9 winograd-like: 8 mfma | 32 v_pk_add_f16", d, sink);
  run<10>("10 = 9 + 56 v_add_f32 (output transform)", d, sink);
  {  // the same loops on random operands: what the DVFS power limit takes
    std::vector<_Float16> hkv(kvn);
    unsigned h = 1;
    for (size_t i = 0; i < kvn; ++i) { h = h * 1664525u + 1013904223u; hkv[i] = (_Float16)(((int)(h >> 8) & 0xffff) * (0.5f / 65536.f) - 0.25f); }
    hipMemcpy(kvbuf, hkv.data(), kvn * 2, hipMemcpyHostToDevice);
    run<0>("0 mfma x16, random", d, sink, -5);
    run<5>("5, random operands", d, sink, -5);
    run<8>("8, random operands", d, sink, -5);
    run<9>("9, random operands", d, sink, -5);
    run<10>("10, random operands", d, sink, -5);
  }
  return 0;
}
