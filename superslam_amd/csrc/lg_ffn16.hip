// k_lg_ffn16: the fused LightGlue FFN block (call sites src/LightGlue.cc:313,446) designed for FOUR waves per
// SIMD and ONE weight pass per 96-128 tokens per CU.  Same arithmetic as k_lg_ffn / k_lg_ffn4 (lg_kernels.hip):
//   x += ffn.3( GELU( LayerNorm( ffn.0( cat[x, ctx] ) ) ) ),  then the projection that consumes the new x (CrossBlock [to_qk|to_v],
//   the next layer's Wqkv with rotary, or final_proj + matchability), out_proj / to_out folded into ffn.0 on the host.
//
// Why another kernel (VERDICT r03 "do this" 1; measurements in profiles/NOTES_r01_r04_design_history.md items 13, 20 and the round-4 entry):
//   * k_lg_ffn4 streams the block's 1.0-1.15 MB of packed weights from L2 ONCE PER 64 TOKENS per workgroup.  The three GEMMs of a
//     64-token tile are 2 048-2 304 MFMAs = 16-18 k clocks of one CU's matrix pipes, and 1.15 MB at the 64 B/clk a CU's vector
//     L1 delivers are 18 k clocks as well: at 64 tokens per weight pass the weight stream alone caps the kernel at the matrix
//     rate, and two co-resident workgroups (each with its own pass) measured the same 27-34 B/clk/CU as one alone.
//   * its LayerNorm / GELU / epilogue phases run at ONE wave's VALU issue rate (one instruction per ~5 clocks; the SIMD could
//     retire one per 2): 256 VGPRs per wave leave room for two waves per SIMD only.
// Here a workgroup is 16 waves (1 024 threads, <= 128 VGPRs each, four per SIMD), ONE per CU, and owns a chunk of NT = 2..4
// 32-token N-tiles (up to 128 tokens, 133 KB of LDS):
//   * every wave owns ONE 32-row M-tile and ALL NT N-tiles of it (accumulators 16 NT registers): a weight fragment fetched from
//     L2 feeds NT MFMAs (256 B per MFMA at NT = 4, half of k_lg_ffn4's 512), and every token-tile fragment read from LDS feeds
//     one MFMA (4 LDS cycles per 8-clock MFMA slot of the CU: 50 %);
//   * ffn.0 (16 M-tiles) runs on all 16 waves; ffn.3 (8 M-tiles) and the third M-tile group of a 768-row projection run on
//     waves 0-7 = two per SIMD, which is enough for an MFMA-bound loop with its operands prefetched one k-step ahead;
//   * the VALU phases (statistics, LayerNorm + GELU, epilogues) run with four waves per SIMD;
//   * the persistent workgroup walks its contiguous range of N-tiles in chunks of 2-4 (9 -> 3+3+3, 10 -> 4+3+3): all CUs finish
//     together, no partly filled last round (1 216 64-token tiles on 512 workgroup slots ran 2.4 rounds).
// LDS-DMA staging, buffer-addressed weight loads / stores, GELU polynomial, bias-initialised accumulators and the fragment-order
// q / k / V^T epilogues are those of k_lg_ffn4.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "lg_ffn.h"

namespace sship {

constexpr int kF16MaxNT = 4;
constexpr int kF16Tile = kF16MaxNT * 32 * kFfnLd * 2;   // 133 120 B: [128 tokens][520 halfs]
constexpr int kF16Par = kF16Tile;                        // floats: b0 512 | gamma 512 | beta 512 | b3 256
constexpr int kF16Pb = kF16Par + 1792 * 4;               // bias of the fused projection, <= 768 floats
constexpr int kF16Red = kF16Pb + 768 * 4;                // [2][16 waves][128 tokens] floats: per-wave sums, sums of squares
constexpr int kF16Mr = kF16Red + 2 * 16 * 128 * 4;       // [128] float2: rstd, -mean rstd
constexpr int kF16Smem = kF16Mr + 128 * 8;               // 160 768 B
static_assert(kF16Smem <= 163840, "LDS budget");

#ifndef SSHIP_FFN16_TRACE_BUILD
#define SSHIP_FFN16_TRACE_BUILD 1
#endif

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: it
// would also wait for every global STORE the wave has in flight (the x / q / k / V^T epilogue stores: 1-2 us of write latency, paid at the
// next barrier by all 16 waves - 4.4 k clocks in the first version's "new x" phase) and for the residual operand it has just requested.
// Cross-wave hand-offs in this kernel go through LDS only; data dependences inside a wave are the compiler's own waitcnts.
__device__ __forceinline__ void bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NEXT_MT, bool HEADS>
__global__ __launch_bounds__(1024) void k_lg_ffn16(const _Float16* __restrict__ ctx, const _Float16* __restrict__ w0p,
                                                    const float* __restrict__ b0, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const _Float16* __restrict__ w3p,
                                                    const float* __restrict__ b3, _Float16* __restrict__ x, FfnTail tail) {
  extern __shared__ __attribute__((aligned(16))) char f16_smem[];
  _Float16* s_x = reinterpret_cast<_Float16*>(f16_smem);
  float* s_par = reinterpret_cast<float*>(f16_smem + kF16Par);
  float* s_pb = reinterpret_cast<float*>(f16_smem + kF16Pb);
  float* s_red = reinterpret_cast<float*>(f16_smem + kF16Red);
  float2* s_mr = reinterpret_cast<float2*>(f16_smem + kF16Mr);
  for (int i = threadIdx.x; i < 1792; i += 1024)
    s_par[i] = i < 512 ? b0[i] : i < 1024 ? gamma[i - 512] : i < 1536 ? beta[i - 1024] : b3[i - 1536];
  for (int i = threadIdx.x; i < NEXT_MT * 256; i += 1024) s_pb[i] = tail.proj.bias[i];

  // this workgroup's contiguous range of 32-token N-tiles, walked in balanced chunks of <= 4
  const int N32 = tail.ntiles, G = gridDim.x, bid = blockIdx.x;
  const int base = N32 / G, extra = N32 - base * G;
  const int first = bid * base + (bid < extra ? bid : extra), cnt = base + (bid < extra ? 1 : 0);
  if (cnt <= 0) return;
  const int nchunks = (cnt + kF16MaxNT - 1) / kF16MaxNT, csz = cnt / nchunks, crem = cnt - csz * nchunks;

  // token rows wave, wave + 16, ... of a chunk by LDS-DMA: lanes 0..31 fetch the 32 16-byte units of x[token], lanes 32..63 those
  // of ctx[token]; the row lands lane-linear at its padded LDS row
  auto stage_tile = [&](int t32, int nt) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* gsrc = reinterpret_cast<const char*>((lane < 32 ? x + lane * 8 : ctx + (lane - 32) * 8) + ((size_t)t32 * 32 + wave) * 256);
#pragma unroll
    for (int k = 0; k < 2 * kF16MaxNT; ++k) {
      if (k < 2 * nt) {
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_x + (wave + 16 * k) * kFfnLd));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
      }
      gsrc += 16 * 512;
    }
  };

  // The same by HALF rows, so that the next chunk can stream in while this one is still in use: a token row of the tile is
  // [x 512 B | ctx 512 B]; from the moment ffn.3 has read the hidden tile its ctx columns are dead (the projection reads the x columns only),
  // and the x columns die with the projection's last MFMA.  One instruction moves 32 lanes x 16 B: the ctx half with lanes 32..63
  // active (LDS address = M0 + lane * 16 = row base + 512 ..), the x half with lanes 0..31.  `part` of `parts` waves share the rows.
  // (all 256 CUs run their chunks in step: the 24-33 MB the chip's CUs fetch per chunk were fully exposed in the first version -
  // 12 us per chunk of a 106 us launch, profiles/r04_d_*)
  auto stage_half = [&](int t32, int nt, bool ctx_half, int part, int parts) {
    const int lane = threadIdx.x & 63;
    const int rows = 32 * nt;
    const bool mine = ctx_half ? lane >= 32 : lane < 32;
    const char* gsrc = reinterpret_cast<const char*>((ctx_half ? ctx + (lane - 32) * 8 : x + lane * 8) + ((size_t)t32 * 32 + part) * 256);
#pragma unroll 1
    for (int r = part; r < rows; r += parts) {
      const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_x + r * kFfnLd));
      if (mine) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
      }
      gsrc += (size_t)parts * 512;
    }
  };

  int t32 = first;
  stage_tile(t32, csz + (0 < crem ? 1 : 0));
  int pend = 0;

#pragma unroll 1
  for (int ci = 0; ci < nchunks; ++ci) {
    const int nt_rt = csz + (ci < crem ? 1 : 0);
    const bool has_next = ci + 1 < nchunks;
    const int t32_next = t32 + nt_rt, nt_next = csz + (ci + 1 < crem ? 1 : 0);
    // this wave's share of the chunk has landed (the DMA is invisible to hipcc's waitcnt bookkeeping).  Memory operations retire in
    // order: the `pend` epilogue stores this wave issued AFTER the DMA may stay in flight (waiting for their write
    // acknowledgements - vmcnt(0) - would put 1-2 us of store latency on every chunk)
    if (SSHIP_FFN16_TRACE_BUILD && tail.trace && ci == 1 && (threadIdx.x & 63) == 0)
      tail.trace[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + 12] = __builtin_readcyclecounter();
    if (pend == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (pend == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (pend == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar_lds();
    if (SSHIP_FFN16_TRACE_BUILD && tail.trace && ci == 1 && (threadIdx.x & 63) == 0)
      tail.trace[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + 13] = __builtin_readcyclecounter();
    {
      // stores the last-pass epilogue of THIS chunk will issue behind the next chunk's DMA: 2 per N-tile for a wave that owns an M-tile there
      constexpr int NMl = 8 * NEXT_MT, last0 = ((NMl + 15) / 16 - 1) * 16;
      const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
      pend = (HEADS && has_next && last0 + wv < NMl) ? 2 * nt_rt : 0;
    }

    auto body = [&](auto nt_c) __attribute__((always_inline)) {
      constexpr int NT = decltype(nt_c)::value;
      const size_t t0 = (size_t)t32 * 32;
      // opaque zero / thread id: keeps the loop-invariant weight / parameter loads and the lane address arithmetic inside the iteration (see k_lg_ffn)
      int zero = 0;
      asm volatile("" : "+s"(zero));
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, hh = lane >> 5;
      const _Float16* bfp = s_x + j * kFfnLd + hh * 8;  // B fragment of N-tile n, k-step ks: bfp + n*32*kFfnLd + ks*16
      typedef unsigned wq_t __attribute__((ext_vector_type(4)));
      const unsigned lane16 = (unsigned)lane * 16u;
      auto wres = [&](const void* basep) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(basep), 0, (int)0x7ffffff0, 0x00020000);
      };
      auto wload = [&](__amdgpu_buffer_rsrc_t r, int halfs) __attribute__((always_inline)) {  // fragment at base + halfs (+ lane * 8)
        return __builtin_bit_cast(h8_t, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, halfs * 2 + zero, 0));
      };
      // (the scalar offset goes through an empty asm: hipcc 7.2 gives raw buffer STORES whose soffsets differ by a constant one
      // soffset register and drops the constant - lg_kernels.hip)
      auto wstore = [&](__amdgpu_buffer_rsrc_t r, unsigned voff, long long halfs, wq_t v) __attribute__((always_inline)) {
        int so = (int)(halfs * 2);
        asm volatile("" : "+s"(so));
        __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, so, 0);
      };
      auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (SSHIP_FFN16_TRACE_BUILD && tail.trace && ci == 1 && lane == 0)
          tail.trace[((size_t)blockIdx.x * 16 + wave) * 16 + slot] = __builtin_readcyclecounter();
      };
      // One GEMM loop for the three layers: acc[n] (+)= W[M-tile][K] . tile[n][K], K = 16 KSTEPS.  The wave's weight fragments come
      // from L2 through a ring of four k-steps of registers (the load of k-step ks + 3 is issued before the MFMAs of k-step ks), the
      // token-tile fragments from LDS (see below).  SWAP: operands exchanged (A = tokens, B = weights) -> D[token][channel],
      // the orientation the V^T epilogue wants.
      auto gemm = [&](auto ksteps_c, auto swap_c, f16x_t (&acc)[NT], __amdgpu_buffer_rsrc_t r, int off0, int kstride) __attribute__((always_inline)) {
        constexpr int KSTEPS = decltype(ksteps_c)::value;
        constexpr bool SWAP = decltype(swap_c)::value;
        h8_t a[4], bf[NT];
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = wload(r, off0 + i * kstride);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          if (ks + 3 < KSTEPS) a[(ks + 3) & 3] = wload(r, off0 + (ks + 3) * kstride);
          __builtin_amdgcn_sched_barrier(0);  // the weight load above stays ABOVE this k-step's MFMAs
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (SWAP) acc[n] = mfma32(bf[n], a[ks & 3], acc[n]);
            else acc[n] = mfma32(a[ks & 3], bf[n], acc[n]);
            // the next k-step's fragment of this N-tile goes into the SAME registers right behind the MFMA that read them: NT MFMAs
            // (>= 64-128 clocks of this wave's matrix work, more with three other waves on the SIMD) cover the LDS latency, and the
            // fragments cost 4 NT registers instead of 8 NT (128 registers per wave is the budget that buys four waves per SIMD)
            if (ks + 1 < KSTEPS) bf[n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd + (ks + 1) * 16);
          }
        }
      };
      const std::integral_constant<int, 32> k32{};
      const std::integral_constant<int, 16> k16{};
      const std::false_type no_swap{};
      const std::true_type do_swap{};
      stamp(0);

      // ---- ffn.0: M-tile `wave` (rows 32 wave .. + 31) x NT N-tiles, K = 512; packed [cb = wave / 2][k16][mt = wave % 2][lane][8] ----
      // the accumulators start from the bias (row 8 g + 4 hh + e of the M-tile is register 4 g + e)
      f16x_t acc[NT];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(s_par + zero + wave * 32 + hh * 4 + g * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc[n][4 * g + 0] = bv.x; acc[n][4 * g + 1] = bv.y; acc[n][4 * g + 2] = bv.z; acc[n][4 * g + 3] = bv.w; }
      }
      gemm(k32, no_swap, acc, wres(w0p), ((wave >> 1) * 64 + (wave & 1)) * 512, 2 * 512);
      stamp(1);

      // ---- LayerNorm(512) statistics: registers -> lane ^ 32 -> the 16 waves through LDS -> one thread per token ----
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          sm += (acc[n][4 * g + 0] + acc[n][4 * g + 1]) + (acc[n][4 * g + 2] + acc[n][4 * g + 3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) sq = fmaf(acc[n][4 * g + e], acc[n][4 * g + e], sq);
        }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (hh == 0) { s_red[wave * 128 + n * 32 + j] = sm; s_red[(16 + wave) * 128 + n * 32 + j] = sq; }
      }
      bar_lds();  // partial statistics complete; every wave has finished reading the input tile
      if (tid < NT * 32) {
        float t = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { t += s_red[w * 128 + tid]; q += s_red[(16 + w) * 128 + tid]; }
        const float mean = t * (1.0f / 512.0f);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(q * (1.0f / 512.0f) - mean * mean, 0.f) + 1e-5f);
        s_mr[tid] = make_float2(rstd, -mean * rstd);  // LayerNorm as two fmas per value: (a rstd - mean rstd) gamma + beta
      }
      bar_lds();
      stamp(2);

      // ---- LayerNorm + GELU on the accumulators; the hidden tile overwrites the input tile ----
      {
        float2 mr[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) mr[n] = s_mr[n * 32 + j];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = wave * 32 + hh * 4 + g * 8;
          const float4 gv = *reinterpret_cast<const float4*>(s_par + zero + 512 + c);
          const float4 be = *reinterpret_cast<const float4*>(s_par + zero + 1024 + c);
          const f2_t g01 = {gv.x, gv.y}, g23 = {gv.z, gv.w}, b01 = {be.x, be.y}, b23 = {be.z, be.w};
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const f2_t a01 = {acc[n][4 * g + 0], acc[n][4 * g + 1]}, a23 = {acc[n][4 * g + 2], acc[n][4 * g + 3]};
            const f2_t o01 = gelu2((a01 * mr[n].x + mr[n].y) * g01 + b01);
            const f2_t o23 = gelu2((a23 * mr[n].x + mr[n].y) * g23 + b23);
            *reinterpret_cast<h4_t*>(s_x + (n * 32 + j) * kFfnLd + c) = to_h4(o01[0], o01[1], o23[0], o23[1]);
          }
        }
      }
      stamp(3);
      // the residual operand of ffn.3's rows comes back from global memory (the x half of the LDS tile is the hidden tile now);
      // requested here, consumed after the barrier - the ffn.0 accumulators are dead, the registers are free
      const bool low = wave < 8;  // waves 0-7 (two per SIMD) own ffn.3's eight M-tiles
      h4_t xr[4][NT];
      if (low) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            xr[g][n] = *reinterpret_cast<const h4_t*>(x + (t0 + n * 32 + j) * 256 + wave * 32 + hh * 4 + g * 8 + zero);
      }
      bar_lds();  // hidden tile complete
      stamp(4);

      // ---- ffn.3: M-tile `wave` (waves 0-7) x NT N-tiles, K = 512; packed [cb = wave][k16][lane][8]; accumulators start from x + b3 ----
      f16x_t ac2[NT];
      if (low) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(s_par + zero + 1536 + wave * 32 + hh * 4 + g * 8);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const h4_t o = xr[g][n];
            ac2[n][4 * g + 0] = (float)o[0] + bv.x; ac2[n][4 * g + 1] = (float)o[1] + bv.y;
            ac2[n][4 * g + 2] = (float)o[2] + bv.z; ac2[n][4 * g + 3] = (float)o[3] + bv.w;
          }
        }
        gemm(k32, no_swap, ac2, wres(w3p), wave * 32 * 512, 512);
      }
      stamp(5);
      bar_lds();  // all waves are done reading the hidden tile: its x columns take the new x
      // ... and its ctx columns the next chunk's ctx rows: waves 8-15 have nothing else to do until the projection
      if (HEADS && has_next && !low) stage_half(t32_next, nt_next, true, wave - 8, 8);
      if (low) {
        // lane (j, hh) holds channels 4 hh + 8 g .. + 3 of token j; v_permlane32_swap pairs the two half-waves' quads so that every
        // lane writes whole 8-channel (16-byte) units to global memory and to the LDS tile
        const __amdgpu_buffer_rsrc_t rx = wres(x);
        const unsigned xrow16 = (unsigned)j * 512u + (unsigned)hh * 16u;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          unsigned lo[4], hi[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const h2_t p01 = {(_Float16)ac2[n][4 * g + 0], (_Float16)ac2[n][4 * g + 1]};
            const h2_t p23 = {(_Float16)ac2[n][4 * g + 2], (_Float16)ac2[n][4 * g + 3]};
            lo[g] = __builtin_bit_cast(unsigned, p01);
            hi[g] = __builtin_bit_cast(unsigned, p23);
          }
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(lo[2 * gp], lo[2 * gp + 1], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(hi[2 * gp], hi[2 * gp + 1], false, false);
            const int c = wave * 32 + (2 * gp + hh) * 8;
            wstore(rx, xrow16, (long long)(t0 + n * 32) * 256 + wave * 32 + 2 * gp * 8, wq_t{s0[0], s1[0], s0[1], s1[1]});
            *reinterpret_cast<uint4*>(s_x + (n * 32 + j) * kFfnLd + c) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
        }
      }
      bar_lds();  // new x in LDS
      stamp(6);

      // ---- fused next projection: 8 NEXT_MT M-tiles, 16 per pass, K = 256.  Tile-interleaved packing (upload_conv): block cb holds
      // M-tile mt = rows (8 mt + cb) * 32 .. + 31, i.e. M-tile T = 8 mt + cb is segment mt (q | k | v, or qk | v); final_proj
      // (!HEADS) is packed plainly, T = cb.  The V segment runs with swapped operands (see k_lg_ffn). ----
      constexpr int NM = 8 * NEXT_MT, NPASS = (NM + 15) / 16;
      const int NP = tail.proj.np, nt32 = NP >> 5;
      const int rope_segs = tail.proj.flags & 0xf;
      const __amdgpu_buffer_rsrc_t rp = wres(tail.proj.wpack);
      const __amdgpu_buffer_rsrc_t rrope = wres(tail.proj.aux);
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int T = pass * 16 + wave;
        const bool active = T < NM;
        const int cb = HEADS ? (T & 7) : T, mt = HEADS ? (T >> 3) : 0;
        const int R0 = T * 32;                      // first output row of this M-tile
        const bool is_v = HEADS && mt == NEXT_MT - 1;
        const bool roped = HEADS && mt < rope_segs;
        f16x_t ac3[NT];
        float4 cs[4];                               // rotary (cos, sin) quads of the N-tile in the epilogue
        auto rope_load = [&](int n) __attribute__((always_inline)) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            cs[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                rrope, (unsigned)j * 256u + (unsigned)hh * 16u, (int)(((t0 + n * 32) * 64 + (R0 & 63) + g * 8) * 4) + zero, 0));
        };
        if (active) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) ac3[n][r] = 0.f;
          const int off0 = (cb * 16 * NEXT_MT + mt) * 512;
          if (is_v) gemm(k16, do_swap, ac3, rp, off0, NEXT_MT * 512);
          else gemm(k16, no_swap, ac3, rp, off0, NEXT_MT * 512);
          if (roped) rope_load(0);  // after the loop: 16 more live registers inside it would not fit the 128-register budget
        }
        stamp(7 + 2 * pass);
        if (pass == NPASS - 1) {
          bar_lds();  // every wave has finished reading the tile
          // q / k / V^T epilogues read no LDS tile data: the next chunk streams in behind them
          if (HEADS && has_next) stage_half(t32_next, nt_next, false, wave, 16);
        }
        if (active) {
          if constexpr (HEADS) {
            const int hd = (R0 >> 6) & 3;
            if (is_v) {
              const int mth = (R0 >> 5) & 1;  // 32-channel half of the head
              const float bv = s_pb[R0 + j];
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                const size_t token = t0 + n * 32;
                const int sq = (int)(token / NP), kt = (int)(token - (size_t)sq * NP) >> 5;
                const long long dst = (((long long)sq * 4 + hd) * nt32 + kt) * 2048;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                  h8_t o;
#pragma unroll
                  for (int e = 0; e < 8; ++e) o[e] = (_Float16)(ac3[n][8 * kk + e] + bv);
                  *reinterpret_cast<h8_t*>(static_cast<_Float16*>(tail.proj.out2) + dst + lane * 8 + (kk * 2 + mth) * 512) = o;
                }
              }
            } else {
              // q / k (or the shared qk of CrossBlock): bias, rotary on interleaved pairs, fp16, then lane ^ 32 pairing so that every
              // lane owns whole 16-byte fragment units: unit u = d / 8 -> [kstep u / 2][lane' = (u & 1) * 32 + token % 32][8]
              const __amdgpu_buffer_rsrc_t rqk = wres(mt == 0 ? tail.proj.out0 : tail.proj.out1);
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                const size_t token = t0 + n * 32;
                const int sq = (int)(token / NP), kt = (int)(token - (size_t)sq * NP) >> 5;
                const long long dst = (((long long)sq * 4 + hd) * nt32 + kt) * 2048;
                unsigned lo[4], hi[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                  const float4 bv = *reinterpret_cast<const float4*>(s_pb + R0 + hh * 4 + g * 8);
                  float v0 = ac3[n][4 * g + 0] + bv.x, v1 = ac3[n][4 * g + 1] + bv.y;
                  float v2 = ac3[n][4 * g + 2] + bv.z, v3 = ac3[n][4 * g + 3] + bv.w;
                  if (roped) {
                    const float4 c = cs[g];
                    const float r0 = v0 * c.x - v1 * c.y, r1 = v1 * c.x + v0 * c.y;
                    const float r2 = v2 * c.z - v3 * c.w, r3 = v3 * c.z + v2 * c.w;
                    v0 = r0; v1 = r1; v2 = r2; v3 = r3;
                  }
                  const h2_t p01 = {(_Float16)v0, (_Float16)v1}, p23 = {(_Float16)v2, (_Float16)v3};
                  lo[g] = __builtin_bit_cast(unsigned, p01);
                  hi[g] = __builtin_bit_cast(unsigned, p23);
                }
                if (roped && n + 1 < NT) rope_load(n + 1);  // the next N-tile's table goes out BEFORE this one's stores (in-order vmcnt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                  const auto s0 = __builtin_amdgcn_permlane32_swap(lo[2 * gp], lo[2 * gp + 1], false, false);
                  const auto s1 = __builtin_amdgcn_permlane32_swap(hi[2 * gp], hi[2 * gp + 1], false, false);
                  wstore(rqk, lane16, dst + ((((R0 & 63) >> 3) + 2 * gp) >> 1) * 512, wq_t{s0[0], s1[0], s0[1], s1[1]});
                }
              }
            }
          } else {
            IgemmArgs pj = tail.proj;
            f16x_t (&one)[1][NT] = *reinterpret_cast<f16x_t (*)[1][NT]>(&ac3);
            EpiF16<false, false>::template run<1, NT>(pj, one, 0, (int)(t0 >> 5), j, R0, hh);
          }
        }
        stamp(8 + 2 * pass);
      }
      if constexpr (!HEADS) {
        if (tail.logsig) {  // matchability head of the last block: 2 NT tokens per wave
          const float* mwq = tail.match_w + zero;
#pragma unroll 1
          for (int tk = wave * 2 * NT; tk < (wave + 1) * 2 * NT; ++tk) {
            const h4_t v = *reinterpret_cast<const h4_t*>(s_x + tk * kFfnLd + lane * 4);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) d += (float)v[e] * mwq[lane * 4 + e];
            const float z = wave_sum(d) + tail.match_b;
            if (lane == 0) tail.logsig[t0 + tk] = fminf(z, 0.f) - log1pf(expf(-fabsf(z)));
          }
        }
        if (has_next) {
          bar_lds();
          stage_tile(t32_next, nt_next);
        }
      }
      stamp(11);
    };
    if (nt_rt == 4) body(std::integral_constant<int, 4>{});
    else if (nt_rt == 3) body(std::integral_constant<int, 3>{});
    else body(std::integral_constant<int, 2>{});
    t32 = t32_next;
  }
}

template <int NEXT_MT, bool HEADS, typename... A>
static hipError_t launch_ffn16_t(int n32, int grid, hipStream_t s, A... args) {
  auto kern = k_lg_ffn16<NEXT_MT, HEADS>;
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kF16Smem);
  if (attr_rc != hipSuccess) return attr_rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), kF16Smem, s, args...);
  return hipGetLastError();
}

// SSHIP_FFN_TRACE=1: mean shader-clock duration of every phase of a workgroup's second chunk
static void ffn16_trace_report(unsigned long long* dev, int nwg, int next_mt, hipStream_t s) {
  std::vector<unsigned long long> h((size_t)nwg * 16 * 16);
  (void)hipStreamSynchronize(s);
  (void)hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
  // stamps: 0 start, 1 ffn.0, 2 stats (two barriers), 3 GELU, 4 x request + barrier, 5 ffn.3 (waves 0-7), 6 barrier + new x + barrier,
  // 7 proj pass 0 MFMA, 8 epilogue 0 (NPASS = 1: after the barrier + next-chunk DMA issue), 9 pass 1 MFMA, 10 epilogue 1, 11 end
  static const char* names[11] = {"ffn.0", "stats", "GELU", "xreq+barrier", "ffn.3", "barrier+newx+barrier", "proj0 MFMA", "epi0", "proj1 MFMA", "epi1", "tail"};
  double sum[2][11] = {{0}}, wait[2] = {0, 0}; long cnt[2] = {0, 0};
  for (int w = 0; w < nwg * 16; ++w) {
    const unsigned long long* t = h.data() + (size_t)w * 16;
    if (!t[0] || !t[11]) continue;
    const int grp = (w & 15) < 8 ? 0 : 1;
    unsigned long long prev = t[0];
    for (int i = 0; i < 11; ++i) {
      unsigned long long b = t[i + 1];
      if (!b) b = prev;  // phases a variant does not have
      sum[grp][i] += (double)(b > prev ? b - prev : 0);
      prev = b;
    }
    wait[grp] += (double)(t[13] > t[12] ? t[13] - t[12] : 0);
    ++cnt[grp];
  }
  for (int grp = 0; grp < 2; ++grp) {
    if (!cnt[grp]) continue;
    fprintf(stderr, "[ffn16 trace next_mt=%d, waves %s, %ld waves]", next_mt, grp ? "8-15" : "0-7", cnt[grp]);
    double tot = 0;
    for (int i = 0; i < 11; ++i) { fprintf(stderr, " %s=%.0f", names[i], sum[grp][i] / cnt[grp]); tot += sum[grp][i] / cnt[grp]; }
    fprintf(stderr, " | chunk=%.0f clk | wait for the chunk's DMA + barrier at the loop top=%.0f\n", tot, wait[grp] / cnt[grp]);
  }
}

// the 16-wave kernel needs at least two N-tiles per workgroup and addresses x / q / k through 32-bit buffer offsets
bool ffn16_applicable(int tokens, int next_mt, bool heads) {
  if (tokens % 32 != 0 || (size_t)tokens * 512 >= 0x7f000000ull) return false;
  if (!((heads && (next_mt == 2 || next_mt == 3)) || (!heads && next_mt == 1))) return false;
  return tokens / 32 >= 2;
}

hipError_t launch_lg_ffn16(int tokens, int next_mt, bool heads, hipStream_t s, const _Float16* ctx, const _Float16* w0p, const float* b0,
                           const float* gamma, const float* beta, const _Float16* w3p, const float* b3, _Float16* x, FfnTail t) {
  const int n32 = tokens / 32;
  const int grid = std::min(cu_count(), n32 / 2);
  t.ntiles = n32;
  static const bool trace_on = dev_env("SSHIP_FFN_TRACE") != nullptr;
  static unsigned long long* trace_buf = nullptr;
  if (trace_on) {
    const size_t bytes = (size_t)cu_count() * 16 * 16 * 8;
    if (!trace_buf) (void)hipMalloc(&trace_buf, bytes);
    (void)hipMemsetAsync(trace_buf, 0, bytes, s);
    t.trace = trace_buf;
  }
  hipError_t rc;
  if (heads && next_mt == 3) rc = launch_ffn16_t<3, true>(n32, grid, s, ctx, w0p, b0, gamma, beta, w3p, b3, x, t);
  else if (heads && next_mt == 2) rc = launch_ffn16_t<2, true>(n32, grid, s, ctx, w0p, b0, gamma, beta, w3p, b3, x, t);
  else rc = launch_ffn16_t<1, false>(n32, grid, s, ctx, w0p, b0, gamma, beta, w3p, b3, x, t);
  if (trace_on && rc == hipSuccess) ffn16_trace_report(trace_buf, grid, next_mt, s);
  return rc;
}

}  // namespace sship
