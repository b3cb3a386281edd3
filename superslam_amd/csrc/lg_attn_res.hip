// LightGlue attention for throughput batches with the keys RESIDENT in LDS (gfx950, wave64).
//
// STATUS: developer build only (build.py DEV_SOURCES; SUPERSLAM_HIP_ATTN=res).  Built for VERDICT r04 "do this" 1, bit-identical to the shipped
// kernel, MEASURED 4-12 % SLOWER (profiles/r05_a_attention_lds_resident_keys_rejected.txt): with every operand in LDS the key loop runs at the
// same ~575 clocks per 32 x 32 score tile as with operands streamed from L2 (it is VALU-issue bound), and the resident form adds a fill that
// cannot overlap the previous workgroup (9.5-16.6 k clocks), a second single-tile pass (19 query tiles on 8 waves) and loses CU sharing with the
// other half-batch's FFN.  The premise in the next paragraphs is the hypothesis the kernel was built to test, kept as written.
//
// What it computes is k_lg_attention<2, 1, 3> (lg_kernels.hip) instruction for instruction: flash-style attention over
// head_dim 64, S^T = K Q^T "swapped" so a lane owns one query column, the reference exponent riding in the QK^T MFMA
// chain, P in fp16, key tiles visited in ascending order by ONE wave per query tile - so the context rows are bit-identical
// to that kernel's (tests/test_gpu_alt_paths.py compares them).  SelfBlock / CrossBlock arithmetic: SURVEY 8(a)-LG; the call
// it sits behind: /root/reference/src/LightGlue.cc:377-457 (one enqueue of the LightGlue engine).
//
// What changed is where the K / V^T fragments come from.  The streaming kernel gives every wave (64 queries) its own pass over
// its head's K and V^T: 10 waves x 152 KB per (sequence, head), 790 MB of L2 -> register traffic per 64-pair launch, one 8-KB
// tile in flight per wave - 1 260 SIMD clocks per 32 x 32 score tile against 288 of MFMA and ~260 of softmax VALU
// (profiles/r04_x_*: 120 k of the launch's 154 k clocks remain with the MFMAs and the exponentials compiled out).
// Here ONE workgroup owns a whole (sequence, head): its 8 waves bring the head's K and V^T (19 tiles x 8 KB at 600 keypoints =
// 152 of the CU's 160 KB) into LDS exactly once, by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass - the
// fragment-ordered global image IS the LDS image), and all of them read fragments by ds_read_b128.  L2 -> CU traffic / 10, the
// fragment latency drops from ~2 k clocks of loaded L2 to ~100 of LDS, and there is NO barrier in the key loop: the fill is
// one-shot (no ring, nothing is overwritten), waves only meet at the few points of the first sweep where a group of four
// tiles has to have landed (counted vmcnt + raw s_barrier - a __syncthreads() would drain the whole DMA queue).
//
// Work split: wave w takes query tiles qlo + w + 8 i.  Two tiles at a time share every fragment read (halves the LDS
// traffic: 3.1 MB per CU and launch, a quarter of the LDS pipe); a third tile (19 = 8 + 8 + 3) is a second, single-tile pass
// over the already resident keys.  More than 19 key tiles (max_keypoints 1024: 32) run as two query halves per (sequence, head)
// and key "epochs" of <= 19 tiles: the online-softmax state simply carries over from one epoch to the next.
#include "kernels.h"

#include <type_traits>
#include <vector>

// developer aid: -DSSHIP_ATTN_RES_TRACE=1 + SSHIP_ATTN_TRACE=1 prints, per wave index, the mean shader clocks of a workgroup's phases
#ifndef SSHIP_ATTN_RES_TRACE
#define SSHIP_ATTN_RES_TRACE 0
#endif

namespace sship {
namespace {

constexpr int kResCapTiles = 19;  // key tiles resident at a time: 19 x (4 KB K + 4 KB V^T) = 152 KB
#ifndef SSHIP_RES_AHEAD
#define SSHIP_RES_AHEAD 1
#endif
constexpr int kResAhead = SSHIP_RES_AHEAD;  // DMA rounds (of four tiles) in flight ahead of the stage being computed; 1..4 (A/B: -DSSHIP_RES_AHEAD=n; 5 = everything at once)

__device__ __forceinline__ float res_max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float res_max_xor32(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return res_max3f(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[1]));
}

// counted wait on this wave's own DMA / loads; n is a multiple of 4 (one DMA group = four 1-KB instructions)
__device__ __forceinline__ void res_wait_vm(int n) {
  switch (n) {
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
__global__ __launch_bounds__(512, 2) void k_lg_attention_res(const _Float16* __restrict__ q, const _Float16* __restrict__ k,
                                                             const _Float16* __restrict__ vt, const int* __restrict__ lens, int NP,
                                                             int cross, _Float16* __restrict__ ctx, int n_wg, int nqsplit, int qper,
                                                             int ept, unsigned long long* __restrict__ trace) {
  unsigned long long tr[6] = {0, 0, 0, 0, 0, 0};  // 0 entry, 1 first stage landed, 2 end of the staged sweep (pass 0), 3 pass 0 stored, 4 end of the kernel; 5 = barrier wait clocks of the sweep
  if (SSHIP_ATTN_RES_TRACE && trace) tr[0] = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) char smem_res[];
  _Float16* sK = reinterpret_cast<_Float16*>(smem_res);  // [ept][4 k-steps][64 lanes][8]   (the global fragment image)
  _Float16* sV = sK + (size_t)ept * 2048;                // [ept][2 kk][2 mt][64 lanes][8]
  // XCD-aware mapping (see k_lg_attention): XCD x runs the consecutive logical workgroups [x n/8, (x+1) n/8) - the two query halves of
  // a (sequence, head), its other heads, then the partner sequence whose K / V^T the cross block reads - all through one L2.
  const int per_xcd = (n_wg + 7) >> 3;
  const int L = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (L >= n_wg) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int qs = L % nqsplit, h = (L / nqsplit) & 3, s = L / (nqsplit * 4);
  const int sk = cross ? (s ^ 1) : s;
  const int nq = min(max(lens[s], 0), NP), nk = min(max(lens[sk], 0), NP);  // device-side counts are clamped to capacity
  const int nt32 = NP >> 5;
  const int nqt = (nq + 31) >> 5, ntk = (nk + 31) >> 5;
  const int qlo = qs * qper, qhi = min(nqt, qlo + qper);
  if (qlo >= qhi) return;  // uniform: no query tile for this workgroup
  const _Float16* Q = q + ((size_t)(s * 4 + h) * nt32) * 2048;
  const _Float16* K = k + ((size_t)(sk * 4 + h) * nt32) * 2048;
  const _Float16* VT = vt + ((size_t)(sk * 4 + h) * nt32) * 2048;
  const unsigned lane16 = lane * 16;
  const unsigned lds_k = (unsigned)(uintptr_t)sK, lds_v = (unsigned)(uintptr_t)sV;
  const int nep = (ntk + ept - 1) / ept;           // key epochs actually needed by this sequence's key count

  // ---- DMA of epoch e (ne tiles): group g = 2 * tile + (0: K, 1: V^T), 4 KB each; in round r wave w moves group 8 r + w, so that round r
  // completes tiles 4 r .. 4 r + 3.  Rounds are issued kResAhead ahead of the stage that reads them, NOT all at once: with every CU of the
  // launch asking for its whole 152 KB at the same moment the memory system delivered the FIRST four tiles of a workgroup after 16 k clocks -
  // about when it delivered the last ones (profiles/r05_b_*: 39 MB in flight at ~5 TB/s) - and nothing was computed meanwhile.
  auto fill_round = [&](int e, int ne, int r) __attribute__((always_inline)) {
    const int g = 8 * r + wave;
    if (g >= 2 * ne) return;
    const int tl = g >> 1, isv = g & 1;
    const unsigned long long ga = (unsigned long long)(uintptr_t)((isv ? VT : K) + (size_t)(e * ept + tl) * 2048);
    const unsigned long long gs = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ga) |
                                  ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ga >> 32)) << 32);
    const unsigned dst = __builtin_amdgcn_readfirstlane((isv ? lds_v : lds_k) + (unsigned)tl * 4096u);
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(dst), "s"(gs) : "memory");
  };

  // ---- per-pass state: up to two query tiles of this wave ----
  h8_t qf[2][4];
  float m[2], l[2];
  f16x_t o[2][2];
  h8_t rf[2];
  h8_t ones_k0;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones_k0[e] = (_Float16)0.f;
  if (hh == 0) { ones_k0[0] = (_Float16)256.f; ones_k0[1] = (_Float16)1.f; }
  const h2_t ones2 = {(_Float16)1.f, (_Float16)1.f};

  // K / V^T fragments of the resident tile tl: ds_read_b128, lane-linear (conflict-free).  ONE register set each: LDS answers in
  // ~100 clocks, so the next tile's K fragments are requested right after this tile's QK^T MFMAs have been issued (they arrive
  // behind the softmax) and this tile's V^T fragments at the top of the tile (they arrive behind QK^T) - the streaming kernel
  // needs a second set (32 VGPRs) to keep a whole tile in flight across ~2 k clocks of L2.
  h8_t kf[4], vf[2][2];
  auto fetch_k = [&](int tl) __attribute__((always_inline)) {
    const _Float16* kp_ = sK + tl * 2048 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const h8_t*>(kp_ + ks * 512);
  };
  auto fetch_v = [&](int tl) __attribute__((always_inline)) {
    const _Float16* vp_ = sV + tl * 2048 + lane * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) vf[kk][mt] = *reinterpret_cast<const h8_t*>(vp_ + (kk * 2 + mt) * 512);
  };
  // resident key tiles [lo, hi) of epoch e for the NQ query tiles of this wave: per tile k_lg_attention's V = 3 body
  auto run_range = [&](auto nq_c, int e, int lo, int hi) __attribute__((always_inline)) {
    constexpr int NQ = decltype(nq_c)::value;
    fetch_k(lo);
    for (int tl = lo; tl < hi; ++tl) {
      const int kt = e * ept + tl, k0 = kt * 32;
      fetch_v(tl);
      const f16x_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f16x_t stq[NQ];
#pragma unroll
      for (int t = 0; t < NQ; ++t) stq[t] = mfma32(ones_k0, rf[t], zero16);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < NQ; ++t) stq[t] = mfma32(kf[ks], qf[t][ks], stq[t]);
      __builtin_amdgcn_sched_barrier(0);  // the QK^T MFMAs of all query tiles stay ahead of the first tile's softmax ...
      fetch_k(min(tl + 1, hi - 1));       // ... and of the reads that refill their K operands
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        f16x_t st = stq[t];
        if (k0 + 32 > nk) {  // only the last (ragged) key tile needs masking - wave-uniform branch
          int kb = k0 + 4 * hh;
          asm volatile("" : "+v"(kb));
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kb + (r & 3) + 8 * (r >> 2) >= nk) st[r] = -INFINITY;
        }
        float tmax = res_max3f(st[0], st[1], st[2]);
        tmax = res_max3f(tmax, st[3], st[4]);
#pragma unroll
        for (int r = 5; r < 15; r += 2) tmax = res_max3f(tmax, st[r], st[r + 1]);
        // st = s - r.  First key tile: r <- fp16 pair of the tile maximum; later: only when the tile's maximum exceeds r by more than 8
        tmax = res_max_xor32(res_max3f(tmax, st[15], st[15]));
        const bool first = kt == 0;
        const bool need = first || tmax > 8.0f;
        if (__any(need)) {
          const float r_tgt = m[t] + tmax;
          const float ra = fabsf(r_tgt) < 2048.f ? 0.f : (float)(_Float16)(fminf(fmaxf(r_tgt * (1.0f / 256.0f), -65000.f), 65000.f));
          const float rb = (float)(_Float16)(r_tgt - 256.0f * ra);
          const float r_new = need ? 256.0f * ra + rb : m[t];
          const float d = r_new - m[t];
          const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
          l[t] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[t][0][r] *= alpha; o[t][1][r] *= alpha; st[r] -= d; }
          m[t] = r_new;
          if (need && hh == 0) { rf[t][0] = (_Float16)(-ra); rf[t][1] = (_Float16)(-rb); }
        }
        float ls0 = 0.f, ls1 = 0.f;
        h8_t pb[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int e2 = 0; e2 < 8; e2 += 2) {
            const h2_t pp = {(_Float16)__builtin_amdgcn_exp2f(st[8 * kk + e2]), (_Float16)__builtin_amdgcn_exp2f(st[8 * kk + e2 + 1])};
            pb[kk][e2] = pp[0]; pb[kk][e2 + 1] = pp[1];
            if (kk == 0) ls0 = __builtin_amdgcn_fdot2(pp, ones2, ls0, false);
            else ls1 = __builtin_amdgcn_fdot2(pp, ones2, ls1, false);
          }
        l[t] += ls0 + ls1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) o[t][mt] = mfma32(vf[kk][mt], pb[kk], o[t][mt]);
      }
    }
  };

  // One pass = NQ query tiles per wave (ta and, at NQ = 2, ta + 8) against every key tile.  `valid` is wave-uniform: a wave without a tile
  // in this pass still moves its share of the DMA and meets the others at the barriers, its key loops run zero times.
  // pidx counts the passes of this workgroup: a single epoch stays resident after the first one.
  auto do_pass = [&](auto nq_c, int ta, bool valid, int pidx) __attribute__((always_inline)) {
    constexpr int NQ = decltype(nq_c)::value;
    // Q fragments by asm loads (hipcc must not count them: its own vmcnt(0) in front of their first use would drain the DMA queue)
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      const unsigned long long ga = (unsigned long long)(uintptr_t)(Q + (size_t)min(ta + 8 * t, nt32 - 1) * 2048);
      const unsigned long long gs = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ga) |
                                    ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ga >> 32)) << 32);
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %4, %5\n\t"
                   "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                   "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                   "global_load_dwordx4 %3, %4, %5 offset:3072"
                   : "=&v"(qf[t][0]), "=&v"(qf[t][1]), "=&v"(qf[t][2]), "=&v"(qf[t][3]) : "v"(lane16), "s"(gs) : "memory");
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      m[t] = 0.f; l[t] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) rf[t][e] = (_Float16)0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[t][0][r] = 0.f; o[t][1][r] = 0.f; }
    }
    auto pin_q = [&]() __attribute__((always_inline)) {  // every consumer of the Q fragments is ordered behind the wait in front of this
      if constexpr (NQ == 2) asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3]) :: "memory");
      else asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]) :: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    const bool resident = nep == 1 && pidx > 0;  // a single epoch stays in LDS for the later passes
    if (resident || nep == 0) { res_wait_vm(0); pin_q(); }
    for (int e = 0; e < nep; ++e) {
      const int ne = min(ept, ntk - e * ept);
      int nst = 1;
      if (!resident) {
        if (pidx > 0 || e > 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the old content
        nst = (2 * ne + 7) >> 3;  // DMA rounds = stages of four tiles
        for (int r = 0; r < min(nst, kResAhead); ++r) fill_round(e, ne, r);
      }
      for (int st = 0; st < nst; ++st) {
        int lo = 0, hi = ne;
        if (!resident) {
          // this wave's DMA groups of the rounds behind st that are already issued may stay in flight (the Q loads are older than every DMA)
          int later = 0;
          for (int r = st + 1; r < min(nst, st + kResAhead); ++r) later += (8 * r + wave < 2 * ne) ? 1 : 0;
          res_wait_vm(4 * later);
          if (st == 0 && e == 0) pin_q();
          unsigned long long tb = 0;
          if (SSHIP_ATTN_RES_TRACE && trace) tb = __builtin_readcyclecounter();
          asm volatile("s_barrier" ::: "memory");  // everybody's share of tiles 4 st .. 4 st + 3 has landed
          if (SSHIP_ATTN_RES_TRACE && trace) { const unsigned long long tn = __builtin_readcyclecounter(); tr[5] += tn - tb; if (st == 0 && e == 0 && pidx == 0) tr[1] = tn; }
          if (st + kResAhead < nst) fill_round(e, ne, st + kResAhead);  // the round this stage's compute hides
          lo = 4 * st; hi = min(ne, 4 * st + 4);
        }
        run_range(nq_c, e, lo, valid ? hi : lo);
      }
    }
    if (SSHIP_ATTN_RES_TRACE && trace && pidx == 0) tr[2] = __builtin_readcyclecounter();
    // ---- normalise and store (the register finalisation of k_lg_attention<.., 1, ..>) ----
    if (valid) {
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        const int q0 = (ta + 8 * t) * 32;
        const float lt = l[t] + __shfl_xor(l[t], 32, 64);
        const float inv = lt > 0.f ? 1.0f / lt : 0.f;
        _Float16* orow = ctx + ((size_t)s * NP + q0 + j) * 256 + h * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<h4_t*>(orow + mt * 32 + 8 * g + 4 * hh) =
                to_h4(o[t][mt][4 * g] * inv, o[t][mt][4 * g + 1] * inv, o[t][mt][4 * g + 2] * inv, o[t][mt][4 * g + 3] * inv);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores and the next pass's counted loads do not mix
    if (SSHIP_ATTN_RES_TRACE && trace && pidx == 0) tr[3] = __builtin_readcyclecounter();
  };
  // Pass plan over the workgroup's ntq query tiles: full passes of 16 tiles (two per wave); a remainder r > 8 adds a two-tile pass for
  // the waves w < r - 8 and a single-tile pass for the others; a remainder r <= 8 one single-tile pass for the waves w < r.
  const int ntq = qhi - qlo, nfull = ntq >> 4, rem = ntq & 15;
  const int n2 = nfull + (rem > 8 ? 1 : 0);
  int pidx = 0;
  for (int i = 0; i < n2; ++i, ++pidx) {
    const int ta = qlo + 16 * i + wave;
    do_pass(std::integral_constant<int, 2>{}, ta, ta + 8 < qhi, pidx);
  }
  if (rem > 0) {
    const int ta = qlo + 16 * nfull + wave;
    do_pass(std::integral_constant<int, 1>{}, ta, rem > 8 ? wave >= rem - 8 : wave < rem, pidx);
  }
  if (SSHIP_ATTN_RES_TRACE && trace && lane == 0) {
    tr[4] = __builtin_readcyclecounter();
    unsigned long long* o_ = trace + ((size_t)L * 8 + wave) * 8;
    o_[0] = tr[1] - tr[0]; o_[1] = tr[2] - tr[1]; o_[2] = tr[3] - tr[2]; o_[3] = tr[4] - tr[3]; o_[4] = tr[5]; o_[5] = tr[4] - tr[0]; o_[6] = 1;
  }
}

}  // namespace

// The resident-key kernel covers the CUs only when there are at least as many (sequence, head[, query half]) units as CUs.
bool lg_attention_res_fits(LgDims d) {
  const int nt32 = d.NP / 32;
  const int nqsplit = nt32 > 24 ? (nt32 + 15) / 16 : 1;
  return d.NP % 32 == 0 && nt32 >= 1 && (long)d.S * 4 * nqsplit >= cu_count();
}

void launch_lg_attention_res(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d, bool cross,
                             _Float16* ctx, hipStream_t s) {
  const int nt32 = d.NP / 32;
  const int nep = (nt32 + kResCapTiles - 1) / kResCapTiles;
  const int ept = (nt32 + nep - 1) / nep;                    // tiles per epoch: 19 at 608 tokens, 16 at 1024
  const int nqsplit = nt32 > 24 ? (nt32 + 15) / 16 : 1;      // query tiles per workgroup: all (<= 24: at most a third, single tile, pass), or 16
  const int qper = (nt32 + nqsplit - 1) / nqsplit;
  const int smem = ept * 8192;
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lg_attention_res),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kResCapTiles * 8192);
  (void)attr_rc;
  const int n_wg = d.S * 4 * nqsplit;
  unsigned long long* tbuf = nullptr;
  static const bool trace_on = SSHIP_ATTN_RES_TRACE && dev_env("SSHIP_ATTN_TRACE") != nullptr;
  if (trace_on) { (void)hipMalloc(&tbuf, (size_t)n_wg * 64 * 8); (void)hipMemsetAsync(tbuf, 0, (size_t)n_wg * 64 * 8, s); }
  hipLaunchKernelGGL(k_lg_attention_res, dim3((n_wg + 7) / 8 * 8), dim3(512), smem, s, q, k, vt, lens, d.NP, cross ? 1 : 0, ctx, n_wg,
                     nqsplit, qper, ept, tbuf);
  if (trace_on) {
    std::vector<unsigned long long> h((size_t)n_wg * 64);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) {
      double sum[6] = {0, 0, 0, 0, 0, 0}; long cnt = 0;
      for (int L = 0; L < n_wg; ++L) {
        const unsigned long long* r = &h[((size_t)L * 8 + w) * 8];
        if (r[6]) { for (int i = 0; i < 6; ++i) sum[i] += (double)r[i]; ++cnt; }
      }
      if (cnt) fprintf(stderr, "[attn res trace cross=%d wave %d] to first stage=%.0f staged sweep (pass 0)=%.0f store=%.0f later passes=%.0f barrier wait=%.0f total=%.0f clk (%ld workgroups)\n",
                       (int)cross, w, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, cnt);
    }
    (void)hipFree(tbuf);
  }
}

}  // namespace sship
