// LightGlue kernels (SURVEY.md 8(a)-LG restates the upstream algorithm; call sites src/LightGlue.cc:313,446).
// Token streams are channels-last fp16 [S*NP][256]; S = 2*pairs sequences (2p = set 0, 2p+1 = set 1 of pair p),
// NP = padded tokens per sequence (a multiple of 32: 600 keypoints -> 608 tokens).  Per-sequence valid lengths live in device memory (`lens`) so the whole
// matcher runs without a host round trip after SuperPoint's on-device top-k.
#include <cstdlib>
#include <type_traits>

#include "igemm.h"
#include "kernels.h"
#include "lg_ffn.h"

namespace sship {

// ---------------------------------------------------------------------------------------------------
// prep: x <- descriptors (zero rows for padding), rotary table <- posenc(normalised keypoints).
// reference: keypoint normalisation src/LightGlue.cc:241-251; LearnableFourierPositionalEncoding(2,64,64).
// rope[token][i] = (cos, sin)(Wr[i,0]*kx + Wr[i,1]*ky), i < 32 (shared by the 4 heads).
// ---------------------------------------------------------------------------------------------------
// Also publishes lens_clamped[s] = clamp(lens[s], 0, max_kp) for the rest of the call and the normalised keypoints (kpn,
// zeros for padding rows) for the parity suite's bit-exact check of the normalisation.
__global__ __launch_bounds__(256) void k_lg_prep(const float* __restrict__ kp, int kp_stride, int kp_seq_stride,
                                                 const int* __restrict__ lens, int max_kp, int* __restrict__ lens_clamped,
                                                 const _Float16* __restrict__ desc,
                                                 size_t desc_seq_stride, const float* __restrict__ wr, float img_w,
                                                 float img_h, int S, int NP, _Float16* __restrict__ x,
                                                 float* __restrict__ rope, float* __restrict__ kpn) {
  if (blockIdx.x == 0 && (int)threadIdx.x < S) lens_clamped[threadIdx.x] = min(max(lens[threadIdx.x], 0), max_kp);
  if (blockIdx.x == 0)
    for (int i = 256 + threadIdx.x; i < S; i += 256) lens_clamped[i] = min(max(lens[i], 0), max_kp);
  const int token = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (token >= S * NP) return;
  const int s = token / NP, n = token % NP;
  const bool valid = n < min(max(lens[s], 0), max_kp);
  h4_t v = to_h4(0.f, 0.f, 0.f, 0.f);
  if (valid) v = *reinterpret_cast<const h4_t*>(desc + (size_t)s * desc_seq_stride + (size_t)n * 256 + lane * 4);
  *reinterpret_cast<h4_t*>(x + (size_t)token * 256 + lane * 4) = v;
  if (lane < 32) {
    float c = 1.f, sn = 0.f, kx = 0.f, ky = 0.f;
    if (valid) {
      const float* k = kp + (size_t)s * kp_seq_stride + (size_t)n * kp_stride;
      const float scale = fmaxf(img_w, img_h) / 2.0f;   // std::max(w, h) / 2.0f
      kx = (k[0] - img_w / 2.0f) / scale;               // (pt.x - cx) / scale : IEEE division, as the host code
      ky = (k[1] - img_h / 2.0f) / scale;
      const float pr = wr[lane * 2 + 0] * kx + wr[lane * 2 + 1] * ky;
      c = cosf(pr);
      sn = sinf(pr);
    }
    if (lane == 0) { kpn[(size_t)token * 2 + 0] = kx; kpn[(size_t)token * 2 + 1] = ky; }
    rope[(size_t)token * 64 + lane * 2 + 0] = c;
    rope[(size_t)token * 64 + lane * 2 + 1] = sn;
  }
}
void launch_lg_prep(const float* kp, int kp_stride, int kp_seq_stride, const int* lens, int max_kp, int* lens_clamped,
                    const _Float16* desc, size_t desc_seq_stride, const float* wr, float img_w, float img_h, LgDims d,
                    _Float16* x, float* rope, float* kpn, hipStream_t s) {
  const int tokens = d.S * d.NP;
  hipLaunchKernelGGL(k_lg_prep, dim3((tokens + 3) / 4), dim3(256), 0, s, kp, kp_stride, kp_seq_stride, lens, max_kp,
                     lens_clamped, desc, desc_seq_stride, wr, img_w, img_h, d.S, d.NP, x, rope, kpn);
}

// ---------------------------------------------------------------------------------------------------
// igemm epilogues for the Linear layers.  A "pixel" is a token: token = y*32 + x (the token stream is an
// image of width 32).  Output rows come in 256-wide segments:
//   SelfBlock  Wqkv : rows permuted on the host to [q | k | v] x [head][64]  -> rope on q,k, v transposed
//   CrossBlock      : [to_qk | to_v] fused into one 512-row GEMM            -> no rope, v transposed
// Softmax scale (and log2 e for exp2) are folded into the q / qk rows on the host.
// ---------------------------------------------------------------------------------------------------
struct EpiHeads {
  template <int MT, int NT>
  static __device__ __forceinline__ void run(const IgemmArgs& p, f16x_t (&acc)[MT][NT], int b, int yb, int x,
                                             int cb0, int hh) {
    const int rope_segs = p.flags & 0xf, t_seg = (p.flags >> 4) & 0xf;
    const int NP = p.np;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int token = (yb + n) * 32 + x;
      if (yb + n >= p.H) continue;
      const int s = token / NP, tn = token - s * NP;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int R0 = cb0 + m * 32 + hh * 4 + g * 8;
          if (R0 >= p.cout) continue;
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + R0);
          float v0 = acc[m][n][4 * g + 0] + bv.x, v1 = acc[m][n][4 * g + 1] + bv.y;
          float v2 = acc[m][n][4 * g + 2] + bv.z, v3 = acc[m][n][4 * g + 3] + bv.w;
          const int seg = R0 >> 8, hd = (R0 >> 6) & 3, d0 = R0 & 63;
          if (seg < rope_segs) {
            // rotate_half on interleaved pairs: out[2i] = x[2i] c_i - x[2i+1] s_i ; out[2i+1] = x[2i+1] c_i + x[2i] s_i
            const float4 cs = *reinterpret_cast<const float4*>(p.aux + (size_t)token * 64 + d0);  // (c0,s0,c1,s1)
            const float r0 = v0 * cs.x - v1 * cs.y, r1 = v1 * cs.x + v0 * cs.y;
            const float r2 = v2 * cs.z - v3 * cs.w, r3 = v3 * cs.z + v2 * cs.w;
            v0 = r0; v1 = r1; v2 = r2; v3 = r3;
          }
          const int nt32 = NP >> 5, kt = tn >> 5, kap = tn & 31;
          const size_t tile = ((size_t)s * 4 + hd) * nt32 + kt;
          if (seg != t_seg) {
            // Q / K: MFMA-fragment order per 32-token tile: [tile][kstep d/16][lane = ((d%16)/8)*32 + token%32][d%8]
            _Float16* dst = static_cast<_Float16*>(seg == 0 ? p.out0 : p.out1);
            *reinterpret_cast<h4_t*>(dst + (tile * 4 + (d0 >> 4)) * 512 + ((((d0 & 15) >> 3) << 5) + kap) * 8 + (d0 & 7)) =
                to_h4(v0, v1, v2, v3);
          } else {
            // V: A-fragment order of the PV MFMA: [tile][kk = key/16][mt = d/32][lane = hh*32 + d%32][e],
            // key%16 = r -> hh = (r%8)/4, e = r%4 + 4*(r/8)   (the key permutation of the swapped QK^T C-layout)
            const int kk = kap >> 4, r = kap & 15;
            const int hh2 = (r & 7) >> 2, ee = (r & 3) + ((r >> 3) << 2);
            _Float16* dst = static_cast<_Float16*>(p.out2) + ((tile * 2 + kk) * 2 + (d0 >> 5)) * 512 + ((hh2 << 5) + (d0 & 31)) * 8 + ee;
            dst[0] = (_Float16)v0; dst[8] = (_Float16)v1; dst[16] = (_Float16)v2; dst[24] = (_Float16)v3;
          }
        }
    }
  }
};

static IgemmArgs token_args(const ConvW& w, const _Float16* in0, int cs0, const _Float16* in1, int cs1, LgDims d) {
  IgemmArgs a{};
  a.in0 = in0; a.in1 = in1 ? in1 : in0; a.cs0 = cs0; a.cs1 = in1 ? cs1 : cs0;
  a.cin0 = in1 ? cs0 : w.cin;
  a.wpack = w.w; a.bias = w.bias;
  a.B = 1; a.H = d.S * d.NP / 32; a.W = 32;
  a.cout = w.cout; a.np = d.NP;
  return a;
}

#if SSHIP_DEV_SWITCHES  // the first Wqkv as a stand-alone implicit GEMM (SUPERSLAM_HIP_LG_QKV0=igemm; the default runs it in the FFN kernel's projection stage)
hipError_t lg_linear_heads(const ConvW& w, const _Float16* x, LgDims d, int rope_segs, int t_seg, const float* rope,
                           _Float16* q, _Float16* k, _Float16* vt, hipStream_t s) {
  IgemmArgs a = token_args(w, x, 256, nullptr, 0, d);
  a.out0 = q; a.out1 = k; a.out2 = vt; a.aux = rope; a.flags = rope_segs | (t_seg << 4);
  return launch_igemm<1, 256, 128, 4, EpiHeads>(a, w.cout_pad, s);
}
#endif
// ---------------------------------------------------------------------------------------------------
// Flash-style attention, one wave per 32 queries, head_dim 64.  Self (keys = own sequence) and cross
// (keys = partner sequence s^1: both directions of CrossBlock in one launch).
// S^T = K Q^T is computed "swapped" so a lane owns ONE query column: the online-softmax statistics are
// lane-local plus one exchange with lane^32.  The key order inside a 32-key tile is whatever the MFMA
// C-layout hands out; V^T is fetched with the same permutation, so P never moves between lanes:
//   reg r of lane (j, hh)  <->  key k0 + (r&3) + 8*(r>>2) + 4*hh
//   PV K-step kk uses regs 8kk..8kk+7 = keys {16kk + 4hh + e, 16kk + 8 + 4hh + e}, e = 0..3.
// Scores arrive pre-scaled by log2(e)/sqrt(64) (folded into the projection weights) -> exp2f.
// ---------------------------------------------------------------------------------------------------
// QT query tiles (32 queries each) per wave share every K / V^T fragment load: the kernel is bound by the L2 -> CU
// fragment traffic (each workgroup streams the whole K/V of its (sequence, head) once: QT = 1 moved 840 MB per launch
// at P = 32), so two query tiles per wave halve it at the cost of 2x accumulator registers.
// v_max3_f32.  This file is built with -fno-honor-nans (build.py): without it fmaxf() canonicalises every MFMA output
// first (one extra v_max_f32 x, x per score).  No inline asm here: hipcc does not insert the MFMA -> VALU wait states
// in front of asm operands.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// max(x, x of lane ^ 32)
__device__ __forceinline__ float max_xor32(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0] = {lo half, lo half}, r[1] = {hi, hi}
  return max3f(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[1]));
}
// KS = waves that split the keys of one query group (flash-decoding style); the workgroup's 4 waves hold 4 / KS query
// groups of 32 QT queries.  KS = 4 (latency mode): 32 QT queries per workgroup, 5 key tiles per wave at 640 keys.
// KS = 2 (throughput): twice the key tiles per wave, so the per-wave prologue (Q + first K/V fragments) and the LDS merge
// of the partials are amortised over twice as many iterations, and half as many partials are merged per query.
// developer aid: -DSSHIP_ATTN_TRACE_BUILD=1 + SSHIP_ATTN_TRACE=1 prints mean clocks of prologue / key loop / merge per wave
#ifndef SSHIP_ATTN_TRACE_BUILD
#define SSHIP_ATTN_TRACE_BUILD 0
#endif
#ifndef SSHIP_ATTN_LDS_STORE
#define SSHIP_ATTN_LDS_STORE 1  // KS == 1: context rows leave as whole 128-byte lines through a wave-private LDS patch (0: per-lane 8-byte stores, A/B builds)
#endif
#ifndef SSHIP_ATTN_REGFIN
#define SSHIP_ATTN_REGFIN 1  // KS == 1: normalise and store the context straight from the accumulators (0: through the merge buffer, A/B builds)
#endif
// V = softmax bookkeeping variant (the loop is VALU-issue bound next to its MFMAs - 55 VALU + 17 transcendentals per 8 MFMAs in
// V = 1 - while the matrix pipe idles three quarters of the time, so V >= 2 move VALU work INTO the matrix pipe):
//   1  classic online softmax: exact running max m, P = exp2(s - m): 16 v_sub per (key tile, query tile), rescale when m moves;
//   2  the reference exponent r rides in the QK^T MFMA chain: a fifth K-step whose A fragment is a column of ones and whose B
//      fragment holds -r of the lane's query (as two fp16 k-slots, 256 a + b), so the accumulator comes out as s - r and
//      P = exp2(acc) directly - no subtractions.  r is a per-query value close to the running max (softmax is shift invariant: any reference within
//      fp16 range of the max is exact; P <= 2^8 stays far inside fp16).  r is set from the first key tile and moved only when
//      a tile's maximum exceeds it by more than 8 (then this tile's scores are re-based in place: 16 v_sub in a rare branch);
//   3  = 2 with the QK^T MFMA chains of both query tiles issued first and interleaved (see `tile`).
//      (another variant with the row sums on the matrix pipe as well - l += ones(32 x 16) P, two more MFMAs per key tile instead of
//      eight v_dot2_f32_f16 - measured no faster than 2 and sat on the mscores0 bar: profiles/r03_a_attention_variants.txt; removed).
// Energy ablations of the attention kernel (build.py --variant ... -DSSHIP_ATTN_ABL=n, scripts/dev/energy_abl.sh attn): 1 no MFMAs (operands still
// delivered), 2 P = the exponent's argument instead of exp2 of it (no v_exp_f32 in the key loop).  Results are wrong by design.
#ifndef SSHIP_ATTN_ABL
#define SSHIP_ATTN_ABL 0
#endif
#ifndef SSHIP_ATTN_TAILSKIP
#define SSHIP_ATTN_TAILSKIP 1  // a unit's query tiles past the sequence end are not computed (0: round 5's behaviour, computed and discarded; A/B builds)
#endif
__device__ __forceinline__ f16x_t mfma32_attn(h8_t a, h8_t b, f16x_t c) {
  if constexpr ((SSHIP_ATTN_ABL & 1) != 0) {
    asm volatile("" :: "v"(a), "v"(b));
    return c;
  } else {
    return mfma32(a, b, c);
  }
}
template <int QT, int KS, int V>
__global__ __launch_bounds__(256, 2) void k_lg_attention(const _Float16* __restrict__ q, const _Float16* __restrict__ k,
                                                         const _Float16* __restrict__ vt, const int* __restrict__ lens,
                                                         int NP, int cross, _Float16* __restrict__ ctx,
                                                         unsigned long long* __restrict__ trace, int gx, int S) {
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  if (SSHIP_ATTN_TRACE_BUILD && trace) tr0 = __builtin_readcyclecounter();
  // One workgroup = 32*QT queries of one (sequence, head); its 4 waves split the KEYS (tile kt -> wave kt & 3,
  // flash-decoding style) and merge their (m, l, O) partials through LDS.
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  float (*s_part)[4][34][64] = reinterpret_cast<float (*)[4][34][64]>(smem_attn);  // [QT][wave][32 O regs + m + l][lane]
  // XCD-aware workgroup mapping.  The gx workgroups of one (sequence, head) stream the same K / V^T (and, in the cross block,
  // the partner sequence's), 156 KB each - more than the launch's L2 footprint allows to stay resident (120 MB of Q / K / V^T per
  // 64-pair launch against 8 x 4 MB of L2).  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, so with
  // a plain (x, head, sequence) grid those gx workgroups land on gx DIFFERENT XCDs and every one of their private L2s fetches the
  // same K / V^T over the fabric (400 MB per launch at 4.1 TB/s: a third of the wave cycles were s_waitcnt, profiles/r02_final_pmc_sq).
  // The grid is therefore 1-D and id -> logical workgroup L = (id % 8) * ceil(N / 8) + id / 8: XCD k runs the CONSECUTIVE logical
  // workgroups [k N/8, (k+1) N/8), i.e. all query blocks of a (sequence, head), then its other heads, then the partner sequence.
  // (Placement is a speed assumption only: any mapping of ids to XCDs computes the same result.)
  // Without a key split (KS == 1, register finalisation) the four waves of a workgroup share nothing - no LDS, no barrier - so
  // the unit of work is the WAVE: gx then counts the QT-tile query units of one (sequence, head) and wave W = 4 L + wave takes
  // unit W % gx of (sequence, head) W / gx.  600 keypoints are 19 query tiles = 10 units: 640 full workgroups per 64 sequences
  // instead of 768 of which every third ran 1.5 of its 4 waves.
  constexpr bool kWaveUnits = KS == 1 && SSHIP_ATTN_REGFIN;
  const int n_wg = (kWaveUnits ? gx : gx * 4) * (S < 0 ? -S : S), per_xcd = (n_wg + 7) >> 3;
  const int L = S < 0 ? (int)blockIdx.x : (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);   // S < 0: identity mapping (A/B runs)
  if (L >= n_wg) return;  // the grid is padded to a multiple of 8
  (void)per_xcd;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int qgrp = wave / KS, ksp = wave % KS;
  const int W = L * 4 + wave;
  const int bx = kWaveUnits ? W % gx : L % gx, h = kWaveUnits ? (W / gx) & 3 : (L / gx) & 3, s = kWaveUnits ? W / (gx * 4) : L / (gx * 4);
  const int qt0 = kWaveUnits ? bx * QT : (bx * (4 / KS) + qgrp) * QT;  // first 32-query tile of this wave's query group
  const int q0 = qt0 * 32;
  const int sk = cross ? (s ^ 1) : s;
  const int nq = min(max(lens[s], 0), NP), nk = min(max(lens[sk], 0), NP);  // device-side counts are clamped to capacity
  if (!kWaveUnits && bx * (4 / KS) * QT * 32 >= nq) return;  // uniform for the whole workgroup
  const bool active = q0 < nq && q0 < NP;  // wave-uniform: a query group past the end only takes part in the barrier
  if (kWaveUnits && !active) return;       // ... and there is no barrier on this path
  // Q/K/V are stored in MFMA-fragment order per 32-token tile (EpiHeads): every operand load below is one
  // fully coalesced 1-KiB wave load (16 B per lane, lane-linear).
  const int nt32 = NP >> 5;
  const _Float16* Q = q + ((size_t)(s * 4 + h) * nt32) * 2048;
  const _Float16* K = k + ((size_t)(sk * 4 + h) * nt32) * 2048;
  const _Float16* VT = vt + ((size_t)(sk * 4 + h) * nt32) * 2048;
  h8_t qf[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[t][ks] = *reinterpret_cast<const h8_t*>(Q + ((size_t)min(qt0 + t, nt32 - 1) * 4 + ks) * 512 + lane * 8);  // NP is a multiple of 32, not of 32 QT; the address does not wait for `lens`
  float m[QT], l[QT];
  f16x_t o[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[t][0][r] = 0.f; o[t][1][r] = 0.f; }
  }
  const int ntiles = active ? (nk + 31) >> 5 : 0;
  // K and V^T fragments are prefetched ONE FULL TILE ahead (loads for tile kt + KS are issued before the MFMAs and
  // softmax of tile kt), so ~1k cycles of L2 latency hide behind a whole iteration instead of a few MFMAs.  Two register
  // sets, used alternately by the two halves of the unrolled key loop: no fragment copies at the end of an iteration
  // (they were 41 of the ~190 VALU instructions of an iteration, and this loop is VALU-bound).
  h8_t kfA[4], vfA[2][2], kfB[4], vfB[2][2];
  auto fetch = [&](h8_t (&kf)[4], h8_t (&vf)[2][2], int kt) __attribute__((always_inline)) {
    const int ktc = min(kt, nt32 - 1);  // clamped: a prefetch past the end re-reads a valid tile and is discarded
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const h8_t*>(K + ((size_t)ktc * 4 + ks) * 512 + lane * 8);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        vf[kk][mt] = *reinterpret_cast<const h8_t*>(VT + (((size_t)ktc * 2 + kk) * 2 + mt) * 512 + lane * 8);
  };
  const h2_t ones2 = {(_Float16)1.f, (_Float16)1.f};
  // V = 2: A fragment "column 0 = 1" (rows = keys; lane (row, hh) holds k = 8 hh + e) and, per query tile, the B fragment
  // "row 0 = -r of the lane's query" (lane (query, hh) holds k = 8 hh + e); m[t] holds r.
  // r is carried as 256 a + b with a, b fp16 (k-slots 0 and 1: A = [256, 1], B = [-a, -b]): exact to fp32 precision for any
  // |r| < 1.6e7 - a single fp16 slot would turn a row whose scaled logits exceed 65 504 into inf (the reference's fp16 engine
  // overflows there as well; this path must not be the narrower one).
  h8_t ones_k0, rf[QT];
#pragma unroll
  for (int e = 0; e < 8; ++e) ones_k0[e] = (_Float16)0.f;
  if (hh == 0) { ones_k0[0] = (_Float16)256.f; ones_k0[1] = (_Float16)1.f; }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int e = 0; e < 8; ++e) rf[t][e] = (_Float16)0.f;
    if (V >= 2) m[t] = 0.f;
  }
  // one key tile for the QT query tiles of this wave
  // V = 3: the QK^T chains of ALL query tiles are issued first, interleaved k-step by k-step (independent accumulators: no
  // dependent-MFMA stall, and the second tile's MFMAs execute while the first tile's softmax VALU issues); V <= 2: tile by tile.
  // NT = the query tiles of this unit that really exist (round 6): 19 query tiles at 600 keypoints are nine units of two tiles and ONE of a
  // single tile - its second tile used to be computed over all key tiles from a clamped address and thrown away at the store (1 / 20 of the
  // launch's MFMAs, exponentials and joules).  The same holds for any image with fewer keypoints than capacity.
  auto tile = [&](auto ntc, const h8_t (&kf)[4], const h8_t (&vf)[2][2], int kt) __attribute__((always_inline)) {
    constexpr int NT = decltype(ntc)::value;
    const int k0 = kt * 32;
    const f16x_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16x_t stq[NT];
    if constexpr (V == 3) {
#pragma unroll
      for (int t = 0; t < NT; ++t) stq[t] = mfma32_attn(ones_k0, rf[t], zero16);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) stq[t] = mfma32_attn(kf[ks], qf[t][ks], stq[t]);
      __builtin_amdgcn_sched_barrier(0);  // the MFMAs above stay ahead of the first tile's softmax
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f16x_t st;
      if constexpr (V == 3) st = stq[t];
      else {
        st = V >= 2 ? mfma32_attn(ones_k0, rf[t], zero16) : zero16;   // -r per query, or 0
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st = mfma32_attn(kf[ks], qf[t][ks], st);
      }
      if (k0 + 32 > nk) {  // only the last (ragged) key tile needs masking - wave-uniform branch
        int kb = k0 + 4 * hh;
        asm volatile("" : "+v"(kb));  // keeps the 16 key indices inside the branch (hipcc hoisted them into every iteration)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb + (r & 3) + 8 * (r >> 2) >= nk) st[r] = -INFINITY;
      }
      // running max on v_max3_f32 (8 instructions; fmaxf() first canonicalises every MFMA output: 16 more), and the
      // lane^32 exchange on v_permlane32_swap instead of an LDS round trip (ds_bpermute + lgkmcnt(0) per query tile)
      float tmax = max3f(st[0], st[1], st[2]);
      tmax = max3f(tmax, st[3], st[4]);
#pragma unroll
      for (int r = 5; r < 15; r += 2) tmax = max3f(tmax, st[r], st[r + 1]);
      if constexpr (V == 1) {
        const float m_new = max3f(m[t], st[15], max_xor32(max3f(tmax, st[15], st[15])));
        // the softmax is VALU-bound at head_dim 64: rescale the 32 output accumulators only when some query's running
        // max actually moved (exact - not the lossy defer-max trick)
        if (__any(m_new > m[t])) {
          const float alpha = __builtin_amdgcn_exp2f(m[t] - m_new);
          l[t] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[t][0][r] *= alpha; o[t][1][r] *= alpha; }
          m[t] = m_new;
        }
      } else {
        // st = s - r.  First tile of this wave (l == 0 and nothing accumulated yet): r <- fp16(max); later: only when the tile's
        // maximum exceeds r by more than 8.  Both halves of a query's lane pair see the same tmax, so they take the same decision.
        tmax = max_xor32(max3f(tmax, st[15], st[15]));
        const bool first = kt == ksp;
        const bool need = first || tmax > 8.0f;
        if (__any(need)) {
          const float r_tgt = m[t] + tmax;
          // the coarse slot comes in at |r| >= 2048: below that a = 0 and r = fp16(running max) is off by <= 0.5, so P <= 2^8.5;
          // above it a single fp16 value would be off by up to 16 (32768 <= |r|) and P = exp2(8 + 16) overflows fp16 -> NaN
          const float ra = fabsf(r_tgt) < 2048.f ? 0.f : (float)(_Float16)(fminf(fmaxf(r_tgt * (1.0f / 256.0f), -65000.f), 65000.f));
          const float rb = (float)(_Float16)(r_tgt - 256.0f * ra);
          const float r_new = need ? 256.0f * ra + rb : m[t];   // = what the two k-slots add up to in the MFMA
          const float d = r_new - m[t];
          // first tile: l = o = 0 and d = r itself, possibly very negative (exp2(-d) = inf, 0 * inf = NaN: found by
          // tests/test_gpu_lightglue_layers.py::test_large_residual_stream_magnitudes) - nothing to rescale yet.
          // Later tiles move r up by more than ~8, so alpha <= 2^-7.
          const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
          l[t] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[t][0][r] *= alpha; o[t][1][r] *= alpha; st[r] -= d; }
          m[t] = r_new;
          if (need && hh == 0) { rf[t][0] = (_Float16)(-ra); rf[t][1] = (_Float16)(-rb); }
        }
      }
      // P in fp16 (the PV operand); the row sum is taken over exactly these rounded values: v_dot2_f32_f16 against ones, fp32
      // accumulate - half the instructions of sixteen fp32 adds
      float ls0 = 0.f, ls1 = 0.f;
      h8_t pb[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float a0 = V == 1 ? st[8 * kk + e] - m[t] : st[8 * kk + e], a1 = V == 1 ? st[8 * kk + e + 1] - m[t] : st[8 * kk + e + 1];
          const h2_t pp = {(_Float16)((SSHIP_ATTN_ABL & 2) ? a0 : __builtin_amdgcn_exp2f(a0)), (_Float16)((SSHIP_ATTN_ABL & 2) ? a1 : __builtin_amdgcn_exp2f(a1))};
          pb[kk][e] = pp[0]; pb[kk][e + 1] = pp[1];
          if (kk == 0) ls0 = __builtin_amdgcn_fdot2(pp, ones2, ls0, false);
          else ls1 = __builtin_amdgcn_fdot2(pp, ones2, ls1, false);
        }
      l[t] += ls0 + ls1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) o[t][mt] = mfma32_attn(vf[kk][mt], pb[kk], o[t][mt]);
    }
  };
  fetch(kfA, vfA, ksp);
  if (SSHIP_ATTN_TRACE_BUILD && trace) tr1 = __builtin_readcyclecounter();
  auto key_loop = [&](auto ntc) __attribute__((always_inline)) {
    for (int kt = ksp; kt < ntiles; kt += 2 * KS) {
      fetch(kfB, vfB, kt + KS);
      __builtin_amdgcn_sched_barrier(0);  // the prefetch stays above this tile's MFMAs / softmax
      tile(ntc, kfA, vfA, kt);
      if (kt + KS < ntiles) {
        fetch(kfA, vfA, kt + 2 * KS);
        __builtin_amdgcn_sched_barrier(0);
        tile(ntc, kfB, vfB, kt + KS);
      }
    }
  };
  // wave-uniform: the unit's live query tiles (the tiles past nq keep l = 0, o = 0 and are never stored: the store loops break at nq)
  if (QT == 1 || !SSHIP_ATTN_TAILSKIP || q0 + 32 * (QT - 1) < nq) key_loop(std::integral_constant<int, QT>{});
  else key_loop(std::integral_constant<int, 1>{});
  if (SSHIP_ATTN_TRACE_BUILD && trace) tr2 = __builtin_readcyclecounter();
  if constexpr (KS == 1 && SSHIP_ATTN_REGFIN) {
    // no key split: nothing to merge - every lane normalises and stores its own accumulators (same values, same stores as the
    // merge path below with one partial: 0 + o * 1 * inv), without the 68 LDS writes, the barrier that made a workgroup's waves
    // wait for its slowest one, and the 66 LDS reads per lane.
    if (!active) return;
    // The context rows leave as WHOLE 128-byte lines: a lane owns one query and 4 consecutive channels per register quad, so storing from
    // the accumulators is 16 eight-byte stores per query tile that each touch 32 different rows (512 sixteen-byte pieces of lines; the store
    // phase was 5.6 k of a wave's ~53 k clocks, profiles/r05_a_*).  Instead the tile goes through a wave-private 32 x 144-byte LDS patch
    // (the kernel uses no other LDS on this path; 36-dword row stride: the 64 lanes' ds_write_b64 hit 64 distinct bank pairs) and comes back
    // as 16 bytes per lane, 8 lanes per row: 4 stores of eight full lines.  Same values, same bytes.
    _Float16* patch = reinterpret_cast<_Float16*>(smem_attn) + wave * (32 * 72);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (q0 + t * 32 >= nq) break;
      const float lt = l[t] + __shfl_xor(l[t], 32, 64);
      const float inv = lt > 0.f ? 1.0f / lt : 0.f;
#if !SSHIP_ATTN_LDS_STORE  // A/B builds: the per-lane 8-byte stores straight from the accumulators
      _Float16* orow = ctx + ((size_t)s * NP + q0 + t * 32 + j) * 256 + h * 64;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<h4_t*>(orow + mt * 32 + 8 * g + 4 * hh) =
              to_h4(o[t][mt][4 * g] * inv, o[t][mt][4 * g + 1] * inv, o[t][mt][4 * g + 2] * inv, o[t][mt][4 * g + 3] * inv);
      continue;
#endif
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<h4_t*>(patch + j * 72 + mt * 32 + 8 * g + 4 * hh) =
              to_h4(o[t][mt][4 * g] * inv, o[t][mt][4 * g + 1] * inv, o[t][mt][4 * g + 2] * inv, o[t][mt][4 * g + 3] * inv);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its LDS operations execute in order; this orders the compiler
      _Float16* obase = ctx + ((size_t)s * NP + q0 + t * 32) * 256 + h * 64 + (lane & 7) * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3);
        *reinterpret_cast<h8_t*>(obase + (size_t)row * 256) = *reinterpret_cast<const h8_t*>(patch + row * 72 + (lane & 7) * 8);
      }
      asm volatile("" ::: "memory");  // the next tile's writes stay behind these reads
    }
    if (SSHIP_ATTN_TRACE_BUILD && trace && lane == 0) {
      unsigned long long* o_ = trace + ((size_t)L * 4 + wave) * 4;
      o_[0] = tr1 - tr0; o_[1] = tr2 - tr1; o_[2] = __builtin_readcyclecounter() - tr2; o_[3] = 1;
    }
    return;
  }
  // ---- merge the KS key-partials of every query group ----
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    l[t] += __shfl_xor(l[t], 32, 64);
    if (V >= 2 && ntiles <= ksp) m[t] = -INFINITY;   // a wave that saw no key tile contributes nothing to the merge
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_part[t][wave][r][lane] = o[t][0][r]; s_part[t][wave][16 + r][lane] = o[t][1][r]; }
    s_part[t][wave][32][lane] = m[t];
    s_part[t][wave][33][lane] = l[t];
  }
  __syncthreads();
  if (!active) return;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (q0 + t * 32 >= nq) break;
    float mw[KS], mt_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < KS; ++w) { mw[w] = s_part[t][qgrp * KS + w][32][lane]; mt_all = fmaxf(mt_all, mw[w]); }
    float sc[KS], lt = 0.f;
#pragma unroll
    for (int w = 0; w < KS; ++w) {
      sc[w] = (mw[w] == -INFINITY) ? 0.f : exp2f(mw[w] - mt_all);
      lt += s_part[t][qgrp * KS + w][33][lane] * sc[w];
    }
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    _Float16* orow = ctx + ((size_t)s * NP + q0 + t * 32 + j) * 256 + h * 64;
    // split ksp finalises combined registers R = (32 / KS) ksp .. + 32 / KS - 1 (R = mt*16 + r): groups of 4 consecutive channels
#pragma unroll
    for (int gq = 0; gq < 8 / KS; ++gq) {
      const int R0 = ksp * (32 / KS) + gq * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) acc += s_part[t][qgrp * KS + w][R0 + e][lane] * sc[w];
        v[e] = acc * inv;
      }
      const int mt = R0 >> 4, r = R0 & 15;
      const int d = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      *reinterpret_cast<h4_t*>(orow + d) = to_h4(v[0], v[1], v[2], v[3]);
    }
  }
  if (SSHIP_ATTN_TRACE_BUILD && trace && lane == 0) {
    unsigned long long* o = trace + ((size_t)L * 4 + wave) * 4;
    o[0] = tr1 - tr0; o[1] = tr2 - tr1; o[2] = __builtin_readcyclecounter() - tr2; o[3] = 1;
  }
}
template <int QT, int KS, int V>
static void launch_attn_v(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d, bool cross,
                        _Float16* ctx, hipStream_t s) {
  constexpr size_t smem = KS == 1 && SSHIP_ATTN_REGFIN ? (size_t)4 * 32 * 72 * sizeof(_Float16) : (size_t)QT * 4 * 34 * 64 * sizeof(float);  // register finalisation: the four waves' 32 x 144-byte store patches
  constexpr int QPB = 32 * QT * (4 / KS);  // queries per workgroup
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lg_attention<QT, KS, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  (void)attr_rc;  // thread-safe one-time opt-in (magic static)
  unsigned long long* tbuf = nullptr;
  constexpr bool kWaveUnits = KS == 1 && SSHIP_ATTN_REGFIN;  // see the kernel: gx = query units per (sequence, head), one per wave
  const int gx = kWaveUnits ? (d.NP / 32 + QT - 1) / QT : (d.NP + QPB - 1) / QPB;
  const size_t nwg = kWaveUnits ? (size_t)gx * d.S : (size_t)gx * 4 * d.S;
  static const bool trace_on = SSHIP_ATTN_TRACE_BUILD && dev_env("SSHIP_ATTN_TRACE") != nullptr;
  if (trace_on) { (void)hipMalloc(&tbuf, nwg * 16 * 8); (void)hipMemsetAsync(tbuf, 0, nwg * 16 * 8, s); }
  static const bool xcd_off = dev_env("SUPERSLAM_HIP_ATTN_XCD") && atoi(dev_env("SUPERSLAM_HIP_ATTN_XCD")) == 0;  // A/B: plain id order
  hipLaunchKernelGGL((k_lg_attention<QT, KS, V>), dim3((nwg + 7) / 8 * 8), dim3(256), smem, s, q, k, vt, lens, d.NP,
                     cross ? 1 : 0, ctx, tbuf, gx, xcd_off ? -d.S : d.S);
  if (trace_on) {
    std::vector<unsigned long long> h(nwg * 16);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[3] = {0, 0, 0}; long cnt = 0;
    for (size_t w = 0; w < nwg * 4; ++w)
      if (h[w * 4 + 3]) { for (int i = 0; i < 3; ++i) sum[i] += (double)h[w * 4 + i]; ++cnt; }
    if (cnt) fprintf(stderr, "[attn trace QT=%d KS=%d cross=%d] prologue=%.0f key loop=%.0f merge=%.0f clk (%ld waves, %zu workgroups)\n", QT, KS,
                     (int)cross, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, cnt, nwg);
    (void)hipFree(tbuf);
  }
}
// softmax bookkeeping variant (see k_lg_attention): SUPERSLAM_HIP_ATTN_V = 1 | 2 | 3 (A/B runs); default kAttnV
constexpr int kAttnV = 3;
template <int QT, int KS>
static void launch_attn(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d, bool cross,
                        _Float16* ctx, hipStream_t s) {
  static const int v_env = dev_env("SUPERSLAM_HIP_ATTN_V") ? atoi(dev_env("SUPERSLAM_HIP_ATTN_V")) : kAttnV;
  if (v_env == 3) launch_attn_v<QT, KS, 3>(q, k, vt, lens, d, cross, ctx, s);
  else if (v_env == 2) launch_attn_v<QT, KS, 2>(q, k, vt, lens, d, cross, ctx, s);
  else launch_attn_v<QT, KS, 1>(q, k, vt, lens, d, cross, ctx, s);
}
void launch_lg_attention(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d, bool cross,
                         _Float16* ctx, hipStream_t s, bool shared_gpu) {
  // throughput batches: two query tiles per wave (half the K/V fragment traffic) and a 2-way key split; a few pairs only:
  // one tile per wave, 4-way key split, so the launch still has enough workgroups to cover the CUs (latency mode).
  // SUPERSLAM_HIP_ATTN_KS=4 keeps the 4-way split for throughput batches (A/B runs).
  static const int ks_env = dev_env("SUPERSLAM_HIP_ATTN_KS") ? atoi(dev_env("SUPERSLAM_HIP_ATTN_KS")) : 0;
  // SUPERSLAM_HIP_ATTN_QT=1 (A/B runs): one query tile per wave also for throughput batches - 144-156 VGPRs, three waves per SIMD
  // instead of two (a wave issues one VALU instruction per ~5 clocks, the SIMD retires one per 2: more waves = more VALU issue),
  // at twice the K / V^T fragment traffic per query
  static const int qt_env = dev_env("SUPERSLAM_HIP_ATTN_QT") ? atoi(dev_env("SUPERSLAM_HIP_ATTN_QT")) : 0;
#if SSHIP_DEV_SWITCHES  // SUPERSLAM_HIP_ATTN=res: the keys of a (sequence, head) resident in LDS (lg_attn_res.hip; measured slower: profiles/r05_a_*)
  static const bool res_env = dev_env("SUPERSLAM_HIP_ATTN") && std::string(dev_env("SUPERSLAM_HIP_ATTN")) == "res";
  if (res_env && d.S * (d.NP / 64) * 4 >= 2 * cu_count() && lg_attention_res_fits(d)) {
    launch_lg_attention_res(q, k, vt, lens, d, cross, ctx, s);
    return;
  }
#endif
  if (d.S * (d.NP / 64) * 4 >= 2 * cu_count()) {
    // shared_gpu: another stream runs the other half-batch's kernels next to this launch (lg_forward), so the partly filled last
    // round of a 256-query workgroup (no key split: no LDS merge, the prologue paid once per 19 key tiles) costs nothing:
    // 92 -> 101 us for a launch on its own, but -1.2 % on the two-stream LightGlue call
    const int ks = ks_env == 4 ? 4 : (ks_env == 1 || (ks_env == 0 && shared_gpu)) ? 1 : 2;
    if (qt_env == 1) {
      if (ks == 4) launch_attn<1, 4>(q, k, vt, lens, d, cross, ctx, s);
      else if (ks == 1) launch_attn<1, 1>(q, k, vt, lens, d, cross, ctx, s);
      else launch_attn<1, 2>(q, k, vt, lens, d, cross, ctx, s);
    } else if (ks == 4) launch_attn<2, 4>(q, k, vt, lens, d, cross, ctx, s);
    else if (ks == 1) launch_attn<2, 1>(q, k, vt, lens, d, cross, ctx, s);
    else launch_attn<2, 2>(q, k, vt, lens, d, cross, ctx, s);
  } else {
    launch_attn<1, 4>(q, k, vt, lens, d, cross, ctx, s);
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused FFN block: x += ffn.3( GELU( LayerNorm( ffn.0( cat[x, ctx] ) ) ) ), one launch per block.
//   * out_proj / to_out is folded into ffn.0 on the host:  W' = [W0a | W0b Wo],  b' = b0 + W0b bo
//     (cat[x, Wo ctx + bo] W0^T  ==  cat[x, ctx] W'^T + b'), so the attention output feeds the FFN directly.
//   * a workgroup (8 waves) owns 64 tokens; cat[x, ctx] (64 x 512 fp16) is staged once in LDS and is the MFMA
//     B operand of ffn.0; wave w owns output rows [64w, 64w+64) and streams its packed A fragments straight
//     from L2 (no sharing between waves -> no point in staging weights through LDS);
//   * LayerNorm statistics: lane-local over the accumulators, lane^32 exchange, then across the 8 waves via LDS
//     (two rounds: mean, then centred variance); exact-erf GELU on the accumulators;
//   * the activated hidden tile overwrites the LDS input tile and is the B operand of ffn.3 (wave w owns 32 output
//     rows); the residual add reads x from global in fp32 and writes it back in place.
// Removes per block: 2 kernel launches and the [T,512] hidden round trip (write + read + write + read).
// ---------------------------------------------------------------------------------------------------
//   * NEXT_MT > 0: the projection that consumes the updated x next (CrossBlock [to_qk|to_v] after a SelfBlock FFN,
//     the next layer's Wqkv after a CrossBlock FFN, final_proj + matchability after the last one) runs in the same
//     launch on the 64-token tile that is already on chip: wave w owns NEXT_MT*32 output rows, K = 256, epilogue =
//     the igemm epilogue of that projection (EpiHeads / plain fp16).
// compile-time ablation of the FFN kernel (build.py --variant -DSSHIP_FFN_ABL=n): 1 skip ffn.0 MFMAs, 2 skip LN/GELU
// math, 4 skip ffn.3, 8 skip the fused projection's MFMAs, 16 skip input staging, 32 skip the projection epilogue,
// 64 skip the residual update.  (Run-time flags put a branch in front of every GELU element.)
#ifndef SSHIP_FFN_ABL
#define SSHIP_FFN_ABL 0
#endif
#ifndef SSHIP_FFN_DEEP_PFK
#define SSHIP_FFN_DEEP_PFK 4  // k-steps of the fused projection requested before the residual phase (0, 4, 8; 16 for NEXT_MT <= 2)
#endif
#ifndef SSHIP_FFN_DEEP
#define SSHIP_FFN_DEEP 1  // build.py --variant -DSSHIP_FFN_DEEP=0: the latency-mode kernel without the deep register prefetch (A/B)
#endif
// NT = 32-token N-tiles per workgroup.  Every weight fragment a wave streams from L2 feeds NT MFMAs; at NT = 2 the
// three GEMMs need 64 B/clk/CU of L2 -> L1 bandwidth to keep the matrix pipe busy (= the TCP's peak, so the kernel
// was bound by the weight stream: 1.18 MB per 64 tokens).  NT = 4 halves the stream per token (throughput batches);
// NT = 2 keeps more workgroups in flight for a few pairs (latency mode).
// PROJ: projection only (the first layer's Wqkv has no FFN in front of it): stage the tile, run the fused projection.
// DEEP (NT = 1, latency mode: one 32-token tile per workgroup, one workgroup per CU, 38 CUs busy for one pair).  The launch is a serial
// chain  ffn.0 stream -> LayerNorm -> GELU -> ffn.3 stream -> residual -> projection stream -> epilogue  in which the three weight
// streams run at the L1's 64 B/clk (18 k clocks for 1.15 MB) and stand still during the VALU phases between them (13 k of a 29-34 k-clock
// tile, profiles/r04_j_*).  Measured (profiles/r04_s_*): requesting ffn.3's whole fragment stream (32 KB = 128 VGPRs per wave) before the
// LayerNorm statistics, or dealt out over the GELU loop four loads per step, only MOVES the time - the 8 x 32 one-KB loads occupy the
// CU's address path for ~4 k clocks and the issuing waves with it (LayerNorm 3.3 -> 6.1 k / GELU 3.7 -> 5.8 k, ffn.3 4.3 -> 2.1 k).
// What is free is small: the first SSHIP_FFN_DEEP_PFK k-steps of the projection requested before the residual phase, whose waits
// (barrier, residual operand) are not address-path time (tail MFMA 5.1 -> 4.25 k / 9.3 -> 7.6 k, residual unchanged), and barriers
// that wait for LDS traffic only (__syncthreads() drains vmcnt too and would stall every wave on its own prefetch).
__device__ __forceinline__ void ffn_bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NEXT_MT, bool HEADS, int NT, bool PROJ>
__global__ __launch_bounds__(512, 2) void k_lg_ffn(const _Float16* __restrict__ ctx, const _Float16* __restrict__ w0p,
                                                const float* __restrict__ b0, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const _Float16* __restrict__ w3p,
                                                const float* __restrict__ b3, _Float16* __restrict__ x, FfnTail tail) {
  constexpr int NTOK = NT * 32;
  constexpr bool DEEP = NT == 1 && !PROJ && SSHIP_FFN_DEEP;
  auto barrier = [&]() __attribute__((always_inline)) { if constexpr (DEEP) ffn_bar_lds(); else __syncthreads(); };
  // ffn.0 k-steps per register-prefetch group.  Two groups are in flight: 4 k-steps x 2 M-tiles x 2 groups = 16 one-KB loads per wave.
  // Round 4 (scripts/ubench/l2_stream.hip, profiles/r04_h_*): a CU streams an L2-resident weight set at 123-132 GB/s with 8 waves and
  // 8-16 loads in flight per wave, and at 56-62 GB/s with 32 - the rate HALVES past 16 per wave.  Groups of 8 k-steps (32 in flight)
  // are what held this kernel at 27-34 B/clk/CU in rounds 1-3.
  constexpr int G0 = 4;
  constexpr int XBUF = NTOK * kFfnLd;  // halfs per token-tile buffer
  extern __shared__ __attribute__((aligned(16))) char ffn_smem[];
  _Float16* s_xbuf = reinterpret_cast<_Float16*>(ffn_smem);                                   // [NBUF][NTOK][kFfnLd]
  float (*s_red)[NTOK] = reinterpret_cast<float (*)[NTOK]>(s_xbuf + (NT <= 2 ? 2 : 1) * XBUF);  // [16][NTOK]: per-wave sums, sums of squares
  // ffn.0 bias, LayerNorm gamma / beta, ffn.3 bias: the same for every tile of this persistent workgroup -> LDS once
  // (they were 28 dependent L2 round trips per lane and tile, right on the critical path between the GEMM phases)
  float* s_par = reinterpret_cast<float*>(s_red) + 16 * NTOK;  // [b0 512 | gamma 512 | beta 512 | b3 256]
  for (int i = threadIdx.x; i < (PROJ ? 0 : 1792); i += 512)
    s_par[i] = i < 512 ? b0[i] : i < 1024 ? gamma[i - 512] : i < 1536 ? beta[i - 1024] : b3[i - 1536];
  // Token-tile staging by LDS-DMA (global_load_lds_dwordx4): one instruction per token row - lanes 0..31 fetch the 32
  // 16-byte units of x[token], lanes 32..63 those of ctx[token]; the row lands lane-linear at its (padded) LDS row.
  // No staging registers, no ds_write pass, and - with two tile buffers (NT = 2) - the NEXT tile streams in while this
  // one is in its GELU / ffn.3 phases.  The DMA is invisible to hipcc's waitcnt bookkeeping: its completion is awaited
  // explicitly (vmcnt(0) where no weight prefetch is in flight), then a barrier, then the reads.
  auto stage_tile = [&](int tile, _Float16* dst, int wave, int lane) {
    if (SSHIP_FFN_ABL & 16) return;
    const size_t tt = (size_t)tile * NTOK;
#pragma unroll
    for (int k = 0; k < NTOK / 8; ++k) {
      const int tok = wave + 8 * k;
      const _Float16* gsrc = (lane < 32 ? x + (tt + tok) * 256 + lane * 8 : ctx + (tt + tok) * 256 + (lane - 32) * 8);
      const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dst + tok * kFfnLd));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  const int tile0 = blockIdx.x;
  const int nwg = tail.n_main > 0 ? tail.n_main : (int)gridDim.x;  // workgroups that walk the tiles
  if (tail.n_main > 0 && tile0 >= nwg) {
    // Prefetch role (latency mode; profiles/NOTES_r01_r04_design_history.md round 4).  One pair's launch covers 38 of 256 CUs and every layer has its own
    // 1.0-1.15 MB of weights (21 MB over the 18 blocks: no XCD's 4 MB L2 keeps them from one frame to the next), so each FFN launch used
    // to stream its weights at HBM / MALL latency: 19-23 us for a launch whose arithmetic is 3 us.  The surplus workgroups of THIS
    // launch read the NEXT launch's weights once per XCD - workgroup ids are dealt round-robin over the 8 XCDs, so prefetch
    // workgroup p serves XCD (n_main + p) % 8 and takes slice p / 8 of the (gridDim.x - n_main) / 8 slices; a different mapping
    // only changes which L2 gets warm, never a result.
    const int p = tile0 - nwg, nsl = ((int)gridDim.x - nwg) >> 3, sl = p >> 3;
    if (sl >= nsl) return;
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const char* base = static_cast<const char*>(tail.pf_ptr[r]);
      const int bytes = tail.pf_bytes[r];
      if (!base) continue;
#pragma unroll 4
      for (int off = (sl * 512 + (int)threadIdx.x) * 16; off < bytes; off += nsl * 512 * 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + off);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    if (acc == 0x9e3779b9u && tail.trace) tail.trace[0] = acc;  // keeps the loads alive; never true in practice and harmless if it is
    return;
  }
  if (tile0 >= tail.ntiles) return;
  stage_tile(tile0, s_xbuf, threadIdx.x >> 6, threadIdx.x & 63);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int it = 0;
#pragma unroll 1
  for (int tile = tile0; tile < tail.ntiles; tile += nwg, ++it) {
  _Float16* s_x = s_xbuf + (NT <= 2 ? (it & 1) : 0) * XBUF;
  _Float16* s_xn = s_xbuf + (NT <= 2 ? ((it + 1) & 1) : 0) * XBUF;
  const bool has_next = tile + nwg < tail.ntiles;
  const size_t t0 = (size_t)tile * NTOK;
  // Everything below that does not depend on the tile (weight fragments, biases, LayerNorm parameters) is loop
  // invariant: LICM would hoist those loads out of the tile loop and spill hundreds of registers.  An opaque zero
  // added to every such pointer keeps them inside the iteration.
  int zero = 0;
  asm volatile("" : "+s"(zero));
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // same for the lane-dependent address arithmetic (dozens of 64-bit offsets)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const _Float16 *w0q = w0p + zero, *w3q = w3p + zero;
  const float *b0q = s_par + zero, *gq = s_par + 512 + zero, *beq = s_par + 1024 + zero, *b3q = s_par + 1536 + zero;
  IgemmArgs pj = tail.proj;
  pj.wpack += zero; pj.bias += zero;
  const float* mwq = tail.match_w + zero;
  const _Float16* bfp = s_x + j * kFfnLd + hh * 8;  // B fragment of N-tile n, k-step ks: bfp + n*32*kFfnLd + ks*16
  auto stamp = [&](int slot) {
    if (tail.trace && it == tail.trace_it && lane == 0) tail.trace[((size_t)blockIdx.x * 8 + wave) * 12 + slot] = __builtin_readcyclecounter();
  };
  stamp(0);
  constexpr int PFK = !DEEP || NEXT_MT == 0 ? 0 : SSHIP_FFN_DEEP_PFK;  // DEEP: k-steps of the fused projection prefetched into atp
  h8_t atp[PFK > 0 ? PFK : 1][NEXT_MT > 0 ? NEXT_MT : 1];
  if constexpr (!PROJ) {
  // ---- ffn.0 : rows [64 wave, +64) x NTOK tokens, K = 512 ----
  f16x_t acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  {
    // Weight fragments stream from L2 (no reuse between waves); keep TWO groups of G0 k-steps in flight in registers
    // so ~1k cycles of L2 latency are covered by the MFMAs of the previous group and the co-resident wave.
    const _Float16* wp = w0q + (size_t)wave * (32 * 2 * 512) + lane * 8;  // packed [cb = wave][k16][mt][lane][8]
    h8_t ab[2][G0][2];
#pragma unroll
    for (int i = 0; i < G0; ++i) {
      ab[0][i][0] = *reinterpret_cast<const h8_t*>(wp + (i * 2 + 0) * 512);
      ab[0][i][1] = *reinterpret_cast<const h8_t*>(wp + (i * 2 + 1) * 512);
    }
#pragma unroll
    for (int grp = 0; grp < 32 / G0; ++grp) {
      if (SSHIP_FFN_ABL & 1) break;
      if (grp + 1 < 32 / G0) {
#pragma unroll
        for (int i = 0; i < G0; ++i) {
          ab[(grp + 1) & 1][i][0] = *reinterpret_cast<const h8_t*>(wp + (((grp + 1) * G0 + i) * 2 + 0) * 512);
          ab[(grp + 1) & 1][i][1] = *reinterpret_cast<const h8_t*>(wp + (((grp + 1) * G0 + i) * 2 + 1) * 512);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next group's loads ABOVE this group's MFMAs (hipcc sinks them otherwise)
#pragma unroll
      for (int i = 0; i < G0; ++i) {
        const int ks = grp * G0 + i;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const h8_t bf = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd + ks * 16);
          acc[0][n] = mfma32(ab[grp & 1][i][0], bf, acc[0][n]);
          acc[1][n] = mfma32(ab[grp & 1][i][1], bf, acc[1][n]);
        }
      }
    }
  }
  stamp(1);
  // ---- bias, LayerNorm(512) over the row dimension (spread over regs, lane^32 and the 8 waves), GELU ----
  float sum[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) sum[n] = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(b0q + wave * 64 + m * 32 + hh * 4 + g * 8);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[m][n][4 * g + 0] += bv.x; acc[m][n][4 * g + 1] += bv.y; acc[m][n][4 * g + 2] += bv.z; acc[m][n][4 * g + 3] += bv.w;
        sum[n] += (acc[m][n][4 * g + 0] + acc[m][n][4 * g + 1]) + (acc[m][n][4 * g + 2] + acc[m][n][4 * g + 3]);
      }
    }
  // one pass: sum and sum of squares together (one barrier round instead of two; 512 fp32 terms of O(1) magnitude)
  float mean[NT], rstd[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float sq = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) sq = fmaf(acc[m][n][r], acc[m][n][r], sq);
    sum[n] += __shfl_xor(sum[n], 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (hh == 0) { s_red[wave][n * 32 + j] = sum[n]; s_red[8 + wave][n * 32 + j] = sq; }
  }
  barrier();
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float t = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { t += s_red[w][n * 32 + j]; q += s_red[8 + w][n * 32 + j]; }
    mean[n] = t * (1.0f / 512.0f);
    rstd[n] = __builtin_amdgcn_rsqf(fmaxf(q * (1.0f / 512.0f) - mean[n] * mean[n], 0.f) + 1e-5f);
  }
  stamp(2);
  // every wave has passed the LayerNorm barriers, i.e. finished the previous tile: its buffer takes the next tile.
  // Issued here because no weight prefetch is in flight (an older DMA would sit in front of it in the in-order vmcnt
  // queue) and the GELU math + ffn.3 that follow cover the HBM latency.
  if (NT <= 2 && has_next) stage_tile(tile + nwg, s_xn, wave, lane);
  // the residual operand (this lane's 16 x values per N-tile) is requested before the GELU math: its L2 round trip
  // used to sit between the ffn.3 loop and the barrier that opens the fused projection
  h4_t xres[4][NT];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int n = 0; n < NT; ++n)
      xres[g][n] = *reinterpret_cast<const h4_t*>(x + (t0 + n * 32 + j) * 256 + wave * 32 + hh * 4 + g * 8 + zero);
  // every wave has passed two barriers since its last read of s_x: the tile can be overwritten with the hidden tile
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = wave * 64 + m * 32 + hh * 4 + g * 8;
      const float4 gv = *reinterpret_cast<const float4*>(gq + c);
      const float4 be = *reinterpret_cast<const float4*>(beq + c);
      const f2_t g01 = {gv.x, gv.y}, g23 = {gv.z, gv.w}, b01 = {be.x, be.y}, b23 = {be.z, be.w};
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const f2_t a01 = {acc[m][n][4 * g + 0], acc[m][n][4 * g + 1]}, a23 = {acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
        const f2_t o01 = (SSHIP_FFN_ABL & 2) ? a01 : gelu2((a01 - mean[n]) * rstd[n] * g01 + b01);
        const f2_t o23 = (SSHIP_FFN_ABL & 2) ? a23 : gelu2((a23 - mean[n]) * rstd[n] * g23 + b23);
        *reinterpret_cast<h4_t*>(s_x + (n * 32 + j) * kFfnLd + c) = to_h4(o01[0], o01[1], o23[0], o23[1]);
      }
    }
  stamp(3);
  barrier();
  stamp(4);
  // ---- ffn.3 : rows [32 wave, +32) x NTOK tokens, K = 512, + residual ----
  f16x_t ac2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) ac2[n][r] = 0.f;
  {
    const _Float16* wp = w3q + (size_t)wave * (32 * 512) + lane * 8;  // packed [cb = wave][k16][mt = 0][lane][8]
    // four groups of 8 k-steps, two in flight (16 loads per wave: see G0)
    h8_t a3[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a3[0][i] = *reinterpret_cast<const h8_t*>(wp + i * 512);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      if (SSHIP_FFN_ABL & 4) break;
      if (grp + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a3[(grp + 1) & 1][i] = *reinterpret_cast<const h8_t*>(wp + ((grp + 1) * 8 + i) * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ks = grp * 8 + i;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const h8_t bf = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd + ks * 16);
          ac2[n] = mfma32(a3[grp & 1][i], bf, ac2[n]);
        }
      }
    }
  }
  stamp(5);
  if (!DEEP || has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next tile has landed (before the barrier below / the next tile's first barrier)
  // DEEP: the first PFK k-steps of the fused projection, requested before the residual phase (the ffn.3 fragments are dead now)
  if constexpr (PFK > 0) {
    const _Float16* wpp = pj.wpack + (size_t)wave * (16 * NEXT_MT * 512) + lane * 8;
#pragma unroll
    for (int i = 0; i < PFK; ++i)
#pragma unroll
      for (int m = 0; m < NEXT_MT; ++m) atp[i][m] = *reinterpret_cast<const h8_t*>(wpp + (i * NEXT_MT + m) * 512);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (NEXT_MT > 0) barrier();  // all waves are done reading the hidden tile: s_x gets the new x
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = wave * 32 + hh * 4 + g * 8;
    const float4 bv = *reinterpret_cast<const float4*>(b3q + c);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      h4_t* px = reinterpret_cast<h4_t*>(x + (t0 + n * 32 + j) * 256 + c);
      if (SSHIP_FFN_ABL & 64) continue;
      const h4_t o = xres[g][n];
      const h4_t xn = to_h4((float)o[0] + (ac2[n][4 * g + 0] + bv.x), (float)o[1] + (ac2[n][4 * g + 1] + bv.y),
                            (float)o[2] + (ac2[n][4 * g + 2] + bv.z), (float)o[3] + (ac2[n][4 * g + 3] + bv.w));
      *px = xn;
      if constexpr (NEXT_MT > 0) *reinterpret_cast<h4_t*>(s_x + (n * 32 + j) * kFfnLd + c) = xn;
    }
  }
  stamp(6);
  }  // !PROJ
  if constexpr (NEXT_MT > 0) {
    barrier();
    stamp(7);
    // ---- fused next projection: rows [NEXT_MT*32*wave, +NEXT_MT*32) x NTOK tokens, K = 256 ----
    f16x_t ac3[NEXT_MT][NT];
#pragma unroll
    for (int m = 0; m < NEXT_MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) ac3[m][n][r] = 0.f;
    const _Float16* wp = pj.wpack + (size_t)wave * (16 * NEXT_MT * 512) + lane * 8;  // [cb = wave][k16][mt][lane][8]
    // M-tiles of the V segment run with SWAPPED operands (A = token tile, B = weights): the accumulator then holds
    // D[token][channel] with lane = channel and 8 consecutive registers = the 8 keys of one PV A-fragment unit, so V^T is
    // written in fragment order with one 16-byte store per lane (the 2-byte transposing stores it replaces were ~16x
    // write-amplified and dominated the kernel's non-MFMA time).
    // The projection's rows are packed tile-interleaved (upload_conv): M-tile m of this wave is rows
    // (8 m + wave) * 32 .. + 31, i.e. one tile of each 256-row segment (self: q | k | v, cross: qk | v).  Every wave
    // has the same mix of rope / plain / transposed-V epilogues, and the V tile is always the last one.
    auto run_tail = [&](auto vmask_c) {
      constexpr int VMASK = decltype(vmask_c)::value;
      // weight fragments: two groups of GT k-steps in flight in registers, pinned above the MFMAs of the previous
      // group (a plain unrolled loop made hipcc wait for every fragment right before its MFMA: 28k clocks for 96 MFMAs)
      constexpr int GT = NEXT_MT >= 3 ? 2 : 4;  // 2 GT NEXT_MT <= 16 loads in flight per wave (see G0)
      // rotary (cos, sin) quads of this lane's q / k rows and the q / k biases: requested HERE, consumed after the MFMA loop.  (Round 4:
      // the generic igemm epilogue loaded them inside its store loops, behind its own stores in the in-order vmcnt queue - 9.8 k clocks
      // for the Wqkv epilogue of a one-pair launch, profiles/r04_j_*.)  The q and the k tile of a wave cover the same head-local
      // channels ((8 m + wave) * 32 mod 64 does not depend on m): one rotary set serves both.
      constexpr int NQK = HEADS ? NEXT_MT - 1 : 0;   // M-tiles with the q / k epilogue (the last one is V)
      const int rope_segs_t = pj.flags & 0xf;
      float4 cs[4][NT], bqk[NQK > 0 ? NQK : 1][4];
      if constexpr (HEADS) {
        if (rope_segs_t > 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              cs[g][n] = *reinterpret_cast<const float4*>(pj.aux + (t0 + n * 32 + j) * 64 + ((wave * 32) & 63) + hh * 4 + g * 8 + zero);
        }
#pragma unroll
        for (int m = 0; m < NQK; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g) bqk[m][g] = *reinterpret_cast<const float4*>(pj.bias + (m * 8 + wave) * 32 + hh * 4 + g * 8);
      }
      constexpr int NG = (16 - PFK) / GT;  // streamed groups behind the PFK prefetched k-steps (DEEP; PFK = 0 otherwise)
      h8_t at[2][GT][NEXT_MT];
      if constexpr (NG > 0) {
#pragma unroll
        for (int i = 0; i < GT; ++i)
#pragma unroll
          for (int m = 0; m < NEXT_MT; ++m) at[0][i][m] = *reinterpret_cast<const h8_t*>(wp + ((PFK + i) * NEXT_MT + m) * 512);
      }
      auto kstep = [&](int ks, const h8_t (&a)[NEXT_MT]) __attribute__((always_inline)) {
        h8_t bf[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd + ks * 16);
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if ((VMASK >> m) & 1) ac3[m][n] = mfma32(bf[n], a[m], ac3[m][n]);
            else ac3[m][n] = mfma32(a[m], bf[n], ac3[m][n]);
          }
        }
      };
      if constexpr (PFK > 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(SSHIP_FFN_ABL & 8)) {
#pragma unroll
          for (int i = 0; i < PFK; ++i) kstep(i, atp[i]);
        }
      }
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        if (SSHIP_FFN_ABL & 8) break;
        if (grp + 1 < NG) {
#pragma unroll
          for (int i = 0; i < GT; ++i)
#pragma unroll
            for (int m = 0; m < NEXT_MT; ++m)
              at[(grp + 1) & 1][i][m] = *reinterpret_cast<const h8_t*>(wp + ((PFK + (grp + 1) * GT + i) * NEXT_MT + m) * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < GT; ++i) kstep(PFK + grp * GT + i, at[grp & 1][i]);
      }
      stamp(8);
      if (PROJ && NT <= 2 && has_next) stage_tile(tile + nwg, s_xn, wave, lane);  // the epilogue covers the DMA
      if (SSHIP_FFN_ABL & 32) return;
      if constexpr (HEADS) {
        const int NP = pj.np, nt32 = NP >> 5;
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) {
          const int R0 = (HEADS ? (m * 8 + wave) : (wave * NEXT_MT + m)) * 32;  // first output row of this M-tile
          if ((VMASK >> m) & 1) {
            const int hd = (R0 >> 6) & 3, mth = (R0 >> 5) & 1;  // head, 32-channel half of the head
            const float bv = pj.bias[R0 + j];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const size_t token = t0 + n * 32;
              const int sq = (int)(token / NP), kt = (int)(token - (size_t)sq * NP) >> 5;
              _Float16* dst = static_cast<_Float16*>(pj.out2) + (((size_t)sq * 4 + hd) * nt32 + kt) * 2048 + lane * 8;
#pragma unroll
              for (int kk = 0; kk < 2; ++kk) {
                h8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)(ac3[m][n][8 * kk + e] + bv);
                *reinterpret_cast<h8_t*>(dst + (kk * 2 + mth) * 512) = o;
              }
            }
          } else {
            // q / k (or the shared qk of CrossBlock): bias, rotary on interleaved pairs, fp16, then lane ^ 32 pairing so that every lane
            // owns whole 16-byte fragment units (k_lg_ffn4's epilogue): unit u = d / 8 -> [kstep u / 2][lane' = (u & 1) * 32 + token % 32][8]
            const int seg = R0 >> 8, hd = (R0 >> 6) & 3;
            const bool roped = seg < rope_segs_t;
            _Float16* obase = static_cast<_Float16*>(seg == 0 ? pj.out0 : pj.out1);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const size_t token = t0 + n * 32;
              const int sq = (int)(token / NP), kt = (int)(token - (size_t)sq * NP) >> 5;
              _Float16* dst = obase + (((size_t)sq * 4 + hd) * nt32 + kt) * 2048;
              unsigned lo[4], hi[4];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float4 bv = bqk[m < NQK ? m : 0][g];
                float v0 = ac3[m][n][4 * g + 0] + bv.x, v1 = ac3[m][n][4 * g + 1] + bv.y;
                float v2 = ac3[m][n][4 * g + 2] + bv.z, v3 = ac3[m][n][4 * g + 3] + bv.w;
                if (roped) {
                  const float4 c = cs[g][n];
                  const float r0 = v0 * c.x - v1 * c.y, r1 = v1 * c.x + v0 * c.y;
                  const float r2 = v2 * c.z - v3 * c.w, r3 = v3 * c.z + v2 * c.w;
                  v0 = r0; v1 = r1; v2 = r2; v3 = r3;
                }
                const h2_t p01 = {(_Float16)v0, (_Float16)v1}, p23 = {(_Float16)v2, (_Float16)v3};
                lo[g] = __builtin_bit_cast(unsigned, p01);
                hi[g] = __builtin_bit_cast(unsigned, p23);
              }
#pragma unroll
              for (int gp = 0; gp < 2; ++gp) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(lo[2 * gp], lo[2 * gp + 1], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(hi[2 * gp], hi[2 * gp + 1], false, false);
                const int u = ((R0 & 63) >> 3) + 2 * gp + hh;
                *reinterpret_cast<uint4*>(dst + (u >> 1) * 512 + (((u & 1) << 5) + j) * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
              }
            }
          }
        }
      } else {
        EpiF16<false, false>::template run<NEXT_MT, NT>(pj, ac3, 0, (int)(t0 >> 5), j, wave * NEXT_MT * 32, hh);
      }
    };
    if constexpr (HEADS) run_tail(std::integral_constant<int, 1 << (NEXT_MT - 1)>{});
    else run_tail(std::integral_constant<int, 0>{});
    if (tail.logsig) {  // matchability head of the last block: one wave per NTOK / 8 tokens
#pragma unroll 1
      for (int tk = wave * (NTOK / 8); tk < (wave + 1) * (NTOK / 8); ++tk) {
        const h4_t v = *reinterpret_cast<const h4_t*>(s_x + tk * kFfnLd + lane * 4);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) d += (float)v[e] * mwq[lane * 4 + e];
        const float z = wave_sum(d) + tail.match_b;
        if (lane == 0) tail.logsig[t0 + tk] = fminf(z, 0.f) - log1pf(expf(-fabsf(z)));
      }
    }
  }
  stamp(9);
  if (NT > 2 && has_next) {  // single tile buffer: synchronous restage
    __syncthreads();
    stage_tile(tile + nwg, s_xn, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (PROJ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (has_next) __syncthreads();  // DMA data (awaited per wave above) visible to every wave; s_red / tile buffers quiescent
  stamp(10);
  }  // tile loop
}
template <int NEXT_MT, bool HEADS, int NT, bool PROJ, typename... A>
static hipError_t launch_ffn_nt(int tokens, int extra_wg, hipStream_t s, A... args) {
  constexpr size_t smem = (size_t)(NT <= 2 ? 2 : 1) * NT * 32 * kFfnLd * 2 + 16 * NT * 32 * 4 + 1792 * 4;
  static_assert(smem <= 163840, "LDS budget");
  auto kern = k_lg_ffn<NEXT_MT, HEADS, NT, PROJ>;
  // thread-safe one-time opt-in to > 64 KiB of dynamic LDS (C++11 magic static; handles may be created on any thread)
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_rc != hipSuccess) return attr_rc;
  const int ntiles = tokens / (NT * 32);
  hipLaunchKernelGGL(kern, dim3((ntiles < cu_count() ? ntiles : cu_count()) + extra_wg), dim3(512), smem, s, args...);  // one persistent workgroup per CU (+ the prefetch workgroups of latency mode)
  return hipGetLastError();
}
// SSHIP_FFN_TRACE=1 (developer aid): mean shader-clock duration of every phase of a workgroup's second tile.
static void ffn_trace_report(unsigned long long* dev, int nwg, int next_mt, hipStream_t s) {
  std::vector<unsigned long long> h((size_t)nwg * 8 * 12);
  (void)hipStreamSynchronize(s);
  (void)hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
  static const char* names[10] = {"ffn.0", "LN stats", "DMA issue+GELU", "barrier", "ffn.3", "vmcnt+barrier+residual", "barrier", "tail MFMA", "tail epilogue", "end barrier"};
  double sum[10] = {0}; long cnt = 0;
  for (int w = 0; w < nwg * 8; ++w) {
    const unsigned long long* t = h.data() + (size_t)w * 12;
    if (!t[0] || !t[10]) continue;
    for (int i = 0; i < 10; ++i) sum[i] += (double)(t[i + 1] > t[i] ? t[i + 1] - t[i] : 0);
    ++cnt;
  }
  if (!cnt) return;
  fprintf(stderr, "[ffn trace next_mt=%d, %ld waves]", next_mt, cnt);
  double tot = 0;
  for (int i = 0; i < 10; ++i) { fprintf(stderr, " %s=%.0f", names[i], sum[i] / cnt); tot += sum[i] / cnt; }
  fprintf(stderr, " | tile=%.0f clk\n", tot);
}
template <int NEXT_MT, bool HEADS, typename... A>
static hipError_t launch_ffn(int nt, int tokens, int extra_wg, hipStream_t s, A... args) {
  // 64-token tiles for throughput, 32-token tiles when the launch cannot even give half of the CUs a workgroup (a few
  // pairs: twice the workgroups in flight, half the MFMA work per weight stream).  128 tokens measured 9 % slower end to
  // end (1.25 tiles per CU at 32 pairs, single tile buffer) and no longer fits the registers: not instantiated.
  return nt == 1 ? launch_ffn_nt<NEXT_MT, HEADS, 1, false>(tokens, extra_wg, s, args...) : launch_ffn_nt<NEXT_MT, HEADS, 2, false>(tokens, extra_wg, s, args...);
}
// ---------------------------------------------------------------------------------------------------
// k_lg_ffn4: the same fused block with FOUR waves per workgroup and TWO workgroups per CU (throughput batches).
//
// The 8-wave kernel above runs one workgroup per CU whose waves move through  ffn.0 -> LayerNorm -> GELU -> ffn.3 ->
// residual -> projection -> epilogue  in lock-step: its phase trace (profiles/r01_v12_ffn_phase_trace.txt) shows the
// matrix pipe busy for ~16 k of a 56-71 k-clock tile - it idles whenever all eight waves are in a VALU / LDS / store phase.
// Here a workgroup is 4 waves (one per SIMD) that own a 64-token tile on their own - wave w computes 128 ffn.0 rows,
// 64 ffn.3 rows and two 32*NEXT_MT-row blocks of the fused projection - and keeps ONE tile buffer (66.5 KB), so two
// workgroups fit a CU (2 x 75.8 KB of LDS, 2 waves per SIMD).  The two workgroups of a CU are independent, drift apart
// and run complementary phases most of the time: one wave's MFMA stream rides beside the other's GELU / epilogue VALU
// on the same SIMD (MI355X_MICROARCH.md: the MFMA and VALU pipes of a SIMD are separate).  Per-wave register tiles are
// twice as large (4 M-tiles x 2 N-tiles for ffn.0), so every LDS B-fragment read feeds twice as many MFMAs as before.
// Weight bytes streamed from L2 per token are unchanged (1.18 MB per 64-token tile).
//   * the rotary table of the q / k epilogue is prefetched into registers BEFORE the projection's MFMA loop (it was 8
//     dependent L2 round trips per M-tile inside the epilogue, the longest phase of the CrossBlock kernel);
//   * q / k rows are written as 16-byte fragment units: v_permlane32_swap pairs the two half-waves' 4-channel quads
//     (they were 8-byte stores);
//   * the residual operand is read from the x half of the LDS tile before it is overwritten (no global re-read);
//   * the next tile's LDS-DMA is issued after the last epilogue: its latency is covered by the other workgroup.
// ---------------------------------------------------------------------------------------------------
// epilogue stores as buffer stores: 1 = x, 2 = V^T, 4 = q / k.  The V^T variant (2) produces wrong matches on the GPU although its
// ISA reads correctly (bisected with this switch; not understood - suspect the same soffset handling in hipcc 7.2 that the empty
// asm in wstore works around), so V^T keeps its flat stores: 4 of the 27 stores of a tile.
#ifndef SSHIP_FFN4_XROWS
#define SSHIP_FFN4_XROWS 1  // the residual stream's new rows are stored from the LDS tile as whole 512-byte rows (0: 32-byte pieces straight from the accumulators; A/B: LightGlue call -2 %, bit-identical, profiles/r05_c_*)
#endif
#ifndef SSHIP_FFN4_BSTORE
#define SSHIP_FFN4_BSTORE 5
#endif
// Energy ablations of the throughput kernel (build.py --variant ... -DSSHIP_FFN4_ABL=n, scripts/dev/energy_abl.sh ffn: rocm-smi power x launch
// time per variant): 1 no MFMAs (operands still delivered), 2 no LayerNorm / GELU math, 4 every weight fragment read from offset 0 (the
// loads of a phase collapse into one: no L2 -> register stream), 8 no projection epilogue (rotary, conversion, stores).  Results are wrong by design.
#ifndef SSHIP_FFN4_ABL
#define SSHIP_FFN4_ABL 0
#endif
__device__ __forceinline__ f16x_t mfma32_abl(h8_t a, h8_t b, f16x_t c) {
  if constexpr ((SSHIP_FFN4_ABL & 1) != 0) {
    asm volatile("" :: "v"(a), "v"(b));
    return c;
  } else {
    return mfma32(a, b, c);
  }
}
template <int NEXT_MT, bool HEADS, bool PROJ>
__global__ __launch_bounds__(256, 2) void k_lg_ffn4(const _Float16* __restrict__ ctx, const _Float16* __restrict__ w0p,
                                                    const float* __restrict__ b0, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const _Float16* __restrict__ w3p,
                                                    const float* __restrict__ b3, _Float16* __restrict__ x, FfnTail tail) {
  constexpr int NT = 2, NTOK = 64;
  extern __shared__ __attribute__((aligned(16))) char ffn_smem[];
  _Float16* s_x = reinterpret_cast<_Float16*>(ffn_smem);                            // [NTOK][kFfnLd]
  float (*s_red)[NTOK] = reinterpret_cast<float (*)[NTOK]>(s_x + NTOK * kFfnLd);     // [8][NTOK]: per-wave sums, sums of squares
  float* s_par = reinterpret_cast<float*>(s_red) + 8 * NTOK;                         // [b0 512 | gamma 512 | beta 512 | b3 256]
  float* s_pb = s_par + 1792;                                                        // bias of the fused projection [<= 768]
  float* s_b0b = s_pb + 768;  // second copy of b0: the accumulators of both N-tiles start from the bias, one ds_read each (no copies)
  for (int i = threadIdx.x; i < (PROJ ? 0 : 1792); i += 256)
    s_par[i] = i < 512 ? b0[i] : i < 1024 ? gamma[i - 512] : i < 1536 ? beta[i - 1024] : b3[i - 1536];
  for (int i = threadIdx.x; i < (PROJ ? 0 : 512); i += 256) s_b0b[i] = b0[i];
  // LDS, not global: a global load in the epilogue sits behind the epilogue's own stores in the in-order vmcnt queue
  // (the first version of this kernel spent 16 k clocks per CrossBlock tile there)
  if constexpr (NEXT_MT > 0 && HEADS)
    for (int i = threadIdx.x; i < NEXT_MT * 256; i += 256) s_pb[i] = tail.proj.bias[i];
  // token rows wave, wave + 4, ... of the tile: the lane's source address is formed once and advanced by 4 KiB per PAIR of rows
  // (the odd row of a pair rides on the instruction offset, which applies to the global and the LDS address alike - M0 is set
  // 2 KiB low for it): 8 address updates per tile instead of 16 selects + 64-bit adds in a kernel that runs at VALU issue rate
  auto stage_tile = [&](int tile, int wave, int lane) {
    const size_t tt = (size_t)tile * NTOK;
    const char* gsrc = reinterpret_cast<const char*>((lane < 32 ? x + lane * 8 : ctx + (lane - 32) * 8) + (tt + wave) * 256);
#pragma unroll
    for (int k = 0; k < NTOK / 4; k += 2) {
      const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_x + (wave + 4 * k) * kFfnLd));
      const unsigned lds1 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_x + (wave + 4 * k + 4) * kFfnLd)) - 2048u;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                   "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds0), "s"(lds1) : "memory");
      gsrc += 4096;
    }
  };
  const int tile0 = blockIdx.x;
  if (tile0 >= tail.ntiles) return;
  stage_tile(tile0, threadIdx.x >> 6, threadIdx.x & 63);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int it = 0;
#pragma unroll 1
  for (int tile = tile0; tile < tail.ntiles; tile += gridDim.x, ++it) {
  const bool has_next = tile + (int)gridDim.x < tail.ntiles;
  const size_t t0 = (size_t)tile * NTOK;
  int zero = 0;
  asm volatile("" : "+s"(zero));  // keeps loop-invariant weight / parameter loads inside the iteration (see k_lg_ffn)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const _Float16 *w0q = w0p + zero, *w3q = w3p + zero;
  const float *b0q = s_par + zero, *gq = s_par + 512 + zero, *beq = s_par + 1024 + zero, *b3q = s_par + 1536 + zero;
  IgemmArgs pj = tail.proj;
  pj.wpack += zero; pj.bias += zero;
  const float* mwq = tail.match_w + zero;
  const _Float16* bfp = s_x + j * kFfnLd + hh * 8;  // B fragment of N-tile n, k-step ks: bfp + n*32*kFfnLd + ks*16
  auto stamp = [&](int slot) {
    if (tail.trace && it == tail.trace_it && lane == 0) tail.trace[((size_t)blockIdx.x * 8 + wave) * 12 + slot] = __builtin_readcyclecounter();
  };
  // Weight fragments are fetched with buffer loads: one lane-offset VGPR (lane * 16 B) for every load of the kernel, the
  // fragment's position as a scalar offset.  With flat 64-bit pointers hipcc spent ~230 VALU instructions per tile on address
  // arithmetic (v_lshl_add_u64, v_add_co / v_addc pairs), and this kernel runs at VALU issue rate (2.7 k VALU instructions
  // against 448 MFMAs per wave and tile, two waves per SIMD).
  typedef unsigned wq_t __attribute__((ext_vector_type(4)));
  const unsigned lane16 = (unsigned)lane * 16u;
  auto wres = [&](const _Float16* base) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, (int)0x7ffffff0, 0x00020000);
  };
  auto wload = [&](__amdgpu_buffer_rsrc_t r, int halfs) __attribute__((always_inline)) {  // fragment at base + halfs (+ lane * 8)
    return __builtin_bit_cast(h8_t, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, ((SSHIP_FFN4_ABL & 4) ? 0 : halfs * 2) + zero, 0));
  };
  // the epilogue's 16-byte stores the same way: lane offset in one VGPR, everything else scalar (q / k / V^T tiles are in fragment
  // order, lane * 16 B again; x rows: token j -> j * 512 B + 16 hh)
  // (the scalar offset goes through an empty asm: hipcc 7.2 otherwise gives raw buffer STORES whose soffsets differ by a constant
  // the same soffset register and drops the constant - two stores land on one address; loads are not affected)
  auto wstore = [&](__amdgpu_buffer_rsrc_t r, unsigned voff, long long halfs, wq_t v) __attribute__((always_inline)) {
    int so = (int)(halfs * 2);
    asm volatile("" : "+s"(so));
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, so, 0);
  };
  const unsigned xrow16 = (unsigned)j * 512u + (unsigned)hh * 16u;
  const __amdgpu_buffer_rsrc_t rx = wres(x), rq = wres(static_cast<const _Float16*>(tail.proj.out0)),
                               rk = wres(static_cast<const _Float16*>(tail.proj.out1)), rv = wres(static_cast<const _Float16*>(tail.proj.out2));
  stamp(0);
  if constexpr (!PROJ) {
  // ---- ffn.0 : rows [128 wave, +128) x 64 tokens, K = 512.  Fragment f = 2 * (64-row block) + m-tile ----
  // the accumulators start from the ffn.0 bias (row 8 g + 4 hh + e of fragment f is register 4 g + e): 128 v_add_f32 per lane and
  // tile less in the statistics phase, which - like GELU - runs at VALU issue rate next to the other workgroup's MFMAs
  f16x_t acc[4][NT];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>((n ? s_b0b + zero : b0q) + wave * 128 + f * 32 + hh * 4 + g * 8);
        acc[f][n][4 * g + 0] = bv.x; acc[f][n][4 * g + 1] = bv.y; acc[f][n][4 * g + 2] = bv.z; acc[f][n][4 * g + 3] = bv.w;
      }
  {
    // Weight fragments stream from L2 through a ring of R0 k-steps of registers: the load of k-step ks + R0 - 1 is issued
    // before the MFMAs of k-step ks, so every fragment has (R0 - 1) x 8 MFMAs (~1 k clocks) to arrive - an L2 hit under load
    // takes 600-900 clocks, and the two-group scheme of k_lg_ffn (one group = 512 clocks of cover) left this kernel's MFMA
    // phases at half rate.  The token-tile B fragments are read from LDS one k-step ahead.
    constexpr int R0 = 4;
    const __amdgpu_buffer_rsrc_t r0 = wres(w0p);  // packed [cb = 2 wave + (f >> 1)][k16][mt = f & 1][lane][8]
    const int w0off = (2 * wave) * (32 * 2 * 512);
    auto frag = [&](int ks, int f) __attribute__((always_inline)) { return wload(r0, w0off + (f >> 1) * (32 * 2 * 512) + (ks * 2 + (f & 1)) * 512); };
    h8_t ab[R0][4], bf[2][NT];
#pragma unroll
    for (int i = 0; i < R0 - 1; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f) ab[i][f] = frag(i, f);
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      if (ks + R0 - 1 < 32) {
#pragma unroll
        for (int f = 0; f < 4; ++f) ab[(ks + R0 - 1) % R0][f] = frag(ks + R0 - 1, f);
      }
      if (ks + 1 < 32) {
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[(ks + 1) & 1][n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd + (ks + 1) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);  // the loads above stay ABOVE this k-step's MFMAs
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[f][n] = mfma32_abl(ab[ks % R0][f], bf[ks & 1][n], acc[f][n]);
    }
  }
  stamp(1);
  // ---- LayerNorm(512) statistics (regs -> lane^32 -> the 4 waves through LDS); the bias is already in ----
  float sum[NT], sq[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { sum[n] = 0.f; sq[n] = 0.f; }
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        sum[n] += (acc[f][n][4 * g + 0] + acc[f][n][4 * g + 1]) + (acc[f][n][4 * g + 2] + acc[f][n][4 * g + 3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) sq[n] = fmaf(acc[f][n][4 * g + e], acc[f][n][4 * g + e], sq[n]);
      }
    }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    sum[n] += __shfl_xor(sum[n], 32, 64);
    sq[n] += __shfl_xor(sq[n], 32, 64);
    if (hh == 0) { s_red[wave][n * 32 + j] = sum[n]; s_red[4 + wave][n * 32 + j] = sq[n]; }
  }
  // the residual operand (this lane's ffn.3 rows of x) comes from the x half of the tile, before GELU overwrites it
  h4_t xres[2][4][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        xres[m][g][n] = *reinterpret_cast<const h4_t*>(s_x + (n * 32 + j) * kFfnLd + wave * 64 + m * 32 + hh * 4 + g * 8);
  __syncthreads();  // statistics complete; every wave has finished reading the input tile
  float mean[NT], rstd[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float t = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { t += s_red[w][n * 32 + j]; q += s_red[4 + w][n * 32 + j]; }
    mean[n] = t * (1.0f / 512.0f);
    rstd[n] = __builtin_amdgcn_rsqf(fmaxf(q * (1.0f / 512.0f) - mean[n] * mean[n], 0.f) + 1e-5f);
    mean[n] = -mean[n] * rstd[n];  // LayerNorm as two fmas per value: (a rstd - mean rstd) gamma + beta
  }
  stamp(2);
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = wave * 128 + f * 32 + hh * 4 + g * 8;
      const float4 gv = *reinterpret_cast<const float4*>(gq + c);
      const float4 be = *reinterpret_cast<const float4*>(beq + c);
      const f2_t g01 = {gv.x, gv.y}, g23 = {gv.z, gv.w}, b01 = {be.x, be.y}, b23 = {be.z, be.w};
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const f2_t a01 = {acc[f][n][4 * g + 0], acc[f][n][4 * g + 1]}, a23 = {acc[f][n][4 * g + 2], acc[f][n][4 * g + 3]};
        const f2_t o01 = (SSHIP_FFN4_ABL & 2) ? a01 : gelu2((a01 * rstd[n] + mean[n]) * g01 + b01);
        const f2_t o23 = (SSHIP_FFN4_ABL & 2) ? a23 : gelu2((a23 * rstd[n] + mean[n]) * g23 + b23);
        *reinterpret_cast<h4_t*>(s_x + (n * 32 + j) * kFfnLd + c) = to_h4(o01[0], o01[1], o23[0], o23[1]);
      }
    }
  stamp(3);
  __syncthreads();  // hidden tile complete
  stamp(4);
  // ---- ffn.3 : rows [64 wave, +64) x 64 tokens, K = 512, + residual ----
  f16x_t ac2[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) ac2[m][n][r] = 0.f;
  {
    constexpr int R3 = 8;
    const __amdgpu_buffer_rsrc_t r3 = wres(w3p);  // packed [cb = 2 wave + m][k16][mt = 0][lane][8]
    const int w3off = (2 * wave) * (32 * 512);
    h8_t a3[R3][2], bf[2][NT];
#pragma unroll
    for (int i = 0; i < R3 - 1; ++i)
#pragma unroll
      for (int m = 0; m < 2; ++m) a3[i][m] = wload(r3, w3off + m * (32 * 512) + i * 512);
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      if (ks + R3 - 1 < 32) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
          a3[(ks + R3 - 1) % R3][m] = wload(r3, w3off + m * (32 * 512) + (ks + R3 - 1) * 512);
      }
      if (ks + 1 < 32) {
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[(ks + 1) & 1][n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd + (ks + 1) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) ac2[m][n] = mfma32_abl(a3[ks % R3][m], bf[ks & 1][n], ac2[m][n]);
    }
  }
  stamp(5);
  if constexpr (NEXT_MT > 0) __syncthreads();  // all waves are done reading the hidden tile: s_x gets the new x
  // x <- x + ffn.3(...) + b3: lane (j, hh) holds channels 4 hh + 8 g .. + 3 of token j; v_permlane32_swap pairs the two
  // half-waves' quads so that every lane writes whole 8-channel (16-byte) units to global memory and to the LDS tile
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      unsigned lo[4], hi[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(b3q + wave * 64 + m * 32 + hh * 4 + g * 8);
        const h4_t o = xres[m][g][n];
        const h2_t p01 = {(_Float16)((float)o[0] + (ac2[m][n][4 * g + 0] + bv.x)), (_Float16)((float)o[1] + (ac2[m][n][4 * g + 1] + bv.y))};
        const h2_t p23 = {(_Float16)((float)o[2] + (ac2[m][n][4 * g + 2] + bv.z)), (_Float16)((float)o[3] + (ac2[m][n][4 * g + 3] + bv.w))};
        lo[g] = __builtin_bit_cast(unsigned, p01);
        hi[g] = __builtin_bit_cast(unsigned, p23);
      }
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(lo[2 * gp], lo[2 * gp + 1], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(hi[2 * gp], hi[2 * gp + 1], false, false);
        const uint4 unit = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        const int c = wave * 64 + m * 32 + (2 * gp + hh) * 8;
        if constexpr (!(SSHIP_FFN4_XROWS && NEXT_MT > 0)) {
          if (SSHIP_FFN4_BSTORE & 1) wstore(rx, xrow16, (long long)(t0 + n * 32) * 256 + wave * 64 + m * 32 + 2 * gp * 8, wq_t{s0[0], s1[0], s0[1], s1[1]});
          else *reinterpret_cast<uint4*>(x + (t0 + n * 32 + j) * 256 + c) = unit;
        }
        if constexpr (NEXT_MT > 0) *reinterpret_cast<uint4*>(s_x + (n * 32 + j) * kFfnLd + c) = unit;
      }
    }
  stamp(6);
  }  // !PROJ
  if constexpr (NEXT_MT > 0) {
    __syncthreads();
    stamp(7);
    if constexpr (SSHIP_FFN4_XROWS && !PROJ) {
      // The new x leaves from the LDS tile as WHOLE 512-byte rows (a wave's stores above would be 32-byte pieces of 32 different rows each: a token's row
      // is assembled by the four waves): wave w copies rows [16 w, 16 w + 16), two rows per instruction, 16 bytes per lane.  The stores are in
      // flight under the projection's MFMAs.
#pragma unroll
      for (int r2 = 0; r2 < NTOK / 8; ++r2) {
        const int row = wave * (NTOK / 4) + 2 * r2 + hh;
        const uint4 u = *reinterpret_cast<const uint4*>(s_x + row * kFfnLd + j * 8);
        *reinterpret_cast<uint4*>(x + (t0 + row) * 256 + j * 8) = u;
      }
    }
    // ---- fused next projection: two passes; pass p = row block cb = wave + 4 p of the tile-interleaved packing
    // (block cb holds M-tile m = rows (8 m + cb) * 32 .. + 31: one tile of each 256-row segment), K = 256 ----
    const int NP = pj.np, nt32 = NP >> 5;
    const int rope_segs = pj.flags & 0xf;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int cb = wave + 4 * pass;
      f16x_t ac3[NEXT_MT][NT];
#pragma unroll
      for (int m = 0; m < NEXT_MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) ac3[m][n][r] = 0.f;
      // rotary (cos, sin) quads of this lane's q / k rows: requested here, consumed after the MFMA loop.  The q and the k
      // tile of a block cover the same head-local channels ((8 m + cb) * 32 mod 64 does not depend on m): one set serves both.
      float4 cs[4][NT];
      const __amdgpu_buffer_rsrc_t rrope = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tail.proj.aux), 0, (int)0x7ffffff0, 0x00020000);
      if constexpr (HEADS) {
        if (rope_segs > 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              cs[g][n] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                  rrope, (unsigned)j * 256u + (unsigned)hh * 16u, (int)(((t0 + n * 32) * 64 + ((cb * 32) & 63) + g * 8) * 4) + zero, 0));
        }
      }
      const __amdgpu_buffer_rsrc_t rp = wres(tail.proj.wpack);  // [cb][k16][mt][lane][8]
      const int wpoff = cb * (16 * NEXT_MT * 512);
      constexpr int VMASK = HEADS ? (1 << (NEXT_MT - 1)) : 0;  // the V tile runs with swapped operands (see k_lg_ffn)
      constexpr int RT = NEXT_MT >= 3 ? 4 : 5;
      h8_t at[RT][NEXT_MT], bf[2][NT];
#pragma unroll
      for (int i = 0; i < RT - 1; ++i)
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) at[i][m] = wload(rp, wpoff + (i * NEXT_MT + m) * 512);
#pragma unroll
      for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + RT - 1 < 16) {
#pragma unroll
          for (int m = 0; m < NEXT_MT; ++m) at[(ks + RT - 1) % RT][m] = wload(rp, wpoff + ((ks + RT - 1) * NEXT_MT + m) * 512);
        }
        if (ks + 1 < 16) {
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[(ks + 1) & 1][n] = *reinterpret_cast<const h8_t*>(bfp + n * 32 * kFfnLd + (ks + 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) {
          const h8_t a = at[ks % RT][m];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if ((VMASK >> m) & 1) ac3[m][n] = mfma32_abl(bf[ks & 1][n], a, ac3[m][n]);
            else ac3[m][n] = mfma32_abl(a, bf[ks & 1][n], ac3[m][n]);
          }
        }
      }
      stamp(8 + pass);
      if constexpr (SSHIP_FFN4_ABL & 8) {
        float keep = 0.f;
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) keep += ac3[m][n][0];
        if (keep == 1.2345e-30f) tail.trace[0] = 1;  // the accumulators stay live; never true
      } else if constexpr (HEADS) {
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) {
          const int R0 = (m * 8 + cb) * 32;  // first output row of this M-tile
          const int hd = (R0 >> 6) & 3;
          if ((VMASK >> m) & 1) {
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
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)(ac3[m][n][8 * kk + e] + bv);
                if (SSHIP_FFN4_BSTORE & 2) wstore(rv, lane16, dst + (kk * 2 + mth) * 512, __builtin_bit_cast(wq_t, o));
                else *reinterpret_cast<h8_t*>(static_cast<_Float16*>(pj.out2) + dst + lane * 8 + (kk * 2 + mth) * 512) = o;
              }
            }
          } else {
            // q / k (or the shared qk of CrossBlock): bias, rotary on interleaved pairs, fp16, then lane^32 pairing so that
            // every lane owns whole 16-byte fragment units: unit u = d / 8 -> [kstep u / 2][lane' = (u & 1) * 32 + token % 32][8]
            const int seg = R0 >> 8;
            const __amdgpu_buffer_rsrc_t rqk = seg == 0 ? rq : rk;
            const bool roped = seg < rope_segs;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const size_t token = t0 + n * 32;
              const int sq = (int)(token / NP), kt = (int)(token - (size_t)sq * NP) >> 5;
              const long long dst = (((long long)sq * 4 + hd) * nt32 + kt) * 2048;
              unsigned lo[4], hi[4];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(s_pb + R0 + hh * 4 + g * 8);
                float v0 = ac3[m][n][4 * g + 0] + bv.x, v1 = ac3[m][n][4 * g + 1] + bv.y;
                float v2 = ac3[m][n][4 * g + 2] + bv.z, v3 = ac3[m][n][4 * g + 3] + bv.w;
                if (roped) {
                  const float4 c = cs[g][n];
                  const float r0 = v0 * c.x - v1 * c.y, r1 = v1 * c.x + v0 * c.y;
                  const float r2 = v2 * c.z - v3 * c.w, r3 = v3 * c.z + v2 * c.w;
                  v0 = r0; v1 = r1; v2 = r2; v3 = r3;
                }
                const h2_t p01 = {(_Float16)v0, (_Float16)v1}, p23 = {(_Float16)v2, (_Float16)v3};
                lo[g] = __builtin_bit_cast(unsigned, p01);
                hi[g] = __builtin_bit_cast(unsigned, p23);
              }
#pragma unroll
              for (int gp = 0; gp < 2; ++gp) {
                // swap(A, B): A' = {A.lo-lanes, B.lo-lanes}, B' = {A.hi-lanes, B.hi-lanes}.  With A = quad g = 2 gp and
                // B = quad 2 gp + 1, lane hh = 0 ends up with the whole unit of quad 2 gp, lane hh = 1 with that of 2 gp + 1.
                const auto s0 = __builtin_amdgcn_permlane32_swap(lo[2 * gp], lo[2 * gp + 1], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(hi[2 * gp], hi[2 * gp + 1], false, false);
                // unit u = ((R0 & 63) >> 3) + 2 gp + hh: (R0 & 63) >> 3 is 0 or 4, so u >> 1 is scalar and (u & 1) * 32 + j is the lane id
                if (SSHIP_FFN4_BSTORE & 4) wstore(rqk, lane16, dst + ((((R0 & 63) >> 3) + 2 * gp) >> 1) * 512, wq_t{s0[0], s1[0], s0[1], s1[1]});
                else {
                  const int u = ((R0 & 63) >> 3) + 2 * gp + hh;
                  *reinterpret_cast<uint4*>(static_cast<_Float16*>(seg == 0 ? pj.out0 : pj.out1) + dst + (u >> 1) * 512 + (((u & 1) << 5) + j) * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
              }
            }
          }
        }
      } else {
        // plain fp16 rows (final_proj): block cb = rows cb * 32 .. + 31 (NEXT_MT = 1, not interleaved)
        EpiF16<false, false>::template run<NEXT_MT, NT>(pj, ac3, 0, (int)(t0 >> 5), j, cb * NEXT_MT * 32, hh);
      }
    }
    if (tail.logsig) {  // matchability head of the last block: one wave per 16 tokens
#pragma unroll 1
      for (int tk = wave * (NTOK / 4); tk < (wave + 1) * (NTOK / 4); ++tk) {
        const h4_t v = *reinterpret_cast<const h4_t*>(s_x + tk * kFfnLd + lane * 4);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) d += (float)v[e] * mwq[lane * 4 + e];
        const float z = wave_sum(d) + tail.match_b;
        if (lane == 0) tail.logsig[t0 + tk] = fminf(z, 0.f) - log1pf(expf(-fabsf(z)));
      }
    }
  }
  stamp(10);
  if (has_next) {
    __syncthreads();  // every wave has finished with the tile buffer
    stage_tile(tile + gridDim.x, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  stamp(11);
  }  // tile loop
}
template <int NEXT_MT, bool HEADS, bool PROJ, typename... A>
static hipError_t launch_ffn4(int tokens, hipStream_t s, A... args) {
  constexpr size_t smem = (size_t)64 * kFfnLd * 2 + 8 * 64 * 4 + (1792 + 768 + 512) * 4;  // 80,896 B: two workgroups per CU
  static_assert(2 * smem <= 163840, "two workgroups must fit the CU's LDS");
  auto kern = k_lg_ffn4<NEXT_MT, HEADS, PROJ>;
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_rc != hipSuccess) return attr_rc;
  const int ntiles = tokens / 64;
  hipLaunchKernelGGL(kern, dim3(ntiles < 2 * cu_count() ? ntiles : 2 * cu_count()), dim3(256), smem, s, args...);
  return hipGetLastError();
}
// the 4-wave kernel serves throughput batches (at least one 64-token tile per workgroup slot); SUPERSLAM_HIP_FFN=8 keeps
// the 8-wave kernel everywhere (A/B runs)
static bool use_ffn4(int tokens) {
  static const int env = dev_env("SUPERSLAM_HIP_FFN") ? atoi(dev_env("SUPERSLAM_HIP_FFN")) : 0;
  // k_lg_ffn4 addresses x / q / k / v^T through buffer resources with 32-bit byte offsets (token * 512 B): beyond ~2 GiB of
  // token stream its out-of-range stores would be dropped silently - such launches use the 8-wave kernel (64-bit addresses)
  if ((size_t)tokens * 512 >= 0x7f000000ull) return false;
  if (env == 8) return false;
  if (env == 4) return tokens % 64 == 0;
  return tokens % 64 == 0 && tokens / 64 >= 2 * cu_count();
}

// SUPERSLAM_HIP_FFN=16: the 16-wave kernel of lg_ffn16.hip for every launch it applies to; default: for throughput batches
// (at least four 32-token N-tiles per CU)
[[maybe_unused]] static bool use_ffn16(int tokens) {
  static const int env = dev_env("SUPERSLAM_HIP_FFN") ? atoi(dev_env("SUPERSLAM_HIP_FFN")) : 0;
  return env == 16;   // measured: no gain over k_lg_ffn4 (profiles/r04_b .. r04_e) - never the default
}
static bool trace_on_is8() { return false; }
// SSHIP_FFN_TRACE=1 with the 4-wave kernel: mean shader-clock duration of every phase of a workgroup's second tile
static void ffn4_trace_report(unsigned long long* dev, int nwg, int next_mt, hipStream_t s) {
  std::vector<unsigned long long> h((size_t)nwg * 8 * 12);
  (void)hipStreamSynchronize(s);
  (void)hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
  // stamps: 0 start, 1 ffn.0 done, 2 stats+barrier, 3 GELU, 4 barrier, 5 ffn.3, 6 residual, 7 barrier, 8 proj pass 0 MFMA,
  // 9 (epilogue 0 +) proj pass 1 MFMA, 10 epilogue 1 (+ matchability), 11 next-tile DMA + barriers
  static const char* names[11] = {"ffn.0", "stats+barrier", "GELU", "barrier", "ffn.3", "barrier+residual", "barrier", "proj0 MFMA",
                                  "epi0+proj1 MFMA", "epi1", "restage"};
  double sum[11] = {0}; long cnt = 0;
  for (int w = 0; w < nwg * 8; ++w) {
    const unsigned long long* t = h.data() + (size_t)w * 12;
    if (t[0] < 1000000ull || !t[11]) continue;
    for (int i = 0; i < 11; ++i) {
      unsigned long long a = t[i], b = t[i + 1];
      if (!b) b = a;  // phases a variant does not have
      sum[i] += (double)(b > a ? b - a : 0);
    }
    ++cnt;
  }
  if (!cnt) return;
  fprintf(stderr, "[ffn4 trace next_mt=%d, %ld waves]", next_mt, cnt);
  double tot = 0;
  for (int i = 0; i < 11; ++i) { fprintf(stderr, " %s=%.0f", names[i], sum[i] / cnt); tot += sum[i] / cnt; }
  fprintf(stderr, " | tile=%.0f clk\n", tot);
}
// next == nullptr: plain FFN.  Otherwise the projection `next` (packed with ct = 32 * next_mt rows per wave) runs on
// the updated tile; heads = true -> EpiHeads (q/k/vt, rope_segs, t_seg), false -> fp16 rows to `out` (+ matchability).
// Latency mode's weight prefetch (FfnTail::n_main): the packed weights the NEXT FFN launch will stream, when the launch leaves CUs free
static int ffn_prefetch_setup(FfnTail& t, int n_main, const ConvW* const* pf) {
  static const bool off = dev_env("SUPERSLAM_HIP_LG_PREFETCH") && atoi(dev_env("SUPERSLAM_HIP_LG_PREFETCH")) == 0;  // A/B runs
  constexpr int kPfWg = 64;  // 8 per XCD
  if (off || !pf || n_main + kPfWg > cu_count()) return 0;
  int k = 0;
  for (int i = 0; i < 3; ++i)
    if (pf[i] && pf[i]->w) { t.pf_ptr[k] = pf[i]->w; t.pf_bytes[k] = pf[i]->cout_pad * pf[i]->cin * 2; ++k; }
  if (!k) return 0;
  t.n_main = n_main;
  return kPfWg;
}

void launch_lg_ffn(const ConvW& w0, const ConvW& w3, const float* gamma, const float* beta, const _Float16* ctx,
                   _Float16* x, LgDims d, const ConvW* next, bool heads, int rope_segs, int t_seg, const float* rope,
                   _Float16* q, _Float16* k, _Float16* vt, _Float16* out, const float* match_w, float match_b,
                   float* logsig, hipStream_t s, const ConvW* const* prefetch) {
  const int tokens = d.S * d.NP;
  FfnTail t{};
  static const int nt_env = dev_env("SUPERSLAM_HIP_FFN_NT") ? atoi(dev_env("SUPERSLAM_HIP_FFN_NT")) : 0;  // A/B: 1 | 2
  const int nt = nt_env == 1 || nt_env == 2 ? nt_env : (tokens / 64 < cu_count() / 2 ? 1 : 2);  // 32-token N-tiles per workgroup tile
  t.ntiles = tokens / (nt * 32);
  static const bool trace_on = dev_env("SSHIP_FFN_TRACE") != nullptr;
  static unsigned long long* trace_buf = nullptr;
  const int trace_wg = t.ntiles < cu_count() ? t.ntiles : cu_count();
  if (trace_on) {
    if (!trace_buf) (void)hipMalloc(&trace_buf, (size_t)2 * cu_count() * 8 * 12 * 8);
    (void)hipMemsetAsync(trace_buf, 0, (size_t)2 * cu_count() * 8 * 12 * 8, s);
    t.trace = trace_buf;
    static const int trace_it = dev_env("SSHIP_FFN_TRACE_IT") ? atoi(dev_env("SSHIP_FFN_TRACE_IT")) : 1;
    t.trace_it = trace_it;
  }
  if (!next) {
    (void)launch_ffn<0, false>(nt, tokens, 0, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
    if (trace_on) ffn_trace_report(trace_buf, trace_wg, 0, s);
    return;
  }
  t.proj = token_args(*next, x, 256, nullptr, 0, d);
  t.proj.out0 = heads ? (void*)q : (void*)out; t.proj.out1 = k; t.proj.out2 = vt; t.proj.aux = rope;
  t.proj.flags = rope_segs | (t_seg << 4); t.proj.ostride = 256;
  t.match_w = match_w; t.match_b = match_b; t.logsig = logsig;
  const int mt = next->cout / 256;  // rows per wave / 32: 768 -> 3, 512 -> 2, 256 -> 1
#if SSHIP_DEV_SWITCHES
  if (use_ffn16(tokens) && ffn16_applicable(tokens, mt, heads)) {
    (void)launch_lg_ffn16(tokens, mt, heads, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
    return;
  }
#endif
  if (use_ffn4(tokens) && !trace_on_is8()) {
    t.ntiles = tokens / 64;
    if (heads && mt == 3) (void)launch_ffn4<3, true, false>(tokens, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
    else if (heads && mt == 2) (void)launch_ffn4<2, true, false>(tokens, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
    else (void)launch_ffn4<1, false, false>(tokens, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
    if (trace_on) ffn4_trace_report(trace_buf, t.ntiles < 2 * cu_count() ? t.ntiles : 2 * cu_count(), mt, s);
    return;
  }
  const int extra = ffn_prefetch_setup(t, trace_wg, prefetch);  // trace_wg = the workgroups that walk the tiles
  if (heads && mt == 3) (void)launch_ffn<3, true>(nt, tokens, extra, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
  else if (heads && mt == 2) (void)launch_ffn<2, true>(nt, tokens, extra, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
  else (void)launch_ffn<1, false>(nt, tokens, extra, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
  if (trace_on) ffn_trace_report(trace_buf, trace_wg, mt, s);
}

// The first SelfBlock's Wqkv (no FFN in front of it): the FFN kernel's fused projection on its own - same persistent
// tile loop, LDS-DMA staging, prefetched weight stream and tile-interleaved rows as the other 17 projections.
hipError_t launch_lg_proj_heads(const ConvW& next, _Float16* x, LgDims d, int rope_segs, int t_seg, const float* rope, _Float16* q,
                                _Float16* k, _Float16* vt, hipStream_t s, const ConvW* const* prefetch) {
  if (next.cout != 768) return hipErrorInvalidValue;
  const int tokens = d.S * d.NP;
  FfnTail t{};
  const int nt = tokens / 64 < cu_count() / 2 ? 1 : 2;  // as launch_lg_ffn: 32-token tiles for a few pairs
  t.ntiles = tokens / (nt * 32);
  t.proj = token_args(next, x, 256, nullptr, 0, d);
  t.proj.out0 = q; t.proj.out1 = k; t.proj.out2 = vt; t.proj.aux = rope;
  t.proj.flags = rope_segs | (t_seg << 4); t.proj.ostride = 256;
  const _Float16* nh = nullptr;
  const float* nf = nullptr;
  if (use_ffn4(tokens)) {
    t.ntiles = tokens / 64;
    return launch_ffn4<3, true, true>(tokens, s, (const _Float16*)x, nh, nf, nf, nf, nh, nf, x, t);
  }
  const int extra = ffn_prefetch_setup(t, t.ntiles < cu_count() ? t.ntiles : cu_count(), prefetch);
  return nt == 1 ? launch_ffn_nt<3, true, 1, true>(tokens, extra, s, (const _Float16*)x, nh, nf, nf, nf, nh, nf, x, t)
                 : launch_ffn_nt<3, true, 2, true>(tokens, extra, s, (const _Float16*)x, nh, nf, nf, nf, nh, nf, x, t);
}

// ---------------------------------------------------------------------------------------------------
// Assignment: sim = md0 md1^T (fp32, [pairs][NP][NP]), then the double log-softmax + mutual arg-max filter.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lg_sim(const _Float16* __restrict__ md, const int* __restrict__ lens, int NP,
                                                float* __restrict__ sim) {
  const int pair = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jl = lane & 31, hh = lane >> 5;
  const int i0 = blockIdx.y * 32, j0 = (blockIdx.x * 4 + wave) * 32;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  if (i0 >= n0 || j0 >= n1) return;
  const _Float16* A = md + ((size_t)(2 * pair) * NP + i0 + jl) * 256 + hh * 8;
  const _Float16* Bm = md + ((size_t)(2 * pair + 1) * NP + j0 + jl) * 256 + hh * 8;
  f16x_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
    acc = mfma32(*reinterpret_cast<const h8_t*>(A + ks * 16), *reinterpret_cast<const h8_t*>(Bm + ks * 16), acc);
  float* out = sim + (size_t)pair * NP * NP;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    out[(size_t)i * NP + j0 + jl] = acc[r];
  }
}
void launch_lg_sim(const _Float16* md, const int* lens, LgDims d, float* sim, hipStream_t s) {
  hipLaunchKernelGGL(k_lg_sim, dim3((d.NP + 127) / 128, d.NP / 32, d.S / 2), dim3(256), 0, s, md, lens, d.NP, sim);
}

#if SSHIP_DEV_SWITCHES  // the four passes over a materialised sim (SUPERSLAM_HIP_LG_ASSIGN=matrix; replaced by k_assign_stream in round 2)
// workspace per pair (floats): [0,NP) lse_row, [NP,2NP) lse_col, [2NP,3NP) max0, [3NP,4NP) m0 (int), [4NP,5NP) m1 (int)
__global__ __launch_bounds__(256) void k_assign_row_lse(const float* __restrict__ sim, const int* __restrict__ lens,
                                                        int NP, float* __restrict__ ws) {
  const int pair = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  if (i >= n0) return;
  const float* row = sim + ((size_t)pair * NP + i) * NP;
  float m = -INFINITY;
  for (int j = lane; j < n1; j += 64) m = fmaxf(m, row[j]);
  m = wave_max(m);
  float sum = 0.f;
  for (int j = lane; j < n1; j += 64) sum += expf(row[j] - m);
  sum = wave_sum(sum);
  if (lane == 0) ws[(size_t)pair * 5 * NP + i] = m + logf(sum);
}
// column passes: 64 columns x kColRG row groups per workgroup (4 groups left every thread a 150-step dependent chain of
// online-softmax updates: 58 us per 64 pairs)
constexpr int kColRG = 16;
__global__ __launch_bounds__(64 * kColRG) void k_assign_col_lse(const float* __restrict__ sim, const int* __restrict__ lens,
                                                        int NP, float* __restrict__ ws) {
  __shared__ float s_m[kColRG][64], s_s[kColRG][64];
  const int pair = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  float m = -INFINITY, sum = 0.f;
  if (j < n1) {
    const float* col = sim + (size_t)pair * NP * NP + j;
    for (int i = rg; i < n0; i += kColRG) {
      const float v = col[(size_t)i * NP];
      const float mn = fmaxf(m, v);
      sum = sum * expf(m - mn) + expf(v - mn);
      m = mn;
    }
  }
  s_m[rg][cl] = m; s_s[rg][cl] = sum;
  __syncthreads();
  if (rg == 0 && j < n1) {
    float M = s_m[0][cl];
    for (int g = 1; g < kColRG; ++g) M = fmaxf(M, s_m[g][cl]);
    float S = 0.f;
    for (int g = 0; g < kColRG; ++g) if (s_m[g][cl] > -INFINITY) S += s_s[g][cl] * expf(s_m[g][cl] - M);
    ws[(size_t)pair * 5 * NP + NP + j] = M + logf(S);
  }
}
// row arg-max of  S_ij = (sim - lse_row_i) + (sim - lse_col_j) + ls0_i + ls1_j ; first index wins ties (torch.max)
__global__ __launch_bounds__(256) void k_assign_row_arg(const float* __restrict__ sim, const float* __restrict__ logsig,
                                                        const int* __restrict__ lens, int NP, float* __restrict__ ws) {
  const int pair = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  if (i >= n0) return;
  float* w = ws + (size_t)pair * 5 * NP;
  const float* row = sim + ((size_t)pair * NP + i) * NP;
  const float* ls1 = logsig + (size_t)(2 * pair + 1) * NP;
  const float li = w[i], ls0 = logsig[(size_t)(2 * pair) * NP + i];
  float best = -INFINITY;
  int bj = 0;  // a row of NaN / -inf scores keeps index 0, like torch.max on a degenerate row (never an out-of-range index)
  for (int j = lane; j < n1; j += 64) {
    const float v = ((row[j] - li) + (row[j] - w[NP + j])) + (ls0 + ls1[j]);
    if (v > best) { best = v; bj = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (ov > best || (ov == best && oj < bj)) { best = ov; bj = oj; }
  }
  if (lane == 0) { w[2 * NP + i] = best; reinterpret_cast<int*>(w)[3 * NP + i] = bj; }
}
__global__ __launch_bounds__(64 * kColRG) void k_assign_col_arg(const float* __restrict__ sim, const float* __restrict__ logsig,
                                                        const int* __restrict__ lens, int NP, float* __restrict__ ws) {
  __shared__ float s_v[kColRG][64];
  __shared__ int s_i[kColRG][64];
  const int pair = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  float* w = ws + (size_t)pair * 5 * NP;
  float best = -INFINITY;
  int bi = 0;
  if (j < n1) {
    const float* col = sim + (size_t)pair * NP * NP + j;
    const float* ls0 = logsig + (size_t)(2 * pair) * NP;
    const float lj = w[NP + j], ls1 = logsig[(size_t)(2 * pair + 1) * NP + j];
    for (int i = rg; i < n0; i += kColRG) {
      const float sv = col[(size_t)i * NP];
      const float v = ((sv - w[i]) + (sv - lj)) + (ls0[i] + ls1);
      if (v > best) { best = v; bi = i; }
    }
  }
  s_v[rg][cl] = best; s_i[rg][cl] = bi;
  __syncthreads();
  if (rg == 0 && j < n1) {
    for (int g = 1; g < kColRG; ++g)
      if (s_v[g][cl] > best || (s_v[g][cl] == best && s_i[g][cl] < bi)) { best = s_v[g][cl]; bi = s_i[g][cl]; }
    reinterpret_cast<int*>(w)[4 * NP + j] = bi;
  }
}
// filter_matches(scores, 0.1): mutual check, mscores0 = mutual ? exp(max0) : 0, matches0 = valid ? m0 : -1.
// (legacy path: the arg-max rows / columns and scores are in ws)
__global__ void k_assign_final(const int* __restrict__ lens, int NP, const float* __restrict__ ws, int max_kp,
                               float thr, int32_t* __restrict__ matches0, float* __restrict__ mscores0) {
  const int pair = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_kp) return;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  const float* w = ws + (size_t)pair * 5 * NP;
  int mj = -1;
  float ms = 0.f;
  if (i < n0 && n1 > 0) {
    const int j = reinterpret_cast<const int*>(w)[3 * NP + i];
    const bool mutual = (unsigned)j < (unsigned)n1 && reinterpret_cast<const int*>(w)[4 * NP + j] == i;
    ms = mutual ? expf(w[2 * NP + i]) : 0.f;
    mj = (mutual && ms > thr) ? j : -1;
  }
  matches0[(size_t)pair * max_kp + i] = mj;
  mscores0[(size_t)pair * max_kp + i] = ms;
}
#endif  // SSHIP_DEV_SWITCHES
// ---------------------------------------------------------------------------------------------------
// Assignment without the similarity matrix (round 2): two streaming passes over md, each recomputing the 32 x 32 tiles of
// sim = md0 md1^T on the matrix cores in BOTH orientations, so that every statistic is lane-local:
//   acc_i = mfma(md1 tile, md0 tile): lane owns row i, its 16 registers are columns j  -> row statistics
//   acc_j = mfma(md0 tile, md1 tile): lane owns column j, its 16 registers are rows i  -> column statistics
// (the two orientations are the same products summed in the same k order: bit-identical values).
// A wave owns a 32-row tile of image 0, keeps its md0 fragments in registers and walks the column tiles of image 1
// (fragments streamed through a two-halves register ring).  PASS 0: online log-sum-exp per row (complete in the wave) and
// per (row tile, column) partials; PASS 1: row arg-max of S_ij (complete) and per (row tile, column) partial arg-max.
// PASS 1 folds PASS 0's partials in its prologue, k_assign_mutual those of PASS 1.  The fp32 [pairs][NP][NP] matrix (92 MB at 64 pairs, written once and
// read four times: 0.22 ms) is never materialised; sship_lg_debug_read(SIM) computes it on demand with k_lg_sim.
//   S_ij = (sim - lse_row_i) + (sim - lse_col_j) + ls0_i + ls1_j = 2 sim + c_i + d_j ; first index wins ties (torch.max).
// The arg-max over j needs only 2 sim + d_j (c_i is added to the winner afterwards), the one over i only 2 sim + c_i:
// one fma, one compare and two selects per entry.
// ---------------------------------------------------------------------------------------------------
// exp(x) for x <= 0 as v_exp_f32(x log2 e): two instructions (libm's expf is twelve, and these passes are instruction-bound);
// relative error ~|x| 2^-24, far inside the assignment's tolerances (the scores are compared at 2e-2)
constexpr float kLog2e = 1.44269504088896340736f;
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {  // (m, s) <- logsumexp-merge with (m2, s2)
  const float M = fmaxf(m, m2);
  const float a = m > -INFINITY ? s * fast_exp(m - M) : 0.f, b = m2 > -INFINITY ? s2 * fast_exp(m2 - M) : 0.f;
  m = M; s = a + b;
}
// kAssignCh column chunks per row tile (blockIdx.z): 4x the waves (a wave per row tile alone leaves ~1 wave per SIMD with every
// dependent-instruction stall exposed: 90 us per pass), at the price of per-chunk row partials next to the per-tile column ones
constexpr int kAssignCh = 4;
// folds of the per-tile / per-chunk partials (ascending part index: one fixed summation order whoever folds them).  PASS 1 of
// k_assign_stream folds the log-sum-exp partials of PASS 0 in its prologue and k_assign_mutual the arg-max partials of PASS 1 -
// the two k_assign_combine launches this replaced were 7 us each in a one-pair call (5 workgroups, two dependent round trips).
__device__ __forceinline__ float fold_lse(const float2* p, int nparts, int NP) {
  float m = -INFINITY, s = 0.f;
  for (int k = 0; k < nparts; ++k) { const float2 v = p[(size_t)k * NP]; lse_merge(m, s, v.x, v.y); }
  return m + logf(s);
}
__device__ __forceinline__ void fold_argmax(const float2* p, int nparts, int NP, float& best, int& bi) {  // ties: smaller index
  best = -INFINITY; bi = 0x7fffffff;
  for (int k = 0; k < nparts; ++k) {
    const float2 v = p[(size_t)k * NP];
    const int i = __float_as_int(v.y);
    if (v.x > best || (v.x == best && i < bi)) { best = v.x; bi = i; }
  }
}
// pcol / prow: the partials this pass WRITES; pcol_in / prow_in (PASS 1): the log-sum-exp partials PASS 0 wrote (different buffers:
// a workgroup of PASS 1 that finishes early must not overwrite partials a later one still folds)
template <int PASS>
__global__ __launch_bounds__(256, 2) void k_assign_stream(const _Float16* __restrict__ md, const float* __restrict__ logsig,
                                                       const int* __restrict__ lens, int NP, float* __restrict__ pcol,
                                                       float* __restrict__ prow, const float* __restrict__ pcol_in,
                                                       const float* __restrict__ prow_in) {
  constexpr int kChunkCols = (kMaxKp / 32 + kAssignCh - 1) / kAssignCh * 32;  // columns of one chunk at most
  __shared__ __attribute__((aligned(16))) float s_lc[PASS ? kChunkCols : 4];  // d_j = ls1_j - lse_col_j of this chunk's columns
  __shared__ float s_lr[PASS ? 128 : 4];                                       // lse_row of the workgroup's 128 rows
  const int pair = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jl = lane & 31, hh = lane >> 5;
  const int NT = NP >> 5, ti = blockIdx.x * 4 + wave, i0 = ti * 32;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  if ((int)blockIdx.x * 128 >= n0) return;     // no row of this workgroup exists (uniform: before any further barrier)
  const bool active = ti < NT && i0 < n0;      // wave-uniform: a wave past the end still stages column tiles and joins the barriers
  const _Float16* A = md + ((size_t)(2 * pair) * NP + min(i0 + jl, NP - 1)) * 256 + hh * 8;
  const _Float16* Bm = md + (size_t)(2 * pair + 1) * NP * 256;
  h8_t fa[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) fa[ks] = *reinterpret_cast<const h8_t*>(A + ks * 16);
  const int my_i = i0 + jl;
  const int ro = 4 * hh;  // register r of this lane <-> tile-local index (r & 3) + 8 (r >> 2) + ro
  // PASS 1 inputs: lse_row / ls0 of the lane's own row, and of the 16 rows its acc_j registers stand for
  float c_i = 0.f, c_r[PASS ? 16 : 1];  // c = ls0 - lse_row
  float rm = -INFINITY, rs = 0.f;  // PASS 0: running (max, sum) of the lane's row;  PASS 1: running best value (rm) ...
  int rj = 0x7fffffff;             // ... and its column
  const int ntj_all = (n1 + 31) >> 5, per = (ntj_all + kAssignCh - 1) / kAssignCh;
  const int tj_lo = blockIdx.z * per, ntj = min(tj_lo + per, ntj_all);  // this workgroup's column tiles: [tj_lo, ntj)
  if (tj_lo >= ntj) return;
  const int jc0 = tj_lo * 32;  // first column of this chunk: s_lc / s_l1 are indexed by j - jc0
  if constexpr (PASS == 1) {
    // PASS 0's partials -> lse_row of this workgroup's rows (over the column chunks that got tiles), lse_col of this chunk's columns
    // (over the row tiles that exist); rows >= n0 / columns >= n1 are masked further down and get a finite dummy
    const int nch = (ntj_all + per - 1) / max(per, 1), nrt = (n0 + 31) >> 5;
    if (threadIdx.x < 128) {
      const int i = blockIdx.x * 128 + threadIdx.x;
      s_lr[threadIdx.x] = i < n0 ? fold_lse(reinterpret_cast<const float2*>(prow_in) + (size_t)pair * kAssignCh * NP + i, nch, NP) : 0.f;
    }
    for (int j = threadIdx.x; j < (ntj - tj_lo) * 32; j += 256) {
      const int jj = jc0 + j;
      s_lc[j] = jj < n1 ? logsig[(size_t)(2 * pair + 1) * NP + jj] - fold_lse(reinterpret_cast<const float2*>(pcol_in) + (size_t)pair * NT * NP + jj, nrt, NP)
                        : 0.f;  // d_j = ls1_j - lse_col_j
    }
    __syncthreads();
    c_i = logsig[(size_t)(2 * pair) * NP + min(my_i, NP - 1)] - s_lr[wave * 32 + jl];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int il = (r & 3) + 8 * (r >> 2) + ro;
      c_r[r] = logsig[(size_t)(2 * pair) * NP + min(i0 + il, NP - 1)] - s_lr[wave * 32 + il];
    }
  }
  // The four waves of the workgroup walk the same column tiles: a tile (32 rows x 512 B of image 1) is fetched ONCE, with
  // coalesced loads (32 lanes = one row), into a padded LDS buffer (row stride 528 B: conflict-free ds_read_b128 fragments)
  // - as fragments straight from global memory every load instruction touched 32 different rows (32 B each) and the
  // texture path, not the matrix pipe, set the pace.  Two buffers; the next tile's loads are in flight during the MFMAs.
  constexpr int kRowH = 264;  // halfs per LDS row
  __shared__ __attribute__((aligned(16))) _Float16 s_b[2][32 * kRowH];
  typedef unsigned stg_t __attribute__((ext_vector_type(4)));  // (HIP's uint4 is a struct with unions: an array of them lands in scratch)
  stg_t stg[4];
  auto fetch = [&](int tj) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = threadIdx.x + 256 * q, row = u >> 5, unit = u & 31;
      stg[q] = *reinterpret_cast<const stg_t*>(Bm + (size_t)min(tj * 32 + row, NP - 1) * 256 + unit * 8);
    }
  };
  auto put = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = threadIdx.x + 256 * q, row = u >> 5, unit = u & 31;
      *reinterpret_cast<stg_t*>(&s_b[buf][row * kRowH + unit * 8]) = stg[q];
    }
  };
  fetch(tj_lo);
  put(0);
  __syncthreads();
  for (int tj = tj_lo; tj < ntj; ++tj) {
    const int j0 = tj * 32, buf = (tj - tj_lo) & 1;
    fetch(min(tj + 1, ntj - 1));  // unconditional (the last one re-reads this tile and is never used): keeps stg in registers
    const f16x_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16x_t ai = zero16, aj = zero16;
    if (active) {
      const _Float16* bt = &s_b[buf][jl * kRowH + hh * 8];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const h8_t fbk = *reinterpret_cast<const h8_t*>(bt + k * 16);
        ai = mfma32(fbk, fa[k], ai);
        aj = mfma32(fa[k], fbk, aj);
      }
    }
    put(buf ^ 1);  // the other buffer: its last readers passed the barrier at the end of the previous iteration
    __syncthreads();
    if (!active) continue;
    const int my_j = j0 + jl;
    // entries outside n0 x n1 become -inf; only the last row tile / column tile can have any (wave-uniform branches)
    if (j0 + 32 > n1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (j0 + (r & 3) + 8 * (r >> 2) + ro >= n1) ai[r] = -INFINITY;
      if (my_j >= n1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) aj[r] = -INFINITY;
      }
    }
    if (i0 + 32 > n0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (i0 + (r & 3) + 8 * (r >> 2) + ro >= n0) aj[r] = -INFINITY;
      if (my_i >= n0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ai[r] = -INFINITY;
      }
    }
    if constexpr (PASS == 0) {
      // ---- row: online log-sum-exp over this tile's 16 columns of the lane
      float tm = ai[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tm = fmaxf(tm, ai[r]);
      if (tm > -INFINITY) {
        float ts = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ts += __builtin_amdgcn_exp2f(fmaf(ai[r], kLog2e, -tm * kLog2e));
        lse_merge(rm, rs, tm, ts);
      }
      // ---- column partial over the 32 rows of this wave
      float cm = aj[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) cm = fmaxf(cm, aj[r]);
      float cs = 0.f;
      if (cm > -INFINITY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cs += __builtin_amdgcn_exp2f(fmaf(aj[r], kLog2e, -cm * kLog2e));
      }
      lse_merge(cm, cs, __shfl_xor(cm, 32, 64), __shfl_xor(cs, 32, 64));
      if (hh == 0 && my_j < n1) *reinterpret_cast<float2*>(pcol + (((size_t)pair * NT + ti) * NP + my_j) * 2) = make_float2(cm, cs);
    } else {
      // ---- row: arg-max over this tile's 16 columns of the lane (ascending j inside the lane)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jb = j0 + 8 * g + ro;
        const float4 d4 = *reinterpret_cast<const float4*>(s_lc + jb - jc0);
        const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(2.f, ai[4 * g + e], dv[e]);
          if (v > rm) { rm = v; rj = jb + e; }  // masked entries are -inf: never greater
        }
      }
      // ---- column partial arg-max over the 32 rows of this wave (ascending i inside the lane)
      const float d_j = s_lc[my_j - jc0];
      float cv = -INFINITY;
      int ci = 0x7fffffff;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2) + ro;
        const float v = fmaf(2.f, aj[r], c_r[r]);
        if (v > cv) { cv = v; ci = i; }
      }
      const float ov = __shfl_xor(cv, 32, 64);
      const int oi = __shfl_xor(ci, 32, 64);
      if (ov > cv || (ov == cv && oi < ci)) { cv = ov; ci = oi; }
      if (hh == 0 && my_j < n1)
        *reinterpret_cast<float2*>(pcol + (((size_t)pair * NT + ti) * NP + my_j) * 2) = make_float2(cv + d_j, __int_as_float(ci));
    }
  }
  float2* pr = reinterpret_cast<float2*>(prow) + ((size_t)pair * kAssignCh + blockIdx.z) * NP + my_i;
  if constexpr (PASS == 0) {
    lse_merge(rm, rs, __shfl_xor(rm, 32, 64), __shfl_xor(rs, 32, 64));
    if (hh == 0 && my_i < n0) *pr = make_float2(rm, rs);
  } else {
    const float ov = __shfl_xor(rm, 32, 64);
    const int oj = __shfl_xor(rj, 32, 64);
    if (ov > rm || (ov == rm && oj < rj)) { rm = ov; rj = oj; }
    if (hh == 0 && my_i < n0) *pr = make_float2(rm + c_i, __int_as_float(rj));
  }
}
// filter_matches(scores, 0.1) on the partials of PASS 1: row i's arg-max column j and score (folded over the column chunks), column
// j's arg-max row (folded over the row tiles), mutual check, mscores0 = mutual ? exp(max0) : 0, matches0 = valid ? j : -1.
// A row / column of NaN / -inf scores keeps index 0, like torch.max on a degenerate row (never an out-of-range index).
__global__ __launch_bounds__(256) void k_assign_mutual(const float* __restrict__ pcol, const float* __restrict__ prow,
                                                       const int* __restrict__ lens, int NP, int max_kp, float thr,
                                                       int32_t* __restrict__ matches0, float* __restrict__ mscores0) {
  const int pair = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_kp) return;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  const int NT = NP >> 5;
  int mj = -1;
  float ms = 0.f;
  if (i < n0 && n1 > 0) {
    const int ntj_all = (n1 + 31) >> 5, per = (ntj_all + kAssignCh - 1) / kAssignCh;
    float best, cbest;
    int j, ci;
    fold_argmax(reinterpret_cast<const float2*>(prow) + (size_t)pair * kAssignCh * NP + i, (ntj_all + per - 1) / max(per, 1), NP, best, j);
    if (j == 0x7fffffff) j = 0;
    bool mutual = false;
    if ((unsigned)j < (unsigned)n1) {
      fold_argmax(reinterpret_cast<const float2*>(pcol) + (size_t)pair * NT * NP + j, (n0 + 31) >> 5, NP, cbest, ci);
      mutual = (ci == 0x7fffffff ? 0 : ci) == i;
    }
    ms = mutual ? expf(best) : 0.f;
    mj = (mutual && ms > thr) ? j : -1;
  }
  matches0[(size_t)pair * max_kp + i] = mj;
  mscores0[(size_t)pair * max_kp + i] = ms;
}

// stage = 0: both passes + the mutual filter (a match call); 1 / 2: one pass only (sship_lg_bench_stage)
void launch_lg_assign(const _Float16* md, const float* logsig, const int* lens, LgDims d, float* ws, float* pcol, int max_kp,
                      int32_t* matches0, float* mscores0, float thr, int stage, hipStream_t s) {
  const int P = d.S / 2, NT = d.NP / 32;
#if SSHIP_DEV_SWITCHES
  static const bool legacy = dev_env("SUPERSLAM_HIP_LG_ASSIGN") && std::string(dev_env("SUPERSLAM_HIP_LG_ASSIGN")) == "matrix";  // A/B
  if (legacy && stage == 0) {  // the four passes over a materialised sim (sim lives in pcol's allocation: P * NP * NP floats)
    float* sim = pcol;
    launch_lg_sim(md, lens, d, sim, s);
    hipLaunchKernelGGL(k_assign_row_lse, dim3(d.NP / 4, P), dim3(256), 0, s, sim, lens, d.NP, ws);
    hipLaunchKernelGGL(k_assign_col_lse, dim3((d.NP + 63) / 64, P), dim3(64 * kColRG), 0, s, sim, lens, d.NP, ws);
    hipLaunchKernelGGL(k_assign_row_arg, dim3(d.NP / 4, P), dim3(256), 0, s, sim, logsig, lens, d.NP, ws);
    hipLaunchKernelGGL(k_assign_col_arg, dim3((d.NP + 63) / 64, P), dim3(64 * kColRG), 0, s, sim, logsig, lens, d.NP, ws);
    hipLaunchKernelGGL(k_assign_final, dim3((max_kp + 255) / 256, P), dim3(256), 0, s, lens, d.NP, ws, max_kp, thr, matches0, mscores0);
    return;
  }
#endif
  {
    // partials inside pcol's allocation (P * NP * NP floats): [P][NT][NP][2] column + [P][kAssignCh][NP][2] row partials of PASS 0, then those of PASS 1
    float* prow = pcol + (size_t)P * NT * d.NP * 2;
    float* pcol1 = prow + (size_t)P * kAssignCh * d.NP * 2;
    float* prow1 = pcol1 + (size_t)P * NT * d.NP * 2;
    if (stage == 0 || stage == 1)
      hipLaunchKernelGGL(k_assign_stream<0>, dim3((NT + 3) / 4, P, kAssignCh), dim3(256), 0, s, md, logsig, lens, d.NP, pcol, prow, nullptr, nullptr);
    if (stage == 0 || stage == 2)
      hipLaunchKernelGGL(k_assign_stream<1>, dim3((NT + 3) / 4, P, kAssignCh), dim3(256), 0, s, md, logsig, lens, d.NP, pcol1, prow1, pcol, prow);
    if (stage == 0)
      hipLaunchKernelGGL(k_assign_mutual, dim3((max_kp + 255) / 256, P), dim3(256), 0, s, pcol1, prow1, lens, d.NP, max_kp, thr, matches0, mscores0);
  }
}

}  // namespace sship
