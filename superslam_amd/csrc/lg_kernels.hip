// LightGlue kernels (SURVEY.md 8(a)-LG restates the upstream algorithm; call sites src/LightGlue.cc:313,446).
// Token streams are channels-last fp16 [S*NP][256]; S = 2*pairs sequences (2p = set 0, 2p+1 = set 1 of pair p),
// NP = padded tokens per sequence.  Per-sequence valid lengths live in device memory (`lens`) so the whole
// matcher runs without a host round trip after SuperPoint's on-device top-k.
#include <cstdlib>
#include <type_traits>

#include "igemm.h"
#include "kernels.h"

namespace sship {

// ---------------------------------------------------------------------------------------------------
// prep: x <- descriptors (zero rows for padding), rotary table <- posenc(normalised keypoints).
// reference: keypoint normalisation src/LightGlue.cc:241-251; LearnableFourierPositionalEncoding(2,64,64).
// rope[token][i] = (cos, sin)(Wr[i,0]*kx + Wr[i,1]*ky), i < 32 (shared by the 4 heads).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lg_prep(const float* __restrict__ kp, int kp_stride, int kp_seq_stride,
                                                 const int* __restrict__ lens, const _Float16* __restrict__ desc,
                                                 size_t desc_seq_stride, const float* __restrict__ wr, float img_w,
                                                 float img_h, int S, int NP, _Float16* __restrict__ x,
                                                 float* __restrict__ rope) {
  const int token = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (token >= S * NP) return;
  const int s = token / NP, n = token % NP;
  const bool valid = n < min(max(lens[s], 0), NP);
  h4_t v = to_h4(0.f, 0.f, 0.f, 0.f);
  if (valid) v = *reinterpret_cast<const h4_t*>(desc + (size_t)s * desc_seq_stride + (size_t)n * 256 + lane * 4);
  *reinterpret_cast<h4_t*>(x + (size_t)token * 256 + lane * 4) = v;
  if (lane < 32) {
    float c = 1.f, sn = 0.f;
    if (valid) {
      const float* k = kp + (size_t)s * kp_seq_stride + (size_t)n * kp_stride;
      const float scale = fmaxf(img_w, img_h) / 2.0f;   // std::max(w, h) / 2.0f
      const float kx = (k[0] - img_w / 2.0f) / scale;   // (pt.x - cx) / scale
      const float ky = (k[1] - img_h / 2.0f) / scale;
      const float pr = wr[lane * 2 + 0] * kx + wr[lane * 2 + 1] * ky;
      c = cosf(pr);
      sn = sinf(pr);
    }
    rope[(size_t)token * 64 + lane * 2 + 0] = c;
    rope[(size_t)token * 64 + lane * 2 + 1] = sn;
  }
}
void launch_lg_prep(const float* kp, int kp_stride, int kp_seq_stride, const int* lens, const _Float16* desc,
                    size_t desc_seq_stride, const float* wr, float img_w, float img_h, LgDims d, _Float16* x,
                    float* rope, hipStream_t s) {
  const int tokens = d.S * d.NP;
  hipLaunchKernelGGL(k_lg_prep, dim3((tokens + 3) / 4), dim3(256), 0, s, kp, kp_stride, kp_seq_stride, lens, desc,
                     desc_seq_stride, wr, img_w, img_h, d.S, d.NP, x, rope);
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

hipError_t lg_linear_heads(const ConvW& w, const _Float16* x, LgDims d, int rope_segs, int t_seg, const float* rope,
                           _Float16* q, _Float16* k, _Float16* vt, hipStream_t s) {
  IgemmArgs a = token_args(w, x, 256, nullptr, 0, d);
  a.out0 = q; a.out1 = k; a.out2 = vt; a.aux = rope; a.flags = rope_segs | (t_seg << 4);
  return launch_igemm<1, 256, 128, 4, EpiHeads>(a, w.cout_pad, s);
}
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
template <int QT>
__global__ __launch_bounds__(256, 2) void k_lg_attention(const _Float16* __restrict__ q, const _Float16* __restrict__ k,
                                                         const _Float16* __restrict__ vt, const int* __restrict__ lens,
                                                         int NP, int cross, _Float16* __restrict__ ctx) {
  // One workgroup = 32*QT queries of one (sequence, head); its 4 waves split the KEYS (tile kt -> wave kt & 3,
  // flash-decoding style) and merge their (m, l, O) partials through LDS.
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  float (*s_part)[4][34][64] = reinterpret_cast<float (*)[4][34][64]>(smem_attn);  // [QT][wave][32 O regs + m + l][lane]
  const int s = blockIdx.z, h = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int q0 = blockIdx.x * 32 * QT;
  const int sk = cross ? (s ^ 1) : s;
  const int nq = min(max(lens[s], 0), NP), nk = min(max(lens[sk], 0), NP);  // device-side counts are clamped to capacity
  if (q0 >= nq) return;  // uniform for the whole workgroup
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
      qf[t][ks] = *reinterpret_cast<const h8_t*>(Q + ((size_t)(blockIdx.x * QT + t) * 4 + ks) * 512 + lane * 8);
  float m[QT], l[QT];
  f16x_t o[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[t][0][r] = 0.f; o[t][1][r] = 0.f; }
  }
  const int ntiles = (nk + 31) >> 5;
  // K and V^T fragments are prefetched ONE FULL TILE ahead (loads for tile kt+4 are issued before the MFMAs and
  // softmax of tile kt), so ~1k cycles of L2 latency hide behind a whole iteration instead of a few MFMAs.
  h8_t kf[4], vf[2][2];
  {
    const int kt0 = min(wave, nt32 - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const h8_t*>(K + ((size_t)kt0 * 4 + ks) * 512 + lane * 8);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        vf[kk][mt] = *reinterpret_cast<const h8_t*>(VT + (((size_t)kt0 * 2 + kk) * 2 + mt) * 512 + lane * 8);
  }
  for (int kt = wave; kt < ntiles; kt += 4) {
    const int k0 = kt * 32;
    h8_t kn[4], vn[2][2];
    const int ktn = min(kt + 4, nt32 - 1);  // clamped: the last prefetch re-reads a valid tile and is discarded
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kn[ks] = *reinterpret_cast<const h8_t*>(K + ((size_t)ktn * 4 + ks) * 512 + lane * 8);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        vn[kk][mt] = *reinterpret_cast<const h8_t*>(VT + (((size_t)ktn * 2 + kk) * 2 + mt) * 512 + lane * 8);
    __builtin_amdgcn_sched_barrier(0);  // the prefetch stays above this tile's MFMAs / softmax
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      f16x_t st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st = mfma32(kf[ks], qf[t][ks], st);
      if (k0 + 32 > nk) {  // only the last (ragged) key tile needs masking - wave-uniform branch
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= nk) st[r] = -INFINITY;
      }
      float tmax = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
#pragma unroll
      for (int r = 4; r < 16; r += 4) tmax = fmaxf(tmax, fmaxf(fmaxf(st[r], st[r + 1]), fmaxf(st[r + 2], st[r + 3])));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m[t], tmax);
      // the softmax is VALU-bound at head_dim 64: rescale the 32 output accumulators only when some query's running
      // max actually moved (exact - not the lossy defer-max trick)
      if (__any(m_new > m[t])) {
        const float alpha = __builtin_amdgcn_exp2f(m[t] - m_new);
        l[t] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[t][0][r] *= alpha; o[t][1][r] *= alpha; }
        m[t] = m_new;
      }
      float ls = 0.f;
      float pr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(st[r] - m[t]); ls += pr[r]; }
      l[t] += ls;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        h8_t pb;
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[e] = (_Float16)pr[8 * kk + e];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) o[t][mt] = mfma32(vf[kk][mt], pb, o[t][mt]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = kn[ks];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) vf[kk][mt] = vn[kk][mt];
  }
  // ---- merge the 4 key-partials ----
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    l[t] += __shfl_xor(l[t], 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_part[t][wave][r][lane] = o[t][0][r]; s_part[t][wave][16 + r][lane] = o[t][1][r]; }
    s_part[t][wave][32][lane] = m[t];
    s_part[t][wave][33][lane] = l[t];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (q0 + t * 32 >= nq) break;
    float mw[4], mt_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = s_part[t][w][32][lane]; mt_all = fmaxf(mt_all, mw[w]); }
    float sc[4], lt = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sc[w] = (mw[w] == -INFINITY) ? 0.f : exp2f(mw[w] - mt_all);
      lt += s_part[t][w][33][lane] * sc[w];
    }
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    _Float16* orow = ctx + ((size_t)s * NP + q0 + t * 32 + j) * 256 + h * 64;
    // wave w finalises combined registers R = 8w .. 8w+7 (R = mt*16 + r): two groups of 4 consecutive channels
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int R0 = wave * 8 + gq * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += s_part[t][w][R0 + e][lane] * sc[w];
        v[e] = acc * inv;
      }
      const int mt = R0 >> 4, r = R0 & 15;
      const int d = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      *reinterpret_cast<h4_t*>(orow + d) = to_h4(v[0], v[1], v[2], v[3]);
    }
  }
}
template <int QT>
static void launch_attn(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d, bool cross,
                        _Float16* ctx, hipStream_t s) {
  constexpr size_t smem = (size_t)QT * 4 * 34 * 64 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lg_attention<QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_lg_attention<QT>), dim3(d.NP / (32 * QT), 4, d.S), dim3(256), smem, s, q, k, vt, lens, d.NP, cross ? 1 : 0,
                     ctx);
}
void launch_lg_attention(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d, bool cross,
                         _Float16* ctx, hipStream_t s) {
  // throughput batches: two query tiles per wave (half the K/V fragment traffic); a few pairs only: one tile per wave
  // so the launch still has enough workgroups to cover the CUs (latency mode)
  if (d.S * (d.NP / 64) * 4 >= 512) launch_attn<2>(q, k, vt, lens, d, cross, ctx, s);
  else launch_attn<1>(q, k, vt, lens, d, cross, ctx, s);
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
constexpr int kFfnTok = 64, kFfnLd = 520;
struct FfnTail {
  int dbg;                                  // ablation (SSHIP_FFN_DBG): 1 skip ffn.0 MFMAs, 2 skip LN/GELU math, 4 skip ffn.3, 8 skip tail
  int copies0, copies3, copiesp;            // weight replicas (workgroup b reads replica b % copies)
  size_t stride0, stride3, stridep;         // halfs between replicas
  IgemmArgs proj;          // epilogue arguments of the fused projection (wpack/bias/outputs/rope/np/flags/cout/H)
  const float* match_w;    // final block only: matchability weights [256] ...
  float match_b;
  float* logsig;           // ... -> logsigmoid(z) per token
};
// NW = waves per workgroup.  8: one 512-thread workgroup per CU (the kernel needs > 128 VGPRs), every wave owns one
// row block per GEMM.  4: two independent 256-thread workgroups per CU, every wave owns RB = 2 row blocks and runs them
// back to back on the same LDS token tile - the staging / LayerNorm barriers / epilogue stores of one workgroup
// overlap the MFMA phases of the other instead of idling the CU.  Same packed weights, same arithmetic order.
template <int NEXT_MT, bool HEADS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_lg_ffn(const _Float16* __restrict__ ctx, const _Float16* __restrict__ w0p,
                                                const float* __restrict__ b0, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const _Float16* __restrict__ w3p,
                                                const float* __restrict__ b3, _Float16* __restrict__ x, FfnTail tail) {
  constexpr int RB = 8 / NW, NTHR = NW * 64;
  constexpr int G0 = 8 / RB;    // ffn.0 k-steps per register-prefetch group
  constexpr int G3 = 16 / RB;   // ffn.3 k-steps per group
  __shared__ __attribute__((aligned(16))) _Float16 s_x[kFfnTok * kFfnLd];
  __shared__ float s_red[NW][kFfnTok];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const size_t t0 = (size_t)blockIdx.x * kFfnTok;
  for (int u = tid; u < kFfnTok * 64; u += NTHR) {
    const int tok = u >> 6, part = u & 63;
    const _Float16* src = part < 32 ? x + (t0 + tok) * 256 + part * 8 : ctx + (t0 + tok) * 256 + (part - 32) * 8;
    *reinterpret_cast<uint4*>(s_x + tok * kFfnLd + part * 8) = (tail.dbg & 16) ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(src);
  }
  __syncthreads();
  // ---- ffn.0 : row blocks cb = wave*RB + rb, rows [64 cb, +64) x 64 tokens, K = 512 ----
  f16x_t acc[RB][2][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][m][n][r] = 0.f;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    // Weight fragments stream from L2 (no reuse between waves); keep TWO groups of G0 k-steps in flight in registers
    // so ~1k cycles of L2 latency are covered by the MFMAs of the previous group and the co-resident waves.
    const _Float16* wp = w0p + (blockIdx.x % tail.copies0) * tail.stride0 + (size_t)(wave * RB + rb) * (32 * 2 * 512) + lane * 8;  // packed [cb][k16][mt][lane][8]
    h8_t ab[2][G0][2];
#pragma unroll
    for (int i = 0; i < G0; ++i) {
      ab[0][i][0] = *reinterpret_cast<const h8_t*>(wp + (i * 2 + 0) * 512);
      ab[0][i][1] = *reinterpret_cast<const h8_t*>(wp + (i * 2 + 1) * 512);
    }
#pragma unroll
    for (int grp = 0; grp < 32 / G0; ++grp) {
      if (tail.dbg & 1) break;
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
        const h8_t bf0 = *reinterpret_cast<const h8_t*>(s_x + j * kFfnLd + ks * 16 + hh * 8);
        const h8_t bf1 = *reinterpret_cast<const h8_t*>(s_x + (32 + j) * kFfnLd + ks * 16 + hh * 8);
        acc[rb][0][0] = mfma32(ab[grp & 1][i][0], bf0, acc[rb][0][0]);
        acc[rb][0][1] = mfma32(ab[grp & 1][i][0], bf1, acc[rb][0][1]);
        acc[rb][1][0] = mfma32(ab[grp & 1][i][1], bf0, acc[rb][1][0]);
        acc[rb][1][1] = mfma32(ab[grp & 1][i][1], bf1, acc[rb][1][1]);
      }
    }
  }
  // ---- bias, LayerNorm(512) over the row dimension (spread over regs, lane^32 and the waves), GELU ----
  float sum[2] = {0.f, 0.f};
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(b0 + (wave * RB + rb) * 64 + m * 32 + hh * 4 + g * 8);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[rb][m][n][4 * g + 0] += bv.x; acc[rb][m][n][4 * g + 1] += bv.y; acc[rb][m][n][4 * g + 2] += bv.z; acc[rb][m][n][4 * g + 3] += bv.w;
          sum[n] += (acc[rb][m][n][4 * g + 0] + acc[rb][m][n][4 * g + 1]) + (acc[rb][m][n][4 * g + 2] + acc[rb][m][n][4 * g + 3]);
        }
      }
  float mean[2], rstd[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    sum[n] += __shfl_xor(sum[n], 32, 64);
    if (hh == 0) s_red[wave][n * 32 + j] = sum[n];
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[w][n * 32 + j];
    mean[n] = t * (1.0f / 512.0f);
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    float sq = 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dlt = acc[rb][m][n][r] - mean[n]; sq += dlt * dlt; }
    sq += __shfl_xor(sq, 32, 64);
    if (hh == 0) s_red[wave][n * 32 + j] = sq;
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[w][n * 32 + j];
    rstd[n] = rsqrtf(t * (1.0f / 512.0f) + 1e-5f);
  }
  // every wave has passed two barriers since its last read of s_x: the tile can be overwritten with the hidden tile
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = (wave * RB + rb) * 64 + m * 32 + hh * 4 + g * 8;
        const float4 gv = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        const float gg[4] = {gv.x, gv.y, gv.z, gv.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = (acc[rb][m][n][4 * g + e] - mean[n]) * rstd[n] * gg[e] + bb[e];
            o[e] = (tail.dbg & 2) ? y : 0.5f * y * (1.0f + fast_erf(y * 0.70710678118654752f));
          }
          *reinterpret_cast<h4_t*>(s_x + (n * 32 + j) * kFfnLd + c) = to_h4(o[0], o[1], o[2], o[3]);
        }
      }
  __syncthreads();
  // ---- ffn.3 : row blocks cb = wave*RB + rb, rows [32 cb, +32) x 64 tokens, K = 512, + residual ----
  f16x_t ac2[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) ac2[rb][n][r] = 0.f;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const _Float16* wp = w3p + (blockIdx.x % tail.copies3) * tail.stride3 + (size_t)(wave * RB + rb) * (32 * 512) + lane * 8;  // packed [cb][k16][mt = 0][lane][8]
    h8_t a3[2][G3];
#pragma unroll
    for (int i = 0; i < G3; ++i) a3[0][i] = *reinterpret_cast<const h8_t*>(wp + i * 512);
#pragma unroll
    for (int grp = 0; grp < 32 / G3; ++grp) {
      if (tail.dbg & 4) break;
      if (grp + 1 < 32 / G3) {
#pragma unroll
        for (int i = 0; i < G3; ++i) a3[(grp + 1) & 1][i] = *reinterpret_cast<const h8_t*>(wp + ((grp + 1) * G3 + i) * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G3; ++i) {
        const int ks = grp * G3 + i;
        const h8_t bf0 = *reinterpret_cast<const h8_t*>(s_x + j * kFfnLd + ks * 16 + hh * 8);
        const h8_t bf1 = *reinterpret_cast<const h8_t*>(s_x + (32 + j) * kFfnLd + ks * 16 + hh * 8);
        ac2[rb][0] = mfma32(a3[grp & 1][i], bf0, ac2[rb][0]);
        ac2[rb][1] = mfma32(a3[grp & 1][i], bf1, ac2[rb][1]);
      }
    }
  }
  if constexpr (NEXT_MT > 0) __syncthreads();  // all waves are done reading the hidden tile: s_x gets the new x
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = (wave * RB + rb) * 32 + hh * 4 + g * 8;
      const float4 bv = *reinterpret_cast<const float4*>(b3 + c);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        h4_t* px = reinterpret_cast<h4_t*>(x + (t0 + n * 32 + j) * 256 + c);
        if (tail.dbg & 64) continue;
        const h4_t o = *px;
        const h4_t xn = to_h4((float)o[0] + (ac2[rb][n][4 * g + 0] + bv.x), (float)o[1] + (ac2[rb][n][4 * g + 1] + bv.y),
                              (float)o[2] + (ac2[rb][n][4 * g + 2] + bv.z), (float)o[3] + (ac2[rb][n][4 * g + 3] + bv.w));
        *px = xn;
        if constexpr (NEXT_MT > 0) *reinterpret_cast<h4_t*>(s_x + (n * 32 + j) * kFfnLd + c) = xn;
      }
    }
  if constexpr (NEXT_MT > 0) {
    __syncthreads();
    // ---- fused next projection: row blocks cb = wave*RB + rb, rows [NEXT_MT*32*cb, +NEXT_MT*32) x 64 tokens, K = 256 ----
    // M-tiles of the V segment run with SWAPPED operands (A = token tile, B = weights): the accumulator then holds
    // D[token][channel] with lane = channel and 8 consecutive registers = the 8 keys of one PV A-fragment unit, so V^T is
    // written in fragment order with one 16-byte store per lane (the 2-byte transposing stores it replaces were ~16x
    // write-amplified and dominated the kernel's non-MFMA time).
    const int t_seg = (tail.proj.flags >> 4) & 0xf;
    // which of a row block's M-tiles belong to the V segment is wave-uniform; the loop is instantiated per pattern
    // (VMASK bit m = tile m is V) so its body stays branch-free: self Wqkv (3 tiles/block): 000, 110 (block 5), 111;
    // cross [to_qk|to_v] (2 tiles/block): 00, 11.
    auto run_tail = [&](auto vmask_c, int cb) {
      constexpr int VMASK = decltype(vmask_c)::value;
      f16x_t ac3[NEXT_MT][2];
#pragma unroll
      for (int m = 0; m < NEXT_MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) ac3[m][n][r] = 0.f;
      const _Float16* wp = tail.proj.wpack + (blockIdx.x % tail.copiesp) * tail.stridep + (size_t)cb * (16 * NEXT_MT * 512) + lane * 8;  // [cb][k16][mt][lane][8]
#pragma unroll 16
      for (int ks = 0; ks < 16; ++ks) {
        if (tail.dbg & 8) break;
        const h8_t bf0 = *reinterpret_cast<const h8_t*>(s_x + j * kFfnLd + ks * 16 + hh * 8);
        const h8_t bf1 = *reinterpret_cast<const h8_t*>(s_x + (32 + j) * kFfnLd + ks * 16 + hh * 8);
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) {
          const h8_t a = *reinterpret_cast<const h8_t*>(wp + (ks * NEXT_MT + m) * 512);
          if ((VMASK >> m) & 1) {
            ac3[m][0] = mfma32(bf0, a, ac3[m][0]);
            ac3[m][1] = mfma32(bf1, a, ac3[m][1]);
          } else {
            ac3[m][0] = mfma32(a, bf0, ac3[m][0]);
            ac3[m][1] = mfma32(a, bf1, ac3[m][1]);
          }
        }
      }
      if (tail.dbg & 32) return;
      if constexpr (HEADS) {
        const int NP = tail.proj.np, nt32 = NP >> 5;
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) {
          const int R0 = (cb * NEXT_MT + m) * 32;  // first output row of this M-tile
          if ((VMASK >> m) & 1) {
            const int hd = (R0 >> 6) & 3, mth = (R0 >> 5) & 1;  // head, 32-channel half of the head
            const float bv = tail.proj.bias[R0 + j];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              const size_t token = t0 + n * 32;
              const int sq = (int)(token / NP), kt = (int)(token - (size_t)sq * NP) >> 5;
              _Float16* dst = static_cast<_Float16*>(tail.proj.out2) + (((size_t)sq * 4 + hd) * nt32 + kt) * 2048 + lane * 8;
#pragma unroll
              for (int kk = 0; kk < 2; ++kk) {
                h8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)(ac3[m][n][8 * kk + e] + bv);
                *reinterpret_cast<h8_t*>(dst + (kk * 2 + mth) * 512) = o;
              }
            }
          } else {
            f16x_t one[1][2] = {{ac3[m][0], ac3[m][1]}};
            EpiHeads::template run<1, 2>(tail.proj, one, 0, (int)(t0 >> 5), j, R0, hh);
          }
        }
      } else {
        EpiF16<false, false>::template run<NEXT_MT, 2>(tail.proj, ac3, 0, (int)(t0 >> 5), j, cb * NEXT_MT * 32, hh);
      }
    };
    constexpr int FULL = (1 << NEXT_MT) - 1;
#pragma unroll 1
    for (int rb = 0; rb < RB; ++rb) {
      const int cb = wave * RB + rb;
      int vmask = 0;
      if constexpr (HEADS) {
#pragma unroll
        for (int m = 0; m < NEXT_MT; ++m) vmask |= ((((cb * NEXT_MT + m) >> 3) == t_seg) ? 1 : 0) << m;
      }
      vmask = __builtin_amdgcn_readfirstlane(vmask);
      if (vmask == 0) run_tail(std::integral_constant<int, 0>{}, cb);
      else if (vmask == FULL) run_tail(std::integral_constant<int, FULL>{}, cb);
      else run_tail(std::integral_constant<int, (FULL & ~1)>{}, cb);  // the only mixed pattern: tile 0 is K, the rest V
    }
    if (tail.logsig) {  // matchability head of the last block: one wave per 64 / NW tokens
#pragma unroll 1
      for (int tk = wave * (kFfnTok / NW); tk < (wave + 1) * (kFfnTok / NW); ++tk) {
        const h4_t v = *reinterpret_cast<const h4_t*>(s_x + tk * kFfnLd + lane * 4);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) d += (float)v[e] * tail.match_w[lane * 4 + e];
        const float z = wave_sum(d) + tail.match_b;
        if (lane == 0) tail.logsig[t0 + tk] = fminf(z, 0.f) - log1pf(expf(-fabsf(z)));
      }
    }
  }
}
template <int NEXT_MT, bool HEADS, typename... A>
static void launch_ffn_nw(int nw, dim3 grid, hipStream_t s, A... args) {
  if (nw == 4) hipLaunchKernelGGL((k_lg_ffn<NEXT_MT, HEADS, 4>), grid, dim3(256), 0, s, args...);
  else hipLaunchKernelGGL((k_lg_ffn<NEXT_MT, HEADS, 8>), grid, dim3(512), 0, s, args...);
}
// next == nullptr: plain FFN.  Otherwise the projection `next` (packed with ct = 32 * next_mt rows per wave) runs on
// the updated tile; heads = true -> EpiHeads (q/k/vt, rope_segs, t_seg), false -> fp16 rows to `out` (+ matchability).
void launch_lg_ffn(const ConvW& w0, const ConvW& w3, const float* gamma, const float* beta, const _Float16* ctx,
                   _Float16* x, LgDims d, const ConvW* next, bool heads, int rope_segs, int t_seg, const float* rope,
                   _Float16* q, _Float16* k, _Float16* vt, _Float16* out, const float* match_w, float match_b,
                   float* logsig, hipStream_t s) {
  const int tokens = d.S * d.NP;
  FfnTail t{};
  static const int dbg = getenv("SSHIP_FFN_DBG") ? atoi(getenv("SSHIP_FFN_DBG")) : 0;
  t.dbg = dbg;
  t.copies0 = w0.copies; t.stride0 = w0.copy_stride; t.copies3 = w3.copies; t.stride3 = w3.copy_stride;
  t.copiesp = 1; t.stridep = 0;
  // SUPERSLAM_HIP_FFN_WAVES=8|4 (default 8; 4 measured 6 % slower end to end): see k_lg_ffn
  static const int nw = (getenv("SUPERSLAM_HIP_FFN_WAVES") && atoi(getenv("SUPERSLAM_HIP_FFN_WAVES")) == 4) ? 4 : 8;
  dim3 grid(tokens / kFfnTok);
  if (!next) {
    launch_ffn_nw<0, false>(nw, grid, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
    return;
  }
  t.proj = token_args(*next, x, 256, nullptr, 0, d);
  t.copiesp = next->copies; t.stridep = next->copy_stride;
  t.proj.out0 = heads ? (void*)q : (void*)out; t.proj.out1 = k; t.proj.out2 = vt; t.proj.aux = rope;
  t.proj.flags = rope_segs | (t_seg << 4); t.proj.ostride = 256;
  t.match_w = match_w; t.match_b = match_b; t.logsig = logsig;
  const int mt = next->cout / 256;  // rows per wave / 32: 768 -> 3, 512 -> 2, 256 -> 1
  if (heads && mt == 3) launch_ffn_nw<3, true>(nw, grid, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
  else if (heads && mt == 2) launch_ffn_nw<2, true>(nw, grid, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
  else launch_ffn_nw<1, false>(nw, grid, s, ctx, w0.w, w0.bias, gamma, beta, w3.w, w3.bias, x, t);
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
  hipLaunchKernelGGL(k_lg_sim, dim3(d.NP / 128, d.NP / 32, d.S / 2), dim3(256), 0, s, md, lens, d.NP, sim);
}

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
__global__ __launch_bounds__(256) void k_assign_col_lse(const float* __restrict__ sim, const int* __restrict__ lens,
                                                        int NP, float* __restrict__ ws) {
  __shared__ float s_m[4][64], s_s[4][64];
  const int pair = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  float m = -INFINITY, sum = 0.f;
  if (j < n1) {
    const float* col = sim + (size_t)pair * NP * NP + j;
    for (int i = rg; i < n0; i += 4) {
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
    for (int g = 1; g < 4; ++g) M = fmaxf(M, s_m[g][cl]);
    float S = 0.f;
    for (int g = 0; g < 4; ++g) if (s_m[g][cl] > -INFINITY) S += s_s[g][cl] * expf(s_m[g][cl] - M);
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
  int bj = 0x7fffffff;
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
__global__ __launch_bounds__(256) void k_assign_col_arg(const float* __restrict__ sim, const float* __restrict__ logsig,
                                                        const int* __restrict__ lens, int NP, float* __restrict__ ws) {
  __shared__ float s_v[4][64];
  __shared__ int s_i[4][64];
  const int pair = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  const int n0 = min(max(lens[2 * pair], 0), NP), n1 = min(max(lens[2 * pair + 1], 0), NP);
  float* w = ws + (size_t)pair * 5 * NP;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (j < n1) {
    const float* col = sim + (size_t)pair * NP * NP + j;
    const float* ls0 = logsig + (size_t)(2 * pair) * NP;
    const float lj = w[NP + j], ls1 = logsig[(size_t)(2 * pair + 1) * NP + j];
    for (int i = rg; i < n0; i += 4) {
      const float sv = col[(size_t)i * NP];
      const float v = ((sv - w[i]) + (sv - lj)) + (ls0[i] + ls1);
      if (v > best) { best = v; bi = i; }
    }
  }
  s_v[rg][cl] = best; s_i[rg][cl] = bi;
  __syncthreads();
  if (rg == 0 && j < n1) {
    for (int g = 1; g < 4; ++g)
      if (s_v[g][cl] > best || (s_v[g][cl] == best && s_i[g][cl] < bi)) { best = s_v[g][cl]; bi = s_i[g][cl]; }
    reinterpret_cast<int*>(w)[4 * NP + j] = bi;
  }
}
// filter_matches(scores, 0.1): mutual check, mscores0 = mutual ? exp(max0) : 0, matches0 = valid ? m0 : -1.
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
    const bool mutual = reinterpret_cast<const int*>(w)[4 * NP + j] == i;
    ms = mutual ? expf(w[2 * NP + i]) : 0.f;
    mj = (mutual && ms > thr) ? j : -1;
  }
  matches0[(size_t)pair * max_kp + i] = mj;
  mscores0[(size_t)pair * max_kp + i] = ms;
}
void launch_lg_assign(const float* sim, const float* logsig, const int* lens, LgDims d, float* ws, int max_kp,
                      int32_t* matches0, float* mscores0, float thr, hipStream_t s) {
  const int P = d.S / 2;
  hipLaunchKernelGGL(k_assign_row_lse, dim3(d.NP / 4, P), dim3(256), 0, s, sim, lens, d.NP, ws);
  hipLaunchKernelGGL(k_assign_col_lse, dim3(d.NP / 64, P), dim3(256), 0, s, sim, lens, d.NP, ws);
  hipLaunchKernelGGL(k_assign_row_arg, dim3(d.NP / 4, P), dim3(256), 0, s, sim, logsig, lens, d.NP, ws);
  hipLaunchKernelGGL(k_assign_col_arg, dim3(d.NP / 64, P), dim3(256), 0, s, sim, logsig, lens, d.NP, ws);
  hipLaunchKernelGGL(k_assign_final, dim3((max_kp + 255) / 256, P), dim3(256), 0, s, lens, d.NP, ws, max_kp, thr,
                     matches0, mscores0);
}

}  // namespace sship
