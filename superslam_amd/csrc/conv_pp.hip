// Ping-pong 3x3 convolution for SuperPoint (utils/convert_superpoint_to_onnx.py:38-49) - the production conv kernel.
//
// The strip kernel (conv_strip.hip) runs ONE workgroup per CU (72 KiB of weights + the input tile fill the LDS) and
// its 8 waves move through  stage -> MFMA -> epilogue  in lock-step, so the matrix pipe idles while they all do
// VALU / LDS / VMEM work (ablation in DESIGN.md: epilogue 19-30 %, conv1a phase 12 %, MFMA loop itself ~85 % efficient).
// CDNA4 puts two waves of a 512-thread workgroup on every SIMD; here they get complementary roles:
//
//   group g = wave >> 2 (waves g*4 .. g*4+3: one per SIMD) owns its own 8 x 32 pixel tile stream and its own LDS input
//   buffer.  In half-step s, group (s & 1) runs the uninterrupted 144-MFMA loop of its current tile while the other
//   group - on the same SIMDs - writes out the epilogue of its previous tile, stages its next tile into LDS and
//   issues the prefetch for the one after.  One workgroup barrier per half-step; the matrix pipe of every SIMD is
//   fed by one wave at a time, the other wave's VALU/LDS/VMEM work rides in its shadow.
//
// LDS: 2 x 43,520 B input tiles (10 x 34 px x 64 ch, NO padding: a 16-byte-unit XOR swizzle  unit ^= (px >> 1) & 7
// makes every ds_read_b128 lane group hit 16 distinct slots) + 73,728 B weights = 160,768 B.
//   CIN = 64 : CT = 64 (MT = 2);   CIN = 128: CT = 32 (MT = 1), two 64-channel chunks per tile (two MFMA half-steps).
//   FUSE1A (conv1b): the tile is produced by conv1a on the matrix cores (K = 9 -> 16) from u8 pixels that each lane
//   prefetched into registers two half-steps earlier; no LDS patch, no extra barrier.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "igemm.h"
#include "kernels.h"

// store units (of 2 MT) the MFMA group writes itself; -1 = measured per-kernel default: none for the 64-channel layers
// (incl. the fused conv1a+conv1b), everything for the 128-channel ones (build.py --variant -DSSHIP_PP_EPI=n to sweep)
#ifndef SSHIP_PP_EPI
#define SSHIP_PP_EPI -1
#endif
// staging geometry hoisted out of the tile loop: -1 = per-kernel default (128-input-channel layers only), 0 / 1 force
#ifndef SSHIP_PP_HOIST
#define SSHIP_PP_HOIST -1
#endif
// role tracing (SSHIP_PP_TRACE=1 at run time) is compiled in only on request: it costs registers in the fused kernel
#ifndef SSHIP_PP_TRACE_BUILD
#define SSHIP_PP_TRACE_BUILD 0
#endif
// s_setprio level of the data-movement role (0 = off; the MFMA role runs at priority 0)
#ifndef SSHIP_PP_PRIO
#define SSHIP_PP_PRIO 2
#endif

namespace sship {

struct PpArgs {
  const _Float16* in;    // channels-last fp16 [B,H,W,CIN]   (unused when FUSE1A)
  const uint8_t* img;    // u8 [B,H,W]                        (FUSE1A)
  const _Float16* w1a;   // conv1a A fragments [2][64][8] fp16 (FUSE1A)
  const float* b1a;      // conv1a bias [64]                  (FUSE1A)
  const _Float16* wpack; // packed weights [cb][chunk][tap][kstep][mt][lane][8]
  const float* bias;
  _Float16* out;
  int B, H, W, cout;
  int dbg;  // ablation (SSHIP_PP_DBG): 1 skip staging, 2 skip epilogue, 4 skip MFMA loop, 8 skip prefetch
  unsigned long long* trace;  // SSHIP_PP_TRACE: [workgroup][group][4] clocks of half-steps 8..9: epilogue, stage, prefetch, mfma
};

constexpr int P_TH = 8, P_TW = 32, P_THH = 10, P_TWH = 34;
constexpr int P_IN_HALFS = P_THH * P_TWH * 64;      // 21,760 halfs = 43,520 B
constexpr int P_W_HALFS = 36864;                    // 72 KiB
constexpr int P_IN_UNITS = P_THH * P_TWH * 8;       // 2720 sixteen-byte units
constexpr int P_IN_IT = (P_IN_UNITS + 255) / 256;   // 11 per thread of a 256-thread group
constexpr int P_NT1A = (P_THH * P_TWH + 31) / 32;   // 11 conv1a N-tiles of 32 halo pixels

// swizzled LDS offset (halfs) of 16-byte unit `unit` of halo pixel (row, col)
__device__ __forceinline__ int pp_lds(int row, int col, int unit) {
  return (row * P_TWH + col) * 64 + ((unit ^ ((col >> 1) & 7)) << 3);
}

template <int CIN, int CT, bool POOL, bool FUSE1A>
__global__ __launch_bounds__(512, 2) void conv3x3_pp(PpArgs p) {
  constexpr int MT = CT / 32, NCHUNK = CIN / 64;
  constexpr int HOIST_GEOMETRY = SSHIP_PP_HOIST < 0 ? (CIN == 128 ? 1 : 0) : SSHIP_PP_HOIST;
  constexpr int EPI_MFMA = SSHIP_PP_EPI < 0 ? (CIN == 128 ? 2 : 0) : (SSHIP_PP_EPI > 2 * MT ? 2 * MT : SSHIP_PP_EPI);
  static_assert(NCHUNK * 9 * 4 * MT * 512 == P_W_HALFS, "weights must fill exactly 72 KiB");
  static_assert(!FUSE1A || CIN == 64, "conv1a fusion feeds a 64-channel layer");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* s_w = reinterpret_cast<_Float16*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);  // wave-uniform role selector (keeps the role branch scalar)
  const int gw = wave & 3, gt = tid & 255;
  _Float16* my_in = s_w + P_W_HALFS + grp * P_IN_HALFS;

  const int tiles_x = (p.W + P_TW - 1) / P_TW, tiles_y = (p.H + P_TH - 1) / P_TH;
  const int ntiles = p.B * tiles_x * tiles_y;
  const int cb = blockIdx.y;
  const int t_begin = (int)((long long)blockIdx.x * ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int n_wg = t_end - t_begin;
  if (n_wg <= 0) return;
  // group g takes tiles t_begin + g, t_begin + g + 2, ...; a work item is (tile, 64-channel chunk)
  const int nw0 = ((n_wg + 1) >> 1) * NCHUNK, nw1 = (n_wg >> 1) * NCHUNK;
  const int NW = grp ? nw1 : nw0;
  const int s_end = max(2 * nw0 - 1, 2 * nw1);  // last half-step in which some group still has an epilogue to write

  {  // weights: once per workgroup, all 512 threads
    const _Float16* wsrc = p.wpack + (size_t)cb * P_W_HALFS;
    for (int u = tid; u < P_W_HALFS / 8; u += 512)
      *reinterpret_cast<uint4*>(s_w + u * 8) = *reinterpret_cast<const uint4*>(wsrc + u * 8);
  }
  // bias of this workgroup's CT channels lives in LDS (registers are the scarce resource here: acc 64 + prefetch 44 +
  // fragment double buffers 32 per lane), read back as float4 in the epilogue
  float* s_bias = reinterpret_cast<float*>(s_w + P_W_HALFS + 2 * P_IN_HALFS);
  if (tid < CT) s_bias[tid] = p.bias[cb * CT + tid];
  if constexpr (FUSE1A) { if (tid >= 64 && tid < 128) s_bias[tid] = p.b1a[tid - 64]; }  // conv1a bias at s_bias[64..127]
  // fragment-read offsets: B fragment of (row n + ky, col j + kx), k-step ks -> unit (2 ks + hh) ^ ((j + kx) >> 1 & 7)
  int boff[3][4];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) boff[kx][ks] = (j + kx) * 64 + (((2 * ks + hh) ^ (((j + kx) >> 1) & 7)) << 3);

  auto tile_of = [&](int item) { return t_begin + grp + 2 * (item / NCHUNK); };
  auto tile_coords = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % tiles_x;
    const int r = t / tiles_x;
    x0 = tx * P_TW; y0 = (r % tiles_y) * P_TH; b = r / tiles_y;
  };

  // ---------------- staging (plain variant): global -> registers (prefetch) -> LDS ----------------
  uint4 rin[FUSE1A ? 1 : P_IN_IT];
  // Edge tiles load from a clamped (always valid) address and are masked when the tile is WRITTEN to LDS one step
  // later: selecting zero here would need the data, i.e. a vmcnt(0) in front of every one of the 11 loads (traced:
  // 30k+ clocks for an edge tile's prefetch against ~1k for an interior one).
  auto prefetch_in = [&](int item) {
    if constexpr (!FUSE1A) {
      int b, y0, x0;
      tile_coords(tile_of(item), b, y0, x0);
      const int chunk = item % NCHUNK;
      int gtv = gt;
      // CIN = 64 (MT = 2, ~250 VGPRs): opaque, keeps the per-iteration geometry from being hoisted into ~50 live VGPRs.
      // CIN = 128 (MT = 1, ~165 VGPRs) has the registers: the offsets are computed once per launch instead of ~330
      // integer instructions per half-step in the (critical) data-movement role.
      if (HOIST_GEOMETRY == 0) asm volatile("" : "+v"(gtv));
      const bool interior = y0 >= 1 && y0 + P_TH + 1 <= p.H && x0 >= 1 && x0 + P_TW + 1 <= p.W;
      if (interior) {
        const _Float16* base = p.in + ((size_t)(b * p.H + (y0 - 1)) * p.W + (x0 - 1)) * CIN + chunk * 64;
#pragma unroll
        for (int i = 0; i < P_IN_IT; ++i) {
          const int u = gtv + i * 256;
          const int pix = u >> 3, part = u & 7;
          const int py = pix / P_TWH, px = pix - py * P_TWH;
          if (i < P_IN_IT - 1 || u < P_IN_UNITS) rin[i] = *reinterpret_cast<const uint4*>(base + (py * p.W + px) * CIN + part * 8);
        }
      } else {
        const _Float16* img = p.in + (size_t)b * p.H * p.W * CIN + chunk * 64;
#pragma unroll
        for (int i = 0; i < P_IN_IT; ++i) {
          const int u = min(gtv + i * 256, P_IN_UNITS - 1);
          const int pix = u >> 3, part = u & 7;
          const int py = pix / P_TWH, px = pix - py * P_TWH;
          const int cy = min(max(y0 - 1 + py, 0), p.H - 1), cx = min(max(x0 - 1 + px, 0), p.W - 1);
          rin[i] = *reinterpret_cast<const uint4*>(img + ((size_t)cy * p.W + cx) * CIN + part * 8);
        }
      }
    }
  };
  auto stage_in = [&](int item) {
    if constexpr (!FUSE1A) {
      int b, y0, x0;
      tile_coords(tile_of(item), b, y0, x0);
      const bool interior = y0 >= 1 && y0 + P_TH + 1 <= p.H && x0 >= 1 && x0 + P_TW + 1 <= p.W;
      int gtv = gt;
      if (HOIST_GEOMETRY == 0) asm volatile("" : "+v"(gtv));
#pragma unroll
      for (int i = 0; i < P_IN_IT; ++i) {
        const int u = gtv + i * 256;
        const int pix = u >> 3, part = u & 7;
        const int py = pix / P_TWH, px = pix - py * P_TWH;
        uint4 v = rin[i];
        if (!interior) {  // zero padding outside the image (uniform branch: only edge tiles pay for the index math)
          const int gy = y0 - 1 + py, gx = x0 - 1 + px;
          if (!(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)) v = make_uint4(0, 0, 0, 0);
        }
        if (i < P_IN_IT - 1 || u < P_IN_UNITS) *reinterpret_cast<uint4*>(my_in + pp_lds(py, px, part)) = v;
      }
    }
  };
  // ---------------- staging (FUSE1A): u8 pixels -> registers (prefetch) -> conv1a MFMA -> LDS ----------------
  // wave gw of the group owns conv1a N-tiles gw, gw + 4, gw + 8 (< 11); lane (j, hh) of N-tile nt is halo pixel
  // q = 32 nt + j and needs taps 8 hh .. 8 hh + 7 of its 3x3 patch (taps 9..15 are the zero padding of K).
  constexpr int NT_W = 3;
  unsigned rp[FUSE1A ? NT_W : 1][3];  // per N-tile: the pixel's 3x3 patch as 3 dwords (bytes 0..2 of patch row r)
  // tile-invariant geometry of this lane's (up to) 3 halo pixels: q = 32 (gw + 4k) + j -> (py, px), its LDS base and
  // the swizzle term; computed once per launch (two VGPRs per N-tile)
  int c1_pypx[FUSE1A ? NT_W : 1], c1_lds[FUSE1A ? NT_W : 1];
  if constexpr (FUSE1A) {
#pragma unroll
    for (int k = 0; k < NT_W; ++k) {
      const int q = min((gw + 4 * k) * 32 + j, P_THH * P_TWH - 1);
      const int py = q / P_TWH, px = q - py * P_TWH;
      c1_pypx[k] = (py << 16) | px;
      c1_lds[k] = ((py * P_TWH + px) * 64 + hh * 4) | (((px >> 1) & 7) << 24);  // half offset | swizzle term
    }
  }
  auto prefetch_u8 = [&](int item) {
    if constexpr (FUSE1A) {
      int b, y0, x0;
      tile_coords(tile_of(item), b, y0, x0);
      const uint8_t* im = p.img + (size_t)b * p.H * p.W;
      // interior: the whole 12 x 36 patch (+3 bytes of dword over-read) lies inside the image -> three unaligned
      // dword loads per pixel, no clamping (24 clamped byte loads per lane made this the longest part of the step)
      const bool interior = y0 >= 2 && y0 + P_TH + 2 <= p.H && x0 >= 2 && x0 + P_TW + 2 + 3 <= p.W;
#pragma unroll
      for (int k = 0; k < NT_W; ++k) {
        const int py = c1_pypx[k] >> 16, px = c1_pypx[k] & 0xffff;
        if (interior) {
          const uint8_t* pp = im + (size_t)(y0 - 2 + py) * p.W + (x0 - 2 + px);
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            unsigned v;
            __builtin_memcpy(&v, pp + (size_t)r * p.W, 4);  // unaligned global_load_dword
            rp[k][r] = v;
          }
        } else {
          // edge tiles: one dword per patch row from a clamped position, fixed up (shift + zero fill) in stage_conv1a once
          // the data is there - 27 clamped byte loads with selects made every edge prefetch wait for HBM nine times over
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int cy = min(max(y0 - 2 + py + r, 0), p.H - 1), cx = min(max(x0 - 2 + px, 0), p.W - 4);
            unsigned v;
            __builtin_memcpy(&v, im + (size_t)cy * p.W + cx, 4);
            rp[k][r] = v;
          }
        }
      }
    }
  };
  // conv1a A fragments; the bias rides in two of the K-padding slots (taps 9 and 10 carry fp16 hi / lo parts of the
  // fp32 bias, the matching B elements are 1.0), so the accumulator comes out of the MFMA already biased.
  h8_t a1a0, a1a1;
  if constexpr (FUSE1A) {
    a1a0 = *reinterpret_cast<const h8_t*>(p.w1a + lane * 8);
    a1a1 = *reinterpret_cast<const h8_t*>(p.w1a + 512 + lane * 8);
  }
  bool tr_stage = false;  // set by the half-step loop for the traced half-steps (SSHIP_PP_TRACE_BUILD)
  auto stage_conv1a = [&](int item) {
    if constexpr (FUSE1A) {
      unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
      if (SSHIP_PP_TRACE_BUILD && tr_stage) ts0 = __builtin_readcyclecounter();
      int b, y0, x0;
      tile_coords(tile_of(item), b, y0, x0);
      const bool interior = y0 >= 1 && y0 + P_TH + 1 <= p.H && x0 >= 1 && x0 + P_TW + 1 <= p.W;
      const bool patch_interior = y0 >= 2 && y0 + P_TH + 2 <= p.H && x0 >= 2 && x0 + P_TW + 2 + 3 <= p.W;  // as in prefetch_u8
      // three passes over this wave's (up to) 3 N-tiles so the six conv1a MFMAs issue back to back and their latency
      // (they queue behind the other group's MFMA stream on the same SIMD) overlaps instead of adding up
      h8_t bf[NT_W];
#pragma unroll
      for (int k = 0; k < NT_W; ++k) {
        if (!patch_interior) {  // undo the clamping of prefetch_u8: byte c of row r must be image column gx + c (0 outside)
          const int py = c1_pypx[k] >> 16, px = c1_pypx[k] & 0xffff;
          const int gx = x0 - 2 + px;
          const int sh = gx - min(max(gx, 0), p.W - 4);  // < 0 at the left border, > 0 at the right one
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int gy = y0 - 2 + py + r;
            unsigned v = rp[k][r];
            v = sh < 0 ? (sh <= -4 ? 0u : v << (-8 * sh)) : (sh >= 4 ? 0u : v >> (8 * sh));
            rp[k][r] = (gy >= 0 && gy < p.H) ? v : 0u;
          }
        }
        // taps (r, c) = byte c of dword r; cv convertTo: float(u8) * (1/255), then the engine's fp16 input
        _Float16 t[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) t[r][c] = (_Float16)((float)((rp[k][r] >> (8 * c)) & 0xffu) * (1.0f / 255.0f));
        // lanes hh = 0: taps 0..7; lanes hh = 1: tap 8, the two bias slots (1.0, 1.0) and zero padding
        bf[k][0] = hh ? t[2][2] : t[0][0];
        bf[k][1] = hh ? (_Float16)1.f : t[0][1];
        bf[k][2] = hh ? (_Float16)1.f : t[0][2];
        bf[k][3] = hh ? (_Float16)0.f : t[1][0];
        bf[k][4] = hh ? (_Float16)0.f : t[1][1];
        bf[k][5] = hh ? (_Float16)0.f : t[1][2];
        bf[k][6] = hh ? (_Float16)0.f : t[2][0];
        bf[k][7] = hh ? (_Float16)0.f : t[2][1];
      }
      if (SSHIP_PP_TRACE_BUILD && tr_stage) ts1 = __builtin_readcyclecounter();
      f16x_t d0[NT_W], d1[NT_W];
#pragma unroll
      for (int k = 0; k < NT_W; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { d0[k][r] = 0.f; d1[k][r] = 0.f; }
        if (gw + 4 * k >= P_NT1A) continue;  // wave-uniform
        d0[k] = mfma32(a1a0, bf[k], d0[k]);
        d1[k] = mfma32(a1a1, bf[k], d1[k]);
      }
      if (SSHIP_PP_TRACE_BUILD && tr_stage) ts2 = __builtin_readcyclecounter();
#pragma unroll
      for (int k = 0; k < NT_W; ++k) {
        if (gw + 4 * k >= P_NT1A) continue;
        if ((gw + 4 * k) * 32 + j < P_THH * P_TWH) {
          bool inside = true;
          if (!interior) {  // conv1b's zero padding is on conv1a's OUTPUT: only edge tiles have outside halo pixels
            const int gy = y0 - 1 + (c1_pypx[k] >> 16), gx = x0 - 1 + (c1_pypx[k] & 0xffff);
            inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
          }
          const int base = c1_lds[k] & 0xffffff, sw = c1_lds[k] >> 24;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // ReLU after the fp16 rounding, two values per instruction (v_pk_max_f16): rounding is monotone and keeps the
            // sign, so relu(fp16(x)) == fp16(relu(x))
            const h4_t z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            h4_t o0 = __builtin_elementwise_max(to_h4(d0[k][4 * g], d0[k][4 * g + 1], d0[k][4 * g + 2], d0[k][4 * g + 3]), z4);
            h4_t o1 = __builtin_elementwise_max(to_h4(d1[k][4 * g], d1[k][4 * g + 1], d1[k][4 * g + 2], d1[k][4 * g + 3]), z4);
            if (!interior && !inside) { o0 = z4; o1 = z4; }  // `interior` is uniform: inner tiles skip the selects
            const int u0 = (g ^ sw) << 3;  // channels 4 hh + 8 g .. (+3): unit g; M-tile 1: unit 4 + g
            *reinterpret_cast<h4_t*>(my_in + base + u0) = o0;
            *reinterpret_cast<h4_t*>(my_in + base + (u0 ^ 32)) = o1;
          }
        }
      }
      if (SSHIP_PP_TRACE_BUILD && tr_stage) {
        unsigned long long* o = p.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 10;
        o[6] = ts1 - ts0; o[7] = ts2 - ts1; o[8] = __builtin_readcyclecounter() - ts2;
      }
    }
  };

  f16x_t acc[MT][2];
  // ---------------- MFMA half-step: 36 k-steps of one 64-channel chunk, fragments double-buffered ----------------
  auto mfma_item = [&](int item) {
    const int chunk = item % NCHUNK;
    if (chunk == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    }
    const _Float16* wc = s_w + chunk * (9 * 4 * MT * 512) + lane * 8;
    const _Float16* ib = my_in + (gw * 2) * P_TWH * 64;
    // one wave per SIMD feeds the matrix pipe here, so LDS latency must be covered by this wave alone: fragments are
    // triple-buffered, the ds_reads of k-step i+2 are issued (and pinned) before the MFMAs of k-step i.
    h8_t fa[3][MT], fb[3][2];
    auto load_frags = [&](int idx, int buf) {
      const int tap = idx >> 2, ks = idx & 3, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[buf][m] = *reinterpret_cast<const h8_t*>(wc + ((tap * 4 + ks) * MT + m) * 512);
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[buf][n] = *reinterpret_cast<const h8_t*>(ib + (n + ky) * P_TWH * 64 + boff[kx][ks]);
    };
    load_frags(0, 0);
    load_frags(1, 1);
#pragma unroll
    for (int idx = 0; idx < 36; ++idx) {
      if (idx + 2 < 36) load_frags(idx + 2, (idx + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][n] = mfma32(fa[idx % 3][m], fb[idx % 3][n], acc[m][n]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---------------- epilogue: bias + ReLU (+ 2x2 max-pool) -> fp16 channels-last, 16-byte stores ----------------
  // units u = 2 m + g / 2 (one 16-byte store per pixel each); [u_lo, u_hi) selects which part of the tile this call writes
  auto epilogue = [&](int item, int u_lo, int u_hi) {
    int b, y0, x0;
    tile_coords(tile_of(item), b, y0, x0);
    const int yb = y0 + gw * 2, x = x0 + j;
    auto pack2 = [](float lo, float hi) -> unsigned {
      const h2_t v = {(_Float16)lo, (_Float16)hi};
      return *reinterpret_cast<const unsigned*>(&v);
    };
    auto store_pair = [&](_Float16* pix, int m, int g, const float (&q0)[4], const float (&q1)[4], bool ok) {
      const unsigned a0 = pack2(q0[0], q0[1]), a1 = pack2(q0[2], q0[3]);
      const unsigned b0 = pack2(q1[0], q1[1]), b1 = pack2(q1[2], q1[3]);
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if (ok) *reinterpret_cast<uint4*>(pix + cb * CT + m * 32 + (g + hh) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    };
    if constexpr (!POOL) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int y = yb + n;
        const bool ok = y < p.H && x < p.W;
        _Float16* pix = p.out + ((size_t)(b * p.H + y) * p.W + x) * p.cout;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            if (2 * m + g / 2 < u_lo || 2 * m + g / 2 >= u_hi) continue;
            float q0[4], q1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              q0[e] = fmaxf(acc[m][n][4 * g + e] + s_bias[m * 32 + hh * 4 + g * 8 + e], 0.f);
              q1[e] = fmaxf(acc[m][n][4 * (g + 1) + e] + s_bias[m * 32 + hh * 4 + (g + 1) * 8 + e], 0.f);
            }
            store_pair(pix, m, g, q0, q1, ok);
          }
      }
    } else {
      const int Ho = p.H >> 1, Wo = p.W >> 1;
      const int yo = yb >> 1, xo = x >> 1;
      const bool ok = !(x & 1) && yo < Ho && xo < Wo;
      _Float16* pix = p.out + ((size_t)(b * Ho + yo) * Wo + xo) * p.cout;
      auto pool4 = [&](int m, int g, float (&q)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float tt = fmaxf(acc[m][0][4 * g + e], acc[m][1][4 * g + e]);
          const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0xB1, 0xF, 0xF, false));
          q[e] = fmaxf(fmaxf(tt, nb) + s_bias[m * 32 + hh * 4 + g * 8 + e], 0.f);
        }
      };
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          if (2 * m + g / 2 < u_lo || 2 * m + g / 2 >= u_hi) continue;
          float q0[4], q1[4];
          pool4(m, g, q0);
          pool4(m, g + 1, q1);
          store_pair(pix, m, g, q0, q1, ok);
        }
    }
  };

  // item w of group g: staged in half-step 2w + g - 1, MFMA in 2w + g, epilogue (after its last chunk) in 2w + g + 1
  if (NW > 0) { if constexpr (FUSE1A) prefetch_u8(0); else prefetch_in(0); }
#pragma unroll 1
  for (int s = -1; s <= s_end; ++s) {
    const bool tr = SSHIP_PP_TRACE_BUILD && p.trace && (s == 8 || s == 9) && gw == 0 && lane == 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (tr) t0 = __builtin_readcyclecounter();
    tr_stage = tr;
    if (((s + 1) & 1) == grp) {
      // ---- data-movement role: epilogue of the item whose MFMA just finished, stage the next item, prefetch ----
      // Static priority for the role (SSHIP_PP_PRIO, default on): this wave shares its SIMD with a wave that has 72-144 MFMAs ready
      // back to back; at equal priority the older wave wins arbitration, and this role's few instructions (conv1a's six MFMAs, the
      // LDS writes, the prefetch loads) queue behind that stream - the role trace had `stage` at 4.8 k clocks for ~1 k of work.
      if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(SSHIP_PP_PRIO);
      const int w_done = (s - 1 - grp) >> 1;  // item whose MFMA ran in half-step s - 1
      if (s - 1 - grp >= 0 && w_done < NW && (w_done % NCHUNK) == NCHUNK - 1 && !(p.dbg & 2)) epilogue(w_done, EPI_MFMA, 2 * MT);
      if (tr) t1 = __builtin_readcyclecounter();
      const int w_next = (s + 1 - grp) >> 1;  // item whose MFMA runs in half-step s + 1
      if (w_next < NW) {
        if (!(p.dbg & 1)) { if constexpr (FUSE1A) stage_conv1a(w_next); else stage_in(w_next); }
        if (tr) t2 = __builtin_readcyclecounter();
        if (w_next + 1 < NW && !(p.dbg & 8)) { if constexpr (FUSE1A) prefetch_u8(w_next + 1); else prefetch_in(w_next + 1); }
      }
      if (tr) {
        t3 = __builtin_readcyclecounter();
        unsigned long long* o = p.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 10;
        o[0] = t1 - t0; o[1] = t2 > t1 ? t2 - t1 : 0; o[2] = t2 ? t3 - t2 : 0;
      }
    } else {
      if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(0);
      const int w = (s - grp) >> 1;
      if (s - grp >= 0 && w < NW && !(p.dbg & 4)) {
        mfma_item(w);
        // the first EPI_MFMA store units are written by the MFMA group itself right after its loop: the data-movement
        // half-step (epilogue + staging + prefetch) is the longer of the two roles, this group would only wait for it
        if (EPI_MFMA > 0 && (w % NCHUNK) == NCHUNK - 1 && !(p.dbg & 2)) epilogue(w, 0, EPI_MFMA);
      }
      if (tr) p.trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 10 + 3] = __builtin_readcyclecounter() - t0;
    }
    if (tr) t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (tr) p.trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 10 + 4 + (((s + 1) & 1) == grp ? 0 : 1)] = __builtin_readcyclecounter() - t1;
  }
}

template <int CIN, int CT, bool POOL, bool FUSE1A>
static hipError_t launch_pp(const PpArgs& a_in, hipStream_t s) {
  PpArgs a = a_in;
  static const int dbg = getenv("SSHIP_PP_DBG") ? atoi(getenv("SSHIP_PP_DBG")) : 0;
  a.dbg = dbg;
  constexpr size_t smem = (size_t)(2 * P_IN_HALFS + P_W_HALFS) * 2 + 128 * 4;
  static_assert(smem <= 163840, "LDS budget");
  auto kern = conv3x3_pp<CIN, CT, POOL, FUSE1A>;
  // thread-safe one-time opt-in to > 64 KiB of dynamic LDS (C++11 magic static; handles may be created on any thread)
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_rc != hipSuccess) return attr_rc;
  const int ncb = a.cout / CT;
  const int ntiles = a.B * ((a.W + P_TW - 1) / P_TW) * ((a.H + P_TH - 1) / P_TH);
  int gx = cu_count() / ncb;  // one persistent workgroup per CU
  if (gx < 1) gx = 1;
  if (gx * 2 > ntiles) gx = (ntiles + 1) / 2;  // every workgroup should feed both of its wave groups
  if (gx < 1) gx = 1;
  static const bool trace_on = SSHIP_PP_TRACE_BUILD && getenv("SSHIP_PP_TRACE") != nullptr;
  static unsigned long long* tbuf = nullptr;
  if (trace_on) {
    if (!tbuf) (void)hipMalloc(&tbuf, 4096 * 2 * 10 * 8);
    (void)hipMemsetAsync(tbuf, 0, 4096 * 2 * 10 * 8, s);
    a.trace = tbuf;
  }
  hipLaunchKernelGGL(kern, dim3(gx, ncb), dim3(512), smem, s, a);
  if (trace_on) {
    std::vector<unsigned long long> h(4096 * 2 * 10);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[10] = {0}; long cnt = 0;
    for (int i = 0; i < gx * ncb * 2; ++i) {
      if (!h[i * 10 + 3]) continue;
      for (int k = 0; k < 10; ++k) sum[k] += (double)h[i * 10 + k];
      ++cnt;
    }
    if (cnt) fprintf(stderr, "[pp trace cin=%d ct=%d pool=%d fuse=%d] epilogue=%.0f stage=%.0f prefetch=%.0f | mfma=%.0f | barrier wait after data=%.0f after mfma=%.0f (clk, %ld groups)\n",
                     CIN, CT, (int)POOL, (int)FUSE1A, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, cnt);
    if (cnt && sum[6] > 0) fprintf(stderr, "[pp trace   conv1a staging] convert=%.0f mfma issue=%.0f relu+lds writes=%.0f clk\n", sum[6] / cnt, sum[7] / cnt, sum[8] / cnt);
  }
  return hipGetLastError();
}

hipError_t sp_conv3x3_pp(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, bool pool, hipStream_t s) {
  PpArgs a{};
  a.in = in; a.wpack = w.w; a.bias = w.bias; a.out = out; a.B = B; a.H = H; a.W = W; a.cout = w.cout;
  if (w.cin == 64 && w.ct == 64) return pool ? launch_pp<64, 64, true, false>(a, s) : launch_pp<64, 64, false, false>(a, s);
  // 128 input channels: the 64-row-tile kernel over 32-channel chunks (conv_pp128.hip) when the layer carries that packing;
  // SUPERSLAM_HIP_CONV128=ct32 keeps the 32-row-tile kernel of this file (A/B runs)
  static const bool ct32 = [] { const char* e = getenv("SUPERSLAM_HIP_CONV128"); return e && std::string(e) == "ct32"; }();
  if (w.cin == 128 && w.w_q && !ct32 && sp_conv3x3_pp128_fits(B, H, W)) return sp_conv3x3_pp128(w, in, out, B, H, W, pool, s);
  if (w.cin == 128 && w.ct == 32) return pool ? launch_pp<128, 32, true, false>(a, s) : launch_pp<128, 32, false, false>(a, s);
  return hipErrorInvalidValue;
}

hipError_t sp_conv1ab_pp(const ConvW& w1b, const _Float16* w1a_frag, const float* b1a, const uint8_t* img, _Float16* out,
                         int B, int H, int W, hipStream_t s) {
  PpArgs a{};
  a.img = img; a.w1a = w1a_frag; a.b1a = b1a; a.wpack = w1b.w; a.bias = w1b.bias; a.out = out;
  a.B = B; a.H = H; a.W = W; a.cout = w1b.cout;
  if (w1b.cin != 64 || w1b.ct != 64) return hipErrorInvalidValue;
  return launch_pp<64, 64, true, true>(a, s);
}

}  // namespace sship
