// Ping-pong 3x3 convolution for SuperPoint (utils/convert_superpoint_to_onnx.py:38-49) - the production conv kernel.
//
// The strip kernel (conv_strip.hip) runs ONE workgroup per CU (72 KiB of weights + the input tile fill the LDS) and
// its 8 waves move through  stage -> MFMA -> epilogue  in lock-step, so the matrix pipe idles while they all do
// VALU / LDS / VMEM work (ablation in profiles/NOTES_r01_r04_design_history.md: epilogue 19-30 %, conv1a phase 12 %, MFMA loop itself ~85 % efficient).
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
#include <type_traits>
#include <vector>

#include "igemm.h"
#include "kernels.h"

// store units (of 2 MT) the MFMA group writes itself; -1 = measured per-kernel default: none for the 64-channel layers
// (incl. the fused conv1a+conv1b), everything for the 128-channel ones (build.py --variant -DSSHIP_PP_EPI=n to sweep)
#ifndef SSHIP_PP_EPI
#define SSHIP_PP_EPI -1
#endif
#ifndef SSHIP_PP_COLMAJOR
#define SSHIP_PP_COLMAJOR 1  // tile walk down the columns (0: raster order, rounds 1-5; A/B builds)
#endif
#ifndef SSHIP_PP_EPI_LAST
#define SSHIP_PP_EPI_LAST 0  // data half-step: 1 = epilogue stores AFTER the staging and the prefetch issue (A/B builds; measured slower, see data_role)
#endif
#ifndef SSHIP_PP_EPI_FUSED  // the same for the fused conv1a + conv1b kernel, whose data half-step also computes conv1a
#define SSHIP_PP_EPI_FUSED 0
#endif
// role tracing (SSHIP_PP_TRACE=1 at run time) is compiled in only on request: it costs registers in the fused kernel
#ifndef SSHIP_PP_TRACE_BUILD
#define SSHIP_PP_TRACE_BUILD 0
#endif
// s_setprio level of the data-movement role (0 = off; the MFMA role runs at priority 0)
#ifndef SSHIP_PP_PRIO
#define SSHIP_PP_PRIO 2
#endif
#ifndef SSHIP_PP_NBUF
#define SSHIP_PP_NBUF 3
#endif
// timing / energy ablation (results are wrong): 1 = the MFMA loop reads its first fragments only and reuses them, 2 = no MFMAs (fragments still read),
// 4 = the epilogue runs without its stores (round 6)
#ifndef SSHIP_PP_ABL
#define SSHIP_PP_ABL 0
#endif
// accumulator initialisation (bias reads): 1 = at the end of the preceding data half-step, 0 = at the start of the MFMA half-step
#ifndef SSHIP_PP_ACC_PRELOAD
#define SSHIP_PP_ACC_PRELOAD 1
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

struct PpWalk { int tx, ty, b; };  // wave-uniform tile coordinates of a group's tile stream (stride 2 tiles)

template <int CIN, int CT, bool POOL, bool FUSE1A>
__global__ __launch_bounds__(512, 2) void conv3x3_pp(PpArgs p) {
  constexpr int MT = CT / 32, NCHUNK = CIN / 64;
  constexpr int EPI_MFMA = FUSE1A ? SSHIP_PP_EPI_FUSED : SSHIP_PP_EPI < 0 ? (CIN == 128 ? 2 * MT : 0) : (SSHIP_PP_EPI > 2 * MT ? 2 * MT : SSHIP_PP_EPI);
  constexpr int PIX_B = CIN * 2;  // bytes per input pixel
  constexpr int NBUF = SSHIP_PP_NBUF;  // fragment ring depth of the MFMA loop: k-step i + NBUF - 1 is requested before the MFMAs of k-step i
  static_assert(NCHUNK * 9 * 4 * MT * 512 == P_W_HALFS, "weights must fill exactly 72 KiB");
  static_assert(!FUSE1A || CIN == 64, "conv1a fusion feeds a 64-channel layer");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* s_w = reinterpret_cast<_Float16*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);  // wave-uniform role selector
  const int gw = wave & 3, gt = tid & 255;
  _Float16* my_in = s_w + P_W_HALFS + grp * P_IN_HALFS;

  const int tiles_x = (p.W + P_TW - 1) / P_TW, tiles_y = (p.H + P_TH - 1) / P_TH;
  const int ntiles = p.B * tiles_x * tiles_y;
  const int cb = blockIdx.y;
  const int t_begin = (int)((long long)blockIdx.x * ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int n_wg = t_end - t_begin;
  if (n_wg <= 0) return;
  // group g takes tiles t_begin + g, t_begin + g + 2, ...
  const int T0 = (n_wg + 1) >> 1, T1 = n_wg >> 1, T_mine = grp ? T1 : T0;

  {  // weights: once per workgroup, all 512 threads
    const _Float16* wsrc = p.wpack + (size_t)cb * P_W_HALFS;
    for (int u = tid; u < P_W_HALFS / 8; u += 512)
      *reinterpret_cast<uint4*>(s_w + u * 8) = *reinterpret_cast<const uint4*>(wsrc + u * 8);
  }
  // bias of this workgroup's CT channels in LDS, one copy per N-tile: the accumulators start from it, each quad by its own
  // ds_read_b128 (two copies so the compiler cannot merge the reads of the two N-tiles and then copy registers)
  float* s_bias = reinterpret_cast<float*>(s_w + P_W_HALFS + 2 * P_IN_HALFS);
  if (tid < 2 * CT) s_bias[tid] = p.bias[cb * CT + (tid % CT)];
  // fragment-read offsets: B fragment of (row n + ky, col j + kx), k-step ks -> unit (2 ks + hh) ^ ((j + kx) >> 1 & 7)
  int boff[3][4];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) boff[kx][ks] = (j + kx) * 64 + (((2 * ks + hh) ^ (((j + kx) >> 1) & 7)) << 3);

  // Tile order inside a workgroup's contiguous range.  SSHIP_PP_COLMAJOR = 1 (round 6): down the tile COLUMNS (ty fastest).  The two groups of
  // a workgroup take alternate tiles, so with this order the tiles in flight are vertical neighbours and the two halo rows they share (2 of
  // 10 input rows) are L2 hits; the 2 of 34 halo columns shared with the next tile column come around tiles_y tiles later.  With the raster
  // order of rounds 1-5 (tx fastest) it was the other way round: the halo rows were re-fetched tiles_x tiles later, long after the XCD's
  // 4-MiB L2 had turned over - conv2a / conv2b read 1.2-1.25 x their input (profiles/r05_final_pmc_traffic.json).  Measured
  // (profiles/r06_g_conv_tile_walk.txt): conv2b's fetch 2 577 -> 2 197 MB per launch (1.04 x its input), conv2a + conv3a -14 %, joules -0.8 % / -1.1 %.
  // NOT for the fused conv1a + conv1b kernel: its input is the u8 image, 34 bytes of a 128-byte line per tile row - in column order the line's
  // other tiles come 47 tiles later and the line is fetched four times (102 -> 386 MB, +0.4 % joules); it keeps the raster order.
  auto walk_init = [&](int t) __attribute__((always_inline)) {
    PpWalk w;
    if constexpr (SSHIP_PP_COLMAJOR != 0 && !FUSE1A) {
      w.ty = t % tiles_y;
      const int r = t / tiles_y;
      w.tx = r % tiles_x; w.b = r / tiles_x;
    } else {
      w.tx = t % tiles_x;
      const int r = t / tiles_x;
      w.ty = r % tiles_y; w.b = r / tiles_y;
    }
    return w;
  };
  auto walk_next = [&](PpWalk& w) __attribute__((always_inline)) {  // two tiles further along the walk
    if constexpr (SSHIP_PP_COLMAJOR != 0 && !FUSE1A) {
      w.ty += 2;
      while (w.ty >= tiles_y) {
        w.ty -= tiles_y;
        if (++w.tx == tiles_x) { w.tx = 0; ++w.b; }
      }
    } else {
      w.tx += 2;
      while (w.tx >= tiles_x) {
        w.tx -= tiles_x;
        if (++w.ty == tiles_y) { w.ty = 0; ++w.b; }
      }
    }
  };
  PpWalk pw = walk_init(t_begin + grp), sw = pw, ew = pw;  // tile of the next prefetch / staging / epilogue

  // ---------------- staging (plain variant): global -> registers (prefetch) -> LDS ----------------
  // The data-movement role shares its SIMD with a wave that has 144 MFMAs queued and pays ~10 clocks per VALU instruction
  // (profiles/r02_pp_role_trace.txt), so the addressing is affine: thread t of the group owns 16-byte unit t of the 272
  // units of halo row i (i = 0..9) - one VGPR offset plus a scalar per-row offset for the ten buffer loads, one LDS address
  // plus an immediate for the ten ds_write_b128; the 16 left-over units per row are an eleventh load of threads 0..159.
  // Out-of-image halo pixels get their voffset pushed out of the buffer's range: the load returns zeros, nothing is masked
  // afterwards.  The buffer is rebuilt per tile on the image (base one row and one pixel before it), offsets stay 32-bit.
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  constexpr unsigned P_OOB = 0xfffffff0u;
  // epilogue: byte offset of this lane's output pixel column inside a tile row (pooled: even columns store, odd ones are dropped)
  const unsigned ep_voff = POOL ? ((j & 1) ? P_OOB : (unsigned)(((j >> 1) * p.cout + hh * 8) * 2)) : (unsigned)((j * p.cout + hh * 8) * 2);
  constexpr int P_ROW_UNITS = P_TWH * 8;  // 272
  u4_t rin[FUSE1A ? 1 : 11];
  const int m_px = gt >> 3, m_part = gt & 7;                                   // main part: unit gt of a halo row
  const int r_row = gt >> 4, r_px = 32 + ((gt & 15) >> 3), r_part = gt & 7;    // remainder: units 256..271 of row gt >> 4
  const bool r_on = gt < 160;
  const unsigned voff_main = (unsigned)(m_px * PIX_B + m_part * 16);
  const unsigned voff_rem = r_on ? (unsigned)((r_row * p.W + r_px) * PIX_B + r_part * 16) : P_OOB;
  _Float16* lds_main = my_in + pp_lds(0, m_px, m_part);  // + i rows for halo row i
  _Float16* lds_rem = my_in + pp_lds(r_on ? r_row : 0, r_px, r_part);
  const unsigned row_bytes = (unsigned)(p.W * PIX_B);
  auto prefetch_in = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (!FUSE1A) {
      const int y0 = pw.ty * P_TH, x0 = pw.tx * P_TW;
      const char* img = reinterpret_cast<const char*>(p.in) + (size_t)pw.b * p.H * p.W * PIX_B - (size_t)(p.W + 1) * PIX_B;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, (int)0x7ffffff0, 0x00020000);
      const unsigned soff = (unsigned)(y0 * p.W + x0) * (unsigned)PIX_B + (unsigned)chunk * 128u;
      const bool interior = y0 >= 1 && y0 + P_TH + 1 <= p.H && x0 >= 1 && x0 + P_TW + 1 <= p.W;
      if (interior) {
#pragma unroll
        for (int i = 0; i < 10; ++i) rin[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_main, soff + i * row_bytes, 0);
        rin[10] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_rem, soff, 0);
      } else {
        // halo row r is inside the image for rlo <= r <= rhi, halo column c for clo <= c <= chi
        const int rlo = y0 == 0 ? 1 : 0, rhi = min(P_THH - 1, p.H - y0), clo = x0 == 0 ? 1 : 0, chi = min(P_TWH - 1, p.W - x0);
        const unsigned vm = (m_px >= clo && m_px <= chi) ? voff_main : P_OOB;
#pragma unroll
        for (int i = 0; i < 10; ++i)
          rin[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i >= rlo && i <= rhi) ? vm : P_OOB, soff + i * row_bytes, 0);
        const bool r_ok = r_px >= clo && r_px <= chi && r_row >= rlo && r_row <= rhi;
        rin[10] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, r_ok ? voff_rem : P_OOB, soff, 0);
      }
      if (chunk == NCHUNK - 1) walk_next(pw);
    }
  };
  auto stage_in = [&]() __attribute__((always_inline)) {
    if constexpr (!FUSE1A) {
#pragma unroll
      for (int i = 0; i < 10; ++i) *reinterpret_cast<u4_t*>(lds_main + i * (P_TWH * 64)) = rin[i];
      if (r_on) *reinterpret_cast<u4_t*>(lds_rem) = rin[10];
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
  // byte selectors of the conv1a B fragment (stage_conv1a): hh = 0: [r0.b0 r0.b1 r0.b2 r1.b0] and [r1.b1 0 0 0] from (X = r1, Y = r0);
  // hh = 1: [r1.b2 r2.b0 r2.b1 r2.b2] and [0xFF 0xFF 0 0] from (X = r2, Y = r1).  v_perm_b32: bytes 0..3 = second source,
  // 4..7 = first source, 0x0c = 0x00, 0x0d = 0xFF.
  const unsigned c1_selA = hh ? 0x06050402u : 0x04020100u, c1_selB = hh ? 0x0c0c0d0du : 0x0c0c0c05u;
  if constexpr (FUSE1A) {
#pragma unroll
    for (int k = 0; k < NT_W; ++k) {
      const int q = min((gw + 4 * k) * 32 + j, P_THH * P_TWH - 1);
      const int py = q / P_TWH, px = q - py * P_TWH;
      c1_pypx[k] = (py << 16) | px;
      c1_lds[k] = ((py * P_TWH + px) * 64 + hh * 4) | (((px >> 1) & 7) << 24);  // half offset | swizzle term
    }
  }
  auto prefetch_u8 = [&]() __attribute__((always_inline)) {
    if constexpr (FUSE1A) {
      const int y0 = pw.ty * P_TH, x0 = pw.tx * P_TW;
      const uint8_t* im = p.img + (size_t)pw.b * p.H * p.W;
      walk_next(pw);
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
  unsigned long long* trow = p.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 10;
  auto stage_conv1a = [&](bool tr_stage) __attribute__((always_inline)) {
    if constexpr (FUSE1A) {
      unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
      if (SSHIP_PP_TRACE_BUILD && tr_stage) ts0 = __builtin_readcyclecounter();
      const int y0 = sw.ty * P_TH, x0 = sw.tx * P_TW;
      walk_next(sw);
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
        // taps (r, c) = byte c of dword r.  K layout (matches w1a_fragb): lanes hh = 0 hold taps 0..4 in slots 0..4, lanes hh = 1
        // taps 5..8 in slots 0..3 and the two bias slots (B = 1.0) in 4, 5 - five or six conversions per lane instead of
        // nine.  The bytes are gathered first (two v_perm_b32 with per-lane selectors); a 0xFF byte converts to exactly
        // 1.0 (255 * fl(1/255) rounds to 1.0f), which is how the bias slots get their ones.
        const unsigned X = hh ? rp[k][2] : rp[k][1], Y = hh ? rp[k][1] : rp[k][0];
        const unsigned q4 = __builtin_amdgcn_perm(X, Y, c1_selA), q2 = __builtin_amdgcn_perm(X, X, c1_selB);
        // cv::Mat::convertTo: float(u8) * (1/255), then the engine's fp16 input
        // (v_mul_f32 by hand: the compiler pairs the products into v_pk_mul_f32, and a packed-fp32 instruction of this wave
        // waited for a gap in the other wave's MFMA stream - the six conv1a MFMAs then issued ~2 k clocks later)
        auto cv8 = [](unsigned w, int byte) __attribute__((always_inline)) {
          float r;
          asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"((float)((w >> (8 * byte)) & 0xffu)), "v"(1.0f / 255.0f));
          return (_Float16)r;
        };
        bf[k][0] = cv8(q4, 0); bf[k][1] = cv8(q4, 1); bf[k][2] = cv8(q4, 2); bf[k][3] = cv8(q4, 3);
        bf[k][4] = cv8(q2, 0); bf[k][5] = cv8(q2, 1); bf[k][6] = (_Float16)0.f; bf[k][7] = (_Float16)0.f;
      }
      if (SSHIP_PP_TRACE_BUILD && tr_stage) ts1 = __builtin_readcyclecounter();
      const f16x_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f16x_t d0[NT_W], d1[NT_W];
#pragma unroll
      for (int k = 0; k < NT_W; ++k) {
        if (gw + 4 * k >= P_NT1A) continue;  // wave-uniform
        d0[k] = mfma32(a1a0, bf[k], zero16);  // C = inline constant 0: no accumulator initialisation
        d1[k] = mfma32(a1a1, bf[k], zero16);
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
          const int base = c1_lds[k] & 0xffffff, swz = c1_lds[k] >> 24;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // ReLU after the fp16 rounding, two values per instruction (v_pk_max_f16): rounding is monotone and keeps the
            // sign, so relu(fp16(x)) == fp16(relu(x))
            const h4_t z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            h4_t o0 = __builtin_elementwise_max(to_h4(d0[k][4 * g], d0[k][4 * g + 1], d0[k][4 * g + 2], d0[k][4 * g + 3]), z4);
            h4_t o1 = __builtin_elementwise_max(to_h4(d1[k][4 * g], d1[k][4 * g + 1], d1[k][4 * g + 2], d1[k][4 * g + 3]), z4);
            if (!interior && !inside) { o0 = z4; o1 = z4; }  // `interior` is uniform: inner tiles skip the selects
            const int u0 = (g ^ swz) << 3;  // channels 4 hh + 8 g .. (+3): unit g; M-tile 1: unit 4 + g
            *reinterpret_cast<h4_t*>(my_in + base + u0) = o0;
            *reinterpret_cast<h4_t*>(my_in + base + (u0 ^ 32)) = o1;
          }
        }
      }
      if (SSHIP_PP_TRACE_BUILD && tr_stage) { trow[6] = ts1 - ts0; trow[7] = ts2 - ts1; trow[8] = __builtin_readcyclecounter() - ts2; }
    }
  };

  f16x_t acc[MT][2];
  // The accumulators start from the bias (row 8 g + 4 hh + e of M-tile m is register 4 g + e: one ds_read_b128 per quad,
  // straight into the accumulator registers).  The reads are issued at the end of the preceding data half-step, so the MFMA
  // half-step opens with its fragment reads only.
  auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(s_bias + n * CT + m * 32 + hh * 4 + g * 8);
          acc[m][n][4 * g + 0] = b4.x; acc[m][n][4 * g + 1] = b4.y; acc[m][n][4 * g + 2] = b4.z; acc[m][n][4 * g + 3] = b4.w;
        }
  };
  // ---------------- MFMA half-step: 36 k-steps of one 64-channel chunk, fragments triple-buffered ----------------
  auto mfma_item = [&](auto chunk_c) __attribute__((always_inline)) {
    constexpr int chunk = decltype(chunk_c)::value;
    const _Float16* wc = s_w + chunk * (9 * 4 * MT * 512) + lane * 8;
    const _Float16* ib = my_in + (gw * 2) * P_TWH * 64;
    // one wave per SIMD feeds the matrix pipe here, so LDS latency must be covered by this wave alone: fragments are
    // triple-buffered, the ds_reads of k-step i+2 are issued (and pinned) before the MFMAs of k-step i.
    h8_t fa[NBUF][MT], fb[NBUF][2];
    auto load_frags = [&](int idx, int buf) __attribute__((always_inline)) {
      // k order of one accumulator.  CIN = 64: tap-major, four k-steps per tap.  CIN = 128: the order of conv3x3_pp128w (32-channel
      // half, kx, k-step, ky) - this kernel is the latency-mode stand-in for that one (sp_conv3x3_pp), and with the same fp16 operands
      // entering the same fp32 accumulator in the same order the two are bit-identical: a frame extracted alone equals the same
      // frame inside a 128-image batch (tests/test_gpu_bench_batch_parity.py, bench.py self_check).
      const int h32 = idx / 18, r18 = idx - 18 * h32, t6 = r18 / 3;
      const int ky = CIN == 128 ? r18 - 3 * t6 : (idx >> 2) / 3, kx = CIN == 128 ? t6 >> 1 : (idx >> 2) - 3 * ky;
      const int tap = ky * 3 + kx, ks = CIN == 128 ? 2 * h32 + (t6 & 1) : idx & 3;
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[buf][m] = *reinterpret_cast<const h8_t*>(wc + ((tap * 4 + ks) * MT + m) * 512);
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[buf][n] = *reinterpret_cast<const h8_t*>(ib + (n + ky) * P_TWH * 64 + boff[kx][ks]);
    };
    if constexpr (CIN == 64 && SSHIP_K_ROWSHARE != 0) {
      // (kx, k-step, ky) order: sub-step t = 3 g + ky of group g = 4 kx + ks multiplies tap row ky; its two N-tiles read input rows ky and ky + 1 of
      // the wave's four, so a group needs rows 0..3 ONCE (row fragment r = 4 g + d, ring of 6) where the tap-major order read 2 per tap row = 6:
      // 2 A + 4/3 B fragment reads per 4 MFMAs instead of 2 + 2.  Fragments of sub-step t + 2 are requested before the MFMAs of sub-step t.
      h8_t ra[3][MT], rbq[6];
      auto loads_for = [&](int t) __attribute__((always_inline)) {
        const int g = t / 3, ky = t - 3 * g, kx = g >> 2, ks = g & 3;
#pragma unroll
        for (int m = 0; m < MT; ++m) ra[t % 3][m] = *reinterpret_cast<const h8_t*>(wc + (((ky * 3 + kx) * 4 + ks) * MT + m) * 512);
        if (ky == 0) {
          rbq[(4 * g) % 6] = *reinterpret_cast<const h8_t*>(ib + boff[kx][ks]);
          rbq[(4 * g + 1) % 6] = *reinterpret_cast<const h8_t*>(ib + P_TWH * 64 + boff[kx][ks]);
        } else {
          rbq[(4 * g + ky + 1) % 6] = *reinterpret_cast<const h8_t*>(ib + (ky + 1) * P_TWH * 64 + boff[kx][ks]);
        }
      };
      loads_for(0); loads_for(1);
      if constexpr (chunk == 0 && !SSHIP_PP_ACC_PRELOAD) acc_init();
#pragma unroll
      for (int t = 0; t < 36; ++t) {
        if (t + 2 < 36 && !(SSHIP_PP_ABL & 1)) loads_for(t + 2);
        __builtin_amdgcn_sched_barrier(0);
        const int g = t / 3, ky = t - 3 * g;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            if constexpr ((SSHIP_PP_ABL & 2) != 0) asm volatile("" :: "v"(ra[t % 3][m]), "v"(rbq[(4 * g + ky + n) % 6]));
            else acc[m][n] = mfma32(ra[t % 3][m], rbq[(4 * g + ky + n) % 6], acc[m][n]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load_frags(i, i);
    if constexpr (chunk == 0 && !SSHIP_PP_ACC_PRELOAD) acc_init();
#pragma unroll
    for (int idx = 0; idx < 36; ++idx) {
      if (idx + NBUF - 1 < 36 && !(SSHIP_PP_ABL & 1)) load_frags(idx + NBUF - 1, (idx + NBUF - 1) % NBUF);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr ((SSHIP_PP_ABL & 2) != 0) asm volatile("" :: "v"(fa[idx % NBUF][m]), "v"(fb[idx % NBUF][n]));
          else acc[m][n] = mfma32(fa[idx % NBUF][m], fb[idx % NBUF][n], acc[m][n]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---------------- epilogue: ReLU (+ 2x2 max-pool) -> fp16 channels-last, 16-byte stores; the bias is already in ----------------
  // units u = 2 m + g / 2 (one 16-byte store per pixel each); [u_lo, u_hi) selects which part of the tile this call writes;
  // the call with u_hi == 2 MT advances the epilogue walker
  auto epilogue = [&](int u_lo, int u_hi) __attribute__((always_inline)) {
    const int y0 = ew.ty * P_TH, x0 = ew.tx * P_TW, b = ew.b;
    if (u_hi == 2 * MT) walk_next(ew);
    const int yb = y0 + __builtin_amdgcn_readfirstlane(gw) * 2;  // wave-uniform, and the compiler must know it (scalar store offsets)
    const h2_t z2 = {(_Float16)0.f, (_Float16)0.f};
    auto relu2 = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {  // two values -> packed fp16, ReLU on the pair
      h2_t v = {(_Float16)lo, (_Float16)hi};
      v = __builtin_elementwise_max(v, z2);
      return *reinterpret_cast<const unsigned*>(&v);
    };
    auto pack2 = [](float lo, float hi) __attribute__((always_inline)) -> unsigned {
      const h2_t v = {(_Float16)lo, (_Float16)hi};
      return *reinterpret_cast<const unsigned*>(&v);
    };
    // lanes hh = 0 / 1 hold channels 4 hh .. + 3 of an 8-channel unit: permlane32_swap pairs them into one 16-byte store each.
    // Buffer stores on the output image of this tile: the lane's pixel column is a tile-invariant VGPR offset (ep_voff; lanes
    // that do not store - odd columns of the pooled variant, columns past the right edge - get an out-of-range offset and the
    // store is dropped), everything else is a scalar offset.  No 64-bit address arithmetic, no exec masking in this role.
    // (scalar offset through an empty asm: see wstore in lg_kernels.hip)
    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.out) + (size_t)b * Ho * Wo * p.cout * 2, 0, (int)0x7ffffff0, 0x00020000);
    const int xo0 = POOL ? x0 >> 1 : x0;                       // first output column of the tile
    unsigned voff = ep_voff;
    if (xo0 + (POOL ? P_TW / 2 : P_TW) > Wo)                    // right-edge tile (wave-uniform): mask the columns past the image
      voff = (xo0 + (POOL ? j >> 1 : j)) < Wo ? ep_voff : P_OOB;
    auto store_pair = [&](int row_off, int m, int g, unsigned a0, unsigned a1, unsigned b0, unsigned b1) __attribute__((always_inline)) {
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      int so = (row_off + cb * CT + m * 32 + g * 8) * 2;
      asm volatile("" : "+s"(so));
      typedef unsigned st4_t __attribute__((ext_vector_type(4)));
      if constexpr ((SSHIP_PP_ABL & 4) != 0) { asm volatile("" :: "v"(r0[0]), "v"(r1[0]), "v"(r0[1]), "v"(r1[1])); return; }  // ablation: no stores
      __builtin_amdgcn_raw_buffer_store_b128(st4_t{r0[0], r1[0], r0[1], r1[1]}, ro, voff, so, 0);
    };
    if constexpr (!POOL) {
      // (flat stores here: measured on one box, the buffer-store form gains 2 % on the pooled layers and loses 1-5 % on these)
      const int x = x0 + j;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int y = yb + n;
        const bool ok = y < p.H && x < p.W;
        _Float16* pix = p.out + ((size_t)(b * p.H + y) * p.W + x) * p.cout + cb * CT;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            if (2 * m + g / 2 < u_lo || 2 * m + g / 2 >= u_hi) continue;
            const f16x_t& a = acc[m][n];
            const auto r0 = __builtin_amdgcn_permlane32_swap(relu2(a[4 * g + 0], a[4 * g + 1]), relu2(a[4 * g + 4], a[4 * g + 5]), false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(relu2(a[4 * g + 2], a[4 * g + 3]), relu2(a[4 * g + 6], a[4 * g + 7]), false, false);
            if constexpr ((SSHIP_PP_ABL & 4) != 0) { asm volatile("" :: "v"(r0[0]), "v"(r1[0]), "v"(r0[1]), "v"(r1[1])); continue; }  // ablation: no stores
            if (ok) *reinterpret_cast<uint4*>(pix + m * 32 + (g + hh) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
      }
    } else {
      const int yo = yb >> 1;
      if (yo >= Ho) voff = P_OOB;  // wave-uniform (the pooling arithmetic below still runs: DPP needs every lane)
      const int pix = (yo * Wo + xo0) * p.cout;
      auto pool1 = [&](int m, int r) __attribute__((always_inline)) -> float {  // max over the wave's two rows and 0 (v_max3), then the column pair (dpp)
        const float tt = fmaxf(fmaxf(acc[m][0][r], acc[m][1][r]), 0.f);
        const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0xB1, 0xF, 0xF, false));
        return fmaxf(tt, nb);
      };
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          if (2 * m + g / 2 < u_lo || 2 * m + g / 2 >= u_hi) continue;
          store_pair(pix, m, g, pack2(pool1(m, 4 * g + 0), pool1(m, 4 * g + 1)), pack2(pool1(m, 4 * g + 2), pool1(m, 4 * g + 3)),
                     pack2(pool1(m, 4 * g + 4), pool1(m, 4 * g + 5)), pack2(pool1(m, 4 * g + 6), pool1(m, 4 * g + 7)));
        }
    }
  };

  // ---------------- schedule ----------------
  // Half-step s: group (s & 1) runs the MFMAs of its item (tile, chunk) s >> 1; the other group - on the same SIMDs - writes
  // the epilogue of the tile it just finished, stages its next item into LDS and issues the prefetch for the one after.
  // Both groups run the SAME straight-line loop (MFMA half-step, barrier, data half-step, barrier), group 1 one barrier
  // behind group 0.  There is no role branch inside the loop: with one, the accumulators were loop-carried through a phi
  // and the compiler moved all 64 of them to other registers and back around every MFMA half-step.
  const bool tr_lane = SSHIP_PP_TRACE_BUILD && p.trace && gw == 0 && lane == 0;
  unsigned long long t0 = 0, t1 = 0;
  // data half-step after MFMA item (it, chunk): epilogue if that was the tile's last chunk; stage item + 1; prefetch item + 2
  auto data_role = [&](bool do_epi, bool do_stage, bool do_prefetch, int pf_chunk, bool init_acc, bool tr) __attribute__((always_inline)) {
    if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(SSHIP_PP_PRIO);
    if (tr) t0 = __builtin_readcyclecounter();
    // Order of the three jobs.  Round 6 tested the hypothesis that the staging and the next prefetch stall behind the epilogue's 16 stores
    // (one in-order vmcnt for loads and stores; removing the stores alone makes conv2a 33 % faster): with the epilogue LAST
    // (-DSSHIP_PP_EPI_LAST=1) conv2a went 1 163 -> 1 183 us and conv3a 548 -> 563 us, the pooled layers did not move - refuted.  The 33 % is the
    // power the written bytes draw (cycles per launch are within 3 % with and without stores; the clock is 45 % higher without them), not a
    // queueing stall (profiles/r06_d_conv_store_energy.txt).  Epilogue first stays.
    if (!SSHIP_PP_EPI_LAST && do_epi && EPI_MFMA < 2 * MT) epilogue(EPI_MFMA, 2 * MT);
    if (tr) { t1 = __builtin_readcyclecounter(); trow[0] = t1 - t0; t0 = t1; }
    if (do_stage) { if constexpr (FUSE1A) stage_conv1a(tr); else stage_in(); }
    if (tr) { t1 = __builtin_readcyclecounter(); trow[1] = t1 - t0; t0 = t1; }
    if (do_prefetch) { if constexpr (FUSE1A) prefetch_u8(); else prefetch_in(pf_chunk); }
    if (SSHIP_PP_EPI_LAST && do_epi && EPI_MFMA < 2 * MT) epilogue(EPI_MFMA, 2 * MT);
    if (SSHIP_PP_ACC_PRELOAD && init_acc) acc_init();  // for the tile whose first MFMA half-step comes next
    if (tr) { t1 = __builtin_readcyclecounter(); trow[2] = t1 - t0; t0 = t1; }
    __syncthreads();
    if (tr) trow[4] = __builtin_readcyclecounter() - t0;
    if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto mfma_role = [&](auto chunk_c, bool tr) __attribute__((always_inline)) {
    constexpr int chunk = decltype(chunk_c)::value;
    if (tr) t0 = __builtin_readcyclecounter();
    mfma_item(chunk_c);
    // the first EPI_MFMA store units are written by the MFMA group itself right after its loop
    if constexpr (chunk == NCHUNK - 1 && EPI_MFMA > 0) epilogue(0, EPI_MFMA);
    if (tr) { t1 = __builtin_readcyclecounter(); trow[3] = t1 - t0; t0 = t1; }
    __syncthreads();
    if (tr) trow[5] = __builtin_readcyclecounter() - t0;
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, NCHUNK - 1> c_last{};
  // barriers per group: (grp) + 1 + 2 NCHUNK T_mine; both groups must reach the larger count
  const int nbar_mine = grp + 1 + 2 * NCHUNK * T_mine, nbar_all = max(1 + 2 * NCHUNK * T0, 2 + 2 * NCHUNK * T1);
  if (T_mine > 0) { if constexpr (FUSE1A) prefetch_u8(); else prefetch_in(0); }
  __syncthreads();  // weights and bias are in LDS
  if (grp == 1) __syncthreads();  // group 1 runs one half-step behind group 0
  // first data half-step: item 0 into LDS, prefetch item 1
  data_role(false, T_mine > 0, NCHUNK > 1 ? T_mine > 0 : T_mine > 1, NCHUNK > 1 ? 1 : 0, T_mine > 0, false);
#pragma unroll 1
  for (int it = 0; it < T_mine; ++it) {
    const bool more = it + 1 < T_mine, tr = tr_lane && it == 2;
    if constexpr (NCHUNK == 1) {
      mfma_role(c0, tr);
      data_role(true, more, it + 2 < T_mine, 0, more, tr);
    } else {
      mfma_role(c0, tr);
      data_role(false, true, more, 0, false, tr);   // stage chunk 1 of this tile, prefetch chunk 0 of the next
      mfma_role(c_last, false);
      data_role(true, more, more, 1, more, false);  // epilogue; stage chunk 0 of the next tile, prefetch its chunk 1
    }
  }
#pragma unroll 1
  for (int k = nbar_mine; k < nbar_all; ++k) __syncthreads();
}

template <int CIN, int CT, bool POOL, bool FUSE1A>
static hipError_t launch_pp(const PpArgs& a_in, hipStream_t s) {
  PpArgs a = a_in;
  constexpr size_t smem = (size_t)(2 * P_IN_HALFS + P_W_HALFS) * 2 + 128 * 4;  // + bias [2][CT]
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
  static const bool trace_on = SSHIP_PP_TRACE_BUILD && dev_env("SSHIP_PP_TRACE") != nullptr;
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
  if ((size_t)H * W * w.cin * 2 >= 0x7f000000ull) return hipErrorInvalidValue;  // staging offsets inside one image are 32-bit
  PpArgs a{};
  a.in = in; a.wpack = w.w; a.bias = w.bias; a.out = out; a.B = B; a.H = H; a.W = W; a.cout = w.cout;
  // SUPERSLAM_HIP_CONV64=dma (A/B runs): the 16-row-tile LDS-DMA kernel of conv_pp128.hip with both 32-channel chunks' weights
  // resident.  Measured at 128 images: conv2a 1 115 -> 1 170 us, conv2b 965 -> 954, conv3a 531 -> 565
  // (profiles/r03_n_conv128_16row_tiles_lds_dma.txt) - with the whole weight set in LDS this file's kernel has no weight DMA to
  // save and its two-stage register staging hides the HBM latency better than a DMA that must land within one half-step.
  static const bool dma64 = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV64"); return e && std::string(e) == "dma"; }();
  if (w.cin == 64 && w.w_q && dma64 && sp_conv3x3_pp128_fits(B, H, W, 64)) return sp_conv3x3_pp128(w, in, out, B, H, W, pool, s);
  if (w.cin == 64 && w.ct == 64) return pool ? launch_pp<64, 64, true, false>(a, s) : launch_pp<64, 64, false, false>(a, s);
  // 128 input channels: the 64-row-tile kernel over 32-channel chunks (conv_pp128.hip) when the layer carries that packing;
  // SUPERSLAM_HIP_CONV128=ct32 keeps the 32-row-tile kernel of this file (A/B runs)
  // (any other value, e.g. th16 / th8, switches the latency-mode choice below off)
  static const std::string c128 = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV128"); return std::string(e ? e : ""); }();
  static const bool ct32 = c128 == "ct32";
  // Latency mode (a frame or two per call): the 16 x 32-pixel tiles of conv_pp128.hip give conv4a at 2 x 47 x 172 cells 36 tiles = 18 x 2
  // workgroups on 256 CUs, each running its two tiles' 2 x 4 x 144 MFMAs per wave back to back.  This file's kernel has 8-row tiles and
  // 32-row cout tiles: 4x the workgroups, a quarter of the serial MFMA chain each.  One pair, us per launch (profiles/r04_q_*):
  // conv3b 27.1 -> 22.6, conv4a / 4b 26.4 -> 12.6, convPa 27.2 -> 16.8.  Taken whenever the 16-row kernel would leave CUs without a workgroup.
  const int tiles16 = B * ((W + 31) / 32) * ((H + 15) / 16);
  const bool few_tiles = c128.empty() && w.ct == 32 && w.w && (tiles16 + 1) / 2 * (w.cout / 64) < cu_count();
  if (w.cin == 128 && w.w_q && !ct32 && !few_tiles && sp_conv3x3_pp128_fits(B, H, W, 128)) return sp_conv3x3_pp128(w, in, out, B, H, W, pool, s);
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
