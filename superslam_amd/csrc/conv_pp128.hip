// Ping-pong 3x3 convolution for the 128-input-channel layers of SuperPoint (conv3b, conv4a, conv4b, convPa;
// utils/convert_superpoint_to_onnx.py:41-46) with a 64-row cout tile.
//
// conv_pp.hip keeps all weights of a workgroup's cout tile in LDS (72 KiB); with 128 input channels that allowed only a
// 32-row tile: one M-tile per wave (1.5 LDS fragment reads per MFMA), every input tile staged by four workgroups.
// Here the K dimension runs in 32-channel chunks:
//   * weights of ONE chunk for 64 output rows are 36 KiB ([tap 9][k-step 2][m-tile 2][lane][8]); two such slots form a ring
//     that is refilled by LDS-DMA (global_load_lds_dwordx4) two half-steps ahead: both wave groups use chunk c in consecutive
//     half-steps, then the slot is free for chunk c + 2;
//   * an input tile chunk is 10 x 34 px x 32 ch = 21.8 KB, staged by cout / 64 workgroups instead of cout / 32;
//   * a work item (tile, chunk) is 72 MFMAs per wave (9 taps x 2 k-steps x 2 M-tiles x 2 N-tiles), with two A and two
//     B fragments per k-step (1.0 LDS fragment reads per MFMA).
// Pixel rows are 64 bytes in LDS (4 sixteen-byte units); unit u of column c sits in slot u ^ ((c >> 2) & 3): the 16 lanes of
// a ds_read_b128 group cover 16 consecutive columns mod 16, i.e. every (c & 3, slot) pair once = 64 distinct banks.
//
// The data-movement role of the ping-pong scheme shares every SIMD with a wave that has 72 MFMAs queued; the role trace
// (profiles/r02_pp_role_trace.txt) shows it pays ~10 clocks per VALU instruction there, so its cost is its instruction count,
// not its bytes.  This kernel keeps that count small:
//   * staging addresses are affine: thread t of a 256-thread group owns unit t of the 272 units of halo-row pair i (i = 0..4),
//     so one VGPR offset + a scalar per-pair offset addresses all five loads (buffer_load ... soffset) and one LDS address +
//     an immediate all five ds_write_b128; the 16 left-over units per pair are a sixth load of threads 0..79;
//   * out-of-image halo pixels are not masked after the fact: their voffset is pushed out of the buffer's range, the load
//     returns zeros (edge tiles only: two compares per load);
//   * tile coordinates are walked incrementally in scalar registers (no division per item);
//   * the bias initialises the accumulators, ReLU runs on packed fp16 after the conversion (rounding is monotone and keeps
//     the sign: relu(fp16(x)) == fp16(relu(x))), the pooled variant uses max3.
// Roles and the half-step schedule are those of conv_pp.hip; the MFMA group writes its own epilogue after a tile's last chunk.
#include "igemm.h"
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#ifndef SSHIP_PP_PRIO
#define SSHIP_PP_PRIO 2
#endif
#ifndef SSHIP_PP_TRACE_BUILD
#define SSHIP_PP_TRACE_BUILD 0  // role tracing (SSHIP_PP_TRACE=1 at run time), as in conv_pp.hip
#endif

// Energy ablations of conv3x3_pp128w (conv3b / conv4a / conv4b / convPa: 3.5 J = 21 % of a 64-pair call, never ablated before round 6;
// build.py --variant pp128abl<n> -DSSHIP_PP128_ABL=<n>, scripts/dev/stage_energy.py under the power poller): 1 no MFMAs (operands still read
// from LDS), 2 no DMA after the first item (neither the input tile chunks nor the weight ring move: no HBM / L2 -> LDS traffic),
// 4 no epilogue (no conversion, no stores), 8 no LDS fragment reads inside the item (one set of fragments read once per item),
// 16 the epilogue's arithmetic without its stores.  Results are wrong by design.  Measured (profiles/r06_c_conv_pp128_energy_ablation.txt):
// MFMAs 53-60 % of a launch's joules, DMA 0-5 %, fragment reads 3-6 %, the output stores 5 % (pooled conv3b) / 33 % (conv4a, convPa).
// The 33 % is not a stall: moving the stores behind the next half-step's DMA wait (so that nothing queues behind their acknowledgements)
// changed nothing (254 -> 256 us); cycles per launch are within 3 % with and without stores, the CLOCK is 22 % higher without them - the
// bytes written cost power (120-270 pJ per byte at the cap), and the layer is priced in joules.
#ifndef SSHIP_PP128_ABL
#define SSHIP_PP128_ABL 0
#endif
#ifndef SSHIP_PP128_COLMAJOR
#define SSHIP_PP128_COLMAJOR 0  // conv3x3_pp128w: 1 = tile walk down the columns (A/B builds; measured: fetch -1 %, joules +-0 - the 16-row tiles share 2 of 18 rows)
#endif

namespace sship {

struct Pp128Args {
  const _Float16* in;     // channels-last fp16 [B,H,W,128]
  const _Float16* wpack;  // packed [cout / 64][chunk 4][tap 9][k-step 2][m-tile 2][lane 64][8]
  const float* bias;
  _Float16* out;
  int B, H, W, cout;
  unsigned long long* trace;  // [workgroup][group][6] clocks of half-steps 8..9: weight DMA issue, stage, prefetch, mfma, barrier waits
  int pairs_on;               // conv3x3_pp128w: shared edge tiles allowed (launch_pp128w)
};

constexpr int Q_TH = 8, Q_TW = 32, Q_THH = 10, Q_TWH = 34;
constexpr int Q_IN_HALFS = Q_THH * Q_TWH * 32;  // 10,880 halfs = 21,760 B per wave group
constexpr int Q_W_SLOT = 9 * 2 * 2 * 512;       // 18,432 halfs = 36,864 B: one (cout tile, chunk)
constexpr int Q_ROW_UNITS = Q_TWH * 4;          // 136 sixteen-byte units per halo row
constexpr int Q_NCHUNK = 4;
constexpr unsigned Q_OOB = 0xfffffff0u;         // voffset beyond num_records: the buffer load returns zeros

__device__ __forceinline__ int pp128_lds(int row, int col, int unit) {
  return (row * Q_TWH + col) * 32 + ((unit ^ ((col >> 2) & 3)) << 3);
}

struct TileWalk { int tx, ty, b; };  // wave-uniform tile coordinates of a group's tile stream (stride 2 tiles)

template <bool POOL>
__global__ __launch_bounds__(512, 2) void conv3x3_pp128(Pp128Args p) {
  constexpr int MT = 2;
  extern __shared__ __attribute__((aligned(16))) char smem128[];
  _Float16* s_w = reinterpret_cast<_Float16*>(smem128);  // [2 slots][Q_W_SLOT]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  const int gw = wave & 3, gt = tid & 255;
  const int gw_u = __builtin_amdgcn_readfirstlane(gw);
  _Float16* my_in = s_w + 2 * Q_W_SLOT + grp * Q_IN_HALFS;
  float* s_bias = reinterpret_cast<float*>(s_w + 2 * Q_W_SLOT + 2 * Q_IN_HALFS);

  const int tiles_x = (p.W + Q_TW - 1) / Q_TW, tiles_y = (p.H + Q_TH - 1) / Q_TH;
  const int ntiles = p.B * tiles_x * tiles_y;
  const int cb = blockIdx.y;
  const int t_begin = (int)((long long)blockIdx.x * ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int n_wg = t_end - t_begin;
  if (n_wg <= 0) return;
  const _Float16* wsrc = p.wpack + (size_t)cb * (Q_NCHUNK * Q_W_SLOT);

  // chunk 0 -> slot 0 before the first half-step (plain copy by all 512 threads)
  for (int u = tid; u < Q_W_SLOT / 8; u += 512) *reinterpret_cast<uint4*>(s_w + u * 8) = *reinterpret_cast<const uint4*>(wsrc + u * 8);
  if (tid < 128) s_bias[tid] = p.bias[cb * 64 + (tid & 63)];  // two copies, one per N-tile (see mfma_item)
  // half `half` (18 fragments of 1 KiB) of weight chunk `cn` -> ring slot cn & 1, by the four waves of a group: wave g moves
  // fragments f0 .. f0 + n - 1 (f0 = 0, 5, 10, 14; n = 5, 5, 4, 4).  One scalar base address, one M0 (LDS base) and one lane
  // offset VGPR per wave; fragments 1..3 ride on the instruction offset (applied to the global AND the LDS address), so M0
  // changes once more at most - rewriting it between DMAs serialised them (traced: ~170 clocks per DMA instruction).
  const unsigned lane16 = lane * 16;
  const int dma_f0 = gw_u < 2 ? gw_u * 5 : 10 + (gw_u - 2) * 4;
  auto fill_weights = [&](int cn, int half) __attribute__((always_inline)) {
    const unsigned long long ga = (unsigned long long)(uintptr_t)(wsrc + (size_t)cn * Q_W_SLOT + (half * 18 + dma_f0) * 512);
    const unsigned long long g = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ga) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ga >> 32)) << 32);  // uniform: say so
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_w + (cn & 1) * Q_W_SLOT + (half * 18 + dma_f0) * 512));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(dst), "s"(g) : "memory");
    if (gw_u < 2)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(lane16), "s"(dst + 4096), "s"(g + 4096) : "memory");
  };
  int boff[3][2];  // B fragment of column j + kx, k-step ksl: unit 2 ksl + hh
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) boff[kx][ksl] = (j + kx) * 32 + (((2 * ksl + hh) ^ (((j + kx) >> 2) & 3)) << 3);

  auto walk_init = [&](int t) __attribute__((always_inline))  {
    TileWalk w;
    w.tx = t % tiles_x;
    const int r = t / tiles_x;
    w.ty = r % tiles_y; w.b = r / tiles_y;
    return w;
  };
  auto walk_next = [&](TileWalk& w) __attribute__((always_inline))  {  // two tiles further along the (b, ty, tx) raster
    w.tx += 2;
    while (w.tx >= tiles_x) {
      w.tx -= tiles_x;
      if (++w.ty == tiles_y) { w.ty = 0; ++w.b; }
    }
  };
  TileWalk pw = walk_init(t_begin + grp), ew = pw;  // tile of the next prefetch / of the next epilogue

  // ---------------- staging geometry (tile-invariant, per thread) ----------------
  // main part: unit gt of the 272 units of a halo-row pair; remainder: units 256..271 of pair (gt >> 4), threads 0..79
  const int m_hi = gt >= Q_ROW_UNITS ? 1 : 0, m_v = gt - m_hi * Q_ROW_UNITS, m_px = m_v >> 2, m_part = m_v & 3;
  const int r_row = 2 * (gt >> 4) + 1, r_px = 30 + ((gt & 15) >> 2), r_part = gt & 3;
  const bool r_on = gt < 80;
  const unsigned voff_main = (unsigned)((m_hi * p.W + m_px) * 256 + m_part * 16);
  const unsigned voff_rem = r_on ? (unsigned)((r_row * p.W + r_px) * 256 + r_part * 16) : Q_OOB;
  _Float16* lds_main = my_in + pp128_lds(m_hi, m_px, m_part);  // + i * (2 rows) for pair i
  _Float16* lds_rem = my_in + pp128_lds(r_on ? r_row : 0, r_px, r_part);
  const unsigned pair_bytes = (unsigned)(2 * p.W * 256);
  // buffer over the whole input, based one row and one pixel before it: halo pixel (r, c) of tile (y0, x0) is at
  // ((b H + y0) W + x0) * 256 + (r W + c) * 256 from there.  Every out-of-image access is masked by voffset (soffset takes no
  // part in the range check), so num_records only has to exceed every valid voffset.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.in) - (size_t)(p.W + 1) * 256), 0, (int)0x7ffffff0, 0x00020000);

  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  u4_t rin[2][6];  // two register sets: the prefetch of item w + 2 is issued before item w + 1 is written to LDS
  auto prefetch_in = [&](int chunk, auto set_c) __attribute__((always_inline))  {
    constexpr int set = decltype(set_c)::value;
    const int y0 = pw.ty * Q_TH, x0 = pw.tx * Q_TW;
    const unsigned soff = (unsigned)((pw.b * p.H + y0) * p.W + x0) * 256u + (unsigned)chunk * 64u;
    const bool interior = y0 >= 1 && y0 + Q_TH + 1 <= p.H && x0 >= 1 && x0 + Q_TW + 1 <= p.W;
    if (interior) {
#pragma unroll
      for (int i = 0; i < 5; ++i) rin[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_main, soff + i * pair_bytes, 0);
      rin[set][5] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_rem, soff, 0);
    } else {
      // halo row r is inside the image for rlo <= r <= rhi, halo column c for clo <= c <= chi
      const int rlo = y0 == 0 ? 1 : 0, rhi = min(Q_THH - 1, p.H - y0), clo = x0 == 0 ? 1 : 0, chi = min(Q_TWH - 1, p.W - x0);
      const bool m_ok = m_px >= clo && m_px <= chi;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int r = 2 * i + m_hi;
        rin[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (m_ok && r >= rlo && r <= rhi) ? voff_main : Q_OOB, soff + i * pair_bytes, 0);
      }
      const bool r_ok = r_px >= clo && r_px <= chi && r_row >= rlo && r_row <= rhi;
      rin[set][5] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, r_ok ? voff_rem : Q_OOB, soff, 0);
    }
    if (chunk == Q_NCHUNK - 1) walk_next(pw);
  };
  auto stage_in = [&](auto set_c) __attribute__((always_inline))  {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < 5; ++i) *reinterpret_cast<u4_t*>(lds_main + i * (2 * Q_TWH * 32)) = rin[set][i];
    if (r_on) *reinterpret_cast<u4_t*>(lds_rem) = rin[set][5];
  };

  f16x_t acc[MT][2];
  // ---------------- MFMA half-step: 18 k-steps of one 32-channel chunk, fragments triple-buffered ----------------
  // chunk 0 starts the accumulators from the bias (row 8 g + 4 hh + e of M-tile m is register 4 g + e), chunks 1..3
  // accumulate in place: the accumulators are defined and consumed inside one tile iteration.
  auto mfma_item = [&](auto chunk_c) __attribute__((always_inline))  {
    constexpr int chunk = decltype(chunk_c)::value;
    const _Float16* wc = s_w + (chunk & 1) * Q_W_SLOT + lane * 8;
    const _Float16* ib = my_in + (gw * 2) * Q_TWH * 32;
    h8_t fa[3][MT], fb[3][2];
    auto load_frags = [&](int idx, int buf) __attribute__((always_inline))  {
      const int tap = idx >> 1, ksl = idx & 1, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[buf][m] = *reinterpret_cast<const h8_t*>(wc + (idx * MT + m) * 512);
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[buf][n] = *reinterpret_cast<const h8_t*>(ib + (n + ky) * Q_TWH * 32 + boff[kx][ksl]);
    };
    load_frags(0, 0);
    load_frags(1, 1);
    if constexpr (chunk == 0) {  // one LDS read per accumulator quad: the ds_read lands in the accumulator registers, no moves
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(s_bias + n * 64 + m * 32 + hh * 4 + g * 8);  // copy n: no CSE, no moves
            acc[m][n][4 * g + 0] = b4.x; acc[m][n][4 * g + 1] = b4.y; acc[m][n][4 * g + 2] = b4.z; acc[m][n][4 * g + 3] = b4.w;
          }
    }
#pragma unroll
    for (int idx = 0; idx < 18; ++idx) {
      if (idx + 2 < 18) load_frags(idx + 2, (idx + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][n] = mfma32(fa[idx % 3][m], fb[idx % 3][n], acc[m][n]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---------------- epilogue: ReLU (+ 2x2 max-pool) -> fp16 channels-last, 16-byte stores; the bias is already in ----------------
  auto epilogue = [&]() __attribute__((always_inline))  {
    const int y0 = ew.ty * Q_TH, x0 = ew.tx * Q_TW, b = ew.b;
    walk_next(ew);
    const int yb = y0 + gw * 2, x = x0 + j;
    const h2_t z2 = {(_Float16)0.f, (_Float16)0.f};
    auto relu2 = [&](float lo, float hi) __attribute__((always_inline))  -> unsigned {  // two values -> packed fp16, ReLU on the pair
      h2_t v = {(_Float16)lo, (_Float16)hi};
      v = __builtin_elementwise_max(v, z2);
      return *reinterpret_cast<const unsigned*>(&v);
    };
    auto pack2 = [](float lo, float hi) -> unsigned {
      const h2_t v = {(_Float16)lo, (_Float16)hi};
      return *reinterpret_cast<const unsigned*>(&v);
    };
    // lanes hh = 0 / 1 hold channels 4 hh .. + 3 of an 8-channel unit: permlane32_swap pairs them into one 16-byte store each
    auto store_pair = [&](_Float16* pix, int m, int g, unsigned a0, unsigned a1, unsigned b0, unsigned b1, bool ok) __attribute__((always_inline))  {
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if (ok) *reinterpret_cast<uint4*>(pix + m * 32 + (g + hh) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    };
    if constexpr (!POOL) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int y = yb + n;
        const bool ok = y < p.H && x < p.W;
        _Float16* pix = p.out + ((size_t)(b * p.H + y) * p.W + x) * p.cout + cb * 64;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const f16x_t& a = acc[m][n];
            store_pair(pix, m, g, relu2(a[4 * g + 0], a[4 * g + 1]), relu2(a[4 * g + 2], a[4 * g + 3]),
                       relu2(a[4 * g + 4], a[4 * g + 5]), relu2(a[4 * g + 6], a[4 * g + 7]), ok);
          }
      }
    } else {
      const int Ho = p.H >> 1, Wo = p.W >> 1;
      const int yo = yb >> 1, xo = x >> 1;
      const bool ok = !(x & 1) && yo < Ho && xo < Wo;
      _Float16* pix = p.out + ((size_t)(b * Ho + yo) * Wo + xo) * p.cout + cb * 64;
      auto pool1 = [&](int m, int r) __attribute__((always_inline))  -> float {  // max over the wave's two rows and 0 (v_max3), then over the column pair (dpp)
        const float tt = fmaxf(fmaxf(acc[m][0][r], acc[m][1][r]), 0.f);
        const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0xB1, 0xF, 0xF, false));
        return fmaxf(tt, nb);
      };
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; g += 2)
          store_pair(pix, m, g, pack2(pool1(m, 4 * g + 0), pool1(m, 4 * g + 1)), pack2(pool1(m, 4 * g + 2), pool1(m, 4 * g + 3)),
                     pack2(pool1(m, 4 * g + 4), pool1(m, 4 * g + 5)), pack2(pool1(m, 4 * g + 6), pool1(m, 4 * g + 7)), ok);
    }
  };

  // ---------------- schedule ----------------
  // Half-step s: group (s & 1) runs the MFMAs of its item s >> 1, the other group moves data: the weight-ring half for the
  // next pair of half-steps (chunk ((s >> 1) + 1) & 3, half s & 1), its next item's tile chunk into LDS, the prefetch of the
  // one after.  Each group has its own straight-line loop over its tiles (4 chunks unrolled) instead of one loop with a role
  // branch: the accumulators are then defined by chunk 0 and consumed by the epilogue inside one iteration - with the role
  // branch they were loop-carried through a phi, which cost 96 register moves per item in the MFMA role.
  const int T0 = (n_wg + 1) >> 1, T1 = n_wg >> 1;  // tiles of group 0 / 1 (T0 - T1 is 0 or 1)
  unsigned long long* trow = p.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 6;
  const bool tr_lane = SSHIP_PP_TRACE_BUILD && p.trace && gw == 0 && lane == 0;
  unsigned long long t0 = 0, t1 = 0;
  // SET: register set holding the item to stage; the prefetch goes to the other one.  Order: weight DMA first (longest latency,
  // ~2 k clocks to land), then the prefetch loads, then the LDS writes of the staged item (their loads are two half-steps old
  // and the oldest in the queue), and only then the wait for the DMA - most of its latency is behind the LDS writes by then.
  auto data_role = [&](auto set_c, int dma_chunk, int dma_half, bool do_stage, bool do_prefetch, int pf_chunk, bool tr) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(SSHIP_PP_PRIO);
    if (tr) t0 = __builtin_readcyclecounter();
    if (dma_chunk >= 0) fill_weights(dma_chunk, dma_half);
    if (tr) { t1 = __builtin_readcyclecounter(); trow[0] = t1 - t0; t0 = t1; }
    if (do_prefetch) prefetch_in(pf_chunk, std::integral_constant<int, SET ^ 1>{});
    if (tr) { t1 = __builtin_readcyclecounter(); trow[2] = t1 - t0; t0 = t1; }
    if (do_stage) stage_in(set_c);
    if (do_prefetch) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // the weight DMA has landed; the six prefetch loads stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tr) { t1 = __builtin_readcyclecounter(); trow[1] = t1 - t0; t0 = t1; }
    __syncthreads();
    if (tr) trow[4] = __builtin_readcyclecounter() - t0;
    if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto mfma_role = [&](auto chunk_c, bool tr) __attribute__((always_inline)) {
    constexpr int chunk = decltype(chunk_c)::value;
    if (tr) t0 = __builtin_readcyclecounter();
    mfma_item(chunk_c);
    if constexpr (chunk == Q_NCHUNK - 1) epilogue();
    if (tr) { t1 = __builtin_readcyclecounter(); trow[3] = t1 - t0; t0 = t1; }
    __syncthreads();
    if (tr) trow[5] = __builtin_readcyclecounter() - t0;
  };
  const std::integral_constant<int, 0> c0{}, setA{};
  const std::integral_constant<int, 1> c1{}, setB{};
  const std::integral_constant<int, 2> c2{};
  const std::integral_constant<int, 3> c3{};
  if ((grp ? T1 : T0) > 0) prefetch_in(0, setA);
  __syncthreads();  // slot 0 and the bias are in LDS
  if (grp == 0) {
    data_role(setA, -1, 0, true, true, 1, false);  // half-step -1: item 0 into LDS, prefetch item 1
#pragma unroll 1
    for (int it = 0; it < T0; ++it) {
      const bool more = it + 1 < T0, tr = tr_lane && it == 1;
      mfma_role(c0, tr);
      data_role(setB, 1, 1, true, true, 2, tr);
      mfma_role(c1, false);
      data_role(setA, 2, 1, true, true, 3, false);
      mfma_role(c2, false);
      data_role(setB, 3, 1, true, more, 0, false);
      mfma_role(c3, false);
      data_role(setA, 0, 1, more, more, 1, false);
    }
  } else {
    __syncthreads();  // half-step -1: nothing to do for this group
#pragma unroll 1
    for (int it = 0; it < T1; ++it) {
      const bool more = it + 1 < T1, tr = tr_lane && it == 1;
      data_role(setA, 1, 0, true, true, 1, tr);
      mfma_role(c0, tr);
      data_role(setB, 2, 0, true, true, 2, false);
      mfma_role(c1, false);
      data_role(setA, 3, 0, true, true, 3, false);
      mfma_role(c2, false);
      data_role(setB, 0, 0, true, more, 0, false);
      mfma_role(c3, false);
    }
    if (T0 > T1) {  // group 0 has one more tile: keep its weight ring filled and keep the barrier count
#pragma unroll 1
      for (int c = 0; c < Q_NCHUNK; ++c) {
        data_role(setA, (c + 1) & 3, 0, false, false, 0, false);
        __syncthreads();
      }
    }
  }
}

template <bool POOL>
static hipError_t launch_pp128(const Pp128Args& a, hipStream_t s) {
  constexpr size_t smem = (size_t)(2 * Q_W_SLOT + 2 * Q_IN_HALFS) * 2 + 128 * 4;
  static_assert(smem <= 163840, "LDS budget");
  auto kern = conv3x3_pp128<POOL>;
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_rc != hipSuccess) return attr_rc;
  const int ncb = a.cout / 64;
  const int ntiles = a.B * ((a.W + Q_TW - 1) / Q_TW) * ((a.H + Q_TH - 1) / Q_TH);
  int gx = cu_count() / ncb;  // one persistent workgroup per CU
  if (gx < 1) gx = 1;
  if (gx * 2 > ntiles) gx = (ntiles + 1) / 2;
  if (gx < 1) gx = 1;
  static const bool trace_on = SSHIP_PP_TRACE_BUILD && dev_env("SSHIP_PP_TRACE") != nullptr;
  static unsigned long long* tbuf = nullptr;
  Pp128Args b = a;
  if (trace_on) {
    if (!tbuf) (void)hipMalloc(&tbuf, 4096 * 2 * 6 * 8);
    (void)hipMemsetAsync(tbuf, 0, 4096 * 2 * 6 * 8, s);
    b.trace = tbuf;
  }
  hipLaunchKernelGGL(kern, dim3(gx, ncb), dim3(512), smem, s, b);
  if (trace_on) {
    std::vector<unsigned long long> h(4096 * 2 * 6);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[6] = {0}; long cnt = 0;
    for (int i = 0; i < gx * ncb * 2; ++i) {
      if (!h[i * 6 + 3]) continue;
      for (int k = 0; k < 6; ++k) sum[k] += (double)h[i * 6 + k];
      ++cnt;
    }
    if (cnt) fprintf(stderr, "[pp128 trace cout=%d pool=%d] weight dma issue=%.0f stage+dma wait=%.0f prefetch issue=%.0f | mfma=%.0f | barrier wait after data=%.0f after mfma=%.0f (clk, %ld groups)\n",
                     a.cout, (int)POOL, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, cnt);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// conv3x3_pp128w: the same ping-pong scheme on 16 x 32-pixel tiles with the data-movement role reduced to LDS-DMA.
//
// The 8-row kernel above spends 3.0-3.6 k clocks per half-step on 72 MFMAs (2 304 clocks of matrix pipe): two barriers, the
// first fragments' LDS latency and the bias reads per 72 MFMAs, one ds_read_b128 per MFMA, and a data role that needs its
// 2.7 k clocks (profiles/r02_h_pp128_role_trace.txt).  Here
//   * a work item (tile, chunk) is 144 MFMAs per wave (4 pixel rows x 2 M-tiles x 18 k-steps): the per-item costs halve, and one
//     36 KiB weight chunk now feeds twice the pixels (half the weight DMA per FLOP);
//   * the B fragments of a k-step are ROWS of the halo tile: tap row ky of output row n reads halo row n + ky, so the three tap
//     rows of one (kx, k-step) share 6 row fragments between their 12 (n, ky) pairs - with the weights' 6 fragments that is
//     12 ds_read_b128 per 24 MFMAs (0.5 per MFMA, was 1.0): the LDS array is busy a quarter of the half-step;
//   * the input tile chunk (18 x 34 px x 32 ch = 39 KB) goes global -> LDS by `buffer_load_dwordx4 ... lds`: no staging
//     registers (they were 48 VGPRs; the accumulators need 128 here), no ds_write_b128, ~15 DMA instructions per wave and
//     half-step.  The swizzled LDS layout is produced by the SOURCE addresses (a DMA writes lane-linear: LDS unit p of the tile
//     is fetched from pixel p / 4, channel unit (p & 3) ^ swizzle); out-of-image halo pixels have their voffset pushed out of
//     the buffer's range and the DMA writes zeros (scripts/ubench/lds_dma_oob.hip).
// LDS: 2 x 36 KiB weight ring + 2 x 39 KiB input tiles + 1 KiB bias = 151 KiB.
// ---------------------------------------------------------------------------------------------------
constexpr int R_TH = 16, R_THH = 18;
constexpr int R_IN_UNITS = R_THH * Q_TWH * 4;      // 2 448 sixteen-byte units per tile chunk
constexpr int R_IN_DMA = (R_IN_UNITS + 63) / 64;   // 39 wave-wide DMA instructions (the last one: 16 units + zero padding)
constexpr int R_IN_HALFS = R_IN_DMA * 64 * 8;      // 19 968 halfs = 39 936 B per wave group
constexpr int R_DMA_PER_WAVE = 10;                 // wave g of a group issues instructions 10 g .. 10 g + 9 (wave 3: 9 of them)
constexpr unsigned R_OOB = 0x80000000u;            // beyond num_records, and no 32-bit wrap with soffset + instruction offset on top

typedef int rsrc4_t __attribute__((ext_vector_type(4)));

// shared edge tiles (see the kernel): no pooling (conv4a / 4b / Pa / Da), an even number of images, an edge strip of at most 15 pixels
// behind at least one full tile column.  SUPERSLAM_HIP_CONV128_PAIRS=0 switches them off (A/B).
__host__ __device__ inline bool pp128w_pairs_shape(bool pool, int B, int W) {
  const int we = W - ((W + Q_TW - 1) / Q_TW - 1) * Q_TW;
  return !pool && (B & 1) == 0 && W > Q_TW && we >= 1 && we <= 15;
}
#define pp128w_pairs(pool, B, W) (p.pairs_on && pp128w_pairs_shape(pool, B, W))
template <bool POOL, int NCH>
__global__ __launch_bounds__(512, 2) void conv3x3_pp128w(Pp128Args p) {
  constexpr int MT = 2, NT = 4;
  constexpr int PB = NCH * 64;  // bytes per input pixel: NCH chunks of 32 channels (cin = 128: four chunks through the weight ring; cin = 64: two, both resident)
  extern __shared__ __attribute__((aligned(16))) char smem128w[];
  _Float16* s_w = reinterpret_cast<_Float16*>(smem128w);  // [2 slots][Q_W_SLOT]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  const int gw = wave & 3;
  const int gw_u = __builtin_amdgcn_readfirstlane(gw);
  _Float16* my_in = s_w + 2 * Q_W_SLOT + grp * R_IN_HALFS;
  float* s_bias = reinterpret_cast<float*>(s_w + 2 * Q_W_SLOT + 2 * R_IN_HALFS);  // [4 copies][64]

  const int tiles_x = (p.W + Q_TW - 1) / Q_TW, tiles_y = (p.H + R_TH - 1) / R_TH;
  // Paired edge strips (pp128w_pairs): when the last tile column holds we = W - 32 (tiles_x - 1) <= 15 pixels, the strips of TWO images
  // (each we + 2 halo columns wide) share one 34-column tile: LDS columns 0 .. we + 1 = image b, we + 2 .. 2 we + 3 = image b + 1.  An MFMA's
  // N dimension is 32 pixels of one halo row whatever they are, so nothing changes between the DMA and the epilogue; conv4a / 4b / Pa at
  // 47 x 172 cells run 33 tiles per image pair instead of 36.  The walk then runs over image PAIRS with a virtual tile row of
  // 2 (tiles_x - 1) + 1 entries: [image 2 bp: columns 0 .. tiles_x - 2 | image 2 bp + 1: the same | the shared edge tile].
  const bool pairs = pp128w_pairs(POOL, p.B, p.W);
  const int txn = tiles_x - 1, we = p.W - txn * Q_TW;                 // full-width tile columns per image, width of the edge strip
  const int tiles_xv = pairs ? 2 * txn + 1 : tiles_x, nb_walk = pairs ? p.B / 2 : p.B;
  const int ntiles = nb_walk * tiles_xv * tiles_y;
  const int cb = blockIdx.y;
  const int t_begin = (int)((long long)blockIdx.x * ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int n_wg = t_end - t_begin;
  if (n_wg <= 0) return;
  const _Float16* wsrc = p.wpack + (size_t)cb * (NCH * Q_W_SLOT);

  // chunk 0 -> slot 0 (NCH == 2: chunk 1 -> slot 1 as well, and the ring never moves: 64 input channels x 64 rows x 9 taps fit)
  for (int u = tid; u < (NCH == 2 ? 2 : 1) * Q_W_SLOT / 8; u += 512) *reinterpret_cast<uint4*>(s_w + u * 8) = *reinterpret_cast<const uint4*>(wsrc + u * 8);
  if (tid < 256) s_bias[tid] = p.bias[cb * 64 + (tid & 63)];  // one copy per N-tile: the accumulators start from it, one ds_read each
  const unsigned lane16 = lane * 16;
  const int dma_f0 = gw_u < 2 ? gw_u * 5 : 10 + (gw_u - 2) * 4;
  auto fill_weights = [&](int cn, int half) __attribute__((always_inline)) {  // as in conv3x3_pp128
    const unsigned long long ga = (unsigned long long)(uintptr_t)(wsrc + (size_t)cn * Q_W_SLOT + (half * 18 + dma_f0) * 512);
    const unsigned long long g = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ga) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ga >> 32)) << 32);
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_w + (cn & 1) * Q_W_SLOT + (half * 18 + dma_f0) * 512));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(dst), "s"(g) : "memory");
    if (gw_u < 2)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(lane16), "s"(dst + 4096), "s"(g + 4096) : "memory");
  };
  int boff[3][2];  // B fragment of column j + kx, k-step ksl: unit 2 ksl + hh
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) boff[kx][ksl] = (j + kx) * 32 + (((2 * ksl + hh) ^ (((j + kx) >> 2) & 3)) << 3);

  // SSHIP_PP128_COLMAJOR = 1 (round 6 experiment, off): the walk runs down the tile columns (ty fastest) as in conv_pp.hip.  Measured: no gain here
  // (fetch 1 270 -> 1 256 MB on conv3b, joules unchanged): profiles/r06_g_conv_tile_walk.txt.
  auto walk_init = [&](int t) __attribute__((always_inline)) {
    TileWalk w;
    if constexpr (SSHIP_PP128_COLMAJOR != 0) {
      w.ty = t % tiles_y;
      const int r = t / tiles_y;
      w.tx = r % tiles_xv; w.b = r / tiles_xv;
    } else {
      w.tx = t % tiles_xv;
      const int r = t / tiles_xv;
      w.ty = r % tiles_y; w.b = r / tiles_y;
    }
    return w;
  };
  auto walk_next = [&](TileWalk& w) __attribute__((always_inline)) {
    if constexpr (SSHIP_PP128_COLMAJOR != 0) {
      w.ty += 2;
      while (w.ty >= tiles_y) {
        w.ty -= tiles_y;
        if (++w.tx == tiles_xv) { w.tx = 0; ++w.b; }
      }
    } else {
      w.tx += 2;
      while (w.tx >= tiles_xv) {
        w.tx -= tiles_xv;
        if (++w.ty == tiles_y) { w.ty = 0; ++w.b; }
      }
    }
  };
  // virtual (tx, b) of the walk -> image, tile column, and whether this is a shared edge tile
  auto decode = [&](const TileWalk& w, int& img, int& tx, bool& shared) __attribute__((always_inline)) {
    if (!pairs) { img = w.b; tx = w.tx; shared = false; return; }
    shared = w.tx == 2 * txn;
    const int second = !shared && w.tx >= txn;
    img = 2 * w.b + second;
    tx = shared ? txn : w.tx - second * txn;
  };
  TileWalk pw = walk_init(t_begin + grp), ew = pw;  // tile of the next input DMA / of the next epilogue

  // ---------------- input DMA geometry (tile-invariant, per lane) ----------------
  // instruction i of this wave fills LDS units (10 gw + i) * 64 + lane of the group's tile buffer.  Unit p = pixel p / 4 of the
  // 18 x 34 halo raster, slot p & 3, which holds channel unit slot ^ swizzle(column) (pp128_lds).  Instructions 4 g .. 4 g + 3
  // share one M0 and ride on the instruction offset (q * 1 KiB), which also moves the global address: voff carries the
  // opposite, and the buffer is based 4 KiB low so that it stays positive.
  // Tile-invariant per lane: the ten voffsets, and which of the ten instructions touch a halo row / column that leaves the image
  // on a tile of the first / last tile row / column - four 10-bit masks (bit i = instruction i stays inside), so that an edge
  // tile costs four scalar-selected ANDs and a bit test per instruction instead of a (row, column) walk with four compares each
  // (the first version spent ~180 VALU instructions per DMA round there, at the ~10 clocks a VALU instruction costs next to the
  // other group's MFMA stream: 2.2-2.8 k clocks of "DMA issue" per half-step, profiles/r03_n_pp128w_role_trace.txt).
  unsigned voff[R_DMA_PER_WAVE];
  unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0;  // bit i: row >= 1 | row <= H - y0(last) | column >= 1 | column <= W - x0(last)
  unsigned m_second = 0, m_dead = 0;  // shared edge tile: bit i = the unit belongs to image b + 1's strip | to no strip (right halo of either, spare columns)
  {
    const int px = (gw * R_DMA_PER_WAVE) * 16 + (lane >> 2);
    int r = (px * 241) >> 13;  // px / 34 for px < 640
    int c = px - r * Q_TWH;
    const int rhi_last = min(R_THH - 1, p.H - (tiles_y - 1) * R_TH), chi_last = min(Q_TWH - 1, p.W - (tiles_x - 1) * Q_TW);
#pragma unroll
    for (int i = 0; i < R_DMA_PER_WAVE; ++i) {
      const int u = (lane & 3) ^ ((c >> 2) & 3);
      voff[i] = r < R_THH ? (unsigned)(4096 - (i & 3) * 1024 + (r * p.W + c) * PB + u * 16) : R_OOB;  // r = 18: padding of the last instruction
      m_top |= (r >= 1 ? 1u : 0u) << i;
      m_bot |= (r <= rhi_last ? 1u : 0u) << i;
      m_left |= (c >= 1 ? 1u : 0u) << i;
      m_right |= (c <= chi_last ? 1u : 0u) << i;
      m_second |= (c >= we + 2 && c < 2 * we + 3 ? 1u : 0u) << i;       // columns we + 2 .. 2 we + 2: left halo + the we pixels of image b + 1
      m_dead |= (c == we + 1 || c >= 2 * we + 3 ? 1u : 0u) << i;        // column W of either image (zero padding) and the unused columns
      c += 16;
      if (c >= Q_TWH) { c -= Q_TWH; ++r; }
    }
  }
  rsrc4_t rs;
  {
    const unsigned long long ba = (unsigned long long)(uintptr_t)p.in - (unsigned long long)(p.W + 1) * (unsigned long long)PB - 4096ull;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
    rs[2] = 0x7ffffff0;
    rs[3] = 0x00020000;
  }
  const unsigned lds_in0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(my_in + (gw_u * R_DMA_PER_WAVE) * 512));
  // one tile chunk -> LDS at m0a / m0b / m0c (the three instruction groups of a wave)
  auto dma_tile = [&](const TileWalk& w, int chunk, unsigned m0a, unsigned m0b, unsigned m0c) __attribute__((always_inline)) {
    int img, txr;
    bool shared;
    decode(w, img, txr, shared);
    const int y0 = w.ty * R_TH, x0 = txr * Q_TW;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((img * p.H + y0) * p.W + x0) * (unsigned)PB + (unsigned)chunk * 64u);
    const bool interior = !shared && y0 >= 1 && y0 + R_TH + 1 <= p.H && x0 >= 1 && x0 + Q_TW + 1 <= p.W;
    unsigned v[R_DMA_PER_WAVE];
    if (interior) {
#pragma unroll
      for (int i = 0; i < R_DMA_PER_WAVE; ++i) v[i] = voff[i];
    } else if (shared) {
      // strip of image b: the ordinary addressing (LDS column c = image column x0 - 1 + c); strip of image b + 1: the same unit we + 2
      // columns further left, one image further on; everything else reads zeros
      unsigned ok = 0x3ffu & ~m_dead;
      if (w.ty == 0) ok &= m_top;
      if (w.ty == tiles_y - 1) ok &= m_bot;
      const unsigned delta = (unsigned)(p.H * p.W - (we + 2)) * (unsigned)PB;
#pragma unroll
      for (int i = 0; i < R_DMA_PER_WAVE; ++i) v[i] = (ok >> i) & 1u ? voff[i] + ((m_second >> i) & 1u ? delta : 0u) : R_OOB;
    } else {
      unsigned ok = 0x3ffu;
      if (w.ty == 0) ok &= m_top;
      if (w.ty == tiles_y - 1) ok &= m_bot;
      if (txr == 0) ok &= m_left;
      if (txr == tiles_x - 1) ok &= m_right;
#pragma unroll
      for (int i = 0; i < R_DMA_PER_WAVE; ++i) v[i] = (ok >> i) & 1u ? voff[i] : R_OOB;
    }
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %6, %7 offen lds\n\t"
                 "buffer_load_dwordx4 %2, %6, %7 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %3, %6, %7 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %4, %6, %7 offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(m0a), "s"(rs), "s"(soff) : "memory");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %6, %7 offen lds\n\t"
                 "buffer_load_dwordx4 %2, %6, %7 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %3, %6, %7 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %4, %6, %7 offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "s"(m0b), "s"(rs), "s"(soff) : "memory");
    if (gw_u < 3)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %4, %5 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %4, %5 offen offset:1024 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(v[8]), "v"(v[9]), "s"(m0c), "s"(rs), "s"(soff) : "memory");
    else  // wave 3: instructions 30..38 - nine of them
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(v[8]), "s"(m0c), "s"(rs), "s"(soff) : "memory");
  };
  // (A pixel is 256 B = two 128-byte lines, a chunk 64 B: the DMA of chunk 0 or 2 is the one that goes to HBM and it has to land
  // inside its own half-step.  Touching those lines one round early - a second DMA round into a scratch area, issued after the
  // real one so that the in-order vmcnt wait skips it - cut the traced DMA wait from 4.3 k to 1.4 k clocks on the 47 x 172 layers
  // and made every layer 3-7 % SLOWER: twice the DMA traffic in half of the rounds.  profiles/r03_n_pp128w_role_trace.txt; removed.)
  auto dma_in = [&](int chunk) __attribute__((always_inline)) {
    dma_tile(pw, chunk, lds_in0, lds_in0 + 4096u, lds_in0 + 8192u);
    if (chunk == NCH - 1) walk_next(pw);
  };

  f16x_t acc[MT][NT];
  // ---------------- MFMA half-step: one 32-channel chunk = 6 (kx, k-step) steps x 3 tap rows, 8 MFMAs per sub-step ----------------
  // The six halo-row fragments of step t (rows 4 gw .. 4 gw + 5, column j + kx, k-step ksl) live in a ring of EIGHT registers
  // sets, row R of step t in slot (6 t + R) & 7: tap row ky reads rows ky .. ky + 3, so row 0 is dead after sub-step 0 and row 1
  // after sub-step 1 - the next step's rows 2 and 3 land there, its rows 0 and 1 in the two spare slots, and rows 4 and 5 are
  // fetched in their own step's first sub-step (first used in the second).  Every fragment is requested at least one sub-step
  // (8 MFMAs = 256 clocks) ahead; af[s % 3][m], the weights of sub-step s, two sub-steps ahead.
  auto mfma_item = [&](auto chunk_c) __attribute__((always_inline)) {
    constexpr int chunk = decltype(chunk_c)::value;
    const _Float16* wc = s_w + (chunk & 1) * Q_W_SLOT + lane * 8;
    const _Float16* ib = my_in + (gw * 4) * Q_TWH * 32;
    h8_t af[3][MT], rows[8];
    auto load_a = [&](int s_, int buf) __attribute__((always_inline)) {
      const int t = s_ / 3, ky = s_ - 3 * t, kx = t >> 1, ksl = t & 1, idx = (ky * 3 + kx) * 2 + ksl;
#pragma unroll
      for (int m = 0; m < MT; ++m) af[buf][m] = *reinterpret_cast<const h8_t*>(wc + (idx * MT + m) * 512);
    };
    auto load_row = [&](int t, int row) __attribute__((always_inline)) {
      rows[(6 * t + row) & 7] = *reinterpret_cast<const h8_t*>(ib + row * Q_TWH * 32 + boff[t >> 1][t & 1]);
    };
#pragma unroll
    for (int row = 0; row < 4; ++row) load_row(0, row);
    load_a(0, 0);
    load_a(1, 1);
    if constexpr (chunk == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(s_bias + n * 64 + m * 32 + hh * 4 + g * 8);  // copy n: no CSE, no moves
            acc[m][n][4 * g + 0] = b4.x; acc[m][n][4 * g + 1] = b4.y; acc[m][n][4 * g + 2] = b4.z; acc[m][n][4 * g + 3] = b4.w;
          }
    }
    if constexpr ((SSHIP_PP128_ABL & 8) != 0) {  // every register the loop reads is defined once, nothing is re-read
      load_a(2, 2);
#pragma unroll
      for (int row = 4; row < 8; ++row) rows[row] = rows[row - 4];
    }
#pragma unroll
    for (int s_ = 0; s_ < 18; ++s_) {
      const int t = s_ / 3, ky = s_ - 3 * t;
      if constexpr ((SSHIP_PP128_ABL & 8) == 0) {
      if (s_ + 2 < 18) load_a(s_ + 2, (s_ + 2) % 3);
      if (ky == 0) {
        load_row(t, 4); load_row(t, 5);
        if (t + 1 < 6) { load_row(t + 1, 0); load_row(t + 1, 1); }
      } else if (t + 1 < 6) load_row(t + 1, ky + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr ((SSHIP_PP128_ABL & 1) != 0) asm volatile("" :: "v"(af[s_ % 3][m]), "v"(rows[(6 * t + n + ky) & 7]));
          else acc[m][n] = mfma32(af[s_ % 3][m], rows[(6 * t + n + ky) & 7], acc[m][n]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---------------- epilogue: ReLU (+ 2x2 max-pool) -> fp16 channels-last, 16-byte stores; the bias is already in ----------------
  auto epilogue = [&]() __attribute__((always_inline)) {
    int b, txr;
    bool shared;
    decode(ew, b, txr, shared);
    const int y0 = ew.ty * R_TH, x0 = txr * Q_TW;
    walk_next(ew);
    const int yb = y0 + gw * 4;
    int x = x0 + j;
    if (shared) {  // output column j < we: image b; we + 2 <= j < 2 we + 2: image b + 1 (its column j - (we + 2)); the rest is nobody's
      const bool second = j >= we + 2;
      b += second;
      x = second ? x0 + j - (we + 2) : (j < we ? x : p.W);
    }
    const h2_t z2 = {(_Float16)0.f, (_Float16)0.f};
    auto relu2 = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {
      h2_t v = {(_Float16)lo, (_Float16)hi};
      v = __builtin_elementwise_max(v, z2);
      return *reinterpret_cast<const unsigned*>(&v);
    };
    auto pack2 = [](float lo, float hi) -> unsigned {
      const h2_t v = {(_Float16)lo, (_Float16)hi};
      return *reinterpret_cast<const unsigned*>(&v);
    };
    auto store_pair = [&](_Float16* pix, int m, int g, unsigned a0, unsigned a1, unsigned b0, unsigned b1, bool ok) __attribute__((always_inline)) {
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if constexpr ((SSHIP_PP128_ABL & 16) != 0) { asm volatile("" :: "v"(r0[0]), "v"(r1[0]), "v"(r0[1]), "v"(r1[1])); return; }  // everything but the store
      if (ok) *reinterpret_cast<uint4*>(pix + m * 32 + (g + hh) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    };
    if constexpr (!POOL) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int y = yb + n;
        const bool ok = y < p.H && x < p.W;
        _Float16* pix = p.out + ((size_t)(b * p.H + y) * p.W + x) * p.cout + cb * 64;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const f16x_t& a = acc[m][n];
            store_pair(pix, m, g, relu2(a[4 * g + 0], a[4 * g + 1]), relu2(a[4 * g + 2], a[4 * g + 3]),
                       relu2(a[4 * g + 4], a[4 * g + 5]), relu2(a[4 * g + 6], a[4 * g + 7]), ok);
          }
      }
    } else {
      const int Ho = p.H >> 1, Wo = p.W >> 1;
      const int xo = x >> 1;
#pragma unroll
      for (int q = 0; q < NT / 2; ++q) {
        const int yo = (yb >> 1) + q;
        const bool ok = !(x & 1) && yo < Ho && xo < Wo;
        _Float16* pix = p.out + ((size_t)(b * Ho + yo) * Wo + xo) * p.cout + cb * 64;
        auto pool1 = [&](int m, int r) __attribute__((always_inline)) -> float {  // max over the row pair and 0 (v_max3), then over the column pair (dpp)
          const float tt = fmaxf(fmaxf(acc[m][2 * q][r], acc[m][2 * q + 1][r]), 0.f);
          const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0xB1, 0xF, 0xF, false));
          return fmaxf(tt, nb);
        };
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int g = 0; g < 4; g += 2)
            store_pair(pix, m, g, pack2(pool1(m, 4 * g + 0), pool1(m, 4 * g + 1)), pack2(pool1(m, 4 * g + 2), pool1(m, 4 * g + 3)),
                       pack2(pool1(m, 4 * g + 4), pool1(m, 4 * g + 5)), pack2(pool1(m, 4 * g + 6), pool1(m, 4 * g + 7)), ok);
      }
    }
  };

  // ---------------- schedule (conv3x3_pp128's, with the two-stage register staging collapsed into one DMA) ----------------
  // Half-step s: group (s & 1) runs the MFMAs of its item s >> 1; the other group issues the weight-ring half for the next pair of
  // half-steps and the tile chunk of ITS next item, waits for both to land and meets the MFMA group at the barrier.
  const int T0 = (n_wg + 1) >> 1, T1 = n_wg >> 1;
  unsigned long long* trow = p.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 6;
  const bool tr_lane = SSHIP_PP_TRACE_BUILD && p.trace && gw == 0 && lane == 0;
  unsigned long long t0 = 0, t1 = 0;
  auto data_role = [&](int dma_chunk, int dma_half, int in_chunk, bool tr) __attribute__((always_inline)) {
    if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(SSHIP_PP_PRIO);  // ~40 instructions that decide when the DMAs start
    if (tr) t0 = __builtin_readcyclecounter();
    if constexpr ((SSHIP_PP128_ABL & 2) != 0) {
      if (in_chunk >= 0 && in_chunk == NCH - 1) walk_next(pw);   // the walk goes on, nothing moves
    } else {
    if (in_chunk >= 0) dma_in(in_chunk);                    // first: the tile chunk comes from HBM half of the time, the weights from L2
    if (NCH == 4 && dma_chunk >= 0) fill_weights(dma_chunk, dma_half);
    }
    if (SSHIP_PP_PRIO) __builtin_amdgcn_s_setprio(0);
    if (tr) { t1 = __builtin_readcyclecounter(); trow[0] = t1 - t0; t0 = t1; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tr) { t1 = __builtin_readcyclecounter(); trow[1] = t1 - t0; trow[2] = 0; t0 = t1; }
    __syncthreads();
    if (tr) trow[4] = __builtin_readcyclecounter() - t0;
  };
  auto mfma_role = [&](auto chunk_c, bool tr) __attribute__((always_inline)) {
    constexpr int chunk = decltype(chunk_c)::value;
    if (tr) t0 = __builtin_readcyclecounter();
    mfma_item(chunk_c);
    if constexpr (chunk == NCH - 1) {
      if constexpr ((SSHIP_PP128_ABL & 4) != 0) {
        walk_next(ew);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) asm volatile("" :: "v"(acc[m][n]));
      } else epilogue();
    }
    if (tr) { t1 = __builtin_readcyclecounter(); trow[3] = t1 - t0; t0 = t1; }
    __syncthreads();
    if (tr) trow[5] = __builtin_readcyclecounter() - t0;
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  const std::integral_constant<int, 3> c3{};
  __syncthreads();  // slot 0 and the bias are in LDS
  if constexpr (NCH == 4) {
    if (grp == 0) {
      data_role(-1, 0, T0 > 0 ? 0 : -1, false);  // half-step -1: item 0 into LDS
#pragma unroll 1
      for (int it = 0; it < T0; ++it) {
        const bool more = it + 1 < T0, tr = tr_lane && it == 1;
        mfma_role(c0, tr);
        data_role(1, 1, 1, tr);
        mfma_role(c1, false);
        data_role(2, 1, 2, false);
        mfma_role(c2, false);
        data_role(3, 1, 3, false);
        mfma_role(c3, false);
        data_role(0, 1, more ? 0 : -1, false);
      }
    } else {
      __syncthreads();  // half-step -1: nothing to do for this group
#pragma unroll 1
      for (int it = 0; it < T1; ++it) {
        const bool tr = tr_lane && it == 1;
        data_role(1, 0, 0, tr);
        mfma_role(c0, tr);
        data_role(2, 0, 1, false);
        mfma_role(c1, false);
        data_role(3, 0, 2, false);
        mfma_role(c2, false);
        data_role(0, 0, 3, false);
        mfma_role(c3, false);
      }
      if (T0 > T1) {  // group 0 has one more tile: keep its weight ring filled and keep the barrier count
#pragma unroll 1
        for (int c = 0; c < Q_NCHUNK; ++c) {
          data_role((c + 1) & 3, 0, -1, false);
          __syncthreads();
        }
      }
    }
  } else {  // two chunks per tile, weights resident: the data role only moves the group's next tile chunk
    if (grp == 0) {
      data_role(-1, 0, T0 > 0 ? 0 : -1, false);
#pragma unroll 1
      for (int it = 0; it < T0; ++it) {
        const bool more = it + 1 < T0, tr = tr_lane && it == 1;
        mfma_role(c0, tr);
        data_role(-1, 0, 1, tr);
        mfma_role(c1, false);
        data_role(-1, 0, more ? 0 : -1, false);
      }
    } else {
      __syncthreads();
#pragma unroll 1
      for (int it = 0; it < T1; ++it) {
        const bool tr = tr_lane && it == 1;
        data_role(-1, 0, 0, tr);
        mfma_role(c0, tr);
        data_role(-1, 0, 1, false);
        mfma_role(c1, false);
      }
      if (T0 > T1) {
        for (int c = 0; c < 2; ++c) { __syncthreads(); __syncthreads(); }
      }
    }
  }
}

template <bool POOL, int NCH>
static hipError_t launch_pp128w(const Pp128Args& a, hipStream_t s) {
  constexpr size_t smem = (size_t)(2 * Q_W_SLOT + 2 * R_IN_HALFS) * 2 + 256 * 4;
  static_assert(smem <= 163840, "LDS budget");
  auto kern = conv3x3_pp128w<POOL, NCH>;
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_rc != hipSuccess) return attr_rc;
  const int ncb = a.cout / 64;
  static const bool pairs_env = !(dev_env("SUPERSLAM_HIP_CONV128_PAIRS") && atoi(dev_env("SUPERSLAM_HIP_CONV128_PAIRS")) == 0);
  const bool pairs = pairs_env && pp128w_pairs_shape(POOL, a.B, a.W);
  const int tiles_x = (a.W + Q_TW - 1) / Q_TW;
  const int ntiles = (pairs ? a.B / 2 * (2 * (tiles_x - 1) + 1) : a.B * tiles_x) * ((a.H + R_TH - 1) / R_TH);  // as the kernel counts them
  int gx = cu_count() / ncb;  // one persistent workgroup per CU
  if (gx < 1) gx = 1;
  if (gx * 2 > ntiles) gx = (ntiles + 1) / 2;
  if (gx < 1) gx = 1;
  static const bool trace_on = SSHIP_PP_TRACE_BUILD && dev_env("SSHIP_PP_TRACE") != nullptr;
  static unsigned long long* tbuf = nullptr;
  Pp128Args b = a;
  b.pairs_on = pairs_env ? 1 : 0;
  if (trace_on) {
    if (!tbuf) (void)hipMalloc(&tbuf, 4096 * 2 * 6 * 8);
    (void)hipMemsetAsync(tbuf, 0, 4096 * 2 * 6 * 8, s);
    b.trace = tbuf;
  }
  hipLaunchKernelGGL(kern, dim3(gx, ncb), dim3(512), smem, s, b);
  if (trace_on) {
    std::vector<unsigned long long> h(4096 * 2 * 6);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[6] = {0}; long cnt = 0;
    for (int i = 0; i < gx * ncb * 2; ++i) {
      if (!h[i * 6 + 3]) continue;
      for (int k = 0; k < 6; ++k) sum[k] += (double)h[i * 6 + k];
      ++cnt;
    }
    if (cnt) fprintf(stderr, "[pp128w trace cin=%d cout=%d pool=%d] dma issue=%.0f dma wait=%.0f | mfma=%.0f | barrier wait after data=%.0f after mfma=%.0f (clk, %ld groups)\n",
                     NCH * 32, a.cout, (int)POOL, sum[0] / cnt, sum[1] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, cnt);
  }
  return hipGetLastError();
}

// byte offsets inside the kernels are 32-bit (buffer addressing): inputs of 2 GiB and more stay on the kernels of conv_pp.hip
bool sp_conv3x3_pp128_fits(int B, int H, int W, int cin) { return (size_t)B * H * W * cin * 2 + (size_t)(W + 1) * cin * 4 + 8192 < 0x7f000000ull; }

// w.w_q: packed by upload_conv_q (cout tile 64, chunk 32); cin = 128, or 64 (16-row-tile kernel only)
hipError_t sp_conv3x3_pp128(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, bool pool, hipStream_t s) {
  if ((w.cin != 128 && w.cin != 64) || !w.w_q || w.cout % 64 || !sp_conv3x3_pp128_fits(B, H, W, w.cin)) return hipErrorInvalidValue;
  Pp128Args a{};
  a.in = in; a.wpack = w.w_q; a.bias = w.bias; a.out = out; a.B = B; a.H = H; a.W = W; a.cout = w.cout;
  if (w.cin == 64) return pool ? launch_pp128w<true, 2>(a, s) : launch_pp128w<false, 2>(a, s);
  // SUPERSLAM_HIP_CONV128=th8 keeps the 8-row-tile kernel with register staging (A/B runs)
  static const bool th8 = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV128"); return e && std::string(e) == "th8"; }();
  if (th8) return pool ? launch_pp128<true>(a, s) : launch_pp128<false>(a, s);
  return pool ? launch_pp128w<true, 4>(a, s) : launch_pp128w<false, 4>(a, s);
}

}  // namespace sship
