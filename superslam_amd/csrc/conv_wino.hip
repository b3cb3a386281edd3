// Winograd F(2x2, 3x3) convolution for the 64 -> 64-channel layers of SuperPoint (conv2a / conv2b; utils/convert_superpoint_to_onnx.py:40-41).
// A/B kernel behind SUPERSLAM_HIP_CONV64=wino (VERDICT r03 "do this" 2).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        16 element-wise positions, each a [64 cout x 64 cin] GEMM over the 2x2 output tiles:
//   16 MFMA-MACs per 2x2 outputs and channel pair instead of 36 - 2.25x fewer matrix instructions, paid for with VALU transforms.
//
// What shapes the kernel on gfx950 (profiles/NOTES_r01_r04_design_history.md item 33):
//   * the 16 transformed weight sets are 128 KB and must be LDS-resident (streamed they would cost 1 KB of L2 traffic per output pixel):
//     32 KB are left for input, so the input tile (10 x 34 halo pixels for 8 x 32 outputs) goes through a ring of three
//     16-CHANNEL chunks (10.9 KB each, LDS-DMA, one MFMA k-step per chunk) instead of being resident whole;
//   * one (32 cout x 32 tiles) output block has 16 positions x 16 = 256 accumulator registers: a wave holds them with the 512-register
//     budget of ONE wave per SIMD.  Workgroup = 4 waves = (cout tile m) x (N-tile n: output rows 4 n .. 4 n + 3 of the 8 x 32 tile),
//     one persistent workgroup per CU;
//   * per chunk and wave: 16 pixel fragments (the lane's 4 x 4 patch, 8 channels) -> B^T d B in packed fp16 (128 v_pk_add_f16) -> 16 MFMAs
//     against 16 weight fragments; after four chunks A^T M A on the accumulators (24 v_add_f32 per register), bias, ReLU,
//     (2 x 2 max-pool = the maximum of the tile's four outputs, lane-local), fp16, 16-byte stores;
//   * LDS layout of a chunk: 16-byte unit (halo row r, column c, half u) at [2 (34 r + 17 (c & 1) + (c >> 1)) + u] ^ ((r >> 1) & 1) - even and
//     odd columns apart (a lane's tile column advances by two pixels) and the half bit flipped every other row pair: every
//     ds_read_b128 lane group hits 16 distinct slots.  The permutation is produced by the DMA's SOURCE addresses.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "kernels.h"

#ifndef SSHIP_WINO_TRACE_BUILD   // the phase trace costs registers (spills) in this 472-register kernel: compiled in on request only
#define SSHIP_WINO_TRACE_BUILD 0
#endif
#ifndef SSHIP_WINO_DBG
#define SSHIP_WINO_DBG 0
#endif
#ifndef SSHIP_WINO_NOP
#define SSHIP_WINO_NOP 0
#endif

namespace sship {

struct WinoArgs {
  const _Float16* in;     // channels-last fp16 [B, H, W, 64]
  const _Float16* upack;  // transformed weights [p 16][chunk 4][m 2][lane 64][8]
  const float* bias;      // [64]
  _Float16* out;          // [B, H, W, 64] or pooled [B, H/2, W/2, 64]
  int B, H, W;
  unsigned long long* trace;  // SSHIP_WINO_TRACE: [workgroup][wave][8] clocks of tile 2: waits + barriers, DMA issue, compute (4 chunks), epilogue
};

constexpr int W_TH = 8, W_TW = 32, W_THH = 10, W_TWH = 34;
constexpr int W_U_HALFS = 16 * 4 * 2 * 512;          // 65 536 halfs = 128 KB
constexpr int W_CHUNK_BYTES = W_THH * W_TWH * 32;    // 10 880 B: 340 pixels x 16 channels
constexpr int W_NDMA = (W_CHUNK_BYTES + 1023) / 1024;  // 11 one-KB LDS-DMA instructions per chunk
constexpr int W_SLOT_BYTES = W_CHUNK_BYTES;          // ring slots are packed: the 11th DMA instruction of a chunk runs with lanes 0..39 only
constexpr int W_RING = 3;
constexpr unsigned W_OOB = 0x80000000u;  // beyond num_records (0x7ffffff0) and does not wrap when an instruction offset (<= 2048) is added
static_assert(W_U_HALFS * 2 + W_RING * W_SLOT_BYTES <= 163840 && W_SLOT_BYTES % 16 == 0, "LDS budget");

typedef int rsrc4w_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2w_t __attribute__((ext_vector_type(2)));

template <bool POOL>
__global__ __launch_bounds__(256) void conv3x3_wino64(WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  _Float16* s_u = reinterpret_cast<_Float16*>(wsm);
  char* s_ring = wsm + W_U_HALFS * 2;  // ring slot s at s_ring + s * W_SLOT_BYTES
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hh = lane >> 5, ty = j >> 4, tx = j & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = wave & 1, n = wave >> 1;

  const int tiles_x = (p.W + W_TW - 1) / W_TW, tiles_y = (p.H + W_TH - 1) / W_TH;
  const int ntiles = p.B * tiles_x * tiles_y;
  const int t_begin = (int)((long long)blockIdx.x * ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int n_wg = t_end - t_begin;
  if (n_wg <= 0) return;

  // transformed weights: once per workgroup
  for (int u = tid; u < W_U_HALFS / 8; u += 256) *reinterpret_cast<uint4*>(s_u + u * 8) = *reinterpret_cast<const uint4*>(p.upack + u * 8);
  // bias of this lane's 16 accumulator rows: row (r & 3) + 8 (r >> 2) + 4 hh of cout tile m
  float breg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) breg[r] = p.bias[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];

  // ---- LDS-DMA of one 16-channel chunk: 11 instructions of 64 lanes x 16 B; wave w issues instructions 3 w, 3 w + 1, 3 w + 2 with ONE M0 and
  // instruction offsets 0 / 1024 / 2048 (the offset moves the global AND the LDS address: the per-lane source offsets are biased by
  // 2048 - 1024 k and the buffer base by -2048).  The first version wrote M0 per instruction: 1.5 k clocks of issue per chunk - rewriting M0
  // between LDS-DMAs serialises them (profiles/NOTES_r01_r04_design_history.md item 16) ----
  // position P = 64 i + lane (16-byte units) holds halo pixel (r, c), half u with  q = P >> 1, r = q / 34, rem = q % 34,
  // c = 2 (rem % 17) + rem / 17, u = (P & 1) ^ ((r >> 1) & 1)
  unsigned voff[3];
  unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
  {
    const int rhi_last = min(W_THH - 1, p.H - (tiles_y - 1) * W_TH), chi_last = min(W_TWH - 1, p.W - (tiles_x - 1) * W_TW);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = 3 * wave + k;
      const int P = i * 64 + lane, q = P >> 1;
      const int r = q / 34, rem = q - r * 34;
      const int c = 2 * (rem % 17) + rem / 17, u = (P & 1) ^ ((r >> 1) & 1);
      const bool real = i < W_NDMA && r < W_THH;
      voff[k] = real ? (unsigned)((r * p.W + c) * 128 + u * 16 + 2048 - 1024 * k) : W_OOB;
      m_top |= (r >= 1 ? 1u : 0u) << k;
      m_bot |= (r <= rhi_last ? 1u : 0u) << k;
      m_left |= (c >= 1 ? 1u : 0u) << k;
      m_right |= (c <= chi_last ? 1u : 0u) << k;
    }
  }
  const int n_dma = wave < 3 ? 3 : 2;  // instructions 9 and 10 are wave 3's; 11 does not exist
  rsrc4w_t rs;
  {
    // base one row and one pixel before the image: halo (0, 0) of tile (0, 0) is offset 0
    const unsigned long long ba = (unsigned long long)(uintptr_t)p.in - (unsigned long long)(p.W + 1) * 128ull - 2048ull;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
    rs[2] = 0x7ffffff0;
    rs[3] = 0x00020000;
  }
  struct Walk { int tx, ty, b; };
  auto walk_init = [&](int t) __attribute__((always_inline)) {
    Walk w;
    w.tx = t % tiles_x;
    const int r = t / tiles_x;
    w.ty = r % tiles_y; w.b = r / tiles_y;
    return w;
  };
  auto walk_next = [&](Walk& w) __attribute__((always_inline)) {
    if (++w.tx == tiles_x) { w.tx = 0; if (++w.ty == tiles_y) { w.ty = 0; ++w.b; } }
  };
  Walk dw = walk_init(t_begin), cw = dw;  // tile of the next DMA / of the computation
  int d_chunk = 0, d_slot = 0;
  auto dma_item = [&]() __attribute__((always_inline)) {  // next (tile, chunk) of the stream into ring slot d_slot
    const int y0 = dw.ty * W_TH, x0 = dw.tx * W_TW;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((dw.b * p.H + y0) * p.W + x0) * 128u + (unsigned)d_chunk * 32u);
    const bool interior = y0 >= 1 && y0 + W_TH + 1 <= p.H && x0 >= 1 && x0 + W_TW + 1 <= p.W;
    unsigned v[3];
    if (interior) {
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = voff[k];
    } else {
      unsigned ok = 7u;
      if (dw.ty == 0) ok &= m_top;
      if (dw.ty == tiles_y - 1) ok &= m_bot;
      if (dw.tx == 0) ok &= m_left;
      if (dw.tx == tiles_x - 1) ok &= m_right;
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = (ok >> k) & 1u ? voff[k] : W_OOB;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_ring + d_slot * W_SLOT_BYTES) + (unsigned)wave * 3072u);
    unsigned keep;
    if (n_dma == 3) {
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %5, %6 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %3, %5, %6 offen offset:2048 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(lds0), "s"(rs), "s"(soff) : "memory");
    } else {
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(v[0]), "s"(lds0), "s"(rs), "s"(soff) : "memory");
      // instruction 10 covers units 640 .. 703 of which 680 .. are past the chunk: lanes 40 .. 63 stay out of it (an out-of-range lane of
      // an LDS-DMA still WRITES zeros, and those 384 bytes are the head of the next ring slot)
      if (lane < 40)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(v[1]), "s"(lds0), "s"(rs), "s"(soff) : "memory");
    }
    if (++d_chunk == 4) { d_chunk = 0; walk_next(dw); }
    if (++d_slot == W_RING) d_slot = 0;
  };

  // ---- fragment addresses ----
  // pixel (a, b) of this lane's patch: unit (base ^ x_a) + 2 (34 a + 17 (b & 1) + (b >> 1)),  base = 2 (34 (4 n + 2 ty) + tx) + hh,
  // x_a = (ty + (a >> 1)) & 1  (delta is even, so the flip commutes with adding it)
  const int ubase = 2 * (34 * (4 * n + 2 * ty) + tx) + hh;
  const unsigned L0 = (unsigned)((ubase ^ (ty & 1)) * 16), L1 = (unsigned)((ubase ^ ((ty + 1) & 1)) * 16);
  const _Float16* ua = s_u + m * 512 + lane * 8;  // weight fragment (p, chunk): ua + ((p * 4 + chunk) * 2) * 512

  f16x_t acc[16];
  const f16x_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  typedef h8_t v8;
  // a - b on packed halfs: v_pk_add_f16 with the negate modifiers on the second operand, written by hand - hipcc lowers a vector fp16
  // subtraction to scalar v_sub_f16 + v_pack_b32_f16 (three instructions per two halfs) and folds fma(b, -1, a) back into that form; this
  // kernel lives on its VALU count.  (The operands come from LDS reads / earlier VALU results, never straight from an MFMA.)
  typedef unsigned u4w __attribute__((ext_vector_type(4)));
  auto vsub = [&](v8 a, v8 b) __attribute__((always_inline)) {
    const u4w x = __builtin_bit_cast(u4w, a), y = __builtin_bit_cast(u4w, b);
    u4w r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned o;
#if SSHIP_WINO_NOP
      asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3" : "=v"(o) : "v"(x[i]), "v"(y[i]));
#else
      asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(x[i]), "v"(y[i]));
#endif
      r[i] = o;
    }
    return __builtin_bit_cast(v8, r);
  };
  auto chunk_compute = [&](auto c_c, const char* slot) __attribute__((always_inline)) {
    constexpr int c = decltype(c_c)::value;
    v8 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        d[a][b] = *reinterpret_cast<const v8*>(slot + (a < 2 ? L0 : L1) + 32 * (34 * a + 17 * (b & 1) + (b >> 1)));
    // B^T d: rows
    v8 t[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      t[0][b] = vsub(d[0][b], d[2][b]);
      t[1][b] = d[1][b] + d[2][b];
      t[2][b] = vsub(d[2][b], d[1][b]);
      t[3][b] = vsub(d[1][b], d[3][b]);
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      v8 V[4];
      V[0] = vsub(t[xi][0], t[xi][2]);
      V[1] = t[xi][1] + t[xi][2];
      V[2] = vsub(t[xi][2], t[xi][1]);
      V[3] = vsub(t[xi][1], t[xi][3]);
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const int pp = xi * 4 + nu;
        const v8 A = *reinterpret_cast<const v8*>(ua + ((pp * 4 + c) * 2) * 512);
        if constexpr (c == 0) acc[pp] = mfma32(A, V[nu], zero16);
        else acc[pp] = mfma32(A, V[nu], acc[pp]);
      }
    }
  };

  // ---- epilogue: A^T M A per accumulator register, bias, ReLU, (pool), fp16, 16-byte buffer stores (masked lanes: out-of-range offset) ----
  const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
  typedef unsigned st4_t __attribute__((ext_vector_type(4)));
  auto epilogue = [&]() __attribute__((always_inline)) {
    const int y0 = cw.ty * W_TH, x0 = cw.tx * W_TW, b = cw.b;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.out) + (size_t)b * Ho * Wo * 128, 0, (int)0x7ffffff0, 0x00020000);
    // Y[i][jj] for the 16 registers
    float Y[2][2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s0[4], s1[4];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        s0[xi] = (acc[xi * 4 + 0][r] + acc[xi * 4 + 1][r]) + acc[xi * 4 + 2][r];
        s1[xi] = (acc[xi * 4 + 1][r] - acc[xi * 4 + 2][r]) - acc[xi * 4 + 3][r];
      }
      Y[0][0][r] = (s0[0] + s0[1]) + s0[2] + breg[r];
      Y[1][0][r] = (s0[1] - s0[2]) - s0[3] + breg[r];
      Y[0][1][r] = (s1[0] + s1[1]) + s1[2] + breg[r];
      Y[1][1][r] = (s1[1] - s1[2]) - s1[3] + breg[r];
    }
#if SSHIP_WINO_DBG == 1   // store path alone: a known pattern instead of the transformed accumulators
#pragma unroll
    for (int r = 0; r < 16; ++r) { Y[0][0][r] = (float)r; Y[0][1][r] = (float)(16 + r); Y[1][0][r] = (float)(32 + r); Y[1][1][r] = (float)(48 + r); }
#endif
    const h2w_t z2 = {(_Float16)0.f, (_Float16)0.f};
    auto relu2 = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {
      h2w_t v = {(_Float16)lo, (_Float16)hi};
      v = __builtin_elementwise_max(v, z2);
      return __builtin_bit_cast(unsigned, v);
    };
    auto store_units = [&](const float (&v)[16], unsigned voff_px, int so_px) __attribute__((always_inline)) {
      // registers 4 g + e = channels 8 g + 4 hh + e: lanes hh = 0 / 1 hold the halves of an 8-channel unit; permlane32_swap pairs them
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned a0 = relu2(v[4 * g + 0], v[4 * g + 1]), b0 = relu2(v[4 * g + 4], v[4 * g + 5]);
        unsigned a1 = relu2(v[4 * g + 2], v[4 * g + 3]), b1 = relu2(v[4 * g + 6], v[4 * g + 7]);
        // This kernel runs ONE wave per SIMD: dependent instructions issue back to back, with no other wave's instructions in between.
        // v_permlane32_swap reading a register the VALU wrote 2-3 instructions earlier got the OLD contents in lanes 12-15 of every
        // 16-lane row (first version: the fp32 temporaries that lived in those registers went out as "fp16" channels 2, 3, 10, 11 of tile
        // columns 12-15); hipcc inserts one wait state at most there.  The same source pattern is fine in conv_pp.hip at two waves per SIMD.
        asm volatile("s_nop 4" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        int so = so_px + (m * 32 + g * 8) * 2;
        // s_nop: v_permlane32_swap's results reach lanes 12-15 of every 16-lane row last; a buffer store two instructions behind it read
        // stale data there (first version of this kernel: channels 2, 3, 10, 11 of tile columns 12-15 wrong) - hipcc inserts no wait states
        // for that pair.  (The asm also keeps hipcc from merging the scalar offsets of these stores: see wstore in lg_kernels.hip.)
        asm volatile("s_nop 7" : "+s"(so));
        __builtin_amdgcn_raw_buffer_store_b128(st4_t{r0[0], r1[0], r0[1], r1[1]}, ro, voff_px, so, 0);
      }
    };
    if constexpr (!POOL) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int y = y0 + 4 * n + 2 * ty + i, x = x0 + 2 * tx + jj;
          const unsigned vo = (y < p.H && x < p.W) ? (unsigned)(((2 * ty) * p.W + 2 * tx) * 128 + hh * 16) : W_OOB;
          store_units(Y[i][jj], vo, ((y0 + 4 * n + i) * p.W + x0 + jj) * 128);
        }
    } else {
      float pm[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pm[r] = fmaxf(fmaxf(Y[0][0][r], Y[0][1][r]), fmaxf(Y[1][0][r], Y[1][1][r]));
      const int yo = (y0 >> 1) + 2 * n + ty, xo = (x0 >> 1) + tx;
      const unsigned vo = (yo < Ho && xo < Wo) ? (unsigned)((ty * Wo + tx) * 128 + hh * 16) : W_OOB;
      store_units(pm, vo, (((y0 >> 1) + 2 * n) * Wo + (x0 >> 1)) * 128);
    }
    walk_next(cw);
  };

  // ---- schedule: item k = (tile k / 4, chunk k % 4) in ring slot k % 3; the DMA of item k + 2 is issued at the start of item k ----
  const int nitems = 4 * n_wg;
  dma_item();
  if (nitems > 1) dma_item();
  __syncthreads();  // weights in LDS
  constexpr int S = POOL ? 2 : 8;  // stores per wave at the end of a tile
  // wait until the DMA of the item that starts now has landed: memory operations retire in order, so the operations this wave issued
  // AFTER it may stay in flight - the next item's DMA and the stores of a tile that ended one or two items ago
  auto wait_item = [&](int chunk, bool last_two) __attribute__((always_inline)) {
    if (last_two) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    const bool st = chunk < 2;  // chunk 0: the tile that just ended stored after the next item's DMA was issued; chunk 1: before it
    if (n_dma == 3) {
      if (st) { if (S == 8) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      if (st) { if (S == 8) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  };
  int slot = 0;
  auto next_slot = [&]() __attribute__((always_inline)) { const char* sp_ = s_ring + slot * W_SLOT_BYTES; if (++slot == W_RING) slot = 0; return sp_; };
#pragma unroll 1
  for (int t = 0; t < n_wg; ++t) {
    const int k0 = 4 * t;
    const bool first_tile = t == 0;
    const bool tr = SSHIP_WINO_TRACE_BUILD && p.trace && t == 2 && lane == 0;
    unsigned long long ts = 0, acc_wait = 0, acc_dma = 0, acc_cmp = 0;
    auto item = [&](auto c_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value;
      const int k = k0 + c;
      if (tr) ts = __builtin_readcyclecounter();
      // the very first tile has no stores behind it: its chunk-0 / chunk-1 waits must not count them
      if (first_tile && c < 2) { if (k + 2 < nitems + 1) { if (n_dma == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else wait_item(c, k + 2 > nitems - 1 + 1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave's share of item k has landed; every wave is done with item k - 1
      if (tr) { const unsigned long long n_ = __builtin_readcyclecounter(); acc_wait += n_ - ts; ts = n_; }
      if (k + 2 < nitems) dma_item();
      if (tr) { const unsigned long long n_ = __builtin_readcyclecounter(); acc_dma += n_ - ts; ts = n_; }
      chunk_compute(c_c, next_slot());
      if (tr) { const unsigned long long n_ = __builtin_readcyclecounter(); acc_cmp += n_ - ts; ts = n_; }
    };
    item(std::integral_constant<int, 0>{});
    item(std::integral_constant<int, 1>{});
    item(std::integral_constant<int, 2>{});
    item(std::integral_constant<int, 3>{});
    epilogue();
    if (tr) {
      unsigned long long* o = p.trace + ((size_t)blockIdx.x * 4 + wave) * 8;
      o[0] = acc_wait; o[1] = acc_dma; o[2] = acc_cmp; o[3] = __builtin_readcyclecounter() - ts;
    }
  }
}

template <bool POOL>
static hipError_t launch_wino(const WinoArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)W_U_HALFS * 2 + (size_t)W_RING * W_SLOT_BYTES;
  static_assert(smem <= 163840, "LDS budget");
  auto kern = conv3x3_wino64<POOL>;
  static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_rc != hipSuccess) return attr_rc;
  const int ntiles = a.B * ((a.W + W_TW - 1) / W_TW) * ((a.H + W_TH - 1) / W_TH);
  const int gx = ntiles < cu_count() ? ntiles : cu_count();
  static const bool trace_on = SSHIP_WINO_TRACE_BUILD && dev_env("SSHIP_WINO_TRACE") != nullptr;
  static unsigned long long* tbuf = nullptr;
  WinoArgs b = a;
  if (trace_on) {
    if (!tbuf) (void)hipMalloc(&tbuf, 1024 * 4 * 8 * 8);
    (void)hipMemsetAsync(tbuf, 0, 1024 * 4 * 8 * 8, s);
    b.trace = tbuf;
  }
  hipLaunchKernelGGL(kern, dim3(gx), dim3(256), smem, s, b);
  if (trace_on) {
    std::vector<unsigned long long> h(1024 * 4 * 8);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[4] = {0}; long cnt = 0;
    for (int i = 0; i < gx * 4; ++i) {
      if (!h[i * 8 + 2]) continue;
      for (int k = 0; k < 4; ++k) sum[k] += (double)h[i * 8 + k];
      ++cnt;
    }
    if (cnt) fprintf(stderr, "[wino trace pool=%d] per tile and wave: waits + barriers=%.0f DMA issue=%.0f compute (4 chunks)=%.0f epilogue=%.0f | tile=%.0f clk (%ld waves)\n",
                     (int)POOL, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, (sum[0] + sum[1] + sum[2] + sum[3]) / cnt, cnt);
  }
  return hipGetLastError();
}

bool sp_conv3x3_wino_fits(int H, int W, int cin, int cout) {
  return cin == 64 && cout == 64 && (size_t)H * W * 128 < 0x7f000000ull && H >= 2 && W >= 2;
}

hipError_t sp_conv3x3_wino(const _Float16* upack, const float* bias, const _Float16* in, _Float16* out, int B, int H, int W, bool pool,
                           hipStream_t s) {
  WinoArgs a{};
  a.in = in; a.upack = upack; a.bias = bias; a.out = out; a.B = B; a.H = H; a.W = W;
  return pool ? launch_wino<true>(a, s) : launch_wino<false>(a, s);
}

}  // namespace sship
