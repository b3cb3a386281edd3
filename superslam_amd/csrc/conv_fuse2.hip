// conv2a -> conv2b -> 2x2 max-pool of SuperPoint (utils/convert_superpoint_to_onnx.py:40-43, 53-55) as ONE kernel: the 64-channel half-resolution
// map between the two layers never leaves the CU.
//
// Why (profiles/r06_d_conv_store_energy.txt): on the 1 400 W cap a launch's time is its joules, and 34 % of conv2a's joules are its OUTPUT STORES
// (2.1 GB per 128-image launch at 120-270 pJ per byte written; the store pattern, the store order and cache blocking do not move it).  conv2b then
// reads the same 2.1 GB back.  Fusing the pair trades that round trip for a one-pixel halo of recomputed conv2a columns.
//
// How: a ROLLING WINDOW down a 30-column strip instead of 2-D tiles, so that only the horizontal halo is recomputed (32 / 30 on conv2a, two idle
// lanes of 32 on conv2b; a tile scheme at 8 x 32 would recompute 33 %), and both layers' intermediate state is a ring of 12 rows each:
//   * a workgroup is 8 waves = two per SIMD: waves 0-3 are conv2a (PRODUCERS), waves 4-7 conv2b (CONSUMERS).  Wave (m, rp) of a role owns the M-tile m
//     (32 output channels) of the row pair rp of a 4-row step: two 32-pixel N-tiles, 2 x 36 MFMAs 32x32x16 per step.
//   * the WEIGHTS live in registers: a wave only ever needs its layer's M-tile m = 36 A fragments = 144 VGPRs, loaded once per launch (the kernel
//     is persistent).  That frees the 144 KiB of LDS the two layers' weights would need, which is what made the fusion impossible as a tile kernel
//     (conv_pp.hip holds ONE layer's 72 KiB next to two 43-KiB tiles), and removes the A-fragment LDS reads: one ds_read_b128 per MFMA is left.
//   * slot t of the pipeline: the producers compute intermediate rows 4t-1 .. 4t+2 of the strip from the INPUT ring into the MID ring, the consumers
//     compute output rows 4(t-2) .. 4(t-2)+3 from the MID ring (rows produced in slots <= t-1), pool them and store; all waves issue the LDS-DMA
//     (buffer_load_dwordx4 ... lds) of the four input rows of slot t+2.  One workgroup barrier per slot; the two roles on a SIMD are both MFMA streams
//     whose fragment reads, epilogues and stores ride under the other's MFMAs.
//   * ring rows: 34 pixels x 144 B (64 channels + 16 B pad: conflict-free ds_read_b128 with a PLAIN layout, so a fragment address is one per-row
//     VGPR base plus an immediate) in 5 120 B = 5 DMA instructions; out-of-image pixels get their voffset pushed out of the buffer's range and the DMA
//     writes zeros (scripts/ubench/lds_dma_oob.hip).  conv2b's zero padding is on conv2a's OUTPUT: the producers write zeros for intermediate pixels
//     outside the image.
//   * same fp16 operands, same k order per accumulator ((kx, k-step, ky), accumulator started from the bias: common.h SSHIP_K_ROWSHARE), same rounding points as
//     conv3x3_pp<64, 64, ...>: the output is BIT-IDENTICAL to the two-launch path (tests/test_gpu_alt_paths.py), which the library keeps for shapes
//     the kernel does not take and the developer build for A/B runs.
//   * a strip is cut into row segments (2 conv2a rows recomputed and a 2-slot pipeline fill per segment) until there are ~2 units per CU: a one-pair call
//     (46 strips) runs 506 segments of 4-5 steps - 0.736 -> 0.710 ms per pair; 128 images run 2 segments per strip = 23 units per CU.
// LDS: 2 x 12 x 5 120 B rings + 512 B bias = 123.4 KB, one workgroup per CU.
#include "common.h"
#include "kernels.h"

namespace sship {

struct F2Args {
  const _Float16* in;    // channels-last fp16 [B,H,W,64]
  const _Float16* wa;    // conv2a packed weights [tap][kstep][mt 2][lane][8] (igemm.h packing, ct = 64)
  const float* ba;
  const _Float16* wb;    // conv2b
  const float* bb;
  _Float16* out;         // [B,H/2,W/2,64]
  int B, H, W;
  int nstrips, nseg, H4; // strips per image row, row segments per strip, 4-row steps per strip
  int nunits;            // B * nstrips * nseg
};

constexpr int F_TW = 30;                  // output columns per strip
constexpr int F_PXB = 144;                // bytes per ring pixel
constexpr int F_ROWB = 5120;              // bytes per ring row
constexpr int F_RING = 12;                // rows per ring
constexpr int F_RINGB = F_RING * F_ROWB;  // 61 440
constexpr unsigned F_OOB = 0x80000000u;   // beyond num_records, and no 32-bit wrap with soffset on top
constexpr size_t F_SMEM = 2 * F_RINGB + 2 * 64 * 4;

struct F2Walk { int u, k, S, b, s, r0; };  // wave-uniform: unit, step inside the unit (0 .. S), steps of the unit, image, strip, first row

// SINGLE = false: the fused pair above (rocprofv3 shows it as conv_roll<false>; the kernel was named conv2ab_fused before it became a template).  SINGLE = true (conv3a_roll): ONE 64 -> 128 layer (conv3a) on the same machinery -
// all 8 waves are "producers" (wave = M-tile m of 4 x row pair rp of 2) that read the INPUT ring and store their rows to global memory; strips are
// 32 columns wide (nothing is recomputed: there is no second layer), the halo is one pixel.  Against conv3x3_pp<64, 64> with its two cout tiles: the
// input is fetched once instead of twice, the weights come from registers instead of LDS, and the row pair 1 waves write their rows at the START of
// their next slot, under the row pair 0 waves' MFMAs.  Same k order, same rounding points: bit-identical - and the same joules (profiles/r06_l_*), so only
// the developer build instantiates it (A/B).
template <bool SINGLE>
__global__ __launch_bounds__(512, 2) void conv_roll(F2Args p) {
  constexpr int TW = SINGLE ? 32 : F_TW, HALO = SINGLE ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char f2_smem[];
  char* s_in = f2_smem;
  char* s_mid = f2_smem + F_RINGB;
  float* s_bias = reinterpret_cast<float*>(f2_smem + 2 * F_RINGB);  // [role][64] (SINGLE: [128])
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = SINGLE ? 0 : wave >> 2, gw = wave & 3, m = SINGLE ? gw : gw & 1, rp = SINGLE ? wave >> 2 : gw >> 1;
  const int Ho = p.H >> 1, Wo = p.W >> 1;

  if (tid < 128) s_bias[tid] = (SINGLE || tid < 64) ? p.ba[tid] : p.bb[tid - 64];
  // this wave's 36 A fragments (M-tile m of its layer), for the whole launch
  h8_t wreg[36];
  {
    const _Float16* w = (SINGLE ? p.wa + (m >> 1) * 36864 + (m & 1) * 512 : (role ? p.wb : p.wa) + m * 512) + lane * 8;  // SINGLE: cout tile m >> 1 of the ct = 64 packing
#pragma unroll
    for (int idx = 0; idx < 36; ++idx) wreg[idx] = *reinterpret_cast<const h8_t*>(w + idx * 1024);
  }

  const int u_begin = (int)((long long)blockIdx.x * p.nunits / gridDim.x), u_end = (int)((long long)(blockIdx.x + 1) * p.nunits / gridDim.x);
  if (u_begin >= u_end) return;
  auto decode = [&](F2Walk& w) __attribute__((always_inline)) {
    const int seg = w.u % p.nseg, t = w.u / p.nseg;
    w.s = t % p.nstrips; w.b = t / p.nstrips;
    const int k0 = seg * p.H4 / p.nseg, k1 = (seg + 1) * p.H4 / p.nseg;
    w.r0 = 4 * k0; w.S = k1 - k0; w.k = 0;
  };
  auto advance = [&](F2Walk& w) __attribute__((always_inline)) {
    if (++w.k > w.S) {
      ++w.u;
      if (w.u < u_end) decode(w);
    }
  };
  int Q = 0;  // pipeline steps of this workgroup: S + 1 per unit
  {
    F2Walk w; w.u = u_begin;
    for (; w.u < u_end; ++w.u) { decode(w); Q += w.S + 1; }
  }

  // ---------------- LDS-DMA of one load step: input rows r0 - 2 + 4k + i (i = 0..3), columns 30 s - 2 .. 30 s + 31, into ring rows (4 q + i) % 12 ----------------
  // instruction n = 5 i + c (c = 0..4) of a step writes LDS units 64 c .. 64 c + 63 of ring row i; wave w issues n = w, w + 8, w + 16 (< 20).
  // unit p = 64 c + lane of a row is pixel p / 9, 16-byte channel unit p % 9 (unit 8 = the pad, units >= 306 = the row's tail: zeros)
  int d_i[3], d_c[3], d_px[3];
  unsigned d_voff[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int n = wave + 8 * e;
    d_i[e] = n / 5; d_c[e] = n - 5 * d_i[e];
    const int pu = d_c[e] * 64 + lane, px = pu / 9, un = pu - 9 * px;
    d_px[e] = (pu < 306 && un < 8) ? px : 1000;  // 1000: never inside the column window
    d_voff[e] = (unsigned)((d_i[e] * p.W + px) * 128 + un * 16);
  }
  const int n_dma = wave < 4 ? 3 : 2;
  typedef int rsrc4_t __attribute__((ext_vector_type(4)));
  auto dma_step = [&](const F2Walk& w, int q3) __attribute__((always_inline)) {
    // buffer of image b, based two rows and two pixels before it: offsets stay non-negative and 32-bit
    const unsigned long long ba = (unsigned long long)(uintptr_t)p.in + ((unsigned long long)w.b * p.H * p.W - (unsigned long long)(HALO * p.W + HALO)) * 128ull;
    rsrc4_t rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
    rs[2] = 0x7ffffff0;
    rs[3] = 0x00020000;
    const int row0 = w.r0 + 4 * w.k;  // image row of i = 0 is row0 - HALO
    const unsigned soff = (unsigned)((row0 * p.W + TW * w.s) * 128);
    const int pxlo = w.s == 0 ? HALO : 0, pxhi = min(34, p.W - TW * w.s + HALO);  // ring pixel px is image column TW s - HALO + px
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      if (e >= n_dma) break;
      const int ri = row0 - HALO + d_i[e];
      const bool rowok = ri >= 0 && ri < p.H;
      const unsigned v = (rowok && d_px[e] >= pxlo && d_px[e] < pxhi) ? d_voff[e] : F_OOB;
      int rr = 4 * q3 + d_i[e];
      rr = rr >= F_RING ? rr - F_RING : rr;
      const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(s_in + rr * F_ROWB + d_c[e] * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(v), "s"(m0v), "s"(rs), "s"(soff) : "memory");
    }
  };

  // ---------------- one 4-row step of this wave's layer: 2 N-tiles (rows 2 rp, 2 rp + 1 of the step) x 36 k-steps ----------------
  const unsigned lane_b = (unsigned)(j * F_PXB + hh * 16);
  f16x_t acc[2];
  auto mfma_step = [&](int q3) __attribute__((always_inline)) {
    const char* ring = role ? s_mid : s_in;
    // ring rows 4 q3 + 2 rp + d (d = 0..3) hold the rows this wave's two N-tiles read (tap row ky of N-tile n: d = n + ky)
    const char* rb[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      int rr = 4 * q3 + 2 * rp + d;
      rr = rr >= F_RING ? rr - F_RING : rr;
      rb[d] = ring + rr * F_ROWB + lane_b;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(s_bias + (SINGLE ? 0 : role * 64) + m * 32 + hh * 4 + g * 8);
        acc[n][4 * g + 0] = b4.x; acc[n][4 * g + 1] = b4.y; acc[n][4 * g + 2] = b4.z; acc[n][4 * g + 3] = b4.w;
      }
    if constexpr (SSHIP_K_ROWSHARE != 0) {
      // (kx, k-step, ky) order, as conv3x3_pp<64, ...>: group g = 4 kx + ks reads its four row fragments once (ring of 6, fragment r = 4 g + d) for the
      // six MFMAs of its three tap rows - 2/3 of a ds_read_b128 per MFMA; the A operand of sub-step t = 3 g + ky is register fragment (3 ky + kx) 4 + ks
      h8_t rbq[6];
      auto loads_for = [&](int t) __attribute__((always_inline)) {
        const int g = t / 3, ky = t - 3 * g, kx = g >> 2, ks = g & 3;
        if (ky == 0) {
          rbq[(4 * g) % 6] = *reinterpret_cast<const h8_t*>(rb[0] + kx * F_PXB + ks * 32);
          rbq[(4 * g + 1) % 6] = *reinterpret_cast<const h8_t*>(rb[1] + kx * F_PXB + ks * 32);
        } else {
          rbq[(4 * g + ky + 1) % 6] = *reinterpret_cast<const h8_t*>(rb[ky + 1] + kx * F_PXB + ks * 32);
        }
      };
      loads_for(0); loads_for(1);
#pragma unroll
      for (int t = 0; t < 36; ++t) {
        if (t + 2 < 36) loads_for(t + 2);
        __builtin_amdgcn_sched_barrier(0);
        const int g = t / 3, ky = t - 3 * g, kx = g >> 2, ks = g & 3;
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[n] = mfma32(wreg[((ky * 3 + kx) << 2) + ks], rbq[(4 * g + ky + n) % 6], acc[n]);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    constexpr int NBUF = 3;
    h8_t fb[NBUF][2];
    auto load_frags = [&](int idx, int buf) __attribute__((always_inline)) {
      const int tap = idx >> 2, ky = tap / 3, kx = tap - 3 * ky, ks = idx & 3;
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[buf][n] = *reinterpret_cast<const h8_t*>(rb[n + ky] + kx * F_PXB + ks * 32);
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load_frags(i, i);
#pragma unroll
    for (int idx = 0; idx < 36; ++idx) {
      if (idx + NBUF - 1 < 36) load_frags(idx + NBUF - 1, (idx + NBUF - 1) % NBUF);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[n] = mfma32(wreg[idx], fb[idx % NBUF][n], acc[n]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // conv2a epilogue: ReLU -> fp16 -> MID ring row (4 q3 + 2 rp + n) % 12, pixel j (image column 30 s - 1 + j), channels 32 m + 8 g + 4 hh .. + 3
  auto epi_producer = [&](const F2Walk& w, int q3) __attribute__((always_inline)) {
    const int col = F_TW * w.s - 1 + j;
    const bool colok = col >= 0 && col < p.W;
    const h4_t z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int ri = w.r0 - 1 + 4 * w.k + 2 * rp + n;
      const bool ok = colok && ri >= 0 && ri < p.H;
      int rr = 4 * q3 + 2 * rp + n;
      rr = rr >= F_RING ? rr - F_RING : rr;
      char* dst = s_mid + rr * F_ROWB + j * F_PXB + m * 64 + hh * 8;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4_t o = __builtin_elementwise_max(to_h4(acc[n][4 * g], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]), z4);
        if (!ok) o = z4;
        *reinterpret_cast<h4_t*>(dst + g * 16) = o;
      }
    }
  };
  // conv2b epilogue: ReLU + 2x2 max-pool of this wave's row pair -> fp16 -> global, one 16-byte store per (even pixel, 8-channel unit).
  // It runs at the START of the consumers' next slot (the accumulators and the store address are carried over): the producers are then in their
  // MFMA loop and the consumers' stores have the whole slot to be acknowledged; the producers' epilogue closes THEIR slot while the consumers are
  // still in their MFMA loop.  In lock-step (both epilogues at the end of the slot) the matrix pipe idled for both.
  _Float16* pend_pix = nullptr;
  bool pend_ok = false;
  auto prep_consumer = [&](const F2Walk& w) __attribute__((always_inline)) {
    const int yo = (w.r0 + 4 * w.k + 2 * rp) >> 1, xo = (F_TW / 2) * w.s + (j >> 1);
    pend_ok = yo < Ho && !(j & 1) && j < F_TW && xo < Wo;
    pend_pix = p.out + ((size_t)(w.b * Ho + yo) * Wo + xo) * 64 + m * 32 + hh * 8;
  };
  auto epi_consumer = [&]() __attribute__((always_inline)) {
    auto pool1 = [&](int r) __attribute__((always_inline)) -> float {  // max over the two rows and 0 (v_max3), then the column pair (dpp quad_perm [1,0,3,2])
      const float tt = fmaxf(fmaxf(acc[0][r], acc[1][r]), 0.f);
      const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0xB1, 0xF, 0xF, false));
      return fmaxf(tt, nb);
    };
    auto pack2 = [](float lo, float hi) __attribute__((always_inline)) -> unsigned {
      const h2_t v = {(_Float16)lo, (_Float16)hi};
      return *reinterpret_cast<const unsigned*>(&v);
    };
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      // lanes hh = 0 / 1 hold channels 4 hh .. + 3 of an 8-channel unit: permlane32_swap pairs them, lane hh stores unit g + hh
      const auto r0 = __builtin_amdgcn_permlane32_swap(pack2(pool1(4 * g + 0), pool1(4 * g + 1)), pack2(pool1(4 * g + 4), pool1(4 * g + 5)), false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(pack2(pool1(4 * g + 2), pool1(4 * g + 3)), pack2(pool1(4 * g + 6), pool1(4 * g + 7)), false, false);
      if (pend_ok) *reinterpret_cast<uint4*>(pend_pix + g * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
  };

  // SINGLE: ReLU -> fp16 -> this wave's two output rows (image rows r0 + 4k + 2 rp + n, column 32 s + j, channels 32 m + 8 (g + hh) .. + 7), 16-byte stores
  bool pend_ok1 = false;
  auto prep_single = [&](const F2Walk& w) __attribute__((always_inline)) {
    const int ro = w.r0 + 4 * w.k + 2 * rp, col = TW * w.s + j;
    pend_ok = col < p.W && ro < p.H;
    pend_ok1 = col < p.W && ro + 1 < p.H;
    pend_pix = p.out + ((size_t)(w.b * p.H + ro) * p.W + col) * 128 + m * 32 + hh * 8;
  };
  auto epi_single = [&]() __attribute__((always_inline)) {
    const h2_t z2 = {(_Float16)0.f, (_Float16)0.f};
    auto relu2 = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {
      h2_t v = {(_Float16)lo, (_Float16)hi};
      v = __builtin_elementwise_max(v, z2);
      return *reinterpret_cast<const unsigned*>(&v);
    };
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const f16x_t& a = acc[n];
        const auto r0 = __builtin_amdgcn_permlane32_swap(relu2(a[4 * g + 0], a[4 * g + 1]), relu2(a[4 * g + 4], a[4 * g + 5]), false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(relu2(a[4 * g + 2], a[4 * g + 3]), relu2(a[4 * g + 6], a[4 * g + 7]), false, false);
        if (n ? pend_ok1 : pend_ok) *reinterpret_cast<uint4*>(pend_pix + (size_t)n * p.W * 128 + g * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      }
  };

  // ---------------- schedule: slot t = DMA of load step t + 2 | producer step t | consumer step t - 2, one barrier ----------------
  F2Walk wd, wr;
  wd.u = u_begin; decode(wd);
  wr = wd;
  int qd3 = 0;  // (DMA step index) % 3
  dma_step(wd, qd3); advance(wd); qd3 = 1;
  if (wd.u < u_end) { dma_step(wd, qd3); advance(wd); }
  qd3 = 2;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int lag = (!SINGLE && role) ? 2 : 0;
  int q3 = 0;  // (this role's step index) % 3
  bool pending = false;  // consumers: the previous step's accumulators wait for their epilogue
#pragma unroll 1
  for (int t = 0; t < Q + (SINGLE ? 0 : 2); ++t) {
    if (wd.u < u_end) { dma_step(wd, qd3); advance(wd); }
    qd3 = qd3 == 2 ? 0 : qd3 + 1;
    if (pending) { if constexpr (SINGLE) epi_single(); else epi_consumer(); pending = false; }
    const int q = t - lag;
    if (q >= 0 && q < Q) {
      // producers: the last step of a unit (k = S) only makes rows r1 - 1, r1 (row pair 0); consumers (and SINGLE, whose k = S step only exists for
      // its load of the halo row below the segment): it is empty
      const bool active = (!SINGLE && role == 0) ? (wr.k < wr.S || rp == 0) : wr.k < wr.S;
      if (active) {
        mfma_step(q3);
        if constexpr (SINGLE) {
          prep_single(wr);
          if (rp == 0) epi_single(); else pending = true;  // row pair 1 stores under row pair 0's next MFMA loop
        } else {
          if (role == 0) epi_producer(wr, q3);
          else { prep_consumer(wr); pending = true; }
        }
      }
      advance(wr);
      q3 = q3 == 2 ? 0 : q3 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (pending) { if constexpr (SINGLE) epi_single(); else epi_consumer(); }
}

// row segments per strip: enough units for ~2 per CU (a pair of frames is 46 strips: cut into 11 segments of 4-5 steps it is 506 units), at least 4
// steps per segment (every segment recomputes 2 conv2a rows and fills a 2-slot pipeline), at least 2 for tall maps (23 units per CU at 128 images).
// A function of the shape only - and segmentation never changes the arithmetic: every split gives the same bits.
static int f2_nseg(int B, int nstrips, int H4) {
#if SSHIP_DEV_SWITCHES
  static const int forced = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV2_NSEG"); return e ? atoi(e) : 0; }();
  if (forced > 0) return forced > H4 ? H4 : forced;
#endif
  const long base = (long)B * nstrips;
  int want = (int)((2l * cu_count() + base - 1) / base);
  const int lo = H4 >= 16 ? 2 : 1, hi = H4 / 4 > 1 ? H4 / 4 : 1;
  if (want < lo) want = lo;
  if (want > hi) want = hi;
  return want;
}
bool sp_conv2ab_fused_fits(int B, int H, int W, bool any_batch) {
  if (H < 8 || W < 8 || B < 1) return false;
  if ((size_t)H * W * 128 >= 0x7f000000ull) return false;  // offsets inside one image are 32-bit
  const int nstrips = (W + F_TW - 1) / F_TW, H4 = (H + 3) / 4;
  // any_batch = false: only batches that give every CU several strip segments without cutting the strips further (round 6's first rule; the library
  // now fuses at every batch size - f2_nseg cuts the strips into as many row segments as it takes - and asks with any_batch = true)
  return any_batch || (long long)B * nstrips * (H4 >= 16 ? 2 : 1) >= 4ll * cu_count();
}

hipError_t sp_conv2ab_fused(const ConvW& wa, const ConvW& wb, const _Float16* in, _Float16* out, int B, int H, int W, hipStream_t s) {
  if (wa.cin != 64 || wa.cout != 64 || wa.ct != 64 || wb.cin != 64 || wb.cout != 64 || wb.ct != 64) return hipErrorInvalidValue;
  F2Args a{};
  a.in = in; a.wa = wa.w; a.ba = wa.bias; a.wb = wb.w; a.bb = wb.bias; a.out = out; a.B = B; a.H = H; a.W = W;
  a.nstrips = (W + F_TW - 1) / F_TW; a.H4 = (H + 3) / 4; a.nseg = f2_nseg(B, a.nstrips, a.H4);
  a.nunits = B * a.nstrips * a.nseg;
  static const hipError_t attr_rc =
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_roll<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_SMEM);
  if (attr_rc != hipSuccess) return attr_rc;
  int gx = cu_count();
  if (gx > a.nunits) gx = a.nunits;
  hipLaunchKernelGGL(conv_roll<false>, dim3(gx), dim3(512), F_SMEM, s, a);
  return hipGetLastError();
}

#if SSHIP_DEV_SWITCHES
// conv3a (64 -> 128, no pool) on the rolling-window kernel: one launch computes all 128 output channels (conv3x3_pp<64, 64> needs two cout tiles,
// i.e. stages the input twice).  Bit-identical to it - and no cheaper: 0.7274 against 0.7276 J per 128-image launch, 521 us both (profiles/r06_l_*):
// conv3a's joules are its MFMAs and its 1.06 GB of output stores, which this form does not change.  Developer build only (SUPERSLAM_HIP_CONV3A=roll).
bool sp_conv3a_roll_fits(int B, int H, int W) {
  return H >= 8 && W >= 8 && B >= 1 && (size_t)H * W * 128 < 0x7f000000ull;  // input offsets inside one image are 32-bit (the output is addressed by pointers)
}
hipError_t sp_conv3a_roll(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, hipStream_t s) {
  if (w.cin != 64 || w.cout != 128 || w.ct != 64) return hipErrorInvalidValue;
  F2Args a{};
  a.in = in; a.wa = w.w; a.ba = w.bias; a.wb = w.w; a.bb = w.bias; a.out = out; a.B = B; a.H = H; a.W = W;
  a.nstrips = (W + 31) / 32; a.H4 = (H + 3) / 4; a.nseg = f2_nseg(B, a.nstrips, a.H4);
  a.nunits = B * a.nstrips * a.nseg;
  static const hipError_t attr_rc =
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_roll<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_SMEM);
  if (attr_rc != hipSuccess) return attr_rc;
  int gx = cu_count();
  if (gx > a.nunits) gx = a.nunits;
  hipLaunchKernelGGL(conv_roll<true>, dim3(gx), dim3(512), F_SMEM, s, a);
  return hipGetLastError();
}
#endif

}  // namespace sship
