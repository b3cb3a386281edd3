// SuperPoint non-GEMM kernels: conv1a (u8 -> 64 ch), heatmap softmax + 9x9 NMS + threshold + candidate
// compaction, radix-select/bitonic top-k, descriptor gather, dense-grid export.  gfx950 only.
#include "kernels.h"

namespace sship {

// ---------------------------------------------------------------------------------------------------
// conv1a: u8 image -> relu(conv3x3(img/255)) , 64 channels, channels-last fp16.
// reference: utils/convert_superpoint_to_onnx.py:38,53 + preprocess src/SuperPoint.cc:768-780.
// Memory-bound (writes 128 B/pixel).  A thread owns 4 consecutive pixels x 8 channels; 8 adjacent lanes
// cover the 64 channels of a pixel so every store instruction writes 1 KiB contiguous.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv1a(const uint8_t* __restrict__ img, const float* __restrict__ w,
                                                const float* __restrict__ bias, _Float16* __restrict__ out,
                                                int B, int H, int W) {
  __shared__ float s_w[9 * 64];  // [tap][cout]
  for (int i = threadIdx.x; i < 576; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int grp = threadIdx.x & 7;
  const int q = blockIdx.x * 32 + (threadIdx.x >> 3);  // quad index over B*H*ceil(W/4)
  const int qw = (W + 3) >> 2;
  if (q >= B * H * qw) return;
  const int xq = q % qw, y = (q / qw) % H, b = q / (qw * H);
  const int x0 = xq * 4;
  float in[3][6];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 6; ++dx) {
      const int yy = y + dy - 1, xx = x0 + dx - 1;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        // OpenCV 8u->32f convertTo: one fp32 multiply; the fp16 engine then sees the fp16-rounded value.
        v = (float)(_Float16)((float)img[((size_t)b * H + yy) * W + xx] * (1.0f / 255.0f));
      }
      in[dy][dx] = v;
    }
  float wr[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) wr[t][c] = s_w[t * 64 + grp * 8 + c];
  float bb[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bb[c] = bias[grp * 8 + c];
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    if (x0 + px >= W) break;
    float a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = fmaf(in[dy][px + dx], wr[dy * 3 + dx][c], a[c]);
    h8_t o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (_Float16)fmaxf(a[c] + bb[c], 0.f);
    *reinterpret_cast<h8_t*>(out + (((size_t)b * H + y) * W + x0 + px) * 64 + grp * 8) = o;
  }
}

void launch_conv1a(const uint8_t* img, const float* w, const float* bias, _Float16* out, int B, int H, int W,
                   hipStream_t s) {
  const int quads = B * H * ((W + 3) / 4);
  hipLaunchKernelGGL(k_conv1a, dim3((quads + 31) / 32), dim3(256), 0, s, img, w, bias, out, B, H, W);
}

// ---------------------------------------------------------------------------------------------------
// Heatmap tile kernel: softmax(65) -> drop dustbin -> depth-to-space -> (2R+1)^2 max-pool NMS ->
// threshold / border -> candidate compaction.   reference: convert_superpoint_to_onnx.py:77-87 (in-graph)
// + src/SuperPoint.cc:696-702 (host scan).  HBM-bound: reads 65 logits per cell once (+ halo cells),
// writes only the surviving candidates (and, on request, the dense post-NMS map for sship_sp_dense).
//
// Tile = 32 x 64 pixels (4 x 8 cells) + an 8-pixel (1-cell) halo; radius <= 8.
// Two loaders share the NMS code: from logits (production) and from a score map (sship_nms / stage tests).
// Candidate key = (score bits << 32) | (h*W + w): descending key order == std::greater<pair<float,
// pair<int,int>>> (SuperPoint.cc:703) because scores are positive floats.
// ---------------------------------------------------------------------------------------------------
constexpr int NT_H = 32, NT_W = 64, NHALO = 8, NLH = NT_H + 2 * NHALO, NLW = NT_W + 2 * NHALO, NLS = NLW + 4;
constexpr int NMS_SLOTS = NT_H * NT_W / 256;  // pixels (candidate slots) per thread

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// 4 sliding 9-wide maxima over 12 consecutive values (radius 4): 5 shared + 3 private max ops each.
__device__ __forceinline__ void window9x4(const float (&v)[12], float (&o)[4]) {
  const float c = fmaxf(max3(v[3], v[4], v[5]), max3(v[6], v[7], v[8]));
  o[0] = fmaxf(c, max3(v[0], v[1], v[2]));
  o[1] = fmaxf(c, max3(v[1], v[2], v[9]));
  o[2] = fmaxf(c, max3(v[2], v[9], v[10]));
  o[3] = fmaxf(c, max3(v[9], v[10], v[11]));
}

// RT = 4: the exporter's radius, register-window separable max with 16-byte LDS reads (the dynamic-radius loops
// were LDS-latency-bound: ~20 us per workgroup).  RT = -1: any radius <= 8 (sship_nms stage API).
//
// The workgroup is persistent over tiles, and the two long latencies of a tile are taken off its critical path (phase
// trace: HBM load 4.6k of 15.4k clocks, slot reservation 5.4k):
//   * the logits of the NEXT tile are requested into registers (17 per lane) before the LDS phases of this one;
//   * the global atomic that reserves the tile's slots in the per-image candidate list is issued and NOT waited for:
//     the tile's candidates are parked in a small LDS list and written out one tile later, when the ticket has long
//     returned (tiles with more than NMS_PEND candidates - constant plateaus - take the synchronous path).
constexpr int NMS_PEND = 256;
#ifndef SSHIP_NMS_XCD
#define SSHIP_NMS_XCD 1  // XCD-aware tile order (0: tile = id + k * grid, rounds 1-5; A/B builds)
#endif

template <int LOADER, int RT>
__global__ __launch_bounds__(256) void k_nms_tile(NmsArgs a) {
  // s_s: scores incl. halo [48][84]; s_r: row maxima [48][64]; the candidate list aliases both once they are dead
  __shared__ __attribute__((aligned(16))) float s_buf[NLH * NLS + NLH * NT_W];
  __shared__ unsigned long long s_pend[NMS_PEND];
  __shared__ int s_cnt, s_base;
  float* s_s = s_buf;
  float* s_r = s_buf + NLH * NLS;
  unsigned long long* s_c = reinterpret_cast<unsigned long long*>(s_buf);
  static_assert(sizeof(s_buf) >= NT_H * NT_W * 8, "candidate list must fit the aliased buffers");
  const int tiles_x = (a.W + NT_W - 1) / NT_W, tiles_y = (a.H + NT_H - 1) / NT_H;
  const int ntiles = a.B * tiles_x * tiles_y;
  const int tid = threadIdx.x, lane = tid & 63;
  // 60 cells (6 x 10 incl. the halo ring) x 4 lanes: each lane owns 16 of the 64 position channels (= 2 rows of
  // the cell's 8 x 8 block), the group of 4 shares max / sum through two DPP exchanges.
  constexpr int NCELL = (NLH / 8) * (NLW / 8);
  const int cell = tid >> 2, qd = tid & 3;
  const int cyl = cell / (NLW / 8), cxl = cell % (NLW / 8);
  float4 pv[4];
  float pd = 0.f;
  bool pin = false;
  auto fetch = [&](int tt) {
    if constexpr (LOADER == 0) {
      const int Hc = a.H >> 3, Wc = a.W >> 3;
      int q = tt;
      const int ftx = q % tiles_x; q /= tiles_x;
      const int fty = q % tiles_y;
      const int fb = q / tiles_y;
      const int cy = ((fty * NT_H - NHALO) >> 3) + cyl, cx = ((ftx * NT_W - NHALO) >> 3) + cxl;  // may be -1
      pin = tid < NCELL * 4 && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
      const int ccy = min(max(cy, 0), Hc - 1), ccx = min(max(cx, 0), Wc - 1);  // clamped: branch-free loads
      const float* lp = a.logits + ((size_t)(fb * Hc + ccy) * Wc + ccx) * a.ls;
#pragma unroll
      for (int i = 0; i < 4; ++i) pv[i] = *reinterpret_cast<const float4*>(lp + qd * 16 + i * 4);
      pd = lp[64];
    }
  };
  int pend_n = 0, pend_b = 0;  // parked candidates of the previous tile (workgroup-uniform)
  int ticket = 0;              // thread 0: first slot reserved for them (return value of the in-flight atomic)
  auto flush_pending = [&]() {  // call after a barrier that follows `s_base = ticket`
    if (pend_n) {
      const int base = s_base;
      for (int i = tid; i < pend_n; i += 256)
        if (base + i < a.cap) a.cand[(size_t)pend_b * a.cap + base + i] = s_pend[i];
    }
  };
  // XCD-aware tile order (round 6).  A tile re-reads its one-cell halo ring: 60 cells for 32.  The dispatcher deals consecutive workgroup ids
  // round-robin over the 8 XCDs, so with `tile = id + k * grid` the workgroups that hold NEIGHBOURING tiles at the same moment sat on eight
  // different XCDs, each with its own L2, and every ring came over the fabric: 590 MB per 128-image launch against 288 MB of logits
  // (FETCH_SIZE calibrated on this access pattern: profiles/r06_e_fetch_size_calibration.json).  Now XCD x owns the CONTIGUOUS tile range
  // [x per, (x + 1) per) and its workgroups walk it together (workgroup q of the XCD takes tiles q, q + Q, ...): at any moment an XCD holds
  // ~160 consecutive tiles = seven tile rows, so a ring cell is fetched once by the XCD's L2 and hit by the neighbours.  Placement is a
  // speed assumption only: any mapping computes the same candidates (their order in the per-image list is arbitrary either way; k_topk sorts).
  const bool xcd_map = SSHIP_NMS_XCD && (gridDim.x & 7) == 0 && gridDim.x >= 8;
  const int xq = xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, xQ = xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int per = xcd_map ? (ntiles + 7) >> 3 : ntiles, xbase = xcd_map ? (int)(blockIdx.x & 7) * per : 0;
  const int t_end = min(per, ntiles - xbase);  // tiles of this workgroup's range (may be <= 0 for the last XCD of a small launch)
  if (xq < t_end) fetch(xbase + xq);
#pragma unroll 1
  for (int tl = xq; tl < t_end; tl += xQ) {
    const int tile = xbase + tl;
    int t = tile;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * NT_W - NHALO, y0 = ty * NT_H - NHALO;  // tile origin incl. halo (multiple of 8)
    if (tid == 0) {
      s_cnt = 0;
      if (pend_n) s_base = ticket;  // the only wait for the reservation, a whole tile after it was issued
    }
    if constexpr (LOADER == 0) {
      if (tid < NCELL * 4) {
        float v[16];
        const bool in = pin;
        const float d = in ? pd : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[4 * i] = in ? pv[i].x : 0.f; v[4 * i + 1] = in ? pv[i].y : 0.f;
          v[4 * i + 2] = in ? pv[i].z : 0.f; v[4 * i + 3] = in ? pv[i].w : 0.f;
        }
        float m = d;
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, v[i]);
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[i] = __expf(v[i] - m); sum += v[i]; }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __expf(d - m);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float* dst = s_s + (cyl * 8 + qd * 2 + r) * NLS + cxl * 8;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float4 o;
            o.x = in ? v[r * 8 + h * 4 + 0] * inv : -INFINITY;
            o.y = in ? v[r * 8 + h * 4 + 1] * inv : -INFINITY;
            o.z = in ? v[r * 8 + h * 4 + 2] * inv : -INFINITY;
            o.w = in ? v[r * 8 + h * 4 + 3] * inv : -INFINITY;
            *reinterpret_cast<float4*>(dst + h * 4) = o;
          }
        }
      }
      if (tl + xQ < t_end) fetch(tile + xQ);  // lands while this tile goes through its LDS phases
    } else {
      for (int i = tid; i < NLH * NLW; i += 256) {
        const int ly = i / NLW, lx = i % NLW;
        const int gy = y0 + ly, gx = x0 + lx;
        float sc = -INFINITY;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) sc = a.scores_in[((size_t)b * a.H + gy) * a.W + gx];
        s_s[ly * NLS + lx] = sc;
      }
    }
    __syncthreads();
    flush_pending();  // previous tile's candidates -> global (s_pend is rewritten only after the next two barriers)
    // ---- row maxima over [x-R, x+R] for the 64 interior columns of all 48 rows
    if constexpr (RT == 4) {
      for (int it = tid; it < NLH * (NT_W / 4); it += 256) {
        const int row = it >> 4, seg = it & 15;
        const float* p = s_s + row * NLS + NHALO - 4 + 4 * seg;
        float v[12], o[4];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float4 t4 = *reinterpret_cast<const float4*>(p + 4 * q);
          v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
        }
        window9x4(v, o);
        *reinterpret_cast<float4*>(s_r + row * NT_W + 4 * seg) = make_float4(o[0], o[1], o[2], o[3]);
      }
    } else {
      const int R = a.radius;
      for (int i = tid; i < NLH * NT_W; i += 256) {
        const int ly = i / NT_W, lx = (i % NT_W) + NHALO;
        float m = -INFINITY;
        for (int d = -R; d <= R; ++d) m = fmaxf(m, s_s[ly * NLS + lx + d]);
        s_r[ly * NT_W + (lx - NHALO)] = m;
      }
    }
    __syncthreads();
    // ---- column maxima, survival test; every thread owns NMS_SLOTS pixels whose verdicts stay in registers
    float sc[NMS_SLOTS];
    unsigned keep = 0;
    unsigned pix[NMS_SLOTS];
    auto verdict = [&](int slot, int iy, int ix, float s, float m) {
      const int gy = y0 + NHALO + iy, gx = x0 + NHALO + ix;
      sc[slot] = s;
      pix[slot] = (unsigned)(gy * a.W + gx);
      if (gy >= a.H || gx >= a.W) return;
      const bool is_max = (RT == 4) ? (s == m) : (a.radius <= 0 || s == m);
      const size_t o = ((size_t)b * a.H + gy) * a.W + gx;
      if (a.scores_raw_out) a.scores_raw_out[o] = s;
      if (a.scores_out) a.scores_out[o] = is_max ? s : 0.0f;
      if (a.cand && is_max && s >= a.thr_f && gy >= a.border && gy < a.H - a.border && gx >= a.border &&
          gx < a.W - a.border)
        keep |= 1u << slot;
    };
    if constexpr (RT == 4) {
#pragma unroll
      for (int k = 0; k < NMS_SLOTS / 4; ++k) {
        const int it = tid + k * 256;
        const int ix = it & 63, rg = it >> 6;  // 4 consecutive interior rows 4*rg .. 4*rg+3 of column ix
        float v[12], o[4];
#pragma unroll
        for (int q = 0; q < 12; ++q) v[q] = s_r[(4 * rg + NHALO - 4 + q) * NT_W + ix];
        window9x4(v, o);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          verdict(k * 4 + q, 4 * rg + q, ix, s_s[(4 * rg + q + NHALO) * NLS + ix + NHALO], o[q]);
      }
    } else {
      const int R = a.radius;
#pragma unroll
      for (int k = 0; k < NMS_SLOTS; ++k) {
        const int i = tid + k * 256;
        const int iy = i / NT_W, ix = i % NT_W;
        float m = -INFINITY;
        for (int d = -R; d <= R; ++d) m = fmaxf(m, s_r[(iy + NHALO + d) * NT_W + ix]);
        verdict(k, iy, ix, s_s[(iy + NHALO) * NLS + ix + NHALO], m);
      }
    }
    pend_n = 0;
    if (a.cand) {
      __syncthreads();  // s_s / s_r are dead: the candidate list takes their place
      // workgroup-local compaction: one LDS atomic per wave, ONE global atomic per tile (per-candidate global atomics
      // on the per-image counter serialise in L2)
      const int mine = __popc(keep);
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      const int total = __shfl(incl, 63, 64);
      int wbase = 0;
      if (lane == 63 && total) wbase = atomicAdd(&s_cnt, total);
      wbase = __shfl(wbase, 63, 64);
      int pos = wbase + incl - mine;
#pragma unroll
      for (int k = 0; k < NMS_SLOTS; ++k)
        if (keep & (1u << k)) s_c[pos++] = ((unsigned long long)__float_as_uint(sc[k]) << 32) | pix[k];
      __syncthreads();
      const int n = s_cnt;
      if (n > NMS_PEND) {  // rare (plateaus): reserve and write synchronously
        if (tid == 0) s_base = atomicAdd(&a.cand_count[b], n);
        __syncthreads();
        const int base = s_base;
        for (int i = tid; i < n; i += 256)
          if (base + i < a.cap) a.cand[(size_t)b * a.cap + base + i] = s_c[i];
      } else if (n > 0) {   // park; the reservation's return value is consumed one tile later
        if (tid < n) s_pend[tid] = s_c[tid];
        if (tid == 0) ticket = atomicAdd(&a.cand_count[b], n);
        pend_n = n; pend_b = b;
      }
    }
    __syncthreads();  // the next tile overwrites s_s / s_cnt (and s_base once the ticket is read)
  }
  if (pend_n) {
    if (tid == 0) s_base = ticket;
    __syncthreads();
    flush_pending();
  }
}

float threshold_as_float(double thr) {
  // smallest float f such that (double)f > thr   (SuperPoint.cc:700: `score > keypoint_threshold_`)
  float f = (float)thr;
  while ((double)f > thr) f = nextafterf(f, -INFINITY);
  while (!((double)f > thr)) f = nextafterf(f, INFINITY);
  return f;
}

void launch_nms_tile(int loader, const NmsArgs& a, hipStream_t s) {
  const int ntiles = a.B * ((a.W + NT_W - 1) / NT_W) * ((a.H + NT_H - 1) / NT_H);
  const int tiles = ntiles < 5 * cu_count() ? ntiles : 5 * cu_count();  // persistent: 5 workgroups per CU (30 KB LDS each)
  const bool r4 = a.radius == 4;
  if (loader == 0) {
    if (r4) hipLaunchKernelGGL((k_nms_tile<0, 4>), dim3(tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_nms_tile<0, -1>), dim3(tiles), dim3(256), 0, s, a);
  } else {
    if (r4) hipLaunchKernelGGL((k_nms_tile<1, 4>), dim3(tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_nms_tile<1, -1>), dim3(tiles), dim3(256), 0, s, a);
  }
}

// ---------------------------------------------------------------------------------------------------
// Top-k: one 1024-thread workgroup per image.  8-pass MSB radix select finds the K-th largest 64-bit key
// (keys are unique), the K keys >= it are compacted into LDS, bitonic-sorted descending, and turned into
// keypoints + cells.   reference: src/SuperPoint.cc:703-719.
// ---------------------------------------------------------------------------------------------------


constexpr int TOPK_CACHE = 8;  // keys per thread held in registers when the image has <= 8192 candidates

__global__ __launch_bounds__(1024) void k_topk(TopkArgs a) {
  __shared__ unsigned long long s_key[kMaxKp];
  __shared__ int s_hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_cnt, s_done;
  __shared__ unsigned long long s_out[1024];  // rank-sort output (K <= 1024)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int n_cand = a.cand_count[b];
  const int M = min(n_cand, a.cap);
  const int K = min(M, a.max_kp);
  if (a.reset_count) {  // every thread has its copy of the counter before it is cleared for the next call
    __syncthreads();
    if (tid == 0) a.reset_count[b] = 0;
  }
  if (a.n_cand_out && tid == 0) a.n_cand_out[b] = n_cand;
  if (tid == 0) a.n_out[b] = K;
  if (K == 0) return;
  const unsigned long long* cand = a.cand + (size_t)b * a.cap;
  // the usual case (a few thousand candidates): every key is read from HBM once and stays in registers for the 8
  // select passes and the compaction; larger sets stream from L2 each pass.
  const bool cached = M <= TOPK_CACHE * 1024;
  unsigned long long kr[TOPK_CACHE];
#pragma unroll
  for (int j = 0; j < TOPK_CACHE; ++j) kr[j] = (cached && tid + j * 1024 < M) ? cand[tid + j * 1024] : 0ull;
  // one LDS atomic per wave when the whole wave agrees on the bin (the top digits of positive float scores nearly
  // always do; 64 same-address atomics serialise)
  auto count = [&](bool valid, int digit) {
    const unsigned long long act = __ballot(valid);
    if (!act) return;
    const int d0 = __shfl(digit, __ffsll((long long)act) - 1, 64);
    const unsigned long long same = __ballot(valid && digit == d0);
    if (same == act) {
      if (lane == __ffsll((long long)act) - 1) atomicAdd(&s_hist[d0], __popcll(act));
    } else if (valid) {
      atomicAdd(&s_hist[digit], 1);
    }
  };
  unsigned long long thresh = 0;
  if (M > K) {
    if (tid == 0) { s_prefix = 0; s_remaining = K; s_done = 0; }
    for (int pass = 0; pass < 8; ++pass) {
      if (tid < 256) s_hist[tid] = 0;
      __syncthreads();
      const unsigned long long prefix = s_prefix;
      const int shift = 56 - 8 * pass;
      if (cached) {
#pragma unroll
        for (int j = 0; j < TOPK_CACHE; ++j) {
          const unsigned long long k = kr[j];
          count(tid + j * 1024 < M && (pass == 0 || (k >> (shift + 8)) == prefix), (int)((k >> shift) & 255));
        }
      } else {
        for (int i0 = 0; i0 < M; i0 += 1024) {
          const int i = i0 + tid;
          const unsigned long long k = i < M ? cand[i] : 0ull;
          count(i < M && (pass == 0 || (k >> (shift + 8)) == prefix), (int)((k >> shift) & 255));
        }
      }
      __syncthreads();
      if (tid < 64) {
        // descending-digit scan by one wave: lane l owns digits 255-4l .. 252-4l
        int h[4], loc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = s_hist[255 - (4 * lane + j)]; loc += h[j]; }
        int incl = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        const int rem = s_remaining;
        int c = incl - loc;
        if (c < rem && rem <= incl) {
          int d = 255 - 4 * lane;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (c + h[j] < rem && d == 255 - (4 * lane + j)) { c += h[j]; --d; }
          s_remaining = rem - c;
          s_prefix = (prefix << 8) | (unsigned)d;
          // The boundary bin is taken WHOLE: every key whose top digits are >= this prefix is selected and there are exactly K of them -
          // the remaining digits of the threshold are zeros and the later passes would only confirm it.  With ~7 k candidates this
          // happens after 3-4 of the 8 passes (three barriers each; the kernel is one workgroup per image and barrier-bound).
          if (rem - c == h[255 - d - 4 * lane]) s_done = pass + 1;
        }
      }
      __syncthreads();
      if (s_done) break;
    }
    thresh = s_done ? s_prefix << (64 - 8 * s_done) : s_prefix;
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  auto emit = [&](bool take, unsigned long long k) {
    const unsigned long long act = __ballot(take);
    if (!act) return;
    int base = 0;
    const int leader = __ffsll((long long)act) - 1;
    if (lane == leader) base = atomicAdd(&s_cnt, __popcll(act));
    base = __shfl(base, leader, 64);
    const int pos = base + __popcll(act & ((1ull << lane) - 1ull));
    if (take && pos < kMaxKp) s_key[pos] = k;
  };
  if (cached) {
#pragma unroll
    for (int j = 0; j < TOPK_CACHE; ++j) emit(tid + j * 1024 < M && kr[j] >= thresh, kr[j]);
  } else {
    for (int i0 = 0; i0 < M; i0 += 1024) {
      const int i = i0 + tid;
      const unsigned long long k = i < M ? cand[i] : 0ull;
      emit(i < M && k >= thresh, k);
    }
  }
  __syncthreads();
  const unsigned long long* sorted = s_key;
  if (K <= 1024) {
    // Rank sort (round 4): key i goes to position #{j : key_j > key_i} - the keys are distinct (the index is part of the key), every
    // thread scans the K selected keys from LDS (all lanes read the same address: a broadcast, 16 bytes = two keys per read) and no
    // barrier is needed until the scatter.  The bitonic network it replaces is 55 barrier rounds for K = 600 (P2 = 1024) in a kernel
    // that is ONE workgroup per image: k_topk 34 us -> see profiles/r04_*; the order is the same total order, so the output is bit-identical.
    if (tid < ((K + 1) & ~1) && tid >= K) s_key[tid] = 0ull;  // pad to an even count
    __syncthreads();
    if (tid < K) {
      const unsigned long long mine = s_key[tid];
      int rank = 0;
      typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
      for (int jj = 0; jj < K; jj += 2) {
        const ull2 two = *reinterpret_cast<const ull2*>(&s_key[jj]);
        rank += (two[0] > mine) + (two[1] > mine);
      }
      s_out[rank] = mine;
    }
    __syncthreads();
    sorted = s_out;
  } else {
  int P2 = 1;
  while (P2 < K) P2 <<= 1;
  for (int i = K + tid; i < P2; i += 1024) s_key[i] = 0ull;
  __syncthreads();
  // bitonic sort, descending
  for (int size = 2; size <= P2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (P2 >> 1); i += 1024) {
        const int lo = ((i / stride) * (stride << 1)) + (i % stride);
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = s_key[lo], y = s_key[hi];
        if ((x < y) == desc) { s_key[lo] = y; s_key[hi] = x; }
      }
      __syncthreads();
    }
  }
  }
  for (int i = tid; i < K; i += 1024) {
    const unsigned long long k = sorted[i];
    const float score = __uint_as_float((unsigned)(k >> 32));
    const unsigned idx = (unsigned)(k & 0xffffffffu);
    const int h = idx / a.score_w, w = idx % a.score_w;
    float* kp = a.kp_xys + ((size_t)b * a.max_kp + i) * 3;
    kp[0] = (float)w * a.scale_x;   // cv::KeyPoint(w * scale_x, h * scale_y, 1, -1, score)
    kp[1] = (float)h * a.scale_y;
    kp[2] = score;
    a.cell_h[(size_t)b * a.max_kp + i] = min(h / 8, a.desc_h - 1);
    a.cell_w[(size_t)b * a.max_kp + i] = min(w / 8, a.desc_w - 1);
  }
}

void launch_topk(const TopkArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_topk, dim3(B), dim3(1024), 0, s, a);
}

// Threshold scan of a dense score map (stage API sship_select_topk; SuperPoint.cc:696-702).
__global__ void k_threshold_scan(const float* __restrict__ scores, int H, int W, float thr_f, int border,
                                 unsigned long long* cand, int* cand_count, int cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int h = i / W, w = i % W;
  if (h < border || h >= H - border || w < border || w >= W - border) return;
  const float s = scores[i];
  if (s >= thr_f) {
    const int idx = atomicAdd(cand_count, 1);
    if (idx < cap) cand[idx] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)i;
  }
}
void launch_threshold_scan(const float* scores, int H, int W, float thr_f, int border, unsigned long long* cand,
                           int* cand_count, int cap, hipStream_t s) {
  hipLaunchKernelGGL(k_threshold_scan, dim3((H * W + 255) / 256), dim3(256), 0, s, scores, H, W, thr_f, border,
                     cand, cand_count, cap);
}

// ---------------------------------------------------------------------------------------------------
// Descriptor gather.  reference: src/DescriptorGather.cu:14-56 (one block / keypoint, 2-byte loads 16 KB
// apart, grid read twice).  Here the grid is channels-last, so a keypoint is ONE contiguous 512-B row: a
// wave reads it with a single 8-B/lane coalesced load, reduces in registers, writes 512 B.  HBM-bound:
// algorithmic bytes = n*(512 + 512 + 8).
// RAW = 1: the row is the un-normalised convDb output; apply the dense F.normalize (exporter :88-89, incl.
// its fp16 rounding) first, then the gather's own renormalisation - the reference's two-step arithmetic,
// evaluated only for the selected cells.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load4(const _Float16* p, float (&v)[4]) {
  const h4_t x = *reinterpret_cast<const h4_t*>(p);
  v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
}

template <bool RAW>
__global__ __launch_bounds__(256) void k_gather_hwc(const _Float16* __restrict__ grid, int C, int gh, int gw,
                                                    size_t img_stride, const int* __restrict__ cell_h,
                                                    const int* __restrict__ cell_w, const int* __restrict__ n_dev,
                                                    int n_host, int max_kp, _Float16* __restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int n = n_dev ? n_dev[b] : n_host;
  if (i >= n) return;
  const int ch = cell_h[(size_t)b * max_kp + i], cw = cell_w[(size_t)b * max_kp + i];
  const _Float16* row = grid + (size_t)b * img_stride + ((size_t)ch * gw + cw) * C;
  _Float16* orow = out + ((size_t)b * max_kp + i) * C;
  {  // C <= 256 (checked by the caller): one trip
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int c = lane * 4;
    if (c < C) load4(row + c, v);
    if (RAW) {
      const float ss = wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
      const float denom = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize(p=2, dim=1, eps=1e-12)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)(_Float16)(v[e] / denom);  // the fp16 dense grid value
    }
    const float ss2 = wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    const float inv = rsqrtf(ss2 + 1e-12f);
    if (c < C) *reinterpret_cast<h4_t*>(orow + c) = to_h4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
  }
}

void launch_gather_hwc(bool raw, const _Float16* grid, int C, int gh, int gw, size_t img_stride, const int* cell_h,
                       const int* cell_w, const int* n_dev, int n_host, int max_kp, int B, _Float16* out,
                       hipStream_t s) {
  const int nmax = n_dev ? max_kp : n_host;
  if (nmax <= 0) return;
  dim3 grid_dim((nmax + 3) / 4, B);
  if (raw)
    hipLaunchKernelGGL(k_gather_hwc<true>, grid_dim, dim3(256), 0, s, grid, C, gh, gw, img_stride, cell_h, cell_w,
                       n_dev, n_host, max_kp, out);
  else
    hipLaunchKernelGGL(k_gather_hwc<false>, grid_dim, dim3(256), 0, s, grid, C, gh, gw, img_stride, cell_h, cell_w,
                       n_dev, n_host, max_kp, out);
}

// CHW grid (the reference engine's layout, kept for the 1:1 launch_gather_descriptors entry point): the
// 2-byte loads are inherently 2*gh*gw bytes apart; each value is read ONCE into a register (the CUDA kernel
// reads the column twice) and a wave, not a 256-thread block + shared-memory tree, owns a keypoint.
__global__ __launch_bounds__(256) void k_gather_chw(const _Float16* __restrict__ grid, int C, int gh, int gw,
                                                    const int* __restrict__ cell_h, const int* __restrict__ cell_w,
                                                    int n, _Float16* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const size_t plane = (size_t)gh * gw;
  const size_t base = (size_t)cell_h[i] * gw + cell_w[i];
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;
  for (int c0 = 0; c0 < C; c0 += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e * 64 + lane;  // lanes walk adjacent channel planes
      const float t = c < C ? (float)grid[c * plane + base] : 0.f;
      ss += t * t;
      if (c0 == 0) v[e] = t;  // C <= 256 (the SuperPoint case): the column is read exactly once
    }
  }
  const float inv = rsqrtf(wave_sum(ss) + 1e-12f);
  for (int c0 = 0; c0 < C; c0 += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e * 64 + lane;
      if (c < C) {
        const float t = (c0 == 0) ? v[e] : (float)grid[c * plane + base];
        out[(size_t)i * C + c] = (_Float16)(t * inv);
      }
    }
  }
}
void launch_gather_chw(const _Float16* grid, int C, int gh, int gw, const int* cell_h, const int* cell_w, int n,
                       _Float16* out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_chw, dim3((n + 3) / 4), dim3(256), 0, s, grid, C, gh, gw, cell_h, cell_w, n, out);
}

// ---------------------------------------------------------------------------------------------------
// Dense exports for sship_sp_dense (stage/parity API; not on the throughput path).
// ---------------------------------------------------------------------------------------------------
// raw channels-last fp16 [B, Hc, Wc, 256] -> F.normalize'd CHW fp16 [B, 256, Hc, Wc] (the engine's layout).
__global__ __launch_bounds__(256) void k_desc_dense_chw(const _Float16* __restrict__ raw, int cells_per_img, int B,
                                                        _Float16* __restrict__ out) {
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (cell >= B * cells_per_img) return;
  const int b = cell / cells_per_img, ci = cell % cells_per_img;
  float v[4];
  load4(raw + (size_t)cell * 256 + lane * 4, v);
  const float ss = wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  const float denom = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    out[((size_t)b * 256 + lane * 4 + e) * cells_per_img + ci] = (_Float16)(v[e] / denom);
}
void launch_desc_dense_chw(const _Float16* raw, int cells_per_img, int B, _Float16* out, hipStream_t s) {
  hipLaunchKernelGGL(k_desc_dense_chw, dim3((B * cells_per_img + 3) / 4), dim3(256), 0, s, raw, cells_per_img, B, out);
}

// logits channels-last padded [B*cells, ls] f32 -> [B, 65, cells] f32
__global__ void k_logits_chw(const float* __restrict__ in, int ls, int cells_per_img, int B, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * cells_per_img * 65) return;
  const int ci = i % cells_per_img, c = (i / cells_per_img) % 65, b = i / (cells_per_img * 65);
  out[i] = in[((size_t)b * cells_per_img + ci) * ls + c];
}
void launch_logits_chw(const float* in, int ls, int cells_per_img, int B, float* out, hipStream_t s) {
  const int n = B * cells_per_img * 65;
  hipLaunchKernelGGL(k_logits_chw, dim3((n + 255) / 256), dim3(256), 0, s, in, ls, cells_per_img, B, out);
}

// 3-channel BGR u8 -> gray u8 (cv::COLOR_BGR2GRAY fixed-point: (B*1868 + G*9617 + R*4899 + 8192) >> 14).
__global__ void k_bgr2gray(const uint8_t* __restrict__ in, int n, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int bb = in[3 * i], g = in[3 * i + 1], r = in[3 * i + 2];
  out[i] = (uint8_t)((bb * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
}
void launch_bgr2gray(const uint8_t* in, int n, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_bgr2gray, dim3((n + 255) / 256), dim3(256), 0, s, in, n, out);
}

}  // namespace sship
