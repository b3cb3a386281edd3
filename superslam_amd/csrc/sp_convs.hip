// SuperPoint's 1x1 layers on the implicit-GEMM template (igemm.h) and the fused descriptor head.
// (The 3x3 layers live in conv_strip.hip / conv_pp.hip.)
//   convDb : 1x1 256->256, raw fp16 - dense grid only for sship_sp_dense; extraction uses k_desc_head_gather below
//   convPb : 1x1 256->65, fp32 logits in a 68-wide row (softmax happens in the heatmap kernel): k_convpb_stream below
// reference: utils/convert_superpoint_to_onnx.py:38-64,77,88.
#include <cstdlib>
#include <string>

#include "igemm.h"
#include "kernels.h"

namespace sship {

static IgemmArgs conv_args(const ConvW& w, const _Float16* in, int B, int H, int W) {
  IgemmArgs a{};
  a.in0 = in; a.in1 = in; a.cin0 = w.cin; a.cs0 = w.cin; a.cs1 = w.cin;
  a.wpack = w.w; a.bias = w.bias; a.B = B; a.H = H; a.W = W; a.cout = w.cout; a.ostride = w.cout;
  return a;
}

hipError_t sp_conv1x1_f16(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, hipStream_t s) {
  IgemmArgs a = conv_args(w, in, B, H, W);
  a.out0 = out;
  if (w.cin != 256) return hipErrorInvalidValue;
  return launch_igemm<1, 256, 128, 4, EpiF16<false, false>>(a, w.cout_pad, s);
}

// ---------------------------------------------------------------------------------------------------
// convPb as a streaming kernel: 1x1, 256 -> 65, fp32 logits.  The layer is HBM-bound (4.1 MB of fp16 in, 2.2 MB of fp32
// out per image, 1.06 GFLOP) and the implicit-GEMM template spent its time on two barriers per 64-channel stage and on
// re-staging 48 KB of weights per 128-pixel workgroup.  Here nothing goes through LDS but one weight M-tile:
//   * v_mfma_f32_16x16x32_f16 with the pixels as the N operand: lane l holds 8 consecutive channels (k-block l >> 4) of
//     pixel l & 15, i.e. a B fragment is a plain 16-byte global load and the four lanes of a pixel cover one 64-byte
//     sector per instruction; a 16-pixel tile is 8 loads per lane, prefetched one tile ahead into a second register set;
//   * the weights of output rows 0..63 live in registers for the whole kernel (4 M-tiles x 8 k-steps = 128 VGPRs,
//     read once per wave), the fifth M-tile (row 64 = the dustbin channel, rows 65..79 zero) comes from LDS (8 KiB);
//   * the accumulators start from the bias (LDS); a lane ends up with 4 consecutive channels of its pixel per M-tile:
//     16-byte stores, 64 contiguous bytes per pixel and instruction.
// 40 MFMAs (640 matrix-pipe clocks) per 16 pixels against ~1.3 KB of HBM traffic per pixel: memory-bound by design.
// ---------------------------------------------------------------------------------------------------
typedef float f4x_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k_convpb_stream(const _Float16* __restrict__ in, const _Float16* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int P,
                                                          int ostride) {
  __shared__ __attribute__((aligned(16))) _Float16 s_a4[8 * 512];  // A fragments of M-tile 4, [k-step][lane][8]
  const int lane = threadIdx.x & 63, n = lane & 15, kb = lane >> 4;
  for (int u = threadIdx.x; u < 8 * 64; u += 256) {
    const int ks = u >> 6, l = u & 63;
    *reinterpret_cast<uint4*>(s_a4 + u * 8) = *reinterpret_cast<const uint4*>(w + (size_t)(64 + (l & 15)) * 256 + ks * 32 + (l >> 4) * 8);
  }
  h8_t a[4][8];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a[m][ks] = *reinterpret_cast<const h8_t*>(w + (size_t)(16 * m + n) * 256 + ks * 32 + kb * 8);
  __shared__ __attribute__((aligned(16))) float s_b[80];  // bias, rows 65..79 zero: the accumulators start from it
  if (threadIdx.x < 80) s_b[threadIdx.x] = threadIdx.x < 65 ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int ntiles = (P + 15) >> 4, nwaves = gridDim.x * 4;
  const _Float16* src = in + kb * 8;
  h8_t fA[8], fB[8];
  auto fetch = [&](h8_t (&f)[8], int t) __attribute__((always_inline)) {
    const int pix = min(min(t, ntiles - 1) * 16 + n, P - 1);  // clamped: a prefetch past the end re-reads a valid pixel
    const _Float16* p = src + (size_t)pix * 256;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) f[ks] = *reinterpret_cast<const h8_t*>(p + ks * 32);
  };
  auto tile = [&](const h8_t (&f)[8], int t) __attribute__((always_inline)) {
    f4x_t acc[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[m] = *reinterpret_cast<const f4x_t*>(s_b + 16 * m + 4 * kb);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const h8_t a4 = *reinterpret_cast<const h8_t*>(s_a4 + (ks * 64 + lane) * 8);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][ks], f[ks], acc[m], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4, f[ks], acc[4], 0, 0, 0);
    }
    const int pix = t * 16 + n;
    if (pix < P) {
      float* o = out + (size_t)pix * ostride + 4 * kb;
#pragma unroll
      for (int m = 0; m < 4; ++m) *reinterpret_cast<f4x_t*>(o + 16 * m) = acc[m];
      if (kb == 0) o[64] = acc[4][0];
    }
  };
  const int t0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  fetch(fA, t0);
  for (int t = t0; t < ntiles; t += 2 * nwaves) {
    fetch(fB, t + nwaves);
    __builtin_amdgcn_sched_barrier(0);  // the prefetch stays above this tile's MFMAs
    tile(fA, t);
    if (t + nwaves < ntiles) {
      fetch(fA, t + 2 * nwaves);
      __builtin_amdgcn_sched_barrier(0);
      tile(fB, t + nwaves);
    }
  }
}

// w.w_q: the plain fp16 weight matrix [80][256] (rows 65..79 zero), w.bias: [>= 80] floats
hipError_t sp_conv1x1_f32(const ConvW& w, const _Float16* in, float* out, int ostride, int B, int H, int W,
                          hipStream_t s) {
  if (w.cin != 256) return hipErrorInvalidValue;
  static const bool use_igemm = dev_env("SUPERSLAM_HIP_CONVPB") && std::string(dev_env("SUPERSLAM_HIP_CONVPB")) == "igemm";  // A/B (developer build)
  if (w.w_q && w.cout == 65 && ostride >= 68 && ostride % 4 == 0 && !use_igemm) {
    const int P = B * H * W;
    int wgs = 2 * cu_count();
    if (wgs * 4 > (P + 15) / 16) wgs = ((P + 15) / 16 + 3) / 4;
    hipLaunchKernelGGL(k_convpb_stream, dim3(wgs), dim3(256), 0, s, in, w.w_q, w.bias, out, P, ostride);
    return hipGetLastError();
  }
#if SSHIP_DEV_SWITCHES
  IgemmArgs a = conv_args(w, in, B, H, W);
  a.out0 = out; a.ostride = ostride;
  return launch_igemm<1, 256, 96, 4, EpiF32>(a, w.cout_pad, s);  // 65 rows: three M-tiles, not four
#else
  return hipErrorInvalidValue;  // one path per layer: the detector's 1x1 is k_convpb_stream (sship_sp_create always uploads its matrix)
#endif
}

// ---------------------------------------------------------------------------------------------------
// Descriptor head at the selected keypoints only: convDb (1x1, 256 -> 256) + F.normalize + the gather's
// renormalisation, evaluated for the <= max_keypoints cells the top-k picked instead of all Hc*Wc cells
// (600 of 8084 at KITTI size: 13x less convDb work, and the dense 256-channel grid is never written).
// reference: convert_superpoint_to_onnx.py:88-89 (dense) + src/DescriptorGather.cu:14-56 (gather); per-cell
// arithmetic is unchanged (fp32 accumulate in the same k order as the dense igemm path, the raw value rounded to
// fp16 where the dense grid would hold it, F.normalize, fp16, rsqrt renormalise), so the staged API
// (sship_sp_dense + sship_gather_normalize) and this kernel agree to the last fp16 ulp.
// A workgroup (8 waves) owns 64 keypoints of one image; wave w owns output channels [32w, 32w+32).
// ---------------------------------------------------------------------------------------------------
constexpr int kDhLd = 264;  // halfs per LDS row (256 + 8: 33 sixteen-byte slots, odd)
__global__ __launch_bounds__(512) void k_desc_head_gather(const _Float16* __restrict__ da, int Hc, int Wc,
                                                          const int* __restrict__ cell_h, const int* __restrict__ cell_w,
                                                          const int* __restrict__ n_dev, int max_kp,
                                                          const _Float16* __restrict__ wp, const float* __restrict__ bias,
                                                          _Float16* __restrict__ out, size_t out_img_stride) {
  __shared__ __attribute__((aligned(16))) _Float16 s_x[64 * kDhLd];
  __shared__ float s_red[8][64];
  const int b = blockIdx.y, i0 = blockIdx.x * 64;
  const int n = n_dev[b];
  if (i0 >= n) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const _Float16* img = da + (size_t)b * Hc * Wc * 256;
  for (int u = tid; u < 64 * 32; u += 512) {
    const int r = u >> 5, part = u & 31;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i0 + r < n) {
      const int ch = cell_h[(size_t)b * max_kp + i0 + r], cw = cell_w[(size_t)b * max_kp + i0 + r];
      v = *reinterpret_cast<const uint4*>(img + ((size_t)ch * Wc + cw) * 256 + part * 8);
    }
    *reinterpret_cast<uint4*>(s_x + r * kDhLd + part * 8) = v;
  }
  h8_t a[16];
  {
    const _Float16* w = wp + (size_t)wave * (16 * 512) + lane * 8;  // packed [cb = wave][k16][lane][8] (ct = 32)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) a[ks] = *reinterpret_cast<const h8_t*>(w + ks * 512);
  }
  __syncthreads();
  f16x_t acc[2];
#pragma unroll
  for (int nn = 0; nn < 2; ++nn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nn][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const h8_t b0 = *reinterpret_cast<const h8_t*>(s_x + j * kDhLd + ks * 16 + hh * 8);
    const h8_t b1 = *reinterpret_cast<const h8_t*>(s_x + (32 + j) * kDhLd + ks * 16 + hh * 8);
    acc[0] = mfma32(a[ks], b0, acc[0]);
    acc[1] = mfma32(a[ks], b1, acc[1]);
  }
  // raw convDb value as the fp16 dense grid would hold it
  float v[2][16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + wave * 32 + hh * 4 + g * 8);
    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[nn][4 * g + e] = (float)(_Float16)(acc[nn][4 * g + e] + bb[e]);
  }
  auto row_sum_sq = [&](float (&tot)[2]) {
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ss += v[nn][r] * v[nn][r];
      ss += __shfl_xor(ss, 32, 64);
      if (hh == 0) s_red[wave][nn * 32 + j] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += s_red[w][nn * 32 + j];
      tot[nn] = t;
    }
    __syncthreads();
  };
  float tot[2];
  row_sum_sq(tot);
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    const float denom = fmaxf(sqrtf(tot[nn]), 1e-12f);  // F.normalize(p=2, dim=1, eps=1e-12)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[nn][r] = (float)(_Float16)(v[nn][r] / denom);  // the fp16 dense descriptor
  }
  row_sum_sq(tot);
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    const int i = i0 + nn * 32 + j;
    if (i >= n) continue;
    const float inv = rsqrtf(tot[nn] + 1e-12f);  // DescriptorGather.cu:46
    _Float16* orow = out + (size_t)b * out_img_stride + (size_t)i * 256 + wave * 32 + hh * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<h4_t*>(orow + g * 8) =
          to_h4(v[nn][4 * g] * inv, v[nn][4 * g + 1] * inv, v[nn][4 * g + 2] * inv, v[nn][4 * g + 3] * inv);
  }
}

// ---------------------------------------------------------------------------------------------------
// Whole descriptor head at the selected keypoints: convDa (3x3, 128 -> 256, ReLU) is ALSO evaluated only where a
// keypoint landed - its 3x3x128 input patch is gathered from the encoder output (a4b) - and feeds convDb + the two
// normalisations above without leaving the CU.  600 keypoints of 8084 cells at KITTI size: 13x less convDa work
// (it was the second most expensive head layer), and neither the 256-channel convDa map nor the descriptor grid is
// written to HBM.  Arithmetic per output is the dense kernel's: same fp16 operands, same k order (64-channel chunk, 32-channel
// half, kx, k-step, ky) into an fp32 accumulator that starts at the bias, ReLU, fp16.  reference: convert_superpoint_to_onnx.py:61-64,88-89.
// A workgroup (8 waves) owns 64 keypoints of one image; wave w owns output channels [32w, 32w+32) of both layers and
// streams its 72 KiB of convDa fragments from L2 (packed exactly as the dense conv kernels read them, ct = 32).
// ---------------------------------------------------------------------------------------------------
constexpr int kDsPatchLd = 584;  // halfs per keypoint and 64-channel chunk: 9 taps x 64 + 8 (1168 B: conflict-free b128 reads)
// NN = 32-keypoint N-tiles per workgroup: 2 for batches, 1 when that would leave most CUs without a workgroup (a stereo pair: 20 -> 40 workgroups,
// half the MFMAs and half the patch gather per workgroup; the per-keypoint arithmetic is the same, so the two are bit-identical)
template <int NN>
__global__ __launch_bounds__(512) void k_desc_head_sparse(const _Float16* __restrict__ a4b, int Hc, int Wc,
                                                          const int* __restrict__ cell_h, const int* __restrict__ cell_w,
                                                          const int* __restrict__ n_dev, int max_kp,
                                                          const _Float16* __restrict__ wda, const float* __restrict__ bda,
                                                          const _Float16* __restrict__ wdb, const float* __restrict__ bdb,
                                                          _Float16* __restrict__ out, size_t out_img_stride) {
  extern __shared__ __attribute__((aligned(16))) char ds_smem[];
  _Float16* s_p = reinterpret_cast<_Float16*>(ds_smem);      // [64][kDsPatchLd] patch chunk
  constexpr int NK = 32 * NN;                                 // keypoints per workgroup
  _Float16* s_x = s_p + NK * kDsPatchLd;                      // [NK][kDhLd] convDa output (convDb input)
  float (*s_red)[64] = reinterpret_cast<float (*)[64]>(s_x + NK * kDhLd);
  int* s_cell = reinterpret_cast<int*>(s_red + 8);            // [64] (cell_h << 16 | cell_w), -1 = no keypoint
  const int b = blockIdx.y, i0 = blockIdx.x * NK;
  const int n = min(max(n_dev[b], 0), max_kp);
  if (i0 >= n) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  if (tid < NK)
    s_cell[tid] = i0 + tid < n ? (cell_h[(size_t)b * max_kp + i0 + tid] << 16) | cell_w[(size_t)b * max_kp + i0 + tid] : -1;
  const _Float16* img = a4b + (size_t)b * Hc * Wc * 128;
  // ---- convDa at the keypoints ----
  f16x_t acc[NN];  // start from the bias, as the dense ping-pong kernel does (conv_pp.hip: same fp32 summation order)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bv = *reinterpret_cast<const float4*>(bda + wave * 32 + hh * 4 + g * 8);
#pragma unroll
    for (int nn = 0; nn < NN; ++nn) { acc[nn][4 * g] = bv.x; acc[nn][4 * g + 1] = bv.y; acc[nn][4 * g + 2] = bv.z; acc[nn][4 * g + 3] = bv.w; }
  }
  const _Float16* wa = wda + (size_t)wave * (72 * 512) + lane * 8;  // [cb = wave][chunk][tap][kstep][lane][8]
  // k order inside a 64-channel chunk: the dense kernels' (conv_pp.hip CIN = 128 / conv_pp128.hip: 32-channel half, kx, k-step, ky)
  auto frag_of = [](int idx, int& tap, int& ks) __attribute__((always_inline)) {  // idx 0 .. 35 within the chunk
    const int h32 = idx / 18, r18 = idx - 18 * h32, t6 = r18 / 3, ky = r18 - 3 * t6;
    tap = ky * 3 + (t6 >> 1); ks = 2 * h32 + (t6 & 1);
  };
  auto wfrag = [&](int lin) __attribute__((always_inline)) {  // lin 0 .. 71 over both chunks
    int tap, ks;
    frag_of(lin % 36, tap, ks);
    return *reinterpret_cast<const h8_t*>(wa + (((lin / 36) * 9 + tap) * 4 + ks) * 512);
  };
  h8_t fa[2][12];  // two groups of 12 fragments in flight
#pragma unroll
  for (int i = 0; i < 12; ++i) fa[0][i] = wfrag(i);
  __syncthreads();  // s_cell
#pragma unroll
  for (int chunk = 0; chunk < 2; ++chunk) {  // unrolled: the fragment double buffer is indexed by the parity of g_lin
    if (chunk) __syncthreads();  // every wave is done reading the previous chunk's patches
    for (int u = tid; u < NK * 72; u += 512) {
      const int kp = u / 72, rem = u - kp * 72, tap = rem >> 3, part = rem & 7;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int c = s_cell[kp];
      const int y = (c >> 16) + ky - 1, x = (c & 0xffff) + kx - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c >= 0 && y >= 0 && y < Hc && x >= 0 && x < Wc)  // conv padding = 1: taps outside the map are zero
        v = *reinterpret_cast<const uint4*>(img + ((size_t)y * Wc + x) * 128 + chunk * 64 + part * 8);
      *reinterpret_cast<uint4*>(s_p + kp * kDsPatchLd + tap * 64 + part * 8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
      const int g_lin = chunk * 3 + grp;  // 0..5 over the whole k range; buffers alternate across the chunk boundary
      if (g_lin + 1 < 6) {
#pragma unroll
        for (int i = 0; i < 12; ++i) fa[(g_lin + 1) & 1][i] = wfrag((g_lin + 1) * 12 + i);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        int tap, ks;
        frag_of(grp * 12 + i, tap, ks);
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          const h8_t bf = *reinterpret_cast<const h8_t*>(s_p + (nn * 32 + j) * kDsPatchLd + tap * 64 + ks * 16 + hh * 8);
          acc[nn] = mfma32(fa[g_lin & 1][i], bf, acc[nn]);
        }
      }
    }
  }
  // convDb fragments: requested now, they land while the convDa epilogue runs
  h8_t a[16];
  {
    const _Float16* w = wdb + (size_t)wave * (16 * 512) + lane * 8;  // packed [cb = wave][k16][lane][8] (ct = 32)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) a[ks] = *reinterpret_cast<const h8_t*>(w + ks * 512);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int nn = 0; nn < NN; ++nn)
      *reinterpret_cast<h4_t*>(s_x + (nn * 32 + j) * kDhLd + wave * 32 + hh * 4 + g * 8) =
          to_h4(fmaxf(acc[nn][4 * g], 0.f), fmaxf(acc[nn][4 * g + 1], 0.f), fmaxf(acc[nn][4 * g + 2], 0.f), fmaxf(acc[nn][4 * g + 3], 0.f));
  }
  __syncthreads();
  // ---- convDb + F.normalize + gather renormalisation (as k_desc_head_gather) ----
#pragma unroll
  for (int nn = 0; nn < NN; ++nn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nn][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
    for (int nn = 0; nn < NN; ++nn) {
      const h8_t bf = *reinterpret_cast<const h8_t*>(s_x + (nn * 32 + j) * kDhLd + ks * 16 + hh * 8);
      acc[nn] = mfma32(a[ks], bf, acc[nn]);
    }
  }
  float v[NN][16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bv = *reinterpret_cast<const float4*>(bdb + wave * 32 + hh * 4 + g * 8);
    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int nn = 0; nn < NN; ++nn)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[nn][4 * g + e] = (float)(_Float16)(acc[nn][4 * g + e] + bb[e]);
  }
  auto row_sum_sq = [&](float (&tot)[NN]) {
#pragma unroll
    for (int nn = 0; nn < NN; ++nn) {
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ss += v[nn][r] * v[nn][r];
      ss += __shfl_xor(ss, 32, 64);
      if (hh == 0) s_red[wave][nn * 32 + j] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < NN; ++nn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += s_red[w][nn * 32 + j];
      tot[nn] = t;
    }
    __syncthreads();
  };
  float tot[NN];
  row_sum_sq(tot);
#pragma unroll
  for (int nn = 0; nn < NN; ++nn) {
    const float denom = fmaxf(sqrtf(tot[nn]), 1e-12f);  // F.normalize(p=2, dim=1, eps=1e-12)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[nn][r] = (float)(_Float16)(v[nn][r] / denom);  // the fp16 dense descriptor
  }
  row_sum_sq(tot);
#pragma unroll
  for (int nn = 0; nn < NN; ++nn) {
    const int i = i0 + nn * 32 + j;
    if (i >= n) continue;
    const float inv = rsqrtf(tot[nn] + 1e-12f);  // DescriptorGather.cu:46
    _Float16* orow = out + (size_t)b * out_img_stride + (size_t)i * 256 + wave * 32 + hh * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<h4_t*>(orow + g * 8) =
          to_h4(v[nn][4 * g] * inv, v[nn][4 * g + 1] * inv, v[nn][4 * g + 2] * inv, v[nn][4 * g + 3] * inv);
  }
}

hipError_t launch_desc_head_sparse(const ConvW& da32, const ConvW& db32, const _Float16* a4b, int Hc, int Wc, const int* cell_h,
                                   const int* cell_w, const int* n_dev, int max_kp, int B, _Float16* out, size_t out_img_stride,
                                   hipStream_t s) {
  if (da32.cin != 128 || da32.cout != 256 || da32.ct != 32 || da32.ks != 3) return hipErrorInvalidValue;
  constexpr size_t smem = (size_t)64 * kDsPatchLd * 2 + (size_t)64 * kDhLd * 2 + 8 * 64 * 4 + 64 * 4;  // NN = 2 (NN = 1 uses half of the tile buffers)
  // thread-safe one-time opt-in to > 64 KiB of dynamic LDS (C++11 magic static; handles may be created on any thread)
  static const hipError_t attr_rc = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_desc_head_sparse<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_desc_head_sparse<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }();
  if (attr_rc != hipSuccess) return attr_rc;
  // latency mode: 32-keypoint workgroups when the 64-keypoint grid would not give half of the CUs a workgroup
  if (B * ((max_kp + 63) / 64) * 2 <= cu_count())
    hipLaunchKernelGGL(k_desc_head_sparse<1>, dim3((max_kp + 31) / 32, B), dim3(512), smem, s, a4b, Hc, Wc, cell_h, cell_w, n_dev,
                       max_kp, da32.w, da32.bias, db32.w, db32.bias, out, out_img_stride);
  else
    hipLaunchKernelGGL(k_desc_head_sparse<2>, dim3((max_kp + 63) / 64, B), dim3(512), smem, s, a4b, Hc, Wc, cell_h, cell_w, n_dev,
                       max_kp, da32.w, da32.bias, db32.w, db32.bias, out, out_img_stride);
  return hipGetLastError();
}

void launch_desc_head_gather(const ConvW& db32, const _Float16* da, int Hc, int Wc, const int* cell_h, const int* cell_w,
                             const int* n_dev, int max_kp, int B, _Float16* out, size_t out_img_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_desc_head_gather, dim3((max_kp + 63) / 64, B), dim3(512), 0, s, da, Hc, Wc, cell_h, cell_w, n_dev,
                     max_kp, db32.w, db32.bias, out, out_img_stride);
}

}  // namespace sship
