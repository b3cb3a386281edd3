// Implicit-GEMM on the gfx950 matrix cores: one kernel template for every 3x3 / 1x1 convolution of
// SuperPoint (utils/convert_superpoint_to_onnx.py:38-49) and every Linear of LightGlue.
//
//   out[pixel][cout] = sum_{ky,kx,cin} in[pixel + (ky,kx)][cin] * W[cout][cin][ky][kx] + bias[cout]
//
// GEMM view (M = cout, N = pixels, K = taps*cin).  Weights are the MFMA A operand, pixels the B operand:
// each lane of the 32x32 accumulator then owns ONE pixel and 4 consecutive output channels per register
// quad, so per-pixel epilogues (bias, ReLU, 2x2 max-pool, rotary, residual, L2 stats) are lane-local and
// channels-last stores are 8-byte vectors.
//
// Data layout
//   activations : channels-last fp16, [B][H][W][C] (tokens are an image with W = 32)
//   weights     : packed on the host into A-fragment order,
//                 [cout_blk][cin_chunk(64)][ky][kx][kstep(16)][mtile(32)][lane(64)][8 halfs]
//                 so a weight stage is a straight coalesced copy into LDS and every A fragment read is a
//                 lane-linear, conflict-free ds_read_b128.
//   LDS         : input tile (TH+KS-1) x (32+KS-1) pixels x 64 channels, pixel stride 144 B (64 ch + 16 B
//                 pad: 9 sixteen-byte slots, odd -> the 16 lanes of a ds_read_b128 group hit 16 distinct
//                 slots), plus one weight stage (one tap row: KS taps x 64 channels x CT couts).
//
// Work decomposition: a workgroup (4 waves) owns a TH x 32 pixel tile and CT output channels; wave w owns
// pixel rows [w*TH/4, (w+1)*TH/4) -> NT = TH/4 N-tiles, MT = CT/32 M-tiles, MT*NT accumulators.
#pragma once
#include "common.h"

namespace sship {

struct IgemmArgs {
  const _Float16* in0;   // source 0, channels [0, cin0)
  const _Float16* in1;   // optional source 1, channels [cin0, CIN)   (cat[x, msg] without materialising it)
  int cin0;              // multiple of 64
  int cs0, cs1;          // elements per pixel in in0 / in1
  const _Float16* wpack; // packed weights (see above)
  const float* bias;     // [cout padded to CT]
  int B, H, W;           // input spatial shape (output: same, or floor(/2) for the pooled epilogue)
  int cout;              // real number of output channels
  int ostride;           // elements per output pixel
  void* out0;
  void* out1;
  void* out2;
  const float* aux;      // epilogue payload (rotary table)
  int np;                // LightGlue: padded tokens per sequence
  int flags;
};

constexpr int kCP = 72;  // halfs per LDS pixel (64 channels + 8 pad)

// SPLITK (EigenPlaces' deep, small-map layers): blockIdx.z owns the 64-channel chunks [z NCHUNK / gridDim.z, (z + 1) NCHUNK / gridDim.z) of
// the reduction and the epilogue (EpiPartial) writes raw fp32 partial sums; a 16 x 16 x 512 layer is 16 workgroups walking 4.7 MB of
// weights otherwise.  false: the code is exactly what it was.
template <int KS, int CIN, int CT, int TH, class Epi, bool SPLITK = false>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs p) {
  constexpr int HALO = KS / 2, TW = 32, TWH = TW + KS - 1, THH = TH + KS - 1;
  constexpr int MT = CT / 32, NT = TH / 4, NCHUNK = CIN / 64;
  constexpr int WSTAGE = KS * 4 * MT * 512;  // halfs per weight stage (one tap row of one 64-channel chunk)
  constexpr int NSTAGE = NCHUNK * KS;
  constexpr int IN_UNITS = THH * TWH * 8, IN_IT = (IN_UNITS + 255) / 256;  // 16-byte units
  constexpr int W_UNITS = WSTAGE / 8, W_IT = (W_UNITS + 255) / 256;
  static_assert(TH % 4 == 0 && CT % 32 == 0 && CIN % 64 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* s_in = reinterpret_cast<_Float16*>(smem);
  _Float16* s_w = s_in + THH * TWH * kCP;

  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int cb = blockIdx.y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;

  f16x_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const _Float16* wbase = p.wpack + (size_t)cb * (NCHUNK * KS * WSTAGE);

  // Software pipeline: the global loads of stage s+1 (weights; the input tile when a new 64-channel chunk
  // starts) are issued into registers BEFORE the MFMAs of stage s and written to LDS after them, so HBM/L2
  // latency hides under the matrix work instead of being exposed once per stage.
  // (ext_vector registers, not HIP's uint4: that is a struct with unions, and an ARRAY of them stays in scratch memory -
  // every instantiation of this kernel used 64-112 bytes of it per lane for these two prefetch arrays)
  typedef unsigned q4_t __attribute__((ext_vector_type(4)));
  q4_t rin[IN_IT], rw[W_IT];
  auto load_in = [&](int chunk) {
    const int c0 = chunk * 64;
    const _Float16* src;
    int cs;
    if (c0 < p.cin0) { src = p.in0 + c0; cs = p.cs0; } else { src = p.in1 + (c0 - p.cin0); cs = p.cs1; }
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int u = tid + i * 256;
      const int pix = u >> 3, part = u & 7;
      const int py = pix / TWH, px = pix - py * TWH;
      const int gy = y0 - HALO + py, gx = x0 - HALO + px;
      q4_t v = {0u, 0u, 0u, 0u};
      if (u < IN_UNITS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
        v = *reinterpret_cast<const q4_t*>(src + ((size_t)(b * p.H + gy) * p.W + gx) * cs + part * 8);
      rin[i] = v;
    }
  };
  auto store_in = [&]() {
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int u = tid + i * 256;
      if (u < IN_UNITS) *reinterpret_cast<q4_t*>(s_in + (u >> 3) * kCP + (u & 7) * 8) = rin[i];
    }
  };
  auto load_w = [&](int stage) {
    const _Float16* wsrc = wbase + (size_t)stage * WSTAGE;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int u = tid + i * 256;
      if (W_UNITS % 256 == 0 || u < W_UNITS) rw[i] = *reinterpret_cast<const q4_t*>(wsrc + u * 8);
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int u = tid + i * 256;
      if (W_UNITS % 256 == 0 || u < W_UNITS) *reinterpret_cast<q4_t*>(s_w + u * 8) = rw[i];
    }
  };

  int s_begin = 0, s_end = NSTAGE;
  if constexpr (SPLITK) {
    s_begin = (int)blockIdx.z * NCHUNK / (int)gridDim.z * KS;
    s_end = ((int)blockIdx.z + 1) * NCHUNK / (int)gridDim.z * KS;
  }
  load_in(s_begin / KS);
  load_w(s_begin);
#pragma unroll 1
  for (int s = s_begin; s < s_end; ++s) {
    const int ky = s % KS;
    __syncthreads();  // every wave is done reading the previous stage's tiles
    if (ky == 0) store_in();
    store_w();
    __syncthreads();
    if (s + 1 < s_end) {
      if ((s + 1) % KS == 0) load_in((s + 1) / KS);
      load_w(s + 1);
    }
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        h8_t a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a[m] = *reinterpret_cast<const h8_t*>(s_w + ((kx * 4 + ks) * MT + m) * 512 + lane * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int row = wave * NT + n + ky;
          const h8_t bf = *reinterpret_cast<const h8_t*>(s_in + (row * TWH + j + kx) * kCP + ks * 16 + hh * 8);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = mfma32(a[m], bf, acc[m][n]);
        }
      }
    }
  }
  Epi::template run<MT, NT>(p, acc, b, y0 + wave * NT, x0 + j, cb * CT, hh);
}

// ---------------------------------------------------------------------------------------------------
// Epilogues.  acc[m][n][4g+e] is channel cb0 + 32m + 4hh + 8g + e of pixel (yb + n, x).
// ---------------------------------------------------------------------------------------------------
// bias + ReLU (+ 2x2 floor max-pool) -> fp16 channels-last.  pool(relu(x + b)) == relu(max(x) + b)
// exactly (fl(a + b), relu and fp16 rounding are all monotone), so the pool runs on raw accumulators.
template <bool RELU, bool POOL>
struct EpiF16 {
  template <int MT, int NT>
  static __device__ __forceinline__ void run(const IgemmArgs& p, f16x_t (&acc)[MT][NT], int b, int yb, int x,
                                             int cb0, int hh) {
    _Float16* out = static_cast<_Float16*>(p.out0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cb0 + m * 32 + hh * 4 + g * 8;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + c);
        if constexpr (!POOL) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int y = yb + n;
            float v0 = acc[m][n][4 * g + 0] + bv.x, v1 = acc[m][n][4 * g + 1] + bv.y;
            float v2 = acc[m][n][4 * g + 2] + bv.z, v3 = acc[m][n][4 * g + 3] + bv.w;
            if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            if (y < p.H && x < p.W && c < p.cout)
              *reinterpret_cast<h4_t*>(out + ((size_t)(b * p.H + y) * p.W + x) * p.ostride + c) = to_h4(v0, v1, v2, v3);
          }
        } else {
          const int Ho = p.H >> 1, Wo = p.W >> 1;
#pragma unroll
          for (int n = 0; n < NT; n += 2) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = fmaxf(acc[m][n][4 * g + e], acc[m][n + 1][4 * g + e]);
              t = fmaxf(t, __shfl_xor(t, 1, 64));
              v[e] = t;
            }
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            if (RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            const int yo = (yb + n) >> 1, xo = x >> 1;
            if (!(x & 1) && yo < Ho && xo < Wo && c < p.cout)
              *reinterpret_cast<h4_t*>(out + ((size_t)(b * Ho + yo) * Wo + xo) * p.ostride + c) = to_h4(v[0], v[1], v[2], v[3]);
          }
        }
      }
  }
};

// bias -> fp32 channels-last (detector logits; the 65 channels live in a 128-wide padded row).
struct EpiF32 {
  template <int MT, int NT>
  static __device__ __forceinline__ void run(const IgemmArgs& p, f16x_t (&acc)[MT][NT], int b, int yb, int x,
                                             int cb0, int hh) {
    float* out = static_cast<float*>(p.out0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cb0 + m * 32 + hh * 4 + g * 8;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + c);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int y = yb + n;
          if (y < p.H && x < p.W && c < p.ostride) {
            float4 v = make_float4(acc[m][n][4 * g + 0] + bv.x, acc[m][n][4 * g + 1] + bv.y,
                                   acc[m][n][4 * g + 2] + bv.z, acc[m][n][4 * g + 3] + bv.w);
            *reinterpret_cast<float4*>(out + ((size_t)(b * p.H + y) * p.W + x) * p.ostride + c) = v;
          }
        }
      }
  }
};

// split-K partial sums: raw fp32 accumulators -> ws[z][pixel][cout] (DECIM: only the even (y, x) pixels, at (y / 2, x / 2)); bias, residual
// and activation are applied by the kernel that adds the gridDim.z partials (ep_kernels.hip: k_ep_splitk_finish)
template <bool DECIM>
struct EpiPartial {
  template <int MT, int NT>
  static __device__ __forceinline__ void run(const IgemmArgs& p, f16x_t (&acc)[MT][NT], int b, int yb, int x, int cb0, int hh) {
    const int Ho = DECIM ? (p.H + 1) >> 1 : p.H, Wo = DECIM ? (p.W + 1) >> 1 : p.W;
    float* out = static_cast<float*>(p.out0) + (size_t)blockIdx.z * Ho * Wo * p.ostride;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cb0 + m * 32 + hh * 4 + g * 8;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int y = yb + n;
          if (y >= p.H || x >= p.W || c >= p.cout) continue;
          if (DECIM && ((y | x) & 1)) continue;
          *reinterpret_cast<float4*>(out + ((size_t)(DECIM ? y >> 1 : y) * Wo + (DECIM ? x >> 1 : x)) * p.ostride + c) =
              make_float4(acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
        }
      }
  }
};

template <int KS, int CIN, int CT, int TH>
constexpr size_t igemm_smem_bytes() {
  return (size_t)((TH + KS - 1) * (32 + KS - 1) * kCP + KS * 4 * (CT / 32) * 512) * 2;
}

// Launch: grid.x = B * tiles, grid.y = cout blocks.  cout_pad = packed (padded) output channels.
template <int KS, int CIN, int CT, int TH, class Epi>
inline hipError_t launch_igemm(const IgemmArgs& a, int cout_pad, hipStream_t stream) {
  constexpr size_t smem = igemm_smem_bytes<KS, CIN, CT, TH>();
  static bool attr_set = false;
  auto kern = igemm_kernel<KS, CIN, CT, TH, Epi>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
  dim3 grid(a.B * tiles_x * tiles_y, (cout_pad + CT - 1) / CT);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
  return hipGetLastError();
}
// split-K launch: grid.z = ksplit (a divisor of CIN / 64); B = 1
template <int KS, int CIN, int CT, int TH, class Epi>
inline hipError_t launch_igemm_splitk(const IgemmArgs& a, int cout_pad, int ksplit, hipStream_t stream) {
  constexpr size_t smem = igemm_smem_bytes<KS, CIN, CT, TH>();
  static bool attr_set = false;
  auto kern = igemm_kernel<KS, CIN, CT, TH, Epi, true>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
  dim3 grid(a.B * tiles_x * tiles_y, (cout_pad + CT - 1) / CT, ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
  return hipGetLastError();
}

}  // namespace sship
