// EigenPlaces (ResNet-18 trunk + L2Norm / GeM / Linear / L2Norm) on gfx950 - SURVEY 8(f) row 4, the place recogniser behind
// superslam::IPlaceRecognizer (src/EigenPlaces.cc:123-174; model: utils/convert_eigenplaces_to_onnx.py:54-60).  It runs once
// per keyframe on the loop-closure thread (src/LoopCloser.cc), not on the per-frame hot path, so it reuses the generic
// implicit-GEMM template (igemm.h) instead of getting kernels of its own:
//   * BatchNorm is folded into the conv weights / bias on the host (eval mode);
//   * the 7x7 stride-2 stem (3 input channels, K = 147) is an im2col into 192-wide fp16 rows + a 1x1 GEMM;
//   * stride-2 3x3 / 1x1 convolutions run at stride 1 and the epilogue keeps the even pixels (out(y, x) of a stride-2, pad-1
//     3x3 conv is the stride-1 result at (2y, 2x)): 4x the MFMA work on three small layers, no new kernel;
//   * residual add + ReLU live in the epilogue; max-pool and the aggregation tail are two small kernels.
#include "igemm.h"
#include "kernels.h"

namespace sship {

// bias (+ residual) (+ ReLU) -> fp16 channels-last; DECIM: only even (y, x) are written, at (y / 2, x / 2)
template <bool RELU, bool RES, bool DECIM>
struct EpiEP {
  template <int MT, int NT>
  static __device__ __forceinline__ void run(const IgemmArgs& p, f16x_t (&acc)[MT][NT], int b, int yb, int x, int cb0, int hh) {
    _Float16* out = static_cast<_Float16*>(p.out0);
    const _Float16* res = static_cast<const _Float16*>(p.out1);
    const int Ho = DECIM ? (p.H + 1) >> 1 : p.H, Wo = DECIM ? (p.W + 1) >> 1 : p.W;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cb0 + m * 32 + hh * 4 + g * 8;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + c);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int y = yb + n;
          if (y >= p.H || x >= p.W || c >= p.cout) continue;
          if (DECIM && ((y | x) & 1)) continue;
          const size_t o = ((size_t)(b * Ho + (DECIM ? y >> 1 : y)) * Wo + (DECIM ? x >> 1 : x)) * p.ostride + c;
          float v0 = acc[m][n][4 * g + 0] + bv.x, v1 = acc[m][n][4 * g + 1] + bv.y;
          float v2 = acc[m][n][4 * g + 2] + bv.z, v3 = acc[m][n][4 * g + 3] + bv.w;
          if (RES) {
            const h4_t r = *reinterpret_cast<const h4_t*>(res + o);
            v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
          }
          if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          *reinterpret_cast<h4_t*>(out + o) = to_h4(v0, v1, v2, v3);
        }
      }
  }
};

template <int KS, int CIN, class Epi>
static hipError_t ep_gemm(const ConvW& w, const _Float16* in, _Float16* out, const _Float16* res, int H, int W, hipStream_t s) {
  IgemmArgs a{};
  a.in0 = in; a.in1 = in; a.cs0 = CIN; a.cs1 = CIN; a.cin0 = CIN;
  a.wpack = w.w; a.bias = w.bias; a.B = 1; a.H = H; a.W = W; a.cout = w.cout; a.ostride = w.cout;
  a.out0 = out; a.out1 = const_cast<_Float16*>(res);
  return launch_igemm<KS, CIN, 64, 8, Epi>(a, w.cout_pad, s);
}

// conv (ks in {1, 3}, cin in {64, 128, 256, 512}; the stem GEMM: ks 1, cin 192) with the epilogue picked at run time
hipError_t ep_conv(const ConvW& w, const _Float16* in, _Float16* out, const _Float16* res, int H, int W, bool relu, bool decim, hipStream_t s) {
#define EP_CASE(KS_, CIN_)                                                                                        \
  if (w.ks == KS_ && w.cin == CIN_) {                                                                             \
    if (res) return ep_gemm<KS_, CIN_, EpiEP<true, true, false>>(w, in, out, res, H, W, s);                       \
    if (decim) return relu ? ep_gemm<KS_, CIN_, EpiEP<true, false, true>>(w, in, out, nullptr, H, W, s)           \
                           : ep_gemm<KS_, CIN_, EpiEP<false, false, true>>(w, in, out, nullptr, H, W, s);         \
    return relu ? ep_gemm<KS_, CIN_, EpiEP<true, false, false>>(w, in, out, nullptr, H, W, s)                     \
                : ep_gemm<KS_, CIN_, EpiEP<false, false, false>>(w, in, out, nullptr, H, W, s);                   \
  }
  EP_CASE(3, 64) EP_CASE(3, 128) EP_CASE(3, 256) EP_CASE(3, 512)
  EP_CASE(1, 64) EP_CASE(1, 128) EP_CASE(1, 256) EP_CASE(1, 192)
#undef EP_CASE
  return hipErrorInvalidValue;
}

// EigenPlaces::preprocess on the device (src/EigenPlaces.cc:123-145): GRAY2RGB / BGR2RGB, cv::resize INTER_LINEAR on 8-bit data, x 1/255,
// ImageNet mean / std, HWC -> CHW.  OpenCV's 8-bit path is fixed-point and this kernel is its restatement integer for integer
// (include/superslam_hip/place_recognizer.hpp::resize_bilinear_u8 is the host form the oracle pins): the tables hold, per output column /
// row, the two source indices and the two 11-bit coefficients (computed on the host, by the same code as the host path - they depend only
// on the two sizes); a thread does one output pixel: horizontal pass in int, vertical pass
//   (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, saturate to u8,
// then the three float operations of the host path in the same order (one multiply, one subtract, one correctly rounded divide: no
// contraction possible).  Output fp32 CHW, bit-identical to sship_ep_preprocess.
// tab: [4][out_w] ints (sx, sx1, ax0, ax1) then [4][out_h] (sy, sy1, by0, by1)
__global__ __launch_bounds__(256) void k_ep_resize_norm(const uint8_t* __restrict__ src, int stride, int ch, const int* __restrict__ tab,
                                                        int out_w, int out_h, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= out_w * out_h) return;
  const int y = idx / out_w, x = idx - y * out_w;
  const int sx = tab[x], sx1 = tab[out_w + x], ax0 = tab[2 * out_w + x], ax1 = tab[3 * out_w + x];
  const int* ty = tab + 4 * out_w;
  const int sy = ty[y], sy1 = ty[out_h + y], by0 = ty[2 * out_h + y], by1 = ty[3 * out_h + y];
  const uint8_t* p0 = src + (size_t)sy * stride;
  const uint8_t* p1 = src + (size_t)sy1 * stride;
  const float kMean[3] = {0.485f, 0.456f, 0.406f}, kStd[3] = {0.229f, 0.224f, 0.225f};
  const size_t hw = (size_t)out_w * out_h;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = ch == 1 ? 0 : 2 - c;  // GRAY2RGB replicates, BGR2RGB swaps (per-channel resize commutes with both)
    const int r0 = p0[sx * ch + sc] * ax0 + p0[sx1 * ch + sc] * ax1;
    const int r1 = p1[sx * ch + sc] * ax0 + p1[sx1 * ch + sc] * ax1;
    int v = (((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : v > 255 ? 255 : v;
    const float f = (float)v * (1.0f / 255.0f);
    out[c * hw + idx] = (f - kMean[c]) / kStd[c];
  }
}
void launch_ep_resize_norm(const uint8_t* src, int stride, int ch, const int* tab, int out_w, int out_h, float* out, hipStream_t s) {
  const int n = out_w * out_h;
  hipLaunchKernelGGL(k_ep_resize_norm, dim3((n + 255) / 256), dim3(256), 0, s, src, stride, ch, tab, out_w, out_h, out);
}

// stem im2col: fp32 CHW [3][H][W] -> fp16 rows [Ho*Wo][192], k = c*49 + ky*7 + kx (PyTorch weight order), zero padded
__global__ __launch_bounds__(256) void k_ep_im2col(const float* __restrict__ x, int H, int W, int Ho, int Wo, _Float16* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one thread per (pixel, 8-wide k group): 24 groups per pixel
  if (idx >= Ho * Wo * 24) return;
  const int pix = idx / 24, kg = idx - pix * 24;
  const int oy = pix / Wo, ox = pix - oy * Wo;
  h8_t v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = kg * 8 + e;
    float f = 0.f;
    if (k < 147) {
      const int c = k / 49, r = k - c * 49, ky = r / 7, kx = r - ky * 7;
      const int iy = 2 * oy + ky - 3, ix = 2 * ox + kx - 3;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) f = x[((size_t)c * H + iy) * W + ix];
    }
    v[e] = (_Float16)f;
  }
  *reinterpret_cast<h8_t*>(out + (size_t)pix * 192 + kg * 8) = v;
}
void launch_ep_im2col(const float* x, int H, int W, int Ho, int Wo, _Float16* out, hipStream_t s) {
  const int n = Ho * Wo * 24;
  hipLaunchKernelGGL(k_ep_im2col, dim3((n + 255) / 256), dim3(256), 0, s, x, H, W, Ho, Wo, out);
}

// MaxPool2d(3, 2, 1) on channels-last fp16 [H][W][64] -> [Ho][Wo][64] (padding = -inf)
__global__ __launch_bounds__(256) void k_ep_maxpool(const _Float16* __restrict__ in, int H, int W, int Ho, int Wo, _Float16* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (pixel, 8-channel group)
  if (idx >= Ho * Wo * 8) return;
  const int pix = idx >> 3, cg = idx & 7;
  const int oy = pix / Wo, ox = pix - oy * Wo;
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int iy = 2 * oy + dy, ix = 2 * ox + dx;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const h8_t v = *reinterpret_cast<const h8_t*>(in + ((size_t)iy * W + ix) * 64 + cg * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
    }
  h8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)m[e];
  *reinterpret_cast<h8_t*>(out + (size_t)pix * 64 + cg * 8) = o;
}
void launch_ep_maxpool(const _Float16* in, int H, int W, int Ho, int Wo, _Float16* out, hipStream_t s) {
  const int n = Ho * Wo * 8;
  hipLaunchKernelGGL(k_ep_maxpool, dim3((n + 255) / 256), dim3(256), 0, s, in, H, W, Ho, Wo, out);
}

// aggregation tail (one workgroup of 512 threads): per-location L2 normalisation over the 512 channels, GeM(p, eps = 1e-6)
// over the npix locations, Linear(512 -> 512) (weights transposed [in][out] fp32), L2 normalisation -> fp32 [512]
__global__ __launch_bounds__(512) void k_ep_tail(const _Float16* __restrict__ feat, int npix, float p, const float* __restrict__ wt,
                                                 const float* __restrict__ bias, float* __restrict__ out) {
  extern __shared__ float s_ep[];  // [npix] inverse norms | [512] pooled | [8] wave partials
  float* s_inv = s_ep;
  float* s_g = s_ep + npix;
  float* s_red = s_g + 512;
  const int t = threadIdx.x;
  for (int px = t; px < npix; px += 512) {
    float ss = 0.f;
    for (int c = 0; c < 512; c += 8) {
      const h8_t v = *reinterpret_cast<const h8_t*>(feat + (size_t)px * 512 + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
    }
    s_inv[px] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(||x||, eps)
  }
  __syncthreads();
  float acc = 0.f;
  for (int px = 0; px < npix; ++px) {
    const float v = fmaxf((float)feat[(size_t)px * 512 + t] * s_inv[px], 1e-6f);
    acc += exp2f(p * log2f(v));
  }
  const float mean = acc / (float)npix;
  s_g[t] = exp2f(log2f(mean) / p);
  __syncthreads();
  float y = bias[t];
  for (int c = 0; c < 512; ++c) y = fmaf(wt[(size_t)c * 512 + t], s_g[c], y);
  float ss = wave_sum(y * y);
  if ((t & 63) == 0) s_red[t >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += s_red[w];
  out[t] = y / fmaxf(sqrtf(tot), 1e-12f);
}
void launch_ep_tail(const _Float16* feat, int npix, float p, const float* wt, const float* bias, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_ep_tail, dim3(1), dim3(512), (size_t)(npix + 512 + 8) * 4, s, feat, npix, p, wt, bias, out);
}

}  // namespace sship
