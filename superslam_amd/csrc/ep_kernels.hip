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

// ---- split-K for the deep layers ----
// out[pix][c] = act(sum_z ws[z][pix][c] + bias[c] (+ res[pix][c])): one thread per (pixel, 4 channels); z ascending (one fixed summation order)
template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void k_ep_splitk_finish(const float* __restrict__ ws, int ksplit, int npix, int cout, const float* __restrict__ bias,
                                                          const _Float16* __restrict__ res, _Float16* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int cq = cout >> 2;
  if (idx >= npix * cq) return;
  const int pix = idx / cq, c = (idx - pix * cq) * 4;
  const size_t o = (size_t)pix * cout + c, zs = (size_t)npix * cout;
  float4 v = *reinterpret_cast<const float4*>(ws + o);
  for (int z = 1; z < ksplit; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(ws + z * zs + o);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  const float4 bv = *reinterpret_cast<const float4*>(bias + c);
  v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
  if (RES) {
    const h4_t r = *reinterpret_cast<const h4_t*>(res + o);
    v.x += (float)r[0]; v.y += (float)r[1]; v.z += (float)r[2]; v.w += (float)r[3];
  }
  if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  *reinterpret_cast<h4_t*>(out + o) = to_h4(v.x, v.y, v.z, v.w);
}
// the split this layer gets: every 64-channel chunk its own workgroup layer once the plain launch would leave most CUs idle.
// The target is a CONSTANT (two workgroups per CU of the full 256-CU part), not the CU count of the device the process happens to see: the
// split fixes the fp32 summation order of the deep layers, and a descriptor's bits must not depend on the partition mode or a CU mask.
constexpr long kEpSplitTargetWgs = 512;
static int ep_ksplit(const ConvW& w, int H, int W, int th) {
  const int nchunk = w.cin / 64;
  if (w.ks != 3 || nchunk < 2) return 1;
  const long wgs = (long)((W + 31) / 32) * ((H + th - 1) / th) * ((w.cout_pad + 63) / 64);
  int k = 1;
  while (k * 2 <= nchunk && wgs * k * 2 <= kEpSplitTargetWgs) k *= 2;
  return k;
}
template <int CIN, int TH>
static hipError_t ep_conv_splitk(const ConvW& w, const _Float16* in, _Float16* out, const _Float16* res, int H, int W, bool relu, bool decim,
                                 int ksplit, float* ws, hipStream_t s) {
  IgemmArgs a{};
  a.in0 = in; a.in1 = in; a.cs0 = CIN; a.cs1 = CIN; a.cin0 = CIN;
  a.wpack = w.w; a.bias = w.bias; a.B = 1; a.H = H; a.W = W; a.cout = w.cout; a.ostride = w.cout;
  a.out0 = ws;
  hipError_t e = decim ? launch_igemm_splitk<3, CIN, 64, TH, EpiPartial<true>>(a, w.cout_pad, ksplit, s)
                       : launch_igemm_splitk<3, CIN, 64, TH, EpiPartial<false>>(a, w.cout_pad, ksplit, s);
  if (e != hipSuccess) return e;
  const int npix = decim ? ((H + 1) / 2) * ((W + 1) / 2) : H * W;
  const int n = npix * (w.cout / 4);
  if (res) hipLaunchKernelGGL((k_ep_splitk_finish<true, true>), dim3((n + 255) / 256), dim3(256), 0, s, ws, ksplit, npix, w.cout, w.bias, res, out);
  else if (relu) hipLaunchKernelGGL((k_ep_splitk_finish<true, false>), dim3((n + 255) / 256), dim3(256), 0, s, ws, ksplit, npix, w.cout, w.bias, res, out);
  else hipLaunchKernelGGL((k_ep_splitk_finish<false, false>), dim3((n + 255) / 256), dim3(256), 0, s, ws, ksplit, npix, w.cout, w.bias, res, out);
  return hipGetLastError();
}
// Exact bound over the layers of a (in_h, in_w) engine: the map sizes follow the real ceil chain (stem stride 2, max-pool stride 2, then one
// stride-2 block per level), and a split layer needs cin / 64 x output pixels x cout floats at most.  (Round 5 sized this as "layer2 + 2x
// headroom"; per-level ceil rounding ate the headroom on engines below ~64 x 64 - ADVICE r05: a 40 x 40 engine wrote 64 KB into 51 KB.)
size_t ep_splitk_workspace_bytes(int in_h, int in_w) {
  int h = (in_h - 1) / 2 + 1, w = (in_w - 1) / 2 + 1;  // stem
  h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;            // MaxPool2d(3, 2, 1): layer1 (one 64-channel chunk, never split)
  size_t need = 0;
  const int planes[3] = {128, 256, 512};
  int cin = 64;
  for (int L = 0; L < 3; ++L) {
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;      // block 0 conv1 decimates; every other conv of the level runs at (ho, wo)
    const size_t npix = (size_t)ho * wo;
    const size_t kmax = (size_t)planes[L] / 64;         // conv2 / block 1: cin = planes; block 0 conv1: cin = the previous level's planes (smaller)
    need = need > kmax * npix * planes[L] ? need : kmax * npix * planes[L];
    (void)cin; cin = planes[L]; h = ho; w = wo;
  }
  return need * sizeof(float);
}

// conv (ks in {1, 3}, cin in {64, 128, 256, 512}; the stem GEMM: ks 1, cin 192) with the epilogue picked at run time
hipError_t ep_conv(const ConvW& w, const _Float16* in, _Float16* out, const _Float16* res, int H, int W, bool relu, bool decim, hipStream_t s,
                   float* ws, size_t ws_bytes) {
  if (ws && w.ks == 3 && w.cin >= 128 && w.cout % 64 == 0) {
    // small maps: 4-row tiles double the workgroup count; then split the reduction until ~2 workgroups per CU
    const bool th4 = H * W <= 64 * 64;
    int ksplit = ep_ksplit(w, H, W, th4 ? 4 : 8);
    // never write past the workspace the caller really holds (ep_splitk_workspace_bytes is exact for the ResNet-18 chain; this guards any other caller)
    const size_t npix_out = decim ? (size_t)((H + 1) / 2) * ((W + 1) / 2) : (size_t)H * W;
    while (ksplit > 1 && (size_t)ksplit * npix_out * w.cout * sizeof(float) > ws_bytes) ksplit >>= 1;
    if (ksplit > 1) {
      if (w.cin == 128) return th4 ? ep_conv_splitk<128, 4>(w, in, out, res, H, W, relu, decim, ksplit, ws, s) : ep_conv_splitk<128, 8>(w, in, out, res, H, W, relu, decim, ksplit, ws, s);
      if (w.cin == 256) return th4 ? ep_conv_splitk<256, 4>(w, in, out, res, H, W, relu, decim, ksplit, ws, s) : ep_conv_splitk<256, 8>(w, in, out, res, H, W, relu, decim, ksplit, ws, s);
      if (w.cin == 512) return th4 ? ep_conv_splitk<512, 4>(w, in, out, res, H, W, relu, decim, ksplit, ws, s) : ep_conv_splitk<512, 8>(w, in, out, res, H, W, relu, decim, ksplit, ws, s);
    }
  }
#define EP_CASE(KS_, CIN_)                                                                                        \
  if (w.ks == KS_ && w.cin == CIN_) {                                                                             \
    if (res) return ep_gemm<KS_, CIN_, EpiEP<true, true, false>>(w, in, out, res, H, W, s);                       \
    if (decim) return relu ? ep_gemm<KS_, CIN_, EpiEP<true, false, true>>(w, in, out, nullptr, H, W, s)           \
                           : ep_gemm<KS_, CIN_, EpiEP<false, false, true>>(w, in, out, nullptr, H, W, s);         \
    return relu ? ep_gemm<KS_, CIN_, EpiEP<true, false, false>>(w, in, out, nullptr, H, W, s)                     \
                : ep_gemm<KS_, CIN_, EpiEP<false, false, false>>(w, in, out, nullptr, H, W, s);                   \
  }
  // layer1 (64 -> 64 at H/4 x W/4, one 64-channel chunk: nothing to split): 4-row tiles double the workgroup count of a small map
  if (w.ks == 3 && w.cin == 64 && !decim && H * W <= 160 * 160) {
    IgemmArgs a{};
    a.in0 = in; a.in1 = in; a.cs0 = 64; a.cs1 = 64; a.cin0 = 64;
    a.wpack = w.w; a.bias = w.bias; a.B = 1; a.H = H; a.W = W; a.cout = w.cout; a.ostride = w.cout;
    a.out0 = out; a.out1 = const_cast<_Float16*>(res);
    if (res) return launch_igemm<3, 64, 64, 4, EpiEP<true, true, false>>(a, w.cout_pad, s);
    return relu ? launch_igemm<3, 64, 64, 4, EpiEP<true, false, false>>(a, w.cout_pad, s) : launch_igemm<3, 64, 64, 4, EpiEP<false, false, false>>(a, w.cout_pad, s);
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

// ---------------------------------------------------------------------------------------------------
// Fused stem (round 6): conv 7x7 / stride 2 / pad 3, 3 -> 64 (BatchNorm folded) + ReLU + MaxPool2d(3, 2, 1) in ONE kernel.
// Rounds 3-5 ran it as im2col (25 MB of fp16 patch rows written and read back: 17.9 us) -> 1x1 GEMM over 192-wide rows (11.8 us) -> max-pool
// (5.6 us; the 8 MB stem map written and read back): 35 us of a 250-us descriptor for 1.2 of its 19 GFLOP.
// Here a workgroup owns an 8 x 8 tile of the POOLED map: it needs the 17 x 17 stem pixels around it and those need a 39 x 39 x 3 input patch,
// which is staged into LDS as fp16 once.  The GEMM runs on the matrix cores with the patch as the B operand built ON THE FLY: k is ordered
// (c, ky, kx padded to 8) so that a lane's 8-element k-slice is 8 CONSECUTIVE input pixels of one patch row - four ds_read_b32 from a 4-byte
// aligned address (the pixel's window starts at the even column 2 sx) - instead of a row of a materialised im2col matrix; kx = 7 carries a zero
// weight.  K = 21 rows x 8 = 168 -> 11 k-steps of 16; the 22 weight fragments (2 M-tiles x 11) live in 88 VGPRs for the whole kernel.
// Stem pixels go to LDS as fp16 (pixels outside the stem map as 0: every pooling window holds at least one real pixel and ReLU outputs are >= 0,
// so 0 is as good as the -inf padding of MaxPool2d), the pooled 8 x 8 x 64 tile leaves as whole 128-byte rows.  Neither the patch matrix nor
// the stem map exists in HBM.
// wfrag: [m 2][s 11][lane 64][8] fp16, lane (row = lane & 31, kg = lane >> 5) of fragment (m, s) = W[32 m + row][idx = 2 s + kg][kx = e]
// (idx = c * 7 + ky; idx = 21 and kx = 7 are zero).  Same rounding points as the GEMM path (fp16 operands, fp32 accumulate, bias, ReLU, fp16).
// ---------------------------------------------------------------------------------------------------
constexpr int kStemPT = 8;                       // pooled tile edge
constexpr int kStemST = 2 * kStemPT + 1;         // 17 stem pixels per edge
constexpr int kStemIT = 2 * kStemST + 5;         // 39 input pixels per edge
constexpr int kStemIW = 40;                      // patch row stride (halfs): column 39 is the zero-weight tap of the last window
constexpr int kStemLd = 72;                      // stem pixel stride in LDS (halfs): 64 channels + 8 (bank spread for the 16-byte pooling reads)
constexpr int kStemNpx = kStemST * kStemST;      // 289 stem pixels = 10 N-tiles of 32 (320 slots)
__global__ __launch_bounds__(256) void k_ep_stem_pool(const float* __restrict__ x, int H, int W, int Ho, int Wo, int Hp, int Wp,
                                                      const _Float16* __restrict__ wfrag, const float* __restrict__ bias,
                                                      _Float16* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) _Float16 s_in[3 * kStemIT * kStemIW];
  __shared__ __attribute__((aligned(16))) _Float16 s_st[320 * kStemLd];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 31, kg = lane >> 5;
  const int tiles_x = (Wp + kStemPT - 1) / kStemPT;
  const int py0 = (blockIdx.x / tiles_x) * kStemPT, px0 = (blockIdx.x % tiles_x) * kStemPT;
  const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;     // first stem pixel of the tile (may be -1: pooling padding)
  const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;     // first input pixel of the patch
  // weights: 22 fragments, once
  h8_t wf[2][11];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int s = 0; s < 11; ++s) wf[m][s] = *reinterpret_cast<const h8_t*>(wfrag + ((size_t)(m * 11 + s) * 64 + lane) * 8);
  // input patch -> LDS (fp16, zeros outside the image and in the pad column)
  for (int i = tid; i < 3 * kStemIT * kStemIW; i += 256) {
    const int c = i / (kStemIT * kStemIW), r = i - c * (kStemIT * kStemIW), py = r / kStemIW, px = r - py * kStemIW;
    const int iy = iy0 + py, ix = ix0 + px;
    float f = 0.f;
    if (px < kStemIT && iy >= 0 && iy < H && ix >= 0 && ix < W) f = x[((size_t)c * H + iy) * W + ix];
    s_in[i] = (_Float16)f;
  }
  __syncthreads();
  for (int nt = wave; nt < 10; nt += 4) {
    const int p = nt * 32 + n;                       // stem pixel of this lane (p >= 289: a spare slot, computed on pixel 288's window, never read)
    const int pc = p < kStemNpx ? p : kStemNpx - 1;
    const int sy = pc / kStemST, sx = pc - sy * kStemST;
    const _Float16* win = s_in + (2 * sy) * kStemIW + 2 * sx;   // window origin: even column -> 4-byte aligned
    f16x_t acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 11; ++s) {
      // patch row idx = 2 s + kg -> (c, ky); idx = 21 (kg = 1 of the last step) has zero weights: read row 20 again
      const int i0 = 2 * s, i1 = 2 * s + 1 < 21 ? 2 * s + 1 : 20;
      const int off0 = ((i0 / 7) * kStemIT + i0 % 7) * kStemIW, off1 = ((i1 / 7) * kStemIT + i1 % 7) * kStemIW;
      const unsigned* q = reinterpret_cast<const unsigned*>(win + (kg ? off1 : off0));
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
      const u4v raw = {q[0], q[1], q[2], q[3]};
      const h8_t b = __builtin_bit_cast(h8_t, raw);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = mfma32(wf[m][s], b, acc[m]);
    }
    // bias + ReLU -> fp16 -> s_st[p][channel]; stem pixels outside the map are 0
    const int gy = sy0 + sy, gx = sx0 + sx;
    const bool inside = p < kStemNpx && gy >= 0 && gy < Ho && gx >= 0 && gx < Wo;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = m * 32 + 8 * g + 4 * kg;      // accumulator register 4 g + e is channel row 8 g + 4 kg + e of the M-tile
        const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
        const float v0 = inside ? fmaxf(acc[m][4 * g + 0] + bv.x, 0.f) : 0.f, v1 = inside ? fmaxf(acc[m][4 * g + 1] + bv.y, 0.f) : 0.f;
        const float v2 = inside ? fmaxf(acc[m][4 * g + 2] + bv.z, 0.f) : 0.f, v3 = inside ? fmaxf(acc[m][4 * g + 3] + bv.w, 0.f) : 0.f;
        *reinterpret_cast<h4_t*>(s_st + p * kStemLd + ch) = to_h4(v0, v1, v2, v3);
      }
  }
  __syncthreads();
  // MaxPool2d(3, 2, 1): pooled pixel (py, px) of the tile = max over stem pixels (2 py + dy, 2 px + dx), dy, dx in 0..2 (tile-local: the
  // tile's stem origin is one pixel before the first window centre); 8 lanes per pooled pixel -> one whole 128-byte row per pixel
  for (int i = tid; i < kStemPT * kStemPT * 8; i += 256) {
    const int pp = i >> 3, cg = i & 7, py = pp / kStemPT, px = pp - py * kStemPT;
    if (py0 + py >= Hp || px0 + px >= Wp) continue;
    h8_t mx = *reinterpret_cast<const h8_t*>(s_st + ((2 * py) * kStemST + 2 * px) * kStemLd + cg * 8);
#pragma unroll
    for (int d = 1; d < 9; ++d) {
      const h8_t v = *reinterpret_cast<const h8_t*>(s_st + ((2 * py + d / 3) * kStemST + 2 * px + d % 3) * kStemLd + cg * 8);
      mx = __builtin_elementwise_max(mx, v);
    }
    *reinterpret_cast<h8_t*>(out + ((size_t)(py0 + py) * Wp + px0 + px) * 64 + cg * 8) = mx;
  }
}
void launch_ep_stem_pool(const float* x, int H, int W, int Ho, int Wo, int Hp, int Wp, const _Float16* wfrag, const float* bias, _Float16* out,
                         hipStream_t s) {
  const int tiles = ((Hp + kStemPT - 1) / kStemPT) * ((Wp + kStemPT - 1) / kStemPT);
  hipLaunchKernelGGL(k_ep_stem_pool, dim3(tiles), dim3(256), 0, s, x, H, W, Ho, Wo, Hp, Wp, wfrag, bias, out);
}

// aggregation tail: per-location L2 normalisation over the 512 channels, GeM(p, eps = 1e-6) over the npix locations, Linear(512 -> 512)
// (weights transposed [in][out] fp32), L2 normalisation -> fp32 [512].
// kTailWg = 8 workgroups of 512 threads, TWO launches (round 6).  Round 5 ran both halves in one kernel around a hand-rolled spin barrier in a
// plain launch: co-residency of the 8 workgroups was assumed, not requested - on the loop-closure thread the kernel shares the GPU with the
// tracker's persistent conv kernels, exactly where 8 workgroups need not start together, and an aborted call left the barrier counter poisoned
// (ADVICE r05).  The Linear needs all 512 pooled values, so the kernel boundary IS the barrier: ~1.5 us of launch gap instead
// of a spin that can last a whole foreign kernel.
//   k_ep_tail_pool (second form of round 6): ONE pass over the map by up to kTailPoolWg workgroups.  A wave owns a location at a time: one coalesced
//     1-KB row (8 channels per lane), wave_sum -> inverse norm, then clamp(x / ||x||, 1e-6)^p of its 8 channels accumulated in registers.  The
//     workgroup adds its 8 waves' sums in wave order and publishes [512] partial sums; the partials are added in workgroup order by the second
//     kernel - one fixed summation order for a given map size.  (The first form had each of 8 workgroups recompute all inverse norms and then
//     walk its 64 channels with 2-byte strided loads: 16.7 us for a 256-KB map.)  It also zeroes the arrival counter of the second kernel, so a
//     previous call that died mid-way cannot poison this one.
//   k_ep_tail_fc: (3) every workgroup finishes the GeM of all 512 channels ((mean)^(1/p): 512 threads, one channel each), then computes its 64
//     outputs, thread = (1/8 of the inputs, output) with its 64 weights all in flight: 128 KB of weights per workgroup; (4) the last workgroup
//     to ARRIVE (an atomic ticket - nobody waits) normalises.
constexpr int kTailWg = 8;
constexpr int kTailPoolWg = 32;
static int ep_tail_pool_wgs(int npix) { const int g = (npix + 7) / 8; return g < kTailPoolWg ? g : kTailPoolWg; }  // a function of the map size only
__global__ __launch_bounds__(512) void k_ep_tail_pool(const _Float16* __restrict__ feat, int npix, float p, float* __restrict__ g_part,
                                                      int* __restrict__ counters) {
  __shared__ float s_part[8 * 512];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, g = blockIdx.x, G = gridDim.x;
  if (g == 0 && t == 0) counters[0] = 0;  // k_ep_tail_fc's ticket (stream-ordered after this kernel)
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int px = g * 8 + wave; px < npix; px += 8 * G) {
    const h8_t v = *reinterpret_cast<const h8_t*>(feat + (size_t)px * 512 + lane * 8);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize(dim = channels): x / max(||x||, 1e-12)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = fmaxf((float)v[e] * inv, 1e-6f);
      acc[e] += __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x));  // x in [1e-6, 1]: no denormal handling needed (v_log_f32 / v_exp_f32)
    }
  }
  *reinterpret_cast<float4*>(s_part + wave * 512 + lane * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(s_part + wave * 512 + lane * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) sum += s_part[q * 512 + t];  // ascending wave: one fixed summation order
  g_part[(size_t)g * 512 + t] = sum;
}
__global__ __launch_bounds__(512) void k_ep_tail_fc(const float* __restrict__ g_part, int nparts, int npix, float p, const float* __restrict__ wt,
                                                    const float* __restrict__ bias, float* __restrict__ y_ws, int* __restrict__ counters,
                                                    float* __restrict__ out) {
  __shared__ float s_g[512], s_part[512], s_red[8];
  __shared__ int s_last;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, g = blockIdx.x;
  // (3) Linear: outputs [64 g, 64 g + 64); thread (part, output) covers inputs [64 part, 64 part + 64): its weights first (independent of the
  // pooled values: the 64 loads overlap the partial-sum reads), then the GeM of channel t
  const int j = 64 * g + lane, part = wave;
  float wv[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) wv[c] = wt[(size_t)(part * 64 + c) * 512 + j];
  {
    float pv[kTailPoolWg];   // all partial sums in flight (a loop over a run-time count waits for one L2 round trip per partial: 16 us)
#pragma unroll
    for (int q = 0; q < kTailPoolWg; ++q) pv[q] = q < nparts ? g_part[(size_t)q * 512 + t] : 0.f;  // written by the previous kernel: visible at the kernel boundary
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < kTailPoolWg; ++q) sum += pv[q];  // ascending workgroup order; the unused rows add +0
    s_g[t] = exp2f(log2f(sum / (float)npix) / p);  // GeM: (mean over locations of clamp(x, 1e-6)^p)^(1/p)
  }
  __syncthreads();
  float d = 0.f;
#pragma unroll
  for (int c = 0; c < 64; ++c) d = fmaf(wv[c], s_g[part * 64 + c], d);
  s_part[t] = d;
  __syncthreads();
  if (t < 64) {
    float y = bias[64 * g + t];
#pragma unroll
    for (int q = 0; q < 8; ++q) y += s_part[q * 64 + t];
    __hip_atomic_store(y_ws + 64 * g + t, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // (4) last workgroup to arrive: final L2 normalisation (a ticket, not a barrier: no workgroup ever waits for another)
  __threadfence();
  __syncthreads();
  if (t == 0) s_last = atomicAdd(counters, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const float y = __hip_atomic_load(y_ws + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // other workgroups' values: device-scope loads, not a stale L1 line
  const float ss = wave_sum(y * y);
  if (lane == 0) s_red[wave] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += s_red[w];
  out[t] = y / fmaxf(sqrtf(tot), 1e-12f);
}
// ws: [kTailPoolWg * 512 partial sums | 512 pre-normalisation outputs] floats (sship_ep_tail_ws_floats), counters: [>= 1] int (any value: the first
// kernel resets it).  Calls of one handle are stream-ordered (include/sship.h).
size_t ep_tail_ws_floats() { return (size_t)kTailPoolWg * 512 + 512; }
void launch_ep_tail(const _Float16* feat, int npix, float p, const float* wt, const float* bias, float* ws, int* counters, float* out,
                    hipStream_t s) {
  const int G = ep_tail_pool_wgs(npix);
  hipLaunchKernelGGL(k_ep_tail_pool, dim3(G), dim3(512), 0, s, feat, npix, p, ws, counters);
  hipLaunchKernelGGL(k_ep_tail_fc, dim3(kTailWg), dim3(512), 0, s, ws, G, npix, p, wt, bias, ws + (size_t)kTailPoolWg * 512, counters, out);
}

}  // namespace sship
