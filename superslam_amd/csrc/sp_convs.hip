// SuperPoint's 1x1 layers on the implicit-GEMM template (igemm.h) and the fused descriptor head.
// (The 3x3 layers live in conv_strip.hip / conv_pp.hip.)
//   convDb : 1x1 256->256, raw fp16 - dense grid only for sship_sp_dense; extraction uses k_desc_head_gather below
//   convPb : 1x1 256->65, fp32 logits in an 80-wide row (softmax happens in the heatmap kernel)
// reference: utils/convert_superpoint_to_onnx.py:38-64,77,88.
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

hipError_t sp_conv1x1_f32(const ConvW& w, const _Float16* in, float* out, int ostride, int B, int H, int W,
                          hipStream_t s) {
  IgemmArgs a = conv_args(w, in, B, H, W);
  a.out0 = out; a.ostride = ostride;
  if (w.cin != 256) return hipErrorInvalidValue;
  return launch_igemm<1, 256, 128, 4, EpiF32>(a, w.cout_pad, s);
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

void launch_desc_head_gather(const ConvW& db32, const _Float16* da, int Hc, int Wc, const int* cell_h, const int* cell_w,
                             const int* n_dev, int max_kp, int B, _Float16* out, size_t out_img_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_desc_head_gather, dim3((max_kp + 63) / 64, B), dim3(512), 0, s, da, Hc, Wc, cell_h, cell_w, n_dev,
                     max_kp, db32.w, db32.bias, out, out_img_stride);
}

}  // namespace sship
