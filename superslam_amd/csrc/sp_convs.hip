// SuperPoint's MFMA layers: instantiations of the implicit-GEMM template (igemm.h).
//   conv1b..conv4b, convPa, convDa : 3x3, bias + ReLU (+ fused 2x2 max-pool after 1b/2b/3b)   [MFMA-bound]
//   convDb : 1x1 256->256, raw fp16 (the L2 normalisation is applied where the grid is consumed)
//   convPb : 1x1 256->65, fp32 logits in a 128-wide row (softmax happens in the heatmap kernel)
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

hipError_t sp_conv3x3(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, bool pool, bool relu,
                      hipStream_t s) {
  IgemmArgs a = conv_args(w, in, B, H, W);
  a.out0 = out;
  if (!relu) return hipErrorInvalidValue;
  if (w.cin == 64) {
    if (pool) return launch_igemm<3, 64, 64, 8, EpiF16<true, true>>(a, w.cout_pad, s);
    return launch_igemm<3, 64, 64, 8, EpiF16<true, false>>(a, w.cout_pad, s);
  }
  if (w.cin == 128) {
    if (pool) return launch_igemm<3, 128, 64, 8, EpiF16<true, true>>(a, w.cout_pad, s);
    return launch_igemm<3, 128, 64, 8, EpiF16<true, false>>(a, w.cout_pad, s);
  }
  return hipErrorInvalidValue;
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

}  // namespace sship
