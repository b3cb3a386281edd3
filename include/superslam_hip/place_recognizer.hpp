// Host layer of the place recogniser's DESCRIPTOR SOURCE: the OpenCV-free half of the reference's EigenPlaces
// (include/EigenPlaces.h:19-30, src/EigenPlaces.cc:123-174) over the C ABI (sship_ep_*): initialize() returns bool,
// compute_global_descriptor returns an empty vector when the recogniser is not initialised; a descriptor is a std::vector<float> where
// the reference has a 1 x D CV_32F cv::Mat.
// The retrieval side - superslam::CosineDescriptorIndex, TemporalConsistencyVoter, IPlaceRecognizer::add / query (include/PlaceRecognizer.h,
// src/PlaceRecognizer.cc) - is GPU-free control plane that stays the reference's own code in libsuperslam_core: the reference-side
// adapter (integration/reference_side/EigenPlaces.h) holds a superslam::CosineDescriptorIndex exactly as include/EigenPlaces.h:30-36,62 does.
// Nothing of it is restated in this product (a restatement for the tests lives in oracle/eigenplaces_ref.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../sship.h"
#include "frontend.hpp"

namespace superslam_hip {

// ---- cv::resize(src, dst, Size(out_w, out_h), 0, 0, INTER_LINEAR) for CV_8UC{1,3}: OpenCV's fixed-point bilinear path
// (imgproc/resize.cpp: coordinate (d + 0.5) * scale - 0.5 in float, 11-bit coefficients saturate_cast<short>(w * 2048),
// horizontal pass in int, vertical pass (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2) ----
// source indices and 11-bit coefficients of one axis (n_src -> n_dst samples); shared by the host resize below and by the device
// kernel's tables (csrc/ep_kernels.hip: k_ep_resize_norm) - they depend only on the two sizes
inline void resize_bilinear_coeffs(int n_dst, int n_src, std::vector<int>& s0, std::vector<int>& s1, std::vector<int>& c0, std::vector<int>& c1) {
  s0.resize(n_dst); s1.resize(n_dst); c0.resize(n_dst); c1.resize(n_dst);
  const double scale = static_cast<double>(n_src) / n_dst;
  for (int d = 0; d < n_dst; ++d) {
    float f = static_cast<float>((d + 0.5) * scale - 0.5);
    int s = static_cast<int>(std::floor(f));
    f -= static_cast<float>(s);
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
    s0[d] = s; s1[d] = std::min(s + 1, n_src - 1);
    c1[d] = static_cast<int>(std::nearbyint(f * 2048.f));            // saturate_cast<short>: round half to even
    c0[d] = static_cast<int>(std::nearbyint((1.f - f) * 2048.f));
  }
}
inline void resize_bilinear_u8(const uint8_t* src, int h, int w, int stride, int ch, int out_h, int out_w, uint8_t* dst) {
  std::vector<int> sx, sx1, ax0, ax1, sy, sy1, by0, by1;
  resize_bilinear_coeffs(out_w, w, sx, sx1, ax0, ax1);
  resize_bilinear_coeffs(out_h, h, sy, sy1, by0, by1);
  std::vector<int> r0(static_cast<size_t>(out_w) * ch), r1(r0.size());
  auto hrow = [&](int y, std::vector<int>& row) {
    const uint8_t* p = src + static_cast<size_t>(y) * stride;
    for (int x = 0; x < out_w; ++x)
      for (int c = 0; c < ch; ++c) row[static_cast<size_t>(x) * ch + c] = p[sx[x] * ch + c] * ax0[x] + p[sx1[x] * ch + c] * ax1[x];
  };
  int cached0 = -1, cached1 = -1;
  for (int y = 0; y < out_h; ++y) {
    if (sy[y] != cached0) { hrow(sy[y], r0); cached0 = sy[y]; }
    if (sy1[y] != cached1) { hrow(sy1[y], r1); cached1 = sy1[y]; }
    for (size_t i = 0; i < r0.size(); ++i) {
      const int v = (((by0[y] * (r0[i] >> 4)) >> 16) + ((by1[y] * (r1[i] >> 4)) >> 16) + 2) >> 2;
      dst[static_cast<size_t>(y) * out_w * ch + i] = static_cast<uint8_t>(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
}

// EigenPlaces::preprocess (src/EigenPlaces.cc:123-145): gray -> RGB replicate / BGR -> RGB, resize, x 1/255, ImageNet
// normalisation, HWC -> CHW.  dst: [3][input_h][input_w] floats.
inline void eigenplaces_preprocess(const Image& image, int input_w, int input_h, float* dst) {
  static const float kMean[3] = {0.485f, 0.456f, 0.406f}, kStd[3] = {0.229f, 0.224f, 0.225f};
  const int stride = image.stride ? image.stride : image.cols * image.channels;
  std::vector<uint8_t> rs(static_cast<size_t>(input_h) * input_w * image.channels);
  resize_bilinear_u8(image.data, image.rows, image.cols, stride, image.channels, input_h, input_w, rs.data());  // per-channel: resize commutes with the channel shuffle
  const size_t hw = static_cast<size_t>(input_h) * input_w;
  for (size_t i = 0; i < hw; ++i)
    for (int c = 0; c < 3; ++c) {
      const uint8_t u = image.channels == 1 ? rs[i] : rs[i * 3 + (2 - c)];  // GRAY2RGB / BGR2RGB
      const float v = static_cast<float>(u) * (1.0f / 255.0f);               // convertTo(CV_32F, 1.0 / 255.0)
      dst[c * hw + i] = (v - kMean[c]) / kStd[c];
    }
}

typedef std::vector<float> GlobalDescriptor;

// ---- include/EigenPlaces.h ----
class EigenPlaces {
public:
  // `engine_file` names the safetensors weight file (utils/convert_eigenplaces_to_onnx.py:99 writes it next to the ONNX)
  EigenPlaces(const std::string& engine_file, int input_width, int input_height)
      : engine_file_(engine_file), input_width_(input_width), input_height_(input_height) {}
  ~EigenPlaces() { if (ep_) sship_ep_destroy(ep_); }
  EigenPlaces(const EigenPlaces&) = delete;
  EigenPlaces& operator=(const EigenPlaces&) = delete;
  bool initialize() {
    if (sship_ep_create(engine_file_.c_str(), input_width_, input_height_, &ep_) != SSHIP_OK) { last_error_ = sship_last_error(); ep_ = nullptr; return false; }
    return true;
  }
  GlobalDescriptor compute_global_descriptor(const Image& image) {
    if (!ep_) return GlobalDescriptor();  // `if (!context_) return cv::Mat();`
    // preprocessing (src/EigenPlaces.cc:123-145) runs on the device: the u8 image goes up (0.5 MB instead of 3 MB of fp32), the fixed-point
    // resize + normalisation is a kernel (bit-identical to eigenplaces_preprocess above, which stays as the host form the tests compare with)
    GlobalDescriptor d(static_cast<size_t>(sship_ep_descriptor_dim(ep_)));
    const int stride = image.stride ? image.stride : image.cols * image.channels;
    if (sship_ep_infer_u8(ep_, image.data, image.rows, image.cols, stride, image.channels, d.data()) != SSHIP_OK) { last_error_ = sship_last_error(); return GlobalDescriptor(); }
    double n = 0.0;  // cv::normalize(desc, desc, 1.0, 0.0, cv::NORM_L2)
    for (float v : d) n += static_cast<double>(v) * v;
    n = std::sqrt(n);
    if (n > 0) for (float& v : d) v = static_cast<float>(v / n);
    return d;
  }
  const std::string& last_error() const { return last_error_; }

private:
  std::string engine_file_;
  int input_width_, input_height_;
  sship_ep* ep_ = nullptr;
  std::string last_error_;
};

}  // namespace superslam_hip
