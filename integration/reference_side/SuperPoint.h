// Reference-side binding (goes into the SuperSLAM tree as include/SuperPoint.h; replaces the TensorRT runner).
// Same class name, constructor and methods as the reference's header (include/SuperPoint.h:36-54), so
// SuperSLAM.cc:68-87, StereoFrontEnd, RgbdFrontEnd and VoEstimator compile and run unchanged.  All work is
// forwarded to libsuperslam_hip.so through include/superslam_hip/frontend.hpp -> include/sship.h.
// Needs OpenCV (as the reference does); not built in this repository's image (no OpenCV here) - the OpenCV-free
// layer it wraps is built and tested in tests/cpp/.
#ifndef SUPERPOINT_HIP_ADAPTER_H_
#define SUPERPOINT_HIP_ADAPTER_H_

#include <memory>
#include <opencv4/opencv2/opencv.hpp>
#include <string>
#include <vector>

#include "DescriptorPool.h"       // the reference's own header (unchanged): superslam::DeviceDescriptors
#include "InferenceInterfaces.h"  // the reference's own header (unchanged): IFeatureExtractor, Features
#include "Logging.h"
#include "SshipLogForward.h"   // library log callback -> SLOG_* (include/Logging.h:21-26)
#include "Profiling.h"
#include "superslam_hip/frontend.hpp"

namespace superslam_hip_adapter {
inline superslam_hip::Image as_image(const cv::Mat& m, cv::Mat& keep) {
  keep = m.isContinuous() ? m : m.clone();
  if (keep.depth() != CV_8U) keep.convertTo(keep, CV_8U);
  return superslam_hip::Image{keep.data, keep.rows, keep.cols, keep.channels(), static_cast<int>(keep.step)};
}
inline void to_cv(const std::vector<superslam_hip::KeyPoint>& in, std::vector<cv::KeyPoint>& out) {
  out.clear();
  out.reserve(in.size());
  for (const auto& k : in) out.emplace_back(k.x, k.y, k.size, k.angle, k.response);
}
inline superslam::DeviceDescriptors to_ref(const superslam_hip::DeviceDescriptors& d) {
  superslam::DeviceDescriptors o;
  o.data = d.data; o.count = d.count; o.dim = d.dim; o.slot = d.slot; o.slot_ref = d.slot_ref;
  return o;
}
// The reference's SUPERSLAM_PROFILE label "sp_gpu_infer" (src/SuperPoint.cc:639: enqueue + scores D2H + sync of the mono
// infer_device path).  Here the network, the selection and the gather are one asynchronous sequence, so the label receives
// the DEVICE time of the library's sp_gpu_infer:* stages (hipEvents inside the library, sship_get_stage_timings) - what the
// reference's wall-clock bracket measured minus its host waits.  Emitted for extract() (the path the reference labels) and
// for extract_stereo() (one batch-2 pass; n = 1 per call) so a SUPERSLAM_PROFILE=1 run keeps every label it had.
inline void profile_begin() {
  if (superslam::Profiler::enabled()) sship_set_profiling(1);
}
inline void profile_emit_gpu_infer() {
  if (!superslam::Profiler::enabled()) return;
  const char* labels[32];
  float ms[32];
  const int n = sship_get_stage_timings(labels, ms, 32);
  double gpu = 0.0;
  for (int i = 0; i < n; ++i)
    if (std::string(labels[i]).compare(0, 13, "sp_gpu_infer:") == 0) gpu += ms[i];
  if (n > 0) superslam::Profiler::instance().add("sp_gpu_infer", gpu);
}
inline superslam::Features to_ref(superslam_hip::Features&& f) {
  superslam::Features o;
  to_cv(f.keypoints, o.keypoints);
  o.descriptors = to_ref(f.descriptors);
  return o;
}
}  // namespace superslam_hip_adapter

class SuperPoint : public superslam::IFeatureExtractor {
public:
  explicit SuperPoint(const std::string& engine_file, int max_keypoints, double keypoint_threshold, int remove_borders)
      : impl_(engine_file, max_keypoints, keypoint_threshold, remove_borders) { superslam_hip_adapter::install_log_forwarding(); }
  bool initialize() {
    const bool ok = impl_.initialize();
    if (!ok) SLOG_ERROR("SuperPoint(HIP): {}", impl_.last_error());
    return ok;
  }
  bool infer(const cv::Mat& image, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors) {
    cv::Mat keep;
    std::vector<superslam_hip::KeyPoint> kp;
    superslam_hip::HostDescriptors d;
    if (!impl_.infer(superslam_hip_adapter::as_image(image, keep), kp, d)) return false;
    superslam_hip_adapter::to_cv(kp, keypoints);
    descriptors = d.rows ? cv::Mat(d.rows, d.cols, CV_32F, d.data.data()).clone() : cv::Mat();
    return true;
  }
  superslam::Features extract(const cv::Mat& image) override {
    cv::Mat keep;
    superslam_hip_adapter::profile_begin();
    auto f = impl_.extract(superslam_hip_adapter::as_image(image, keep));
    superslam_hip_adapter::profile_emit_gpu_infer();  // src/SuperPoint.cc:639
    return superslam_hip_adapter::to_ref(std::move(f));
  }
  std::pair<superslam::Features, superslam::Features> extract_stereo(const cv::Mat& left, const cv::Mat& right) override {
    SUPERSLAM_PROFILE_SCOPE("sp_extract_stereo");
    cv::Mat kl, kr;
    superslam_hip_adapter::profile_begin();
    auto lr = impl_.extract_stereo(superslam_hip_adapter::as_image(left, kl), superslam_hip_adapter::as_image(right, kr));
    superslam_hip_adapter::profile_emit_gpu_infer();
    return {superslam_hip_adapter::to_ref(std::move(lr.first)), superslam_hip_adapter::to_ref(std::move(lr.second))};
  }

private:
  superslam_hip::SuperPoint impl_;
};
typedef std::shared_ptr<SuperPoint> SuperPointPtr;
#endif
