// Reference-side binding (goes into the SuperSLAM tree as include/LightGlue.h; replaces the TensorRT runner).
// Same public surface as include/LightGlue.h:28-63: LightGlueEngine, both constructors, initialize(),
// shared_engine(), the 5-argument match, the two IFeatureMatcher overloads and descriptors_to_host.
#ifndef LIGHTGLUE_HIP_ADAPTER_H_
#define LIGHTGLUE_HIP_ADAPTER_H_

#include <memory>
#include <opencv4/opencv2/opencv.hpp>
#include <string>
#include <vector>

#include "InferenceInterfaces.h"  // MatchResult, superslam::IFeatureMatcher (the reference's own header)
#include "Logging.h"
#include "SshipLogForward.h"   // library log callback -> SLOG_* (include/Logging.h:21-26)
#include "superslam_hip/frontend.hpp"

typedef superslam_hip::LightGlueEngine LightGlueEngine;  // shareable weights (one load, many matchers)

class LightGlue : public superslam::IFeatureMatcher {
public:
  explicit LightGlue(const std::string& engine_file, int image_width, int image_height)
      : impl_(engine_file, image_width, image_height) { superslam_hip_adapter::install_log_forwarding(); }
  LightGlue(std::shared_ptr<LightGlueEngine> shared_engine, int image_width, int image_height)
      : impl_(std::move(shared_engine), image_width, image_height) { superslam_hip_adapter::install_log_forwarding(); }
  bool initialize() {
    const bool ok = impl_.initialize();
    if (!ok) SLOG_ERROR("LightGlue(HIP): {}", impl_.last_error());
    return ok;
  }
  std::shared_ptr<LightGlueEngine> shared_engine() const { return impl_.shared_engine(); }

  bool match(const std::vector<cv::KeyPoint>& kp0, const cv::Mat& d0, const std::vector<cv::KeyPoint>& kp1,
             const cv::Mat& d1, MatchResult& result) {
    superslam_hip::MatchResult r;
    const bool ok = impl_.match(from_cv(kp0), from_cv(d0), from_cv(kp1), from_cv(d1), r);
    to_cv(r, result);
    return ok;
  }
  MatchResult match(const std::vector<cv::KeyPoint>& kp0, const cv::Mat& d0, const std::vector<cv::KeyPoint>& kp1,
                    const cv::Mat& d1) override {
    MatchResult r;
    match(kp0, d0, kp1, d1, r);
    return r;
  }
  MatchResult match(const std::vector<cv::KeyPoint>& kp0, const superslam::DeviceDescriptors& d0,
                    const std::vector<cv::KeyPoint>& kp1, const superslam::DeviceDescriptors& d1) override {
    MatchResult out;
    to_cv(impl_.match(from_cv(kp0), from_ref(d0), from_cv(kp1), from_ref(d1)), out);
    return out;
  }
  cv::Mat descriptors_to_host(const superslam::DeviceDescriptors& d) override {
    superslam_hip::HostDescriptors h = impl_.descriptors_to_host(from_ref(d));
    return h.rows ? cv::Mat(h.rows, h.cols, CV_32F, h.data.data()).clone() : cv::Mat();
  }

private:
  static std::vector<superslam_hip::KeyPoint> from_cv(const std::vector<cv::KeyPoint>& in) {
    std::vector<superslam_hip::KeyPoint> out(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
      out[i].x = in[i].pt.x; out[i].y = in[i].pt.y; out[i].size = in[i].size; out[i].angle = in[i].angle;
      out[i].response = in[i].response;
    }
    return out;
  }
  static superslam_hip::HostDescriptors from_cv(const cv::Mat& m) {
    superslam_hip::HostDescriptors h;
    cv::Mat f;
    if (m.type() == CV_32F) f = m.isContinuous() ? m : m.clone(); else m.convertTo(f, CV_32F);  // LightGlue.cc:268-275
    h.rows = f.rows; h.cols = f.cols;
    h.data.assign(reinterpret_cast<const float*>(f.data), reinterpret_cast<const float*>(f.data) + f.total());
    return h;
  }
  static superslam_hip::DeviceDescriptors from_ref(const superslam::DeviceDescriptors& d) {
    superslam_hip::DeviceDescriptors o;
    o.data = d.data; o.count = d.count; o.dim = d.dim; o.slot = d.slot; o.slot_ref = d.slot_ref;
    return o;
  }
  static void to_cv(const superslam_hip::MatchResult& in, MatchResult& out) {
    out.matches.clear();
    for (const auto& m : in.matches) {
      cv::DMatch dm;
      dm.queryIdx = m.queryIdx; dm.trainIdx = m.trainIdx; dm.distance = m.distance;
      out.matches.push_back(dm);
    }
  }
  superslam_hip::LightGlue impl_;
};
typedef std::shared_ptr<LightGlue> LightGluePtr;
#endif
