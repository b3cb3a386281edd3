// Reference-side binding (goes into the SuperSLAM tree as include/EigenPlaces.h; replaces the TensorRT runner).
// Same class name, constructor and methods as the reference's header (include/EigenPlaces.h:19-40): SuperSLAM.cc:116-133 and
// LoopCloser compile and run unchanged against superslam::IPlaceRecognizer.  All work is forwarded to libsuperslam_hip.so
// through include/superslam_hip/place_recognizer.hpp -> include/sship.h (sship_ep_*).
#ifndef EIGENPLACES_HIP_ADAPTER_H_
#define EIGENPLACES_HIP_ADAPTER_H_

#include <memory>
#include <opencv4/opencv2/core.hpp>
#include <string>
#include <vector>

#include "Logging.h"
#include "PlaceRecognizer.h"  // the reference's own header (unchanged): superslam::IPlaceRecognizer, LoopCandidate
#include "superslam_hip/place_recognizer.hpp"

class EigenPlaces : public superslam::IPlaceRecognizer {
public:
  EigenPlaces(const std::string& engine_file, int input_width, int input_height) : impl_(engine_file, input_width, input_height) {}
  bool initialize() {
    const bool ok = impl_.initialize();
    if (!ok) SLOG_ERROR("EigenPlaces(HIP): {}", impl_.last_error());
    return ok;
  }
  cv::Mat compute_global_descriptor(const cv::Mat& image) override {
    cv::Mat keep = image.isContinuous() ? image : image.clone();
    if (keep.depth() != CV_8U) keep.convertTo(keep, CV_8U);
    const superslam_hip::GlobalDescriptor d = impl_.compute_global_descriptor(
        superslam_hip::Image{keep.data, keep.rows, keep.cols, keep.channels(), static_cast<int>(keep.step)});
    if (d.empty()) return cv::Mat();
    cv::Mat out(1, static_cast<int>(d.size()), CV_32F);
    for (size_t i = 0; i < d.size(); ++i) out.ptr<float>(0)[i] = d[i];
    return out;
  }
  void add(size_t keyframe_id, const cv::Mat& global_descriptor) override { impl_.add(keyframe_id, to_vec(global_descriptor)); }
  std::vector<superslam::LoopCandidate> query(const cv::Mat& global_descriptor, size_t excludeRecent, int topK) override {
    std::vector<superslam::LoopCandidate> out;
    for (const auto& c : impl_.query(to_vec(global_descriptor), excludeRecent, topK)) {
      superslam::LoopCandidate lc;
      lc.keyframe_id = c.keyframe_id; lc.score = c.score;
      out.push_back(lc);
    }
    return out;
  }

private:
  static superslam_hip::GlobalDescriptor to_vec(const cv::Mat& m) {
    cv::Mat f;
    if (m.type() == CV_32F) f = m; else m.convertTo(f, CV_32F);
    superslam_hip::GlobalDescriptor v(f.total());
    for (int r = 0, k = 0; r < f.rows; ++r)
      for (int c = 0; c < f.cols; ++c) v[k++] = f.ptr<float>(r)[c];
    return v;
  }
  superslam_hip::EigenPlaces impl_;
};
#endif
