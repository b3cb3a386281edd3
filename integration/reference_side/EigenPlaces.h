// Reference-side binding (goes into the SuperSLAM tree as include/EigenPlaces.h; replaces the TensorRT runner).
// Same class name, constructor and methods as the reference's header (include/EigenPlaces.h:19-40): SuperSLAM.cc:116-133 and
// LoopCloser compile and run unchanged against superslam::IPlaceRecognizer.  The descriptor is computed by libsuperslam_hip.so
// (include/superslam_hip/place_recognizer.hpp -> include/sship.h, sship_ep_*); retrieval stays the reference's own
// superslam::CosineDescriptorIndex (src/PlaceRecognizer.cc in libsuperslam_core), held and used exactly as include/EigenPlaces.h:30-36,53,62.
#ifndef EIGENPLACES_HIP_ADAPTER_H_
#define EIGENPLACES_HIP_ADAPTER_H_

#include <cstdlib>
#include <memory>
#include <opencv4/opencv2/core.hpp>
#include <string>
#include <vector>

#include "Logging.h"
#include "SshipLogForward.h"   // library log callback -> SLOG_* (include/Logging.h:21-26)
#include "PlaceRecognizer.h"  // the reference's own header (unchanged): superslam::IPlaceRecognizer, LoopCandidate
#include "superslam_hip/place_recognizer.hpp"

class EigenPlaces : public superslam::IPlaceRecognizer {
public:
  EigenPlaces(const std::string& engine_file, int input_width, int input_height) : impl_(engine_file, input_width, input_height) {
    superslam_hip_adapter::install_log_forwarding();
    if (const char* s = std::getenv("SUPERSLAM_LOOP_MIN_SCORE")) min_score_ = static_cast<float>(std::atof(s));  // src/EigenPlaces.cc:33-34
  }
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
  void add(size_t keyframe_id, const cv::Mat& global_descriptor) override { index_.add(keyframe_id, global_descriptor); }
  std::vector<superslam::LoopCandidate> query(const cv::Mat& global_descriptor, size_t excludeRecent, int topK) override {
    return index_.query(global_descriptor, excludeRecent, topK, min_score_);
  }

private:
  superslam_hip::EigenPlaces impl_;
  float min_score_ = 0.75f;                  // include/EigenPlaces.h:53
  superslam::CosineDescriptorIndex index_;   // include/EigenPlaces.h:62
};
#endif
