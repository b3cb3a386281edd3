// superslam_hip/frontend.hpp - header-only C++17 host layer above the C ABI (include/sship.h).
//
// Mirrors the reference's inference classes with OpenCV-free value types so that it builds anywhere:
//   superslam_hip::DescriptorPool / DeviceDescriptors   <- include/DescriptorPool.h:13-24,46-91 (the FreeList of :25-44 is sship_pool_*)
//   superslam_hip::Features, MatchResult, IFeatureExtractor, IFeatureMatcher <- include/InferenceInterfaces.h:12-59
//   superslam_hip::SuperPoint   <- include/SuperPoint.h:36-54   (ctor, initialize, infer, extract, extract_stereo)
//   superslam_hip::LightGlue    <- include/LightGlue.h:33-63    (ctors, initialize, shared_engine, match x3,
//                                                               descriptors_to_host)
// KeyPoint / DMatch carry exactly the cv::KeyPoint / cv::DMatch fields the reference reads, so the OpenCV
// adapter in integration/reference_side/ is a field-for-field copy.  Error behaviour is the reference's:
// initialize()/infer() return bool, interface methods never throw and return empty results on failure.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../sship.h"

namespace superslam_hip {

struct KeyPoint {  // cv::KeyPoint(x, y, size = 1, angle = -1, response)   (src/SuperPoint.cc:715)
  float x = 0, y = 0, size = 1, angle = -1, response = 0;
};
struct DMatch {    // cv::DMatch{queryIdx, trainIdx, distance}            (src/LightGlue.cc:355-359)
  int queryIdx = -1, trainIdx = -1;
  float distance = 0;
};
struct Image {     // what the adapter extracts from a cv::Mat: u8, 1 or 3 (BGR) channels
  const uint8_t* data = nullptr;
  int rows = 0, cols = 0, channels = 1;
  int stride = 0;  // bytes per row (0 -> cols * channels)
};
struct MatchResult {
  std::vector<DMatch> matches;
};

// ---- include/DescriptorPool.h -------------------------------------------------------------------
struct DeviceDescriptors {
  void* data = nullptr;
  int count = 0;
  int dim = 0;
  int slot = -1;
  std::shared_ptr<void> slot_ref;  // deleter releases `slot` back to the pool
  bool empty() const { return data == nullptr || count == 0; }
};

// Slot bookkeeping (the reference's FreeList, DescriptorPool.h:25-44: LIFO, -1 when exhausted) lives behind the C ABI:
// sship_pool_acquire / sship_pool_release / sship_pool_in_use.  There is no host-side copy of it here.

// Pool handle shared by every DeviceDescriptors copy.  It holds its own REFERENCE on the C-side bookkeeping
// (sship_pool_retain), so a handle may outlive the extractor object - whose sship_sp_destroy frees the device slots and
// drops only the extractor's reference - without its deleter touching freed memory (DescriptorPool.h:71-75: the
// deleter captures the shared FreeList, not `this`).  Device memory still dies with the pool (DescriptorPool.cc:27-32).
struct PoolRef {
  sship_pool* pool = nullptr;
  bool owned = false;   // this object created the pool (stand-alone DescriptorPool): it also frees the device slots
  ~PoolRef() {
    if (!pool) return;
    if (owned) sship_pool_destroy(pool);   // frees the slots, drops the creator's reference
    else sship_pool_release_ref(pool);     // borrowed from an extractor: drop the reference taken in the constructor
  }
};

class DescriptorPool {
public:
  DescriptorPool(int num_slots, int max_keypoints, int dim) : ref_(std::make_shared<PoolRef>()), dim_(dim), max_kp_(max_keypoints) {
    if (sship_pool_create(num_slots, max_keypoints, dim, &ref_->pool) == SSHIP_OK) ref_->owned = true;
  }
  explicit DescriptorPool(sship_pool* borrowed, int max_keypoints, int dim)
      : ref_(std::make_shared<PoolRef>()), dim_(dim), max_kp_(max_keypoints) {
    ref_->pool = borrowed;
    sship_pool_retain(borrowed);
  }
  DescriptorPool(const DescriptorPool&) = delete;
  DescriptorPool& operator=(const DescriptorPool&) = delete;

  DeviceDescriptors make(int count) { return wrap(sship_pool_acquire(ref_->pool), count); }
  // Wrap an already-acquired slot (the C ABI acquires inside sship_sp_extract*).
  DeviceDescriptors wrap(int slot, int count) {
    DeviceDescriptors d;
    d.count = count; d.dim = dim_; d.slot = slot;
    if (slot < 0) return d;  // exhausted
    d.data = sship_pool_slot_ptr(ref_->pool, slot);
    auto ref = ref_;
    d.slot_ref = std::shared_ptr<void>(d.data, [ref, slot](void*) { sship_pool_release(ref->pool, slot); });
    return d;
  }
  void* slot_ptr(int slot) const { return sship_pool_slot_ptr(ref_->pool, slot); }
  int dim() const { return dim_; }
  int max_keypoints() const { return max_kp_; }
  int in_use() const { return sship_pool_in_use(ref_->pool); }
  bool ok() const { return ref_->pool != nullptr; }

private:
  std::shared_ptr<PoolRef> ref_;
  int dim_, max_kp_;
};

// ---- include/InferenceInterfaces.h ---------------------------------------------------------------
struct Features {
  std::vector<KeyPoint> keypoints;
  DeviceDescriptors descriptors;  // [N x 256] fp16 in a pool slot
};
struct HostDescriptors {          // CV_32F [rows x cols], row-major (the loop-closure keyframe DB format)
  std::vector<float> data;
  int rows = 0, cols = 0;
  bool empty() const { return rows == 0; }
};

class IFeatureExtractor {
public:
  virtual ~IFeatureExtractor() = default;
  virtual Features extract(const Image& image) = 0;
  virtual std::pair<Features, Features> extract_stereo(const Image& left, const Image& right) {
    return {extract(left), extract(right)};
  }
};
class IFeatureMatcher {
public:
  virtual ~IFeatureMatcher() = default;
  virtual MatchResult match(const std::vector<KeyPoint>& kp0, const HostDescriptors& d0,
                            const std::vector<KeyPoint>& kp1, const HostDescriptors& d1) = 0;
  virtual MatchResult match(const std::vector<KeyPoint>& kp0, const DeviceDescriptors& d0,
                            const std::vector<KeyPoint>& kp1, const DeviceDescriptors& d1) = 0;
  virtual HostDescriptors descriptors_to_host(const DeviceDescriptors& d) = 0;
};

// ---- include/SuperPoint.h ------------------------------------------------------------------------
class SuperPoint : public IFeatureExtractor {
public:
  // `engine_file` names the safetensors weight file (the .engine's replacement, src/SuperSLAM.cc:72-78).
  SuperPoint(const std::string& engine_file, int max_keypoints, double keypoint_threshold, int remove_borders)
      : engine_file_(engine_file), max_keypoints_(max_keypoints), keypoint_threshold_(keypoint_threshold),
        remove_borders_(remove_borders) {}
  ~SuperPoint() override {
    pool_.reset();
    if (sp_) sship_sp_destroy(sp_);
  }
  SuperPoint(const SuperPoint&) = delete;
  SuperPoint& operator=(const SuperPoint&) = delete;

  bool initialize() {
    sship_sp_config cfg{};
    cfg.weights_path = engine_file_.c_str();
    cfg.max_keypoints = max_keypoints_;
    cfg.keypoint_threshold = keypoint_threshold_;
    cfg.remove_borders = remove_borders_;
    cfg.nms_radius = 4;
    cfg.pool_slots = descriptor_pool_slots;
    cfg.max_batch = 2;
    if (sship_sp_create(&cfg, &sp_) != SSHIP_OK) { last_error_ = sship_last_error(); sp_ = nullptr; return false; }
    pool_.reset(new DescriptorPool(sship_sp_pool(sp_), max_keypoints_, descriptor_dim));
    return true;
  }
  // Host path (src/SuperPoint.cc:322-348): keypoints + CV_32F [N,256] descriptors.
  bool infer(const Image& image, std::vector<KeyPoint>& keypoints, HostDescriptors& descriptors) {
    keypoints.clear();
    descriptors = HostDescriptors();
    if (!sp_) return false;
    std::vector<float> kp(3 * static_cast<size_t>(max_keypoints_)), desc(static_cast<size_t>(max_keypoints_) * descriptor_dim);
    int n = 0;
    if (sship_sp_infer_host(sp_, image.data, image.rows, image.cols, stride_of(image), image.channels, kp.data(),
                            desc.data(), &n) != SSHIP_OK) { last_error_ = sship_last_error(); return false; }
    fill(keypoints, kp.data(), n);
    desc.resize(static_cast<size_t>(n) * descriptor_dim);
    descriptors.data = std::move(desc); descriptors.rows = n; descriptors.cols = n ? descriptor_dim : 0;
    return true;
  }
  Features extract(const Image& image) override {
    Features f;
    if (!sp_) return f;
    std::vector<float> kp(3 * static_cast<size_t>(max_keypoints_));
    sship_features out{kp.data(), 0, nullptr, -1};
    if (sship_sp_extract(sp_, image.data, image.rows, image.cols, stride_of(image), image.channels, &out) != SSHIP_OK)
      last_error_ = sship_last_error();
    fill(f.keypoints, kp.data(), out.n);
    f.descriptors = pool_->wrap(out.slot, out.n);
    return f;
  }
  std::pair<Features, Features> extract_stereo(const Image& left, const Image& right) override {
    Features l, r;
    if (!sp_) return {l, r};
    if (left.rows != right.rows || left.cols != right.cols || left.channels != right.channels) {
      last_error_ = "SuperPoint: stereo pair must share resolution (rectified)";  // src/SuperPoint.cc:762-765
      return {l, r};
    }
    std::vector<float> kl(3 * static_cast<size_t>(max_keypoints_)), kr(kl.size());
    sship_features ol{kl.data(), 0, nullptr, -1}, orr{kr.data(), 0, nullptr, -1};
    if (sship_sp_extract_stereo(sp_, left.data, right.data, left.rows, left.cols, stride_of(left), left.channels, &ol,
                                &orr) != SSHIP_OK)
      last_error_ = sship_last_error();
    fill(l.keypoints, kl.data(), ol.n);
    fill(r.keypoints, kr.data(), orr.n);
    l.descriptors = pool_->wrap(ol.slot, ol.n);
    r.descriptors = pool_->wrap(orr.slot, orr.n);
    return {std::move(l), std::move(r)};
  }
  // Decode-ahead upload ring (sship_sp_ring_*, include/sship.h): a decoder thread fills ring_host(slot, 0 / 1) and calls
  // ring_upload(slot); the tracking thread then calls extract_stereo_ring(slot).
  bool ring_create(int depth, int rows, int cols, int channels) {
    if (!sp_ || sship_sp_ring_create(sp_, depth, rows, cols, channels) != SSHIP_OK) { last_error_ = sship_last_error(); return false; }
    return true;
  }
  uint8_t* ring_host(int slot, int image) { return sp_ ? sship_sp_ring_host(sp_, slot, image) : nullptr; }
  bool ring_upload(int slot) { return sp_ && sship_sp_ring_upload(sp_, slot) == SSHIP_OK; }
  // enqueue the extraction of an uploaded slot ahead of time (sship_sp_ring_submit): extract_stereo_ring(slot) then only waits
  bool ring_submit(int slot) {
    if (!sp_) return false;
    if (sship_sp_ring_submit(sp_, slot) != SSHIP_OK) { last_error_ = sship_last_error(); return false; }
    return true;
  }
  std::pair<Features, Features> extract_stereo_ring(int slot) {
    Features l, r;
    if (!sp_) return {l, r};
    std::vector<float> kl(3 * static_cast<size_t>(max_keypoints_)), kr(kl.size());
    sship_features ol{kl.data(), 0, nullptr, -1}, orr{kr.data(), 0, nullptr, -1};
    if (sship_sp_extract_stereo_ring(sp_, slot, &ol, &orr) != SSHIP_OK) last_error_ = sship_last_error();
    fill(l.keypoints, kl.data(), ol.n);
    fill(r.keypoints, kr.data(), orr.n);
    l.descriptors = pool_->wrap(ol.slot, ol.n);
    r.descriptors = pool_->wrap(orr.slot, orr.n);
    return {std::move(l), std::move(r)};
  }
  const std::string& last_error() const { return last_error_; }
  int pool_in_use() const { return pool_ ? pool_->in_use() : 0; }
  sship_sp* handle() const { return sp_; }

  static constexpr int descriptor_dim = 256;
  static constexpr int descriptor_pool_slots = 8;

private:
  static int stride_of(const Image& im) { return im.stride ? im.stride : im.cols * im.channels; }
  static void fill(std::vector<KeyPoint>& out, const float* kp, int n) {
    out.clear();
    out.reserve(n);
    for (int i = 0; i < n; ++i) {
      KeyPoint k;
      k.x = kp[3 * i]; k.y = kp[3 * i + 1]; k.size = 1.0f; k.angle = -1.0f; k.response = kp[3 * i + 2];
      out.push_back(k);
    }
  }
  std::string engine_file_;
  int max_keypoints_;
  double keypoint_threshold_;
  int remove_borders_;
  sship_sp* sp_ = nullptr;
  std::unique_ptr<DescriptorPool> pool_;
  std::string last_error_;
};
typedef std::shared_ptr<SuperPoint> SuperPointPtr;

// ---- include/LightGlue.h -------------------------------------------------------------------------
// Shareable weights: one load backs the tracking matcher and the loop-closure matcher (LightGlue.h:28-31).
struct LightGlueEngine {
  sship_lg_weights* weights = nullptr;
  ~LightGlueEngine() { if (weights) sship_lg_weights_release(weights); }
};

class LightGlue : public IFeatureMatcher {
public:
  LightGlue(const std::string& engine_file, int image_width, int image_height, int max_keypoints = 1024)
      : engine_file_(engine_file), image_width_(image_width), image_height_(image_height), max_keypoints_(max_keypoints) {}
  LightGlue(std::shared_ptr<LightGlueEngine> shared_engine, int image_width, int image_height, int max_keypoints = 1024)
      : image_width_(image_width), image_height_(image_height), max_keypoints_(max_keypoints), engine_(std::move(shared_engine)) {}
  ~LightGlue() override { if (lg_) sship_lg_destroy(lg_); }
  LightGlue(const LightGlue&) = delete;
  LightGlue& operator=(const LightGlue&) = delete;

  bool initialize() {
    if (!engine_) {
      auto e = std::make_shared<LightGlueEngine>();
      if (sship_lg_weights_load(engine_file_.c_str(), &e->weights) != SSHIP_OK) { last_error_ = sship_last_error(); return false; }
      engine_ = std::move(e);
    }
    if (!engine_->weights) return false;
    if (sship_lg_create(engine_->weights, image_width_, image_height_, max_keypoints_, 1, &lg_) != SSHIP_OK) {
      last_error_ = sship_last_error(); lg_ = nullptr; return false;
    }
    return true;
  }
  std::shared_ptr<LightGlueEngine> shared_engine() const { return engine_; }

  // 5-argument form (src/LightGlue.cc:285-324): false for an uninitialised matcher or an empty set.
  bool match(const std::vector<KeyPoint>& kp0, const HostDescriptors& d0, const std::vector<KeyPoint>& kp1,
             const HostDescriptors& d1, MatchResult& result) {
    result.matches.clear();
    if (!lg_) return false;
    const int n0 = static_cast<int>(kp0.size()), n1 = static_cast<int>(kp1.size());
    if (n0 == 0 || n1 == 0) return false;
    std::vector<int32_t> m0(n0);
    std::vector<float> ms0(n0);
    if (sship_lg_match_host(lg_, &kp0[0].x, kp_stride, n0, d0.data.data(), &kp1[0].x, kp_stride, n1, d1.data.data(),
                            m0.data(), ms0.data()) != SSHIP_OK) { last_error_ = sship_last_error(); return false; }
    postprocess(m0, ms0, result);
    return true;
  }
  MatchResult match(const std::vector<KeyPoint>& kp0, const HostDescriptors& d0, const std::vector<KeyPoint>& kp1,
                    const HostDescriptors& d1) override {
    MatchResult r;
    match(kp0, d0, kp1, d1, r);
    return r;
  }
  MatchResult match(const std::vector<KeyPoint>& kp0, const DeviceDescriptors& d0, const std::vector<KeyPoint>& kp1,
                    const DeviceDescriptors& d1) override {
    MatchResult r;
    if (!lg_ || d0.empty() || d1.empty()) return r;  // src/LightGlue.cc:381-383
    const int n0 = static_cast<int>(kp0.size()), n1 = static_cast<int>(kp1.size());
    if (n0 == 0 || n1 == 0) return r;
    std::vector<int32_t> m0(n0);
    std::vector<float> ms0(n0);
    if (sship_lg_match_device(lg_, &kp0[0].x, kp_stride, n0, d0.data, &kp1[0].x, kp_stride, n1, d1.data, m0.data(),
                              ms0.data()) != SSHIP_OK) { last_error_ = sship_last_error(); return r; }
    postprocess(m0, ms0, r);
    return r;
  }
  HostDescriptors descriptors_to_host(const DeviceDescriptors& d) override {
    HostDescriptors out;
    if (d.empty()) return out;  // empty handle -> empty Mat
    out.data.resize(static_cast<size_t>(d.count) * d.dim);
    if (sship_desc_to_host(d.data, d.count, d.dim, out.data.data()) != SSHIP_OK) { last_error_ = sship_last_error(); return HostDescriptors(); }
    out.rows = d.count; out.cols = d.dim;
    return out;
  }
  const std::string& last_error() const { return last_error_; }

private:
  static constexpr int kp_stride = static_cast<int>(sizeof(KeyPoint) / sizeof(float));
  static void postprocess(const std::vector<int32_t>& m0, const std::vector<float>& ms0, MatchResult& r) {
    const int n0 = static_cast<int>(m0.size());
    std::vector<int> q(n0), t(n0);
    std::vector<float> dist(n0);
    const int k = sship_filter_matches(m0.data(), ms0.data(), n0, q.data(), t.data(), dist.data());
    for (int i = 0; i < k; ++i) { DMatch dm; dm.queryIdx = q[i]; dm.trainIdx = t[i]; dm.distance = dist[i]; r.matches.push_back(dm); }
  }
  std::string engine_file_;
  int image_width_, image_height_, max_keypoints_;
  std::shared_ptr<LightGlueEngine> engine_;
  sship_lg* lg_ = nullptr;
  std::string last_error_;
};
typedef std::shared_ptr<LightGlue> LightGluePtr;

}  // namespace superslam_hip
