// The reference-side binding, COMPILED: integration/reference_side/{SuperPoint,LightGlue}.h (this repository's adapters with the
// reference's class names) + the reference's own unmodified headers and src/StereoFrontEnd.cc (from /root/reference at build
// time), against the stand-in OpenCV / GTSAM / spdlog declarations of tests/cpp/shim (this image has none of the three).
//   no arguments  : CPU checks - the adapters are concrete IFeatureExtractor / IFeatureMatcher implementations, the reference's
//                   two StereoFrontEnd cases (tests/test_stereo_frontend.cc:49-73) hold on the compiled reference source, and
//                   initialize() fails cleanly without a GPU;
//   <sp> <lg> args: GPU - the same two cases (disparity -> uR, zero disparity -> no depth) through the REAL adapters:
//                   StereoFrontEnd::process(SuperPoint(HIP), LightGlue(HIP)) on a synthetic shifted pair.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "EigenPlaces.h"     // integration/reference_side (first on the include path)
#include "LightGlue.h"       // integration/reference_side
#include "SuperPoint.h"      // integration/reference_side
#include "StereoFrontEnd.h"  // /root/reference/include

// src/Logging.cc needs spdlog's sink headers; the two statics are all the binding uses
std::shared_ptr<spdlog::logger> superslam::Logger::logger_;
bool superslam::Logger::initialized_ = false;
void superslam::Logger::initialize() { if (!logger_) logger_ = std::make_shared<spdlog::logger>(); initialized_ = true; }
std::shared_ptr<spdlog::logger> superslam::Logger::getLogger() { if (!logger_) initialize(); return logger_; }

static int g_fail = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } } while (0)

using namespace superslam;

namespace {
// the reference test's mocks (tests/test_stereo_frontend.cc:12-46): left/right keypoints on successive extract() calls,
// identity index matcher
struct AlternatingExtractor : IFeatureExtractor {
  float disparity; int call = 0;
  explicit AlternatingExtractor(float d) : disparity(d) {}
  Features extract(const cv::Mat&) override {
    Features f;
    const float dx = (call++ % 2 == 0) ? 0.f : disparity;
    f.keypoints = {cv::KeyPoint(100.f - dx, 50.f, 1), cv::KeyPoint(200.f - dx, 80.f, 1)};
    return f;
  }
};
struct IdMatcher : IFeatureMatcher {
  static MatchResult identity(size_t na, size_t nb) {
    MatchResult r;
    for (int i = 0; i < static_cast<int>(std::min(na, nb)); ++i) r.matches.push_back(cv::DMatch(i, i, 0.f));
    return r;
  }
  MatchResult match(const std::vector<cv::KeyPoint>& a, const cv::Mat&, const std::vector<cv::KeyPoint>& b, const cv::Mat&) override { return identity(a.size(), b.size()); }
  MatchResult match(const std::vector<cv::KeyPoint>& a, const DeviceDescriptors&, const std::vector<cv::KeyPoint>& b, const DeviceDescriptors&) override { return identity(a.size(), b.size()); }
  cv::Mat descriptors_to_host(const DeviceDescriptors&) override { return cv::Mat(); }
};

// procedural texture: blocks of random grey on a noisy ramp (deterministic LCG), blurred once
cv::Mat make_image(int h, int w, unsigned seed) {
  cv::Mat m(h, w, CV_8UC1);
  unsigned s = seed * 2654435761u + 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  std::vector<float> f(static_cast<size_t>(h) * w);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) f[static_cast<size_t>(y) * w + x] = 60.f + 40.f * std::sin(x * 0.013f) + 30.f * std::cos(y * 0.021f) + (rnd() % 16);
  for (int r = 0; r < 260; ++r) {
    const int rh = 4 + rnd() % (h / 6), rw = 4 + rnd() % (w / 10), y0 = rnd() % (h - rh), x0 = rnd() % (w - rw);
    const float v = 20.f + (rnd() % 200);
    for (int y = y0; y < y0 + rh; ++y) for (int x = x0; x < x0 + rw; ++x) f[static_cast<size_t>(y) * w + x] = 0.35f * f[static_cast<size_t>(y) * w + x] + 0.65f * v;
  }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float a = 0.f; int n = 0;
      for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) { a += f[static_cast<size_t>(yy) * w + xx]; ++n; }
      }
      const float v = a / n;
      m.ptr<unsigned char>(y)[x] = static_cast<unsigned char>(v < 0 ? 0 : v > 255 ? 255 : v + 0.5f);
    }
  return m;
}
cv::Mat shift_left(const cv::Mat& src, int d) {  // right image of a fronto-parallel scene: x_R = x_L - d
  cv::Mat m(src.rows, src.cols, CV_8UC1);
  for (int y = 0; y < src.rows; ++y)
    for (int x = 0; x < src.cols; ++x) m.ptr<unsigned char>(y)[x] = src.ptr<unsigned char>(y)[std::min(x + d, src.cols - 1)];
  return m;
}
}  // namespace

static void cpu_checks() {
  static_assert(std::is_base_of<IFeatureExtractor, SuperPoint>::value && !std::is_abstract<SuperPoint>::value, "SuperPoint adapter");
  static_assert(std::is_base_of<IFeatureMatcher, LightGlue>::value && !std::is_abstract<LightGlue>::value, "LightGlue adapter");
  static_assert(std::is_same<SuperPointPtr, std::shared_ptr<SuperPoint>>::value && std::is_same<LightGluePtr, std::shared_ptr<LightGlue>>::value, "typedefs");
  auto K = gtsam::Cal3_S2Stereo(500, 500, 0, 320, 240, 0.5);
  {  // tests/test_stereo_frontend.cc:49-63 on the compiled reference StereoFrontEnd.cc
    AlternatingExtractor ext(10.f); IdMatcher matcher;
    StereoFrontEnd fe(&ext, &matcher, K, 1.0f);
    StereoFrame f = fe.process(cv::Mat::zeros(480, 640, CV_8U), cv::Mat::zeros(480, 640, CV_8U), 1.0);
    CHECK(f.keypoints_left.size() == 2 && f.has_depth[0] == 1);
    CHECK(std::fabs(f.stereo[0].uL() - 100.0) < 1e-6 && std::fabs(f.stereo[0].uR() - 90.0) < 1e-6 && std::fabs(f.stereo[0].v() - 50.0) < 1e-6);
  }
  {  // :65-73
    AlternatingExtractor ext(0.f); IdMatcher matcher;
    StereoFrontEnd fe(&ext, &matcher, K, 1.0f);
    StereoFrame f = fe.process(cv::Mat::zeros(480, 640, CV_8U), cv::Mat::zeros(480, 640, CV_8U), 1.0);
    CHECK(f.has_depth[0] == 0);
  }
  // error conventions of the boundary: a missing weights file (or no GPU) -> initialize() returns false, extract() returns
  // empty Features, match() an empty MatchResult - nothing throws (src/SuperPoint.cc:895-899, src/LightGlue.cc:381-391)
  SuperPoint sp("/nonexistent/superpoint.safetensors", 600, 0.005, 4);
  CHECK(!sp.initialize());
  Features f = sp.extract(cv::Mat::zeros(64, 64, CV_8U));
  CHECK(f.keypoints.empty() && f.descriptors.empty());
  LightGlue lg("/nonexistent/lightglue.safetensors", 640, 480);
  CHECK(!lg.initialize());
  MatchResult r = lg.match({cv::KeyPoint(1, 1, 1)}, DeviceDescriptors(), {cv::KeyPoint(1, 1, 1)}, DeviceDescriptors());
  CHECK(r.matches.empty());
  CHECK(lg.descriptors_to_host(DeviceDescriptors()).empty());
  // the place recogniser adapter is a concrete superslam::IPlaceRecognizer with the reference's constructor
  static_assert(std::is_base_of<IPlaceRecognizer, EigenPlaces>::value && !std::is_abstract<EigenPlaces>::value, "EigenPlaces adapter");
  EigenPlaces ep("/nonexistent/eigenplaces.safetensors", 512, 512);
  CHECK(!ep.initialize());
  CHECK(ep.compute_global_descriptor(cv::Mat::zeros(64, 64, CV_8U)).empty());      // `if (!context_) return cv::Mat();`
  CHECK(ep.query(cv::Mat::zeros(1, 512, CV_32F), 0, 5).empty());
  // add / query of the adapter run the reference's OWN retrieval code (src/PlaceRecognizer.cc, compiled from /root/reference into this
  // binary): the index cases of the reference's tests/test_place_recognizer.cc:22-80 through the IPlaceRecognizer interface
  auto desc = [](int dim, int seed, float jitter) { cv::Mat d = cv::Mat::zeros(1, dim, CV_32F); d.at<float>(0, seed % dim) = 1.0f; d.at<float>(0, (seed + 1) % dim) = 0.5f + jitter; return d; };
  {
    setenv("SUPERSLAM_LOOP_MIN_SCORE", "0.0", 1);   // src/EigenPlaces.cc:33-34: the gate is read at construction
    EigenPlaces idx("/nonexistent/eigenplaces.safetensors", 512, 512);
    IPlaceRecognizer& r = idx;
    r.add(0, desc(16, 3, 0.f)); r.add(1, desc(16, 9, 0.f));
    std::vector<LoopCandidate> res = r.query(desc(16, 3, 0.01f), 0, 5);
    CHECK(!res.empty() && res.front().keyframe_id == 0u && res.front().score > 0.95f);
    if (res.size() > 1) CHECK(res[1].score < res.front().score);
    for (int i = 2; i < 6; ++i) r.add(i, desc(16, i + 20, 0.f));
    for (const auto& c : r.query(desc(16, 25, 0.f), 2, 5)) CHECK(c.keyframe_id < 4u);   // excludeRecent skips the last two insertions
    CHECK(r.query(desc(16, 3, 0.f), 0, 2).size() <= 2u);                                 // topK
    CHECK(r.query(desc(16, 3, 0.f), 6, 5).empty());                                      // everything excluded
    unsetenv("SUPERSLAM_LOOP_MIN_SCORE");
    EigenPlaces gated("/nonexistent/eigenplaces.safetensors", 512, 512);                 // default gate 0.75 (include/EigenPlaces.h:53)
    gated.add(0, desc(16, 3, 0.f)); gated.add(1, desc(16, 9, 0.f));
    for (const auto& c : gated.query(desc(16, 3, 0.f), 0, 5)) CHECK(c.score >= 0.75f && c.keyframe_id == 0u);
  }
}

static void gpu_checks(const char* spw, const char* lgw) {
  const int H = 376, W = 1240, D = 16;  // true disparity of every pixel (a multiple of the 8-px cell: the seeded random weights are only shift-equivariant in cell steps)
  SuperPoint sp(spw, 600, 0.005, 4);
  LightGlue lg(lgw, W, H);
  CHECK(sp.initialize());
  CHECK(lg.initialize());
  if (g_fail) return;
  auto K = gtsam::Cal3_S2Stereo(718.856, 718.856, 0, 607.19, 185.21, 0.537);
  StereoFrontEnd fe(&sp, &lg, K, 1.0f);
  cv::Mat left = make_image(H, W, 7), right = shift_left(left, D);
  int n_depth = 0, n_true = 0;
  {
    StereoFrame f = fe.process(left, right, 1.0);
    CHECK(f.keypoints_left.size() == 600);
    CHECK(!f.descriptors_left.empty() && f.descriptors_left.dim == 256 && f.descriptors_left.count == 600 && f.descriptors_left.slot >= 0);
    CHECK(f.stereo.size() == f.keypoints_left.size() && f.has_depth.size() == f.keypoints_left.size());
    for (size_t i = 0; i < f.has_depth.size(); ++i) {
      CHECK(std::fabs(f.stereo[i].uL() - f.keypoints_left[i].pt.x) < 1e-6 && std::fabs(f.stereo[i].v() - f.keypoints_left[i].pt.y) < 1e-6);
      if (!f.has_depth[i]) { CHECK(std::isnan(f.stereo[i].uR())); continue; }
      ++n_depth;
      CHECK(f.stereo[i].uL() - f.stereo[i].uR() >= 1.0);                       // disparity floor (StereoFrontEnd.cc:41-42)
      if (std::fabs((f.stereo[i].uL() - f.stereo[i].uR()) - D) <= 1.0) ++n_true;  // the scene's disparity
    }
    std::printf("binding/gpu: %zu keypoints, %d with depth, %d at the true disparity %d px\n", f.keypoints_left.size(), n_depth, n_true, D);
    CHECK(n_depth >= 5 && n_true >= (n_depth * 8) / 10);
    // second matcher on the shared engine (SuperSLAM.cc:129-133) + host-descriptor overload == device overload
    LightGlue lg2(lg.shared_engine(), W, H);
    CHECK(lg2.initialize());
    Features L = sp.extract(left), R = sp.extract(right);
    MatchResult dev = lg.match(L.keypoints, L.descriptors, R.keypoints, R.descriptors);
    cv::Mat dl = lg.descriptors_to_host(L.descriptors), dr = lg.descriptors_to_host(R.descriptors);
    CHECK(dl.rows == 600 && dl.cols == 256 && dl.type() == CV_32F);
    MatchResult host = lg2.match(L.keypoints, dl, R.keypoints, dr);
    CHECK(dev.matches.size() == host.matches.size() && !dev.matches.empty());
    for (size_t i = 0; i < std::min(dev.matches.size(), host.matches.size()); ++i) {
      CHECK(dev.matches[i].queryIdx == host.matches[i].queryIdx && dev.matches[i].trainIdx == host.matches[i].trainIdx);
      if (i) CHECK(dev.matches[i].queryIdx > dev.matches[i - 1].queryIdx);     // ascending queryIdx (LightGlue.cc:351-361)
      CHECK(dev.matches[i].distance >= 0.f && dev.matches[i].distance <= 0.9f + 1e-6f);   // 1 - score, score > 0.1
    }
    MatchResult five;
    CHECK(lg.match(L.keypoints, dl, R.keypoints, dr, five) && five.matches.size() == dev.matches.size());
    CHECK(!lg.match({}, cv::Mat(), R.keypoints, dr, five) && five.matches.empty());   // n = 0 -> false (LightGlue.cc:294-295)
  }
  {  // zero disparity: left == right -> every match is rejected by the disparity floor (tests/test_stereo_frontend.cc:65-73)
    StereoFrame f = fe.process(left, left, 2.0);
    int nd = 0;
    for (char c : f.has_depth) nd += c;
    std::printf("binding/gpu: identical pair -> %d with depth (expected 0)\n", nd);
    CHECK(f.keypoints_left.size() == 600 && nd == 0);
  }
  // 3-channel input goes through the adapter's as_image (BGR) and infer() returns CV_32F rows
  cv::Mat bgr(H, W, CV_8UC3);
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int c = 0; c < 3; ++c) bgr.ptr<unsigned char>(y)[3 * x + c] = left.ptr<unsigned char>(y)[x];
  std::vector<cv::KeyPoint> kp; cv::Mat desc;
  CHECK(sp.infer(bgr, kp, desc) && kp.size() == 600 && desc.rows == 600 && desc.cols == 256);
}

static void gpu_place_recognizer(const char* epw) {
  EigenPlaces ep(epw, 512, 512);
  CHECK(ep.initialize());
  cv::Mat a = make_image(376, 1240, 3), b = make_image(376, 1240, 4);
  cv::Mat da = ep.compute_global_descriptor(a), db = ep.compute_global_descriptor(b), da2 = ep.compute_global_descriptor(a);
  CHECK(da.rows == 1 && da.cols == 512 && da.type() == CV_32F);
  double n = 0, same = 0;
  for (int i = 0; i < 512; ++i) { n += da.ptr<float>(0)[i] * da.ptr<float>(0)[i]; same += std::fabs(da.ptr<float>(0)[i] - da2.ptr<float>(0)[i]); }
  CHECK(std::fabs(n - 1.0) < 1e-4 && same == 0.0);               // L2-normalised, deterministic
  ep.add(0, da); ep.add(1, db);
  std::vector<LoopCandidate> r = ep.query(da, 0, 5);
  CHECK(!r.empty() && r.front().keyframe_id == 0u && r.front().score > 0.999f);
  CHECK(ep.query(da, 2, 5).empty());                              // everything excluded as too recent
  std::printf("binding/gpu: EigenPlaces adapter, self score %.6f, %zu candidates\n", r.empty() ? 0.f : r.front().score, r.size());
}

int main(int argc, char** argv) {
  cpu_checks();
  if (argc >= 3) gpu_checks(argv[1], argv[2]);
  if (argc >= 4) gpu_place_recognizer(argv[3]);
  if (g_fail) { std::printf("%d check(s) failed\n", g_fail); return 1; }
  std::printf("reference binding: all checks passed (%s)\n", argc >= 3 ? "cpu + gpu" : "cpu");
  return 0;
}
