// C++ host-layer tests.  CPU part: "fails loudly without a device" (pool, extractor).  GPU part: the reference's
// tests/test_descriptor_pool.cc cases on the C ABI's pool and (argv[1] = sp weights, argv[2] = lg weights) the
// StereoFrontEnd consumer semantics of the reference's tests/test_stereo_frontend.cc, driven through the real
// extractor / matcher behind the IFeatureExtractor / IFeatureMatcher interfaces.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "superslam_hip/frontend.hpp"

using namespace superslam_hip;

static int g_fail = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

// The reference's tests/test_descriptor_pool.cc cases (FreeList.AcquireReleaseRoundTrips, FreeList.EmptyAndFull) on the
// bookkeeping the product actually uses: the C ABI's pool (sship_pool_acquire / release / in_use).  Needs a device (the pool
// owns device slots); without one sship_pool_create must fail loudly.
static void test_pool_bookkeeping(bool have_device) {
  sship_pool* p = nullptr;
  if (!have_device) {
    EXPECT(sship_pool_create(3, 16, 256, &p) == SSHIP_ERR_NO_DEVICE && p == nullptr);
    EXPECT(sship_pool_acquire(nullptr) == -1 && sship_pool_in_use(nullptr) == 0);
    return;
  }
  {  // AcquireReleaseRoundTrips
    EXPECT(sship_pool_create(3, 16, 256, &p) == SSHIP_OK);
    const int a = sship_pool_acquire(p), b = sship_pool_acquire(p), c = sship_pool_acquire(p);
    EXPECT(a >= 0); EXPECT(b >= 0); EXPECT(c >= 0);
    EXPECT(a != b && b != c && a != c);
    EXPECT(sship_pool_acquire(p) == -1);
    EXPECT(sship_pool_in_use(p) == 3);
    sship_pool_release(p, b);
    EXPECT(sship_pool_in_use(p) == 2);
    EXPECT(sship_pool_acquire(p) == b);   // LIFO: the slot released last is handed out first
    EXPECT(sship_pool_slot_ptr(p, a) != nullptr && sship_pool_slot_ptr(p, 3) == nullptr);
    sship_pool_destroy(p);
  }
  {  // EmptyAndFull
    EXPECT(sship_pool_create(2, 16, 256, &p) == SSHIP_OK);
    EXPECT(sship_pool_in_use(p) == 0);
    const int a = sship_pool_acquire(p), b = sship_pool_acquire(p);
    EXPECT(sship_pool_in_use(p) == 2);
    sship_pool_release(p, a); sship_pool_release(p, b);
    EXPECT(sship_pool_in_use(p) == 0);
    sship_pool_destroy(p);
  }
}

// procedural frame (value noise is overkill here: blocks + gradients give SuperPoint plenty of corners)
static std::vector<uint8_t> make_image(int h, int w, int shift) {
  // smooth multi-octave value noise + a few hash-placed rectangles (a C++ cousin of superslam_amd/synth.py)
  auto hash = [](unsigned a, unsigned b, unsigned c) {
    unsigned x = a * 73856093u ^ b * 19349663u ^ c * 83492791u; x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15; return (x & 0xffff) / 65535.0f;
  };
  auto noise = [&](float fx, float fy, unsigned oct) {
    const int x0 = static_cast<int>(std::floor(fx)), y0 = static_cast<int>(std::floor(fy));
    float tx = fx - x0, ty = fy - y0;
    tx = tx * tx * (3 - 2 * tx); ty = ty * ty * (3 - 2 * ty);
    const float a = hash(x0, y0, oct), b = hash(x0 + 1, y0, oct), c = hash(x0, y0 + 1, oct), d = hash(x0 + 1, y0 + 1, oct);
    return (a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty;
  };
  std::vector<uint8_t> img(static_cast<size_t>(h) * w);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float xs = static_cast<float>(x + shift) + 1000.f, ys = static_cast<float>(y) + 1000.f;
      float v = 0.f, amp = 1.f, cell = 48.f, norm = 0.f;
      for (unsigned o = 0; o < 5; ++o) { v += amp * noise(xs / cell, ys / cell, o); norm += amp; amp *= 0.6f; cell *= 0.5f; }
      v /= norm;
      for (unsigned r = 0; r < 40; ++r) {  // rectangles in pattern coordinates
        const float rx = 1000.f + hash(r, 1, 99) * (w + 40), ry = 1000.f + hash(r, 2, 99) * h;
        const float rw = 6.f + hash(r, 3, 99) * 30.f, rh = 5.f + hash(r, 4, 99) * 24.f;
        if (xs >= rx && xs < rx + rw && ys >= ry && ys < ry + rh) v = 0.35f * v + 0.65f * hash(r, 5, 99);
      }
      img[static_cast<size_t>(y) * w + x] = static_cast<uint8_t>(std::min(255.f, std::max(0.f, v * 255.f)));
    }
  return img;
}

static int run_gpu(const char* spw, const char* lgw) {
  const int H = 240, W = 320;
  SuperPoint sp(spw, 300, 0.005, 4);
  EXPECT(sp.initialize());
  LightGlue lg(lgw, W, H, 300);
  EXPECT(lg.initialize());
  LightGlue lg2(lg.shared_engine(), W, H, 300);  // second matcher on the shared weights (loop-closure thread)
  EXPECT(lg2.initialize());
  if (g_fail) return 1;
  const auto left = make_image(H, W, 0), right = make_image(H, W, 16);  // right = left shifted by 16 px (2 SuperPoint cells: the net is shift-equivariant only in 8-px steps)
  Image L{left.data(), H, W, 1, 0}, R{right.data(), H, W, 1, 0};
  IFeatureExtractor* ext = &sp;
  IFeatureMatcher* matcher = &lg;
  auto lr = ext->extract_stereo(L, R);
  EXPECT(!lr.first.keypoints.empty());
  EXPECT(lr.first.descriptors.count == static_cast<int>(lr.first.keypoints.size()));
  EXPECT(sp.pool_in_use() == 2);
  MatchResult m = matcher->match(lr.first.keypoints, lr.first.descriptors, lr.second.keypoints, lr.second.descriptors);
  // StereoFrontEnd::process gates (src/StereoFrontEnd.cc:35-48)
  int with_depth = 0, disparity16 = 0, ascending = 1, last = -1;
  for (const DMatch& d : m.matches) {
    if (d.queryIdx <= last) ascending = 0;
    last = d.queryIdx;
    const KeyPoint& a = lr.first.keypoints[d.queryIdx];
    const KeyPoint& b = lr.second.keypoints[d.trainIdx];
    if (a.x - b.x < 1.0f) continue;
    if (std::fabs(a.y - b.y) > 2.0f) continue;
    ++with_depth;
    if (std::fabs((a.x - b.x) - 16.0f) < 0.5f) ++disparity16;
  }
  std::printf("cpp: %zu/%zu keypoints, %zu matches, %d with depth, %d at disparity 16\n", lr.first.keypoints.size(),
              lr.second.keypoints.size(), m.matches.size(), with_depth, disparity16);
  EXPECT(ascending);
  EXPECT(with_depth > 0);
  EXPECT(disparity16 * 10 >= with_depth * 6);  // the synthetic pair has a constant 16 px disparity
  // zero disparity: left == right -> every gated match is rejected (MarksBelowFloorDisparityAsNoDepth)
  auto ll = ext->extract_stereo(L, L);
  MatchResult mz = matcher->match(ll.first.keypoints, ll.first.descriptors, ll.second.keypoints, ll.second.descriptors);
  int below = 0;
  for (const DMatch& d : mz.matches) if (ll.first.keypoints[d.queryIdx].x - ll.second.keypoints[d.trainIdx].x < 1.0f) ++below;
  EXPECT(!mz.matches.empty() && below == static_cast<int>(mz.matches.size()));
  // host-descriptor overload through the second matcher == device overload
  HostDescriptors h0 = lg.descriptors_to_host(lr.first.descriptors), h1 = lg.descriptors_to_host(lr.second.descriptors);
  EXPECT(h0.rows == static_cast<int>(lr.first.keypoints.size()) && h0.cols == 256);
  MatchResult mh = lg2.match(lr.first.keypoints, h0, lr.second.keypoints, h1);
  EXPECT(mh.matches.size() == m.matches.size());
  for (size_t i = 0; i < std::min(mh.matches.size(), m.matches.size()); ++i) EXPECT(mh.matches[i].trainIdx == m.matches[i].trainIdx);
  // empty inputs never throw: empty results
  std::vector<KeyPoint> none;
  EXPECT(matcher->match(none, DeviceDescriptors(), lr.second.keypoints, lr.second.descriptors).matches.empty());
  EXPECT(lg.descriptors_to_host(DeviceDescriptors()).empty());
  MatchResult r5;
  EXPECT(!lg.match(none, HostDescriptors(), none, HostDescriptors(), r5));
  // pool exhaustion (8 slots): handles held -> the 5th stereo call fails softly and frees nothing it does not own
  std::vector<std::pair<Features, Features>> held;
  for (int i = 0; i < 3; ++i) held.push_back(ext->extract_stereo(L, R));  // 2 + 2 (ll) + 6 = 10 > 8
  EXPECT(sp.pool_in_use() == 8);
  EXPECT(held.back().second.descriptors.empty() || held.back().first.descriptors.empty());
  held.clear(); ll = {}; lr = {};
  EXPECT(sp.pool_in_use() == 0);
  return g_fail ? 1 : 0;
}

int main(int argc, char** argv) {
  const bool have_device = sship_init(-1) == SSHIP_OK;
  test_pool_bookkeeping(have_device);
  if (argc < 3) {
    // CPU box: the library must fail loudly, never fall back
    if (have_device) { std::printf("cpp: a GPU is visible, pass weight paths to run the GPU part\n"); return g_fail ? 1 : 0; }
    SuperPoint sp("missing.safetensors", 600, 0.005, 4);
    EXPECT(!sp.initialize());
    EXPECT(sp.last_error().find("no HIP device") != std::string::npos);
    Features f = sp.extract(Image{});
    EXPECT(f.keypoints.empty() && f.descriptors.empty());
    std::printf("cpp: host-layer CPU checks %s\n", g_fail ? "FAILED" : "ok");
    return g_fail ? 1 : 0;
  }
  const int rc = run_gpu(argv[1], argv[2]);
  std::printf("cpp: host-layer GPU checks %s\n", rc ? "FAILED" : "ok");
  return rc;
}
