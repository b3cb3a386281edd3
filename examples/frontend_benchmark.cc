// Per-frame benchmark of the HIP front-end on a stereo sequence - the counterpart of the reference's
// examples/stereo/benchmark.cc:46-108 / examples/stereo/kitti.cc:70-129 for the part of SuperSLAM this library replaces
// (SURVEY 8(f) rows 1 and 3).  Per frame it runs exactly StereoFrontEnd's unit of work (src/StereoFrontEnd.cc:10-48):
// extract_stereo, the device LightGlue match, the consumer's disparity gate (uL - uR >= 1, |vL - vR| <= 2) - and, with
// --keyframe-match, the second LightGlue call of the tracker against the previous frame's left features
// (src/VoEstimator.cc:243-246).  Output mirrors the reference benchmark: frames, per-frame ms mean / p50 / p95 / max, fps over
// wall time, the real-time (>= 10 fps) verdict; timing brackets the front-end call only, as the reference times
// `track_stereo` after `cv::imread`.
//
// Input, as the reference's runners read it: <sequence>/times.txt (one timestamp per line; optional here - without it frames
// are read until one is missing), <sequence>/image_0/%06d.png and image_1/%06d.png (8/16-bit gray or RGB PNG, decoded by
// include/superslam_hip/image_io.hpp; .pgm is the fallback extension), or --synthetic N procedurally generated 1376x376 pairs.
//
// Decode runs AHEAD on a second thread, straight into the extractor's pinned upload ring (sship_sp_ring_*): the decoder
// writes pixels into pinned host memory and starts the H2D copy of frame i + 1..i + 3 on the ring's copy stream while the
// tracking thread is still computing frame i; --no-ring uses the copying host API (the reference's in-line staging) instead.
// What this runner does NOT link is the reference's GPU-free core (VoEstimator / WindowSmoother need GTSAM, absent here):
// it produces the StereoFrame inputs, not poses.
//
// build:  g++ -std=c++17 -O2 -Iinclude examples/frontend_benchmark.cc -o frontend_benchmark
//             -Lsuperslam_amd/lib -lsuperslam_hip -Wl,-rpath,$PWD/superslam_amd/lib -Wl,-rpath,/opt/rocm/lib -lpthread -lz
// run:    ./frontend_benchmark --sp sp.safetensors --lg lg.safetensors (--sequence DIR | --synthetic 200) [--keyframe-match]
//                              [--max-kp 600] [--threshold 0.005] [--border 4] [--no-ring] [--no-pipeline]
//
// Cross-frame pipelining (default with the ring; --no-pipeline turns it off): as soon as frame t's extraction has returned, frame
// t+1's extraction is ENQUEUED on the extractor's stream (sship_sp_ring_submit) - before frame t's LightGlue match - so the next
// frame's SuperPoint kernels share the GPU with this frame's matcher, whose launches cover a fraction of the CUs at one pair.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "superslam_hip/frontend.hpp"
#include "superslam_hip/image_io.hpp"

namespace sh = superslam_hip;

namespace {

// deterministic texture: two octaves of value noise plus filled rectangles (corners for the detector)
void synth_pair(int idx, int rows, int cols, uint8_t* left, uint8_t* right) {
  uint32_t s = 0x9E3779B9u * (uint32_t)(idx + 1);
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
  const int gw = cols / 16 + 2, gh = rows / 16 + 2;
  std::vector<float> grid((size_t)gw * gh);
  for (auto& g : grid) g = (float)(rnd() & 0xffff) / 65535.f;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const float fx = x / 16.f, fy = y / 16.f;
      const int x0 = (int)fx, y0 = (int)fy;
      const float ax = fx - x0, ay = fy - y0;
      const float v = (grid[y0 * gw + x0] * (1 - ax) + grid[y0 * gw + x0 + 1] * ax) * (1 - ay) +
                      (grid[(y0 + 1) * gw + x0] * (1 - ax) + grid[(y0 + 1) * gw + x0 + 1] * ax) * ay;
      left[(size_t)y * cols + x] = (uint8_t)(40.f + 120.f * v);
    }
  for (int r = 0; r < 160; ++r) {
    const int w = 8 + (int)(rnd() % 40), h = 8 + (int)(rnd() % 40);
    const int x0 = (int)(rnd() % (uint32_t)(cols - w)), y0 = (int)(rnd() % (uint32_t)(rows - h));
    const uint8_t c = (uint8_t)(rnd() % 256);
    for (int y = y0; y < y0 + h; ++y) std::memset(&left[(size_t)y * cols + x0], c, (size_t)w);
  }
  const int d = 16 + (int)(rnd() % 3) * 16;  // SuperPoint on synthetic weights is shift-equivariant in 8-px steps
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) right[(size_t)y * cols + x] = left[(size_t)y * cols + std::min(cols - 1, x + d)];
}

float percentile(std::vector<float> v, double p) {  // same definition as the reference benchmark
  if (v.empty()) return 0.f;
  std::sort(v.begin(), v.end());
  return v[std::min(v.size() - 1, (size_t)(p * (v.size() - 1)))];
}

struct Source {
  std::string sequence, ext = "png";
  int synthetic = 0;
  size_t count = 0;  // frames announced by times.txt (0: until a file is missing)
  // frame ni into caller-provided memory of rows * cols bytes each; false at the end of the sequence
  bool load(int ni, int rows, int cols, uint8_t* left, uint8_t* right) const {
    if (synthetic > 0) {
      if (ni >= synthetic) return false;
      // eight distinct pairs, generated once and cycled: "decoding" a synthetic frame then costs a memcpy, like reading an
      // already-decoded image - the procedural generator itself (several ms per pair) would be the slowest stage of the run
      static std::vector<std::vector<uint8_t>> cache;
      static int crows = 0, ccols = 0;
      if (cache.empty() || crows != rows || ccols != cols) {
        cache.assign(16, std::vector<uint8_t>((size_t)rows * cols));
        for (int k = 0; k < 8; ++k) synth_pair(k, rows, cols, cache[2 * k].data(), cache[2 * k + 1].data());
        crows = rows; ccols = cols;
      }
      std::memcpy(left, cache[2 * (ni % 8)].data(), (size_t)rows * cols);
      std::memcpy(right, cache[2 * (ni % 8) + 1].data(), (size_t)rows * cols);
      return true;
    }
    if (count && (size_t)ni >= count) return false;
    char name[32];
    std::snprintf(name, sizeof name, "%06d.%s", ni, ext.c_str());
    int r = 0, c = 0;
    auto into = [&](uint8_t* dst) { return [=](int rr, int cc) -> uint8_t* { return (rr == rows && cc == cols) ? dst : nullptr; }; };
    return sh::read_gray_image(sequence + "/image_0/" + name, r, c, into(left)) &&
           sh::read_gray_image(sequence + "/image_1/" + name, r, c, into(right));
  }
};

}  // namespace

int main(int argc, char** argv) {
  std::string sp_path, lg_path;
  Source src;
  int max_kp = 600, border = 4;
  double thr = 0.005;
  bool keyframe = false, use_ring = true, pipeline = true;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--sp") sp_path = next();
    else if (a == "--lg") lg_path = next();
    else if (a == "--sequence") src.sequence = next();
    else if (a == "--synthetic") src.synthetic = std::atoi(next());
    else if (a == "--max-kp") max_kp = std::atoi(next());
    else if (a == "--threshold") thr = std::atof(next());
    else if (a == "--border") border = std::atoi(next());
    else if (a == "--keyframe-match") keyframe = true;
    else if (a == "--no-ring") use_ring = false;
    else if (a == "--no-pipeline") pipeline = false;
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (sp_path.empty() || lg_path.empty() || (src.sequence.empty() && src.synthetic <= 0)) {
    std::fprintf(stderr, "usage: %s --sp W.safetensors --lg W.safetensors (--sequence DIR | --synthetic N) [--keyframe-match] "
                         "[--max-kp 600] [--threshold 0.005] [--border 4] [--no-ring] [--no-pipeline]\n", argv[0]);
    return 2;
  }

  // ---- frame geometry from the first frame; timestamps as the reference reads them ----
  int rows = 376, cols = 1376;
  std::vector<double> ts;
  if (src.synthetic <= 0) {
    ts = sh::read_times(src.sequence + "/times.txt");
    src.count = ts.size();
    std::vector<uint8_t> probe;
    if (!sh::read_gray_image(src.sequence + "/image_0/000000.png", probe, rows, cols)) {
      src.ext = "pgm";
      if (!sh::read_gray_image(src.sequence + "/image_0/000000.pgm", probe, rows, cols)) {
        std::fprintf(stderr, "Not a KITTI sequence dir (need image_0/000000.png or .pgm): %s\n", src.sequence.c_str());
        return 1;
      }
    }
  }
  sh::SuperPoint extractor(sp_path, max_kp, thr, border);
  sh::LightGlue matcher(lg_path, cols, rows, max_kp);
  if (!extractor.initialize() || !matcher.initialize()) {
    std::fprintf(stderr, "initialisation failed: %s\n", sship_last_error());
    return 1;
  }
  constexpr int kDepth = 4;
  if (use_ring && !extractor.ring_create(kDepth, rows, cols, 1)) {
    std::fprintf(stderr, "upload ring: %s\n", extractor.last_error().c_str());
    return 1;
  }

  // ---- decode-ahead producer: frame ni -> ring slot ni % kDepth (pinned memory) -> asynchronous upload ----
  std::vector<std::vector<uint8_t>> plain(use_ring ? 0 : 2 * kDepth, std::vector<uint8_t>((size_t)rows * cols));
  std::mutex mu;
  std::condition_variable cv_free, cv_ready;
  int produced = 0, consumed = 0;  // frames decoded / frames whose slot has been handed back
  bool done = false;
  std::thread producer([&]() {
    for (int ni = 0;; ++ni) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_free.wait(lk, [&] { return ni - consumed < kDepth; });  // the slot's previous frame has been extracted
      }
      const int slot = ni % kDepth;
      uint8_t* l = use_ring ? extractor.ring_host(slot, 0) : plain[2 * slot].data();
      uint8_t* r = use_ring ? extractor.ring_host(slot, 1) : plain[2 * slot + 1].data();
      if (!src.load(ni, rows, cols, l, r)) break;
      if (use_ring) extractor.ring_upload(slot);
      std::lock_guard<std::mutex> lk(mu);
      produced = ni + 1;
      cv_ready.notify_one();
    }
    std::lock_guard<std::mutex> lk(mu);
    done = true;
    cv_ready.notify_one();
  });

  std::vector<float> ms;
  long stereo_points = 0, stereo_matches = 0, track_matches = 0, submitted_ahead = 0;
  sh::Features prev_left;
  const auto wall0 = std::chrono::steady_clock::now();
  for (int ni = 0;; ++ni) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_ready.wait(lk, [&] { return produced > ni || done; });
      if (produced <= ni) break;
    }
    const int slot = ni % kDepth;
    const auto t1 = std::chrono::steady_clock::now();
    std::pair<sh::Features, sh::Features> feats;
    if (use_ring) {
      feats = extractor.extract_stereo_ring(slot);   // a submitted slot: waits for its completion event only
      if (pipeline) {  // frame ni + 1 already decoded and uploading?  enqueue its extraction now, ahead of this frame's match
        bool have_next;
        { std::lock_guard<std::mutex> lk(mu); have_next = produced > ni + 1; }
        if (have_next && extractor.ring_submit((ni + 1) % kDepth)) ++submitted_ahead;
      }
    } else feats = extractor.extract_stereo(sh::Image{plain[2 * slot].data(), rows, cols, 1, 0}, sh::Image{plain[2 * slot + 1].data(), rows, cols, 1, 0});
    sh::MatchResult lr = matcher.match(feats.first.keypoints, feats.first.descriptors, feats.second.keypoints, feats.second.descriptors);
    for (const sh::DMatch& m : lr.matches) {  // StereoFrontEnd's gate
      const sh::KeyPoint &kl = feats.first.keypoints[m.queryIdx], &kr = feats.second.keypoints[m.trainIdx];
      if (kl.x - kr.x >= 1.0f && std::abs(kl.y - kr.y) <= 2.0f) ++stereo_points;
    }
    stereo_matches += (long)lr.matches.size();
    if (keyframe && !prev_left.descriptors.empty())
      track_matches += (long)matcher.match(feats.first.keypoints, feats.first.descriptors, prev_left.keypoints, prev_left.descriptors).matches.size();
    const auto t2 = std::chrono::steady_clock::now();
    ms.push_back((float)std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count() / 1000.0f);
    if (keyframe) prev_left = std::move(feats.first);  // holds its pool slot until the next frame replaces it
    {
      std::lock_guard<std::mutex> lk(mu);
      consumed = ni + 1;  // the extract call that read this slot has returned: the decoder may overwrite it
      cv_free.notify_one();
    }
  }
  const double wall = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - wall0).count() / 1000.0;
  producer.join();
  if (ms.empty()) { std::fprintf(stderr, "no frames\n"); return 1; }

  const double mean = std::accumulate(ms.begin(), ms.end(), 0.0) / ms.size();
  std::printf("=========== SuperSLAM HIP front-end benchmark ===========\n");
  std::printf("source           : %s, %dx%d, %s%s\n", src.synthetic > 0 ? "synthetic" : (src.ext == "png" ? "PNG sequence" : "PGM sequence"), cols, rows,
              use_ring ? "pinned upload ring" : "copying host API", ts.empty() ? "" : ", times.txt");
  std::printf("frames           : %zu\n", ms.size());
  if (use_ring) std::printf("pipelined        : %s (%ld of %zu extractions enqueued one frame ahead)\n", pipeline ? "yes" : "no", submitted_ahead, ms.size());
  std::printf("per-frame ms      mean=%.2f p50=%.2f p95=%.2f max=%.2f\n", mean, percentile(ms, 0.50), percentile(ms, 0.95), percentile(ms, 1.0));
  std::printf("throughput        : %.2f fps over %.1fs wall\n", wall > 0 ? ms.size() / wall : 0.0, wall);
  std::printf("real-time (>=10fps): %s\n", mean > 0 && (1000.0 / mean) >= 10.0 ? "YES" : "NO");
  if (!ts.empty() && ts.size() >= ms.size() && ms.size() > 1)
    std::printf("sequence rate     : %.2f fps (times.txt), processed at %.1fx real time\n", (ms.size() - 1) / (ts[ms.size() - 1] - ts[0]),
                (ts[ms.size() - 1] - ts[0]) / std::max(1e-9, wall));
  std::printf("stereo matches    : %.1f per frame, %.1f pass the disparity gate\n", (double)stereo_matches / ms.size(), (double)stereo_points / ms.size());
  if (keyframe) std::printf("keyframe matches  : %.1f per frame\n", ms.size() > 1 ? (double)track_matches / (ms.size() - 1) : 0.0);
  std::printf("=========================================================\n");
  return 0;
}
