// Per-frame benchmark of the HIP front-end on a stereo sequence - the counterpart of the reference's
// examples/stereo/benchmark.cc:46-108 for the part of SuperSLAM this library replaces.
// Per frame it runs exactly StereoFrontEnd's unit of work (src/StereoFrontEnd.cc:10-48): extract_stereo, the device
// LightGlue match, the consumer's disparity gate (uL - uR >= 1, |vL - vR| <= 2) - and, with --keyframe-match, the
// second LightGlue call of the tracker against the previous frame's left features (src/VoEstimator.cc:243-246).
// Output mirrors the reference benchmark: frames, per-frame ms mean / p50 / p95 / max, fps over wall time, the
// real-time (>= 10 fps) verdict.  Timing brackets the front-end call only; image decode runs ahead on a second thread
// into a small ring, as the reference times `track_stereo` after `cv::imread`.
//
// Input: a KITTI-style directory  <sequence>/image_0/000000.pgm, <sequence>/image_1/000000.pgm ...  (binary PGM, "P5";
// this image ships neither OpenCV nor libpng - `mogrify -format pgm *.png` converts a KITTI sequence once), or
// --synthetic N: N procedurally generated 1376x376 pairs (value noise + rectangles, right = left shifted by 16..48 px).
//
// build:  g++ -std=c++17 -O2 -Iinclude examples/frontend_benchmark.cc -o frontend_benchmark
//             -Lsuperslam_amd/lib -lsuperslam_hip -Wl,-rpath,$PWD/superslam_amd/lib -Wl,-rpath,/opt/rocm/lib -lpthread
// run:    ./frontend_benchmark --sp sp.safetensors --lg lg.safetensors (--sequence DIR | --synthetic 200) [--keyframe-match]
//                              [--max-kp 600] [--threshold 0.005] [--border 4]
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "superslam_hip/frontend.hpp"

namespace sh = superslam_hip;

namespace {

struct Frame {
  std::vector<uint8_t> left, right;
  int rows = 0, cols = 0;
  bool ok = false;
};

bool read_pgm(const std::string& path, std::vector<uint8_t>& px, int& rows, int& cols) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string magic;
  f >> magic;
  if (magic != "P5") return false;
  int vals[3], got = 0;
  while (got < 3 && f) {  // width, height, maxval with '#' comments in between
    f >> std::ws;
    if (f.peek() == '#') { std::string skip; std::getline(f, skip); continue; }
    f >> vals[got++];
  }
  if (got < 3 || vals[2] != 255) return false;
  f.get();  // the single whitespace byte after maxval
  cols = vals[0]; rows = vals[1];
  px.resize((size_t)rows * cols);
  f.read(reinterpret_cast<char*>(px.data()), (std::streamsize)px.size());
  return (size_t)f.gcount() == px.size();
}

// deterministic texture: two octaves of value noise plus filled rectangles (corners for the detector)
void synth_pair(int idx, int rows, int cols, Frame& fr) {
  fr.rows = rows; fr.cols = cols; fr.left.assign((size_t)rows * cols, 0); fr.right.assign((size_t)rows * cols, 0);
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
      fr.left[(size_t)y * cols + x] = (uint8_t)(40.f + 120.f * v);
    }
  for (int r = 0; r < 160; ++r) {
    const int w = 8 + (int)(rnd() % 40), h = 8 + (int)(rnd() % 40);
    const int x0 = (int)(rnd() % (uint32_t)(cols - w)), y0 = (int)(rnd() % (uint32_t)(rows - h));
    const uint8_t c = (uint8_t)(rnd() % 256);
    for (int y = y0; y < y0 + h; ++y) std::memset(&fr.left[(size_t)y * cols + x0], c, (size_t)w);
  }
  const int d = 16 + (int)(rnd() % 3) * 16;  // SuperPoint on synthetic weights is shift-equivariant in 8-px steps
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) fr.right[(size_t)y * cols + x] = fr.left[(size_t)y * cols + std::min(cols - 1, x + d)];
  fr.ok = true;
}

float percentile(std::vector<float> v, double p) {  // same definition as the reference benchmark
  if (v.empty()) return 0.f;
  std::sort(v.begin(), v.end());
  return v[std::min(v.size() - 1, (size_t)(p * (v.size() - 1)))];
}

}  // namespace

int main(int argc, char** argv) {
  std::string sp_path, lg_path, sequence;
  int synthetic = 0, max_kp = 600, border = 4;
  double thr = 0.005;
  bool keyframe = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--sp") sp_path = next();
    else if (a == "--lg") lg_path = next();
    else if (a == "--sequence") sequence = next();
    else if (a == "--synthetic") synthetic = std::atoi(next());
    else if (a == "--max-kp") max_kp = std::atoi(next());
    else if (a == "--threshold") thr = std::atof(next());
    else if (a == "--border") border = std::atoi(next());
    else if (a == "--keyframe-match") keyframe = true;
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (sp_path.empty() || lg_path.empty() || (sequence.empty() && synthetic <= 0)) {
    std::fprintf(stderr, "usage: %s --sp W.safetensors --lg W.safetensors (--sequence DIR | --synthetic N) [--keyframe-match] "
                         "[--max-kp 600] [--threshold 0.005] [--border 4]\n", argv[0]);
    return 2;
  }

  // ---- decode-ahead producer: a ring of at most 4 decoded pairs ----
  std::deque<Frame> ring;
  std::mutex mu;
  std::condition_variable cv_not_full, cv_not_empty;
  bool done = false;
  std::thread producer([&]() {
    for (int ni = 0;; ++ni) {
      Frame fr;
      if (synthetic > 0) {
        if (ni >= synthetic) break;
        synth_pair(ni, 376, 1376, fr);
      } else {
        char name[32];
        std::snprintf(name, sizeof name, "%06d.pgm", ni);
        int r2 = 0, c2 = 0;
        if (!read_pgm(sequence + "/image_0/" + name, fr.left, fr.rows, fr.cols) ||
            !read_pgm(sequence + "/image_1/" + name, fr.right, r2, c2) || r2 != fr.rows || c2 != fr.cols)
          break;  // end of the sequence (or an unreadable / mismatched pair)
        fr.ok = true;
      }
      std::unique_lock<std::mutex> lk(mu);
      cv_not_full.wait(lk, [&] { return ring.size() < 4; });
      ring.push_back(std::move(fr));
      cv_not_empty.notify_one();
    }
    std::lock_guard<std::mutex> lk(mu);
    done = true;
    cv_not_empty.notify_one();
  });
  auto pop = [&](Frame& fr) {
    std::unique_lock<std::mutex> lk(mu);
    cv_not_empty.wait(lk, [&] { return !ring.empty() || done; });
    if (ring.empty()) return false;
    fr = std::move(ring.front());
    ring.pop_front();
    cv_not_full.notify_one();
    return true;
  };

  Frame fr;
  if (!pop(fr)) { std::fprintf(stderr, "no frames\n"); producer.join(); return 1; }
  sh::SuperPoint extractor(sp_path, max_kp, thr, border);
  sh::LightGlue matcher(lg_path, fr.cols, fr.rows, max_kp);
  if (!extractor.initialize() || !matcher.initialize()) {
    std::fprintf(stderr, "initialisation failed: %s\n", sship_last_error());
    { std::lock_guard<std::mutex> lk(mu); done = true; ring.clear(); }
    cv_not_full.notify_all();
    producer.detach();
    return 1;
  }

  std::vector<float> ms;
  long stereo_points = 0, stereo_matches = 0, track_matches = 0;
  sh::Features prev_left;
  const auto wall0 = std::chrono::steady_clock::now();
  do {
    const sh::Image l{fr.left.data(), fr.rows, fr.cols, 1, 0}, r{fr.right.data(), fr.rows, fr.cols, 1, 0};
    const auto t1 = std::chrono::steady_clock::now();
    auto feats = extractor.extract_stereo(l, r);
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
  } while (pop(fr));
  const double wall = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - wall0).count() / 1000.0;
  producer.join();

  const double mean = ms.empty() ? 0.0 : std::accumulate(ms.begin(), ms.end(), 0.0) / ms.size();
  std::printf("=========== SuperSLAM HIP front-end benchmark ===========\n");
  std::printf("frames           : %zu\n", ms.size());
  std::printf("per-frame ms      mean=%.2f p50=%.2f p95=%.2f max=%.2f\n", mean, percentile(ms, 0.50), percentile(ms, 0.95), percentile(ms, 1.0));
  std::printf("throughput        : %.2f fps over %.1fs wall\n", wall > 0 ? ms.size() / wall : 0.0, wall);
  std::printf("real-time (>=10fps): %s\n", mean > 0 && (1000.0 / mean) >= 10.0 ? "YES" : "NO");
  std::printf("stereo matches    : %.1f per frame, %.1f pass the disparity gate\n", ms.empty() ? 0.0 : (double)stereo_matches / ms.size(),
              ms.empty() ? 0.0 : (double)stereo_points / ms.size());
  if (keyframe) std::printf("keyframe matches  : %.1f per frame\n", ms.size() > 1 ? (double)track_matches / (ms.size() - 1) : 0.0);
  std::printf("=========================================================\n");
  return 0;
}
