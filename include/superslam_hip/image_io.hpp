// Image file readers for the dataset runners: the step in front of the hot path.  The reference reads its
// sequences with cv::imread(path, cv::IMREAD_UNCHANGED) (examples/stereo/benchmark.cc:74-75, examples/stereo/kitti.cc:104-105)
// and SuperSLAM::track_stereo converts 3-channel frames with cv::COLOR_BGR2GRAY (src/SuperSLAM.cc:155-160); this image has no
// OpenCV, so the two formats the KITTI / EuRoC runners need are decoded here:
//   * PNG  - 8/16-bit gray, gray+alpha, RGB, RGBA, non-interlaced (zlib inflate + the five PNG row filters);
//   * PGM  - binary "P5", maxval 255.
// Every reader returns 8-bit gray: RGB goes through OpenCV's fixed-point BGR2GRAY weights
// ((R*4899 + G*9617 + B*1868 + 8192) >> 14), 16-bit samples keep their high byte.
// Header-only; link with -lz.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace superslam_hip {

// Decodes into `out` (rows * stride bytes, stride >= cols; caller-provided so a pinned ring slot can be the target).
// `alloc(rows, cols)` is called once the size is known and returns the destination pointer (or nullptr to abort).
template <class Alloc>
inline bool read_pgm(const std::string& path, int& rows, int& cols, Alloc&& alloc) {
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
  if (got < 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) return false;
  f.get();  // the single whitespace byte after maxval
  cols = vals[0]; rows = vals[1];
  uint8_t* dst = alloc(rows, cols);
  if (!dst) return false;
  f.read(reinterpret_cast<char*>(dst), static_cast<std::streamsize>(rows) * cols);
  return f.gcount() == static_cast<std::streamsize>(rows) * cols;
}

namespace png_detail {
inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace png_detail

template <class Alloc>
inline bool read_png(const std::string& path, int& rows, int& cols, Alloc&& alloc) {
  using namespace png_detail;
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return false;
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
    if (pos + 12 + len > file.size()) return false;
    const uint8_t* data = &file[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return false;
      w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!w || !h || w > 32768 || h > 32768 || !ch || (depth != 8 && depth != 16) || interlace != 0) return false;
  const size_t bpp = static_cast<size_t>(ch) * (depth / 8), row_bytes = bpp * w;
  std::vector<uint8_t> raw((row_bytes + 1) * h);
  uLongf raw_len = static_cast<uLongf>(raw.size());
  if (uncompress(raw.data(), &raw_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || raw_len != raw.size()) return false;
  rows = static_cast<int>(h); cols = static_cast<int>(w);
  uint8_t* dst = alloc(rows, cols);
  if (!dst) return false;
  std::vector<uint8_t> prev(row_bytes, 0), cur(row_bytes);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* in = &raw[(row_bytes + 1) * y];
    const int ft = in[0];
    ++in;
    for (size_t i = 0; i < row_bytes; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = in[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: return false;
      }
      cur[i] = static_cast<uint8_t>(v);
    }
    uint8_t* out = dst + static_cast<size_t>(y) * w;
    const size_t sb = depth / 8;  // sample bytes: 16-bit samples are big-endian, the high byte comes first
    for (uint32_t x = 0; x < w; ++x) {
      const uint8_t* px = &cur[x * bpp];
      if (ch <= 2) out[x] = px[0];
      else out[x] = static_cast<uint8_t>((px[0] * 4899 + px[sb] * 9617 + px[2 * sb] * 1868 + 8192) >> 14);
    }
    prev.swap(cur);
  }
  return true;
}

// By extension (.png / .pgm); `alloc` as above.
template <class Alloc>
inline bool read_gray_image(const std::string& path, int& rows, int& cols, Alloc&& alloc) {
  const size_t dot = path.rfind('.');
  const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
  if (ext == "png" || ext == "PNG") return read_png(path, rows, cols, alloc);
  if (ext == "pgm" || ext == "PGM") return read_pgm(path, rows, cols, alloc);
  return false;
}
inline bool read_gray_image(const std::string& path, std::vector<uint8_t>& px, int& rows, int& cols) {
  return read_gray_image(path, rows, cols, [&](int r, int c) { px.resize(static_cast<size_t>(r) * c); return px.data(); });
}

// KITTI times.txt (examples/stereo/benchmark.cc:55-61): one timestamp per line, stops at the first empty line.
inline std::vector<double> read_times(const std::string& path) {
  std::vector<double> ts;
  std::ifstream f(path);
  std::string s;
  while (std::getline(f, s) && !s.empty()) ts.push_back(std::strtod(s.c_str(), nullptr));
  return ts;
}

}  // namespace superslam_hip
