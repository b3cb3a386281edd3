// Minimal stand-in for <opencv2/core.hpp> (tests/cpp/shim/README.md): only what the binding, StereoFrontEnd.cc and PlaceRecognizer.cc touch.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> 3) & 511) + 1)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {
struct Point2f {
  float x = 0.f, y = 0.f;
  Point2f() = default;
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct KeyPoint {
  Point2f pt;
  float size = 0.f, angle = -1.f, response = 0.f;
  int octave = 0, class_id = -1;
  KeyPoint() = default;
  KeyPoint(float x, float y, float size_, float angle_ = -1.f, float response_ = 0.f, int octave_ = 0, int class_id_ = -1)
      : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 3.4e38f;
  DMatch() = default;
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), distance(d) {}
};
class Mat {
public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;  // bytes per row
  Mat() = default;
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* external, size_t step_ = 0) : rows(r), cols(c), data(static_cast<unsigned char*>(external)), type_(type) {
    step = step_ ? step_ : static_cast<size_t>(c) * elemSize();
  }
  static Mat zeros(int r, int c, int type) { Mat m(r, c, type); if (m.data) std::memset(m.data, 0, m.step * r); return m; }
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type; step = static_cast<size_t>(c) * elemSize();
    own_ = std::shared_ptr<unsigned char>(new unsigned char[(step * r) != 0 ? step * r : 1], std::default_delete<unsigned char[]>());
    data = own_.get();
  }
  int type() const { return type_; }
  int depth() const { return CV_MAT_DEPTH(type_); }
  int channels() const { return CV_MAT_CN(type_); }
  size_t elemSize() const { return (depth() == CV_32F ? 4u : 1u) * static_cast<size_t>(channels()); }
  size_t total() const { return static_cast<size_t>(rows) * cols; }
  bool empty() const { return data == nullptr || total() == 0; }
  bool isContinuous() const { return step == static_cast<size_t>(cols) * elemSize(); }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * r); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * r); }
  Mat clone() const {
    Mat m;
    if (!data) return m;
    m.create(rows, cols, type_);
    for (int r = 0; r < rows; ++r) std::memcpy(m.data + m.step * r, data + step * r, static_cast<size_t>(cols) * elemSize());
    return m;
  }
  void convertTo(Mat& dst, int rtype) const {  // saturating u8 <-> f32, same channel count
    const int ddepth = CV_MAT_DEPTH(rtype);
    Mat out(rows, cols, CV_MAKETYPE(ddepth, channels()));
    const int n = cols * channels();
    for (int r = 0; r < rows; ++r)
      for (int i = 0; i < n; ++i) {
        const float v = depth() == CV_32F ? ptr<float>(r)[i] : static_cast<float>(ptr<unsigned char>(r)[i]);
        if (ddepth == CV_32F) out.ptr<float>(r)[i] = v;
        else out.ptr<unsigned char>(r)[i] = static_cast<unsigned char>(v < 0.f ? 0.f : v > 255.f ? 255.f : v + 0.5f);
      }
    dst = out;
  }
  // ---- the CV_32F row-matrix operations src/PlaceRecognizer.cc uses (reshape / push_back / rowRange / t / at; norm, /, * below) ----
  Mat reshape(int /*cn*/, int new_rows) const {  // single channel, continuous data only
    Mat m = *this;
    m.rows = new_rows; m.cols = new_rows ? static_cast<int>(total() * channels() / new_rows) : 0;
    m.type_ = CV_MAKETYPE(depth(), 1); m.step = static_cast<size_t>(m.cols) * m.elemSize();
    return m;
  }
  void push_back(const Mat& row) {  // append the rows of `row` (same type and width)
    Mat m(rows + row.rows, row.cols, row.type());
    for (int r = 0; r < rows; ++r) std::memcpy(m.data + m.step * r, data + step * r, m.step);
    for (int r = 0; r < row.rows; ++r) std::memcpy(m.data + m.step * (rows + r), row.data + row.step * r, m.step);
    *this = m;
  }
  Mat rowRange(int a, int b) const { Mat m = *this; m.rows = b - a; m.data = data + step * a; return m; }  // shares the buffer
  Mat t() const {
    Mat m(cols, rows, type_);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) m.ptr<float>(c)[r] = ptr<float>(r)[c];
    return m;
  }
  template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }

private:
  int type_ = 0;
  std::shared_ptr<unsigned char> own_;  // shared like cv::Mat's refcounted buffer
};
inline double norm(const Mat& m) {  // NORM_L2 over a CV_32F matrix
  double n = 0.0;
  for (int r = 0; r < m.rows; ++r) for (int c = 0; c < m.cols; ++c) n += static_cast<double>(m.ptr<float>(r)[c]) * m.ptr<float>(r)[c];
  return std::sqrt(n);
}
inline Mat operator/(const Mat& m, double d) {
  Mat o(m.rows, m.cols, m.type());
  for (int r = 0; r < m.rows; ++r) for (int c = 0; c < m.cols; ++c) o.ptr<float>(r)[c] = static_cast<float>(m.ptr<float>(r)[c] / d);
  return o;
}
inline Mat operator*(const Mat& a, const Mat& b) {  // CV_32F GEMM
  Mat o(a.rows, b.cols, CV_32F);
  for (int r = 0; r < a.rows; ++r)
    for (int c = 0; c < b.cols; ++c) {
      double acc = 0.0;
      for (int k = 0; k < a.cols; ++k) acc += static_cast<double>(a.ptr<float>(r)[k]) * b.ptr<float>(k)[c];
      o.ptr<float>(r)[c] = static_cast<float>(acc);
    }
  return o;
}
}  // namespace cv
