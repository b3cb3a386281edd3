// Minimal stand-in for gtsam::Cal3_S2Stereo (tests/cpp/shim/README.md)
#pragma once
namespace gtsam {
class Cal3_S2Stereo {
public:
  Cal3_S2Stereo() = default;
  Cal3_S2Stereo(double fx, double fy, double s, double u0, double v0, double b) : fx_(fx), fy_(fy), s_(s), u0_(u0), v0_(v0), b_(b) {}
  double fx() const { return fx_; }
  double fy() const { return fy_; }
  double skew() const { return s_; }
  double px() const { return u0_; }
  double py() const { return v0_; }
  double baseline() const { return b_; }
private:
  double fx_ = 1, fy_ = 1, s_ = 0, u0_ = 0, v0_ = 0, b_ = 1;
};
}  // namespace gtsam
