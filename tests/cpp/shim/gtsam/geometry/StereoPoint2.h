#pragma once
namespace gtsam {
class StereoPoint2 {
public:
  StereoPoint2() = default;
  StereoPoint2(double uL, double uR, double v) : uL_(uL), uR_(uR), v_(v) {}
  double uL() const { return uL_; }
  double uR() const { return uR_; }
  double v() const { return v_; }
private:
  double uL_ = 0, uR_ = 0, v_ = 0;
};
}  // namespace gtsam
