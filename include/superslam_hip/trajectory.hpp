// Trajectory writers with the reference's exact text formats (src/SuperSLAM.cc:192-218), for runners that produce poses from
// the HIP front-end's StereoFrames: KITTI = camera-to-world 3x4 [Rwc | twc] row-major per line, TUM = "timestamp tx ty tz qx qy qz qw";
// both std::fixed with 9 decimals.  superslam_amd/trajectory.py reads these files back and scores them (ATE, KITTI segments).
#pragma once
#include <array>
#include <fstream>
#include <iomanip>
#include <string>
#include <vector>

namespace superslam_hip {

typedef std::array<double, 12> Pose3x4;  // R00 R01 R02 tx R10 R11 R12 ty R20 R21 R22 tz (Twc)

inline bool save_trajectory_kitti(const std::string& path, const std::vector<Pose3x4>& poses) {
  std::ofstream f(path);
  if (!f) return false;
  f << std::fixed << std::setprecision(9);
  for (const Pose3x4& p : poses) {
    for (int i = 0; i < 12; ++i) f << p[i] << (i == 11 ? "\n" : " ");
  }
  return static_cast<bool>(f);
}

// quaternions are (x, y, z, w), as Eigen::Quaternion's accessors are printed by the reference
inline bool save_trajectory_tum(const std::string& path, const std::vector<double>& timestamps, const std::vector<std::array<double, 3>>& t,
                                const std::vector<std::array<double, 4>>& q_xyzw) {
  std::ofstream f(path);
  if (!f || t.size() != q_xyzw.size()) return false;
  f << std::fixed << std::setprecision(9);
  for (size_t i = 0; i < t.size(); ++i) {
    const double ts = i < timestamps.size() ? timestamps[i] : static_cast<double>(i);
    f << ts << " " << t[i][0] << " " << t[i][1] << " " << t[i][2] << " " << q_xyzw[i][0] << " " << q_xyzw[i][1] << " " << q_xyzw[i][2] << " "
      << q_xyzw[i][3] << "\n";
  }
  return static_cast<bool>(f);
}

}  // namespace superslam_hip
