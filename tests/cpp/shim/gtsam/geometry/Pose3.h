#pragma once
#include "Point3.h"
namespace gtsam {
struct Pose3 { double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; Point3 t; };  // identity by default, as gtsam::Pose3()
}  // namespace gtsam
