#pragma once
namespace gtsam {
struct Point3 { double x = 0, y = 0, z = 0; };
}  // namespace gtsam
