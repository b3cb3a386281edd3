// Decoder test driver for include/superslam_hip/image_io.hpp: prints "rows cols\n" + the raw gray pixels of an image file
// (tests/test_frontend_benchmark.py compares them with the arrays it encoded), or the timestamps of a times.txt.
#include <cstdio>
#include <cstring>

#include "superslam_hip/image_io.hpp"

int main(int argc, char** argv) {
  if (argc == 3 && !std::strcmp(argv[1], "--times")) {
    const auto ts = superslam_hip::read_times(argv[2]);
    std::printf("%zu", ts.size());
    for (double t : ts) std::printf(" %.6f", t);
    std::printf("\n");
    return 0;
  }
  if (argc != 2) return 2;
  std::vector<uint8_t> px;
  int rows = 0, cols = 0;
  if (!superslam_hip::read_gray_image(argv[1], px, rows, cols)) return 1;
  std::printf("%d %d\n", rows, cols);
  std::fwrite(px.data(), 1, px.size(), stdout);
  return 0;
}
