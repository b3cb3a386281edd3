// include/superslam_hip/place_recognizer.hpp: EigenPlaces' error conventions without a GPU (the retrieval index is the reference's own
// code and is exercised, compiled from the reference tree, by tests/cpp/test_reference_binding.cc).  With a weights path (GPU): descriptor of a
// procedural image written to stdout (the Python test compares it with the oracle and with the Python mirror).
#include <cstdio>

#include "superslam_hip/place_recognizer.hpp"

using namespace superslam_hip;
static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); ++g_fail; } } while (0)

int main(int argc, char** argv) {
  {  // error conventions: missing weights (or no GPU) -> initialize() false, empty descriptor
    EigenPlaces ep("/nonexistent/eigenplaces.safetensors", 512, 512);
    CHECK(!ep.initialize());
    std::vector<uint8_t> px(64 * 64, 7);
    CHECK(ep.compute_global_descriptor(Image{px.data(), 64, 64, 1, 0}).empty());
  }
  if (argc >= 2) {  // GPU: descriptor of a deterministic gray ramp + blocks image, 376 x 1241
    EigenPlaces ep(argv[1], 512, 512);
    CHECK(ep.initialize());
    const int H = 376, W = 1241;
    std::vector<uint8_t> img(static_cast<size_t>(H) * W);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) img[static_cast<size_t>(y) * W + x] = static_cast<uint8_t>((x * 7 + y * 13 + ((x / 40 + y / 30) % 5) * 37) & 255);
    const GlobalDescriptor d = ep.compute_global_descriptor(Image{img.data(), H, W, 1, 0});
    CHECK(d.size() == 512);
    std::printf("DESC");
    for (float v : d) std::printf(" %.9g", v);
    std::printf("\n");
  }
  if (g_fail) { std::printf("%d check(s) failed\n", g_fail); return 1; }
  std::printf("place recognizer: all checks passed\n");
  return 0;
}
