// include/superslam_hip/place_recognizer.hpp: the reference's six PlaceRecognizer cases (tests/test_place_recognizer.cc:22-106),
// restated without gtest / OpenCV, plus EigenPlaces' error conventions without a GPU.  With a weights path (GPU): descriptor of a
// procedural image written to stdout (the Python test compares it with the oracle and with the Python mirror).
#include <cstdio>

#include "superslam_hip/place_recognizer.hpp"

using namespace superslam_hip;
static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); ++g_fail; } } while (0)

static GlobalDescriptor desc(int dim, int seed, float jitter = 0.f) {
  GlobalDescriptor d(dim, 0.f);
  d[seed % dim] = 1.0f;
  d[(seed + 1) % dim] = 0.5f + jitter;
  return d;
}

int main(int argc, char** argv) {
  {  // RanksNearDuplicateAboveDistinct
    CosineDescriptorIndex idx;
    idx.add(0, desc(16, 3)); idx.add(1, desc(16, 9));
    auto res = idx.query(desc(16, 3, 0.01f), 0, 5, 0.0f);
    CHECK(!res.empty() && res.front().keyframe_id == 0u && res.front().score > 0.95f);
    if (res.size() > 1) CHECK(res[1].score < res.front().score);
  }
  {  // ExcludeRecentSkipsTemporalNeighbours
    CosineDescriptorIndex idx;
    for (int i = 0; i < 5; ++i) idx.add(i, desc(16, i));
    for (const auto& c : idx.query(desc(16, 4), 2, 5, 0.0f)) CHECK(c.keyframe_id < 3u);
  }
  {  // TopKAndMinScoreGate
    CosineDescriptorIndex idx;
    for (int i = 0; i < 6; ++i) idx.add(i, desc(16, i));
    CHECK(idx.query(desc(16, 0), 0, 2, -1.0f).size() <= 2u);
    for (const auto& c : idx.query(desc(16, 0), 0, 10, 0.99f)) CHECK(c.score >= 0.99f);
    auto all = idx.query(desc(16, 0), 0, 0, -1.0f);
    for (size_t i = 1; i < all.size(); ++i) CHECK(all[i - 1].score >= all[i].score);   // descending
  }
  {  // EmptyOrAllExcludedReturnsNothing
    CosineDescriptorIndex idx;
    CHECK(idx.query(desc(16, 0), 0, 5, 0.0f).empty());
    idx.add(0, desc(16, 0));
    CHECK(idx.query(desc(16, 0), 1, 5, 0.0f).empty());
  }
  {  // RequiresConsecutiveConsistentVotes
    TemporalConsistencyVoter voter(3, 2);
    LoopCandidate a{10, 0.9f}, b{11, 0.9f};
    CHECK(!voter.vote(&a)); CHECK(!voter.vote(&b)); CHECK(voter.vote(&a));
  }
  {  // ResetsOnGapOrInconsistency
    TemporalConsistencyVoter voter(2, 1);
    LoopCandidate a{10, 0.9f}, far{99, 0.9f};
    CHECK(!voter.vote(&a)); CHECK(!voter.vote(nullptr)); CHECK(!voter.vote(&a)); CHECK(!voter.vote(&far)); CHECK(voter.vote(&far));
  }
  {  // error conventions: missing weights (or no GPU) -> initialize() false, empty descriptor, query on an empty index
    EigenPlaces ep("/nonexistent/eigenplaces.safetensors", 512, 512);
    CHECK(!ep.initialize());
    std::vector<uint8_t> px(64 * 64, 7);
    CHECK(ep.compute_global_descriptor(Image{px.data(), 64, 64, 1, 0}).empty());
    CHECK(ep.query(desc(16, 0), 0, 5).empty());
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
    ep.add(0, d);
    CHECK(ep.query(d, 0, 5).size() == 1);
  }
  if (g_fail) { std::printf("%d check(s) failed\n", g_fail); return 1; }
  std::printf("place recognizer: all checks passed\n");
  return 0;
}
