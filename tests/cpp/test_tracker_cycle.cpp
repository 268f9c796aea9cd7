// C++ walk through the tracker-side cycle clip() -> align -> merge() (S/trackers/tracker_slice_processor_impl.cpp:111-205)
// written against the C++ mirror of include/srrg2_slam_amd.hpp: SceneClipperBall, MultiAligner3DQR,
// MergerCorrespondenceHomo, with the clouds staying in device memory.  Plain main(), driven by tests/test_cpp_mirror.py
// (compiled on CPU, run on the GPU box).  Checks the contract of the reference classes (status values, merge counts,
// scene growth, pose recovery); the bit-level parity with the oracle is tests/test_gpu_scene.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "srrg2_slam_amd.hpp"

using namespace srrg2_slam_amd;
using Isometry3f = Isometry<3>;

static int g_failures = 0;
#define ASSERT_TRUE(cond)                                                        \
  do {                                                                           \
    if (!(cond)) {                                                               \
      std::fprintf(stderr, "%s:%d: ASSERT failed: %s\n", __FILE__, __LINE__, #cond); \
      ++g_failures;                                                              \
    }                                                                            \
  } while (0)

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  double uniform() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double) (z >> 11) * (1.0 / 9007199254740992.0);
  }
};

// a room: floor z = 0, walls x = 0 and y = 0 (three orthogonal planes constrain all six degrees of freedom)
static void sampleRoom(uint64_t seed, int n, const Isometry3f& robot_in_world, std::vector<float>& pts, std::vector<float>& nrm) {
  Rng r(seed);
  const Isometry3f W = robot_in_world.inverse();
  pts.resize((size_t) n * 3);
  nrm.resize((size_t) n * 3);
  for (int i = 0; i < n; ++i) {
    const int plane = i % 3;
    float p[3] = {(float) (4 * r.uniform()), (float) (4 * r.uniform()), (float) (3 * r.uniform())}, m[3] = {0, 0, 0};
    p[plane == 0 ? 2 : (plane == 1 ? 0 : 1)] = 0.f;
    m[plane == 0 ? 2 : (plane == 1 ? 0 : 1)] = 1.f;
    for (int a = 0; a < 3; ++a) {
      pts[(size_t) i * 3 + a] = W.m[a * 4 + 0] * p[0] + W.m[a * 4 + 1] * p[1] + W.m[a * 4 + 2] * p[2] + W.m[a * 4 + 3];
      nrm[(size_t) i * 3 + a] = W.m[a * 4 + 0] * m[0] + W.m[a * 4 + 1] * m[1] + W.m[a * 4 + 2] * m[2];
    }
  }
}

static Isometry3f smallMotion(double tx, double ty, double yaw) {
  Isometry3f T = Isometry3f::Identity();
  T.m[0] = (float) std::cos(yaw); T.m[1] = (float) -std::sin(yaw);
  T.m[4] = (float) std::sin(yaw); T.m[5] = (float) std::cos(yaw);
  T.m[3] = (float) tx; T.m[7] = (float) ty;
  return T;
}

int main() {
  const int N = 20000;
  Scene<3> scene, clipped, measurement;
  SceneClipperBall<3> clipper;
  clipper.param_range = 20.f;
  MergerCorrespondenceHomo<3> merger;
  merger.param_maximum_distance_geometry_squared = 0.01f;
  merger.param_target_number_of_merges           = 1000000;  // always add what was not merged
  MultiAligner3DQR aligner;
  srrg2_slice_config c = MultiAligner3DQR::defaultSliceConfig();
  c.kind                      = SRRG2_SLICE_P2PLANE;
  c.finder_max_distance       = 0.3f;
  c.robustifier               = SRRG2_ROBUST_CAUCHY;
  c.robustifier_chi_threshold = 0.05f;
  const int slice = aligner.addSlice(c);

  std::vector<float> pts, nrm;
  Isometry3f robot_in_world = Isometry3f::Identity();
  Isometry3f estimate       = Isometry3f::Identity();
  robot_in_world.m[3] = 2.f; robot_in_world.m[7] = 2.f; robot_in_world.m[11] = 1.f;
  estimate = robot_in_world;

  // frame 0 opens the local map: merge without correspondences (merger_correspondence_homo_impl.cpp:30-41)
  sampleRoom(1, N, robot_in_world, pts, nrm);
  measurement.set(pts.data(), 12, nrm.data(), 12, N);
  merger.setScene(&scene);
  merger.setMeasurement(&measurement);
  merger.setMeasurementInScene(estimate);
  merger.setCorrespondences(nullptr);
  merger.compute();
  ASSERT_TRUE(merger.status() == MergerCorrespondenceHomo<3>::Success);
  ASSERT_TRUE(scene.size() == N && merger.last().num_added == N);

  for (int k = 1; k <= 3; ++k) {
    robot_in_world = robot_in_world * smallMotion(0.04, -0.03, 0.01);
    sampleRoom(1 + (uint64_t) k, N, robot_in_world, pts, nrm);
    measurement.set(pts.data(), 12, nrm.data(), 12, N);
    // clip(): the scene around the last estimate, in robot coordinates (scene_clipper.h:106-107)
    clipper.setFullScene(&scene);
    clipper.setClippedSceneInRobot(&clipped);
    clipper.setRobotInLocalMap(estimate);
    clipper.compute();
    ASSERT_TRUE(clipper.status() == SceneClipperBall<3>::Successful);
    ASSERT_TRUE(clipped.size() == scene.size());
    ASSERT_TRUE((int) clipper.globalIndices().size() == clipped.size());
    // align: measurement = fixed, clipped scene = moving, both fed from device memory
    const float *cp, *cn, *mp, *mn;
    int n, m;
    clipped.deviceArrays(cp, cn, n);
    measurement.deviceArrays(mp, mn, m);
    aligner.setMoving(slice, cp, 16, cn, 16, n, SRRG2_MEM_DEVICE);
    aligner.setFixed(slice, mp, 16, mn, 16, m, SRRG2_MEM_DEVICE);
    aligner.setMovingInFixed(Isometry3f::Identity());
    aligner.compute();
    ASSERT_TRUE(aligner.status() == AlignerBase::Success);
    estimate = estimate * aligner.movingInFixed().inverse();  // previous robot frame -> current one
    // merge(): correspondences flipped and mapped to the global scene on the device (:160-186)
    const int before = scene.size();
    merger.setMeasurementInScene(estimate);
    merger.computeFromAligner(aligner, slice, clipped);
    ASSERT_TRUE(merger.status() == MergerCorrespondenceHomo<3>::Success);
    ASSERT_TRUE(merger.last().num_correspondences > N / 2);
    ASSERT_TRUE(merger.last().num_merged > N / 10);
    ASSERT_TRUE(merger.last().num_merged + merger.last().num_added <= N);
    ASSERT_TRUE(scene.size() == before + merger.last().num_added);
    double err = 0;
    for (int i = 0; i < 12; ++i) err = std::fmax(err, std::fabs(estimate.m[i] - robot_in_world.m[i]));
    ASSERT_TRUE(err < 2e-2);
  }
  // misuse throws, as in the reference
  bool thrown = false;
  try {
    MergerCorrespondenceHomo<3> m2;
    m2.compute();
  } catch (const std::runtime_error&) {
    thrown = true;
  }
  ASSERT_TRUE(thrown);
  std::printf("%s (%d failed checks)\n", g_failures ? "FAILED" : "PASSED", g_failures);
  return g_failures ? 1 : 0;
}
