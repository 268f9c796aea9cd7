// C++ walk through the loop-closure path of the reference on the GPU, written against the C++ mirror
// (include/srrg2_slam_amd_loop_closure.hpp): LocalMapSelectorBreadthFirst proposes candidates from the pose graph,
// MultiLoopDetectorBruteForce aligns them against the current local map (ONE batched compute through the C ABI) and
// applies the reference's accept gates, GraphSLAMLifecycle adds the closures disabled, validates them and runs the
// global solver; MultiRelocalizer picks the relocalization map.  Mirrors MultiGraphSLAM_::loopDetect / loopValidate /
// optimize (S/system/multi_graph_slam_impl.cpp:190-317) with the detectors of S/registration/loop_detector and
// S/registration/relocalization.  Plain main(), driven by tests/test_cpp_mirror.py (compiled on CPU, run on the GPU box).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <thread>
#include <vector>

#include "srrg2_slam_amd_loop_closure.hpp"

using namespace srrg2_slam_amd;

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

// a room: floor z = 0, walls x = 0 and y = 0, seen from `robot_in_world`, expressed in the robot (local map) frame
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

static Isometry3f motion(double tx, double ty, double tz, double yaw) {
  Isometry3f T = Isometry3f::Identity();
  T.m[0] = (float) std::cos(yaw); T.m[1] = (float) -std::sin(yaw);
  T.m[4] = (float) std::sin(yaw); T.m[5] = (float) std::cos(yaw);
  T.m[3] = (float) tx; T.m[7] = (float) ty; T.m[11] = (float) tz;
  return T;
}

static double maxAbsDiff(const Isometry3f& A, const Isometry3f& B) {
  double e = 0;
  for (int i = 0; i < 12; ++i) e = std::fmax(e, std::fabs((double) A.m[i] - (double) B.m[i]));
  return e;
}

int main() {
  const int M = 8, N = 6000;
  // ground truth: the robot drives a small loop and comes back next to where it started; local map k sits at gt[k]
  std::vector<Isometry3f> gt((size_t) M);
  gt[0] = motion(2.0, 2.0, 1.0, 0.0);
  const double yaw_step = 2.0 * M_PI / M;
  for (int k = 1; k < M; ++k) gt[(size_t) k] = gt[(size_t) k - 1] * motion(0.10, 0.0, 0.0, yaw_step);
  // every local map's cloud in its own frame
  std::vector<std::vector<float>> pts((size_t) M), nrm((size_t) M);
  for (int k = 0; k < M; ++k) sampleRoom(100 + (uint64_t) k, N, gt[(size_t) k], pts[(size_t) k], nrm[(size_t) k]);

  // the pose graph: odometry with a systematic error (drift), estimates = odometry integration
  PoseGraph3D graph;
  graph.param_pcg_tolerance = 1e-9f;
  GraphSLAMLifecycle<PoseGraph3D> slam(graph);
  std::vector<Isometry3f> odom_est((size_t) M);
  std::vector<LocalMapSelectorBreadthFirst<3>::Factor> factors;
  odom_est[0] = gt[0];
  ASSERT_TRUE(slam.makeNewMap(odom_est[0], Isometry3f::Identity()) == 0);
  for (int k = 1; k < M; ++k) {
    const Isometry3f Z = motion(0.10 + 0.004, 0.003, -0.002, yaw_step + 0.004);  // biased odometry
    odom_est[(size_t) k] = odom_est[(size_t) k - 1] * Z;
    ASSERT_TRUE(slam.makeNewMap(odom_est[(size_t) k], Z) == k);
    factors.push_back({k - 1, k, true});
  }
  const int source = M - 1;
  const double drift_before = maxAbsDiff(odom_est[(size_t) source], gt[(size_t) source]);
  ASSERT_TRUE(drift_before > 0.02);

  // LocalMapSelectorBreadthFirst_: candidates around the current local map, hop-count visit over the enabled factors
  LocalMapSelectorBreadthFirst<3> selector;
  selector.param_max_local_map_distance = 0.25f;
  std::map<int, Isometry3f> estimates;
  for (int k = 0; k < M; ++k) estimates[k] = odom_est[(size_t) k];
  std::vector<ClosureHint<3>> hints = selector.compute(estimates, factors, source, odom_est[(size_t) source]);
  ASSERT_TRUE(!hints.empty());
  ASSERT_TRUE(selector.costs().at(0) == M - 1 && selector.costs().at(source - 1) == 1);
  bool has_start = false;
  for (ClosureHint<3>& h : hints) {
    ASSERT_TRUE(h.local_map_id != source);
    has_start = has_start || h.local_map_id == 0;
    h.moving = pts[(size_t) h.local_map_id].data();  // the candidate's slice
    h.moving_normals = nrm[(size_t) h.local_map_id].data();
    h.size = N;
  }
  ASSERT_TRUE(has_start);  // the loop: local map 0 is 7 hops away on the graph but within range
  // a hint whose local map does not carry the slice is skipped (multi_loop_detector_brute_force_impl.cpp:71-75)
  ClosureHint<3> empty_hint;
  empty_hint.local_map_id = 1234;
  hints.push_back(empty_hint);

  // MultiLoopDetectorBruteForce_: one batched alignment of all candidates against the current local map
  MultiAligner3DQR aligner;
  srrg2_slice_config c = MultiAligner3DQR::defaultSliceConfig();
  c.kind                      = SRRG2_SLICE_P2PLANE;
  c.finder_max_distance       = 0.3f;
  c.finder_normal_cos         = 0.8f;
  c.robustifier               = SRRG2_ROBUST_CAUCHY;
  c.robustifier_chi_threshold = 0.05f;
  aligner.addSlice(c);
  aligner.param_max_iterations = 15;
  MultiLoopDetectorBruteForce<MultiAligner3DQR> detector;
  bool thrown = false;
  try {
    detector.compute(source, hints);
  } catch (const std::runtime_error&) {
    thrown = true;  // no aligner (:52-54)
  }
  ASSERT_TRUE(thrown);
  detector.param_relocalize_aligner = &aligner;
  detector.setFixed(pts[(size_t) source].data(), nrm[(size_t) source].data(), N);
  const std::vector<LoopClosure<3>>& closures = detector.compute(source, hints);
  ASSERT_TRUE(detector.attemptedClosures().size() == hints.size() - 1);
  ASSERT_TRUE(!closures.empty());
  ASSERT_TRUE(closures.size() + detector.drops().size() == detector.attemptedClosures().size());
  for (const LoopClosure<3>& cl : closures) {
    ASSERT_TRUE(cl.source_graph_id == source && cl.num_inliers >= 500 && cl.chi_inliers <= 0.005f);
    ASSERT_TRUE((float) cl.num_inliers / (float) cl.num_correspondences >= 0.7f);
    // measurement = moving_in_fixed = target local map in the current one: against the ground truth
    const Isometry3f expect = gt[(size_t) source].inverse() * gt[(size_t) cl.target_graph_id];
    ASSERT_TRUE(maxAbsDiff(cl.measurement, expect) < 1.5e-2);
    ASSERT_TRUE(cl.information[0] == 1.f && cl.information[7] == 1.f && cl.information[1] == 0.f);  // Omega = I (:120-131)
    ASSERT_TRUE(cl.aligner_information[0] > 0.f);  // H of the alignment travels with the record
    ASSERT_TRUE(!cl.enabled);
  }

  // ---- the same detection spread over TWO aligner handles (k -> handle k mod 2, one host thread each; here both on
  // device 0, on a node one per GPU): closures, drops and their order are those of the one-handle run, byte for byte,
  // and the record table (every handle fills its rows of a zero table, the tables are summed: the all-reduce of
  // SURVEY.md 8e done in host memory) holds every record of the one-handle batch
  {
    MultiAligner3DQR aligner_b;
    aligner_b.addSlice(c);
    aligner_b.param_max_iterations = aligner.param_max_iterations;
    const std::vector<LoopClosure<3>> single = closures;  // (copy: the detector's vector is rewritten below)
    const std::vector<std::pair<int, std::string>> single_drops = detector.drops();
    detector.param_relocalize_aligners = {&aligner_b};
    const std::vector<LoopClosure<3>>& both = detector.compute(source, hints);
    ASSERT_TRUE(both.size() == single.size() && detector.drops() == single_drops);
    for (size_t k = 0; k < both.size() && k < single.size(); ++k) {
      ASSERT_TRUE(both[k].target_graph_id == single[k].target_graph_id);
      ASSERT_TRUE(std::memcmp(both[k].measurement.data(), single[k].measurement.data(), sizeof(float) * 12) == 0);
      ASSERT_TRUE(std::memcmp(both[k].aligner_information, single[k].aligner_information, sizeof(both[k].aligner_information)) == 0);
      ASSERT_TRUE(both[k].num_inliers == single[k].num_inliers && both[k].chi_inliers == single[k].chi_inliers);
    }
    // the batch itself and its record table
    std::vector<const float*> cl, nr;
    std::vector<int> sz;
    std::vector<Isometry3f> gs;
    for (const ClosureHint<3>& h : hints)
      if (h.moving) {
        cl.push_back(h.moving); nr.push_back(h.moving_normals); sz.push_back(h.size); gs.push_back(h.initial_guess);
      }
    aligner.setFixed(0, pts[(size_t) source].data(), 12, nrm[(size_t) source].data(), 12, N);
    const std::vector<srrg2_batch_result> one = aligner.computeBatch(cl, sz, nr, gs);
    ShardedAligners<MultiAligner3DQR> two({&aligner, &aligner_b});
    two.setFixed(0, pts[(size_t) source].data(), 12, nrm[(size_t) source].data(), 12, N);
    const std::vector<srrg2_batch_result> res2 = two.computeBatch(cl, sz, nr, gs);
    ASSERT_TRUE(one.size() == res2.size());
    for (size_t k = 0; k < one.size() && k < res2.size(); ++k) ASSERT_TRUE(std::memcmp(&one[k], &res2[k], sizeof(srrg2_batch_result)) == 0);
    const std::vector<double> table = two.recordTable(res2, SRRG2_SE3_QUAT_RIGHT);
    for (size_t k = 0; k < one.size(); ++k) {
      double rec[SRRG2_RECORD_FLOATS];
      ASSERT_TRUE(srrg2_multi_gpu_pack_record((int) k, SRRG2_SE3_QUAT_RIGHT, &one[k], rec) == 0);
      ASSERT_TRUE(std::memcmp(rec, table.data() + k * SRRG2_RECORD_FLOATS, sizeof(rec)) == 0);
    }
    // ---- EIGHT handles (the shape of a node: one per GPU, here all on device 0), 64 candidates, 8 per handle: the batch
    // and its record table equal the one-handle batch byte for byte
    {
      std::vector<std::unique_ptr<MultiAligner3DQR>> extra;
      std::vector<MultiAligner3DQR*> eight = {&aligner, &aligner_b};
      for (int g = 2; g < 8; ++g) {
        extra.emplace_back(new MultiAligner3DQR());
        extra.back()->addSlice(c);
        extra.back()->param_max_iterations = aligner.param_max_iterations;
        eight.push_back(extra.back().get());
      }
      std::vector<const float*> cl8, nr8;
      std::vector<int> sz8;
      std::vector<Isometry3f> gs8;
      for (int k = 0; k < 64; ++k) {  // the candidates again and again, ragged: candidate k loses its last 37 k points
        const size_t j = (size_t) k % cl.size();
        cl8.push_back(cl[j]); nr8.push_back(nr[j]); sz8.push_back(sz[j] - 37 * k); gs8.push_back(gs[j]);
      }
      aligner.setFixed(0, pts[(size_t) source].data(), 12, nrm[(size_t) source].data(), 12, N);
      const std::vector<srrg2_batch_result> one8 = aligner.computeBatch(cl8, sz8, nr8, gs8);
      ShardedAligners<MultiAligner3DQR> all8(eight);
      all8.setFixed(0, pts[(size_t) source].data(), 12, nrm[(size_t) source].data(), 12, N);
      const std::vector<srrg2_batch_result> res8 = all8.computeBatch(cl8, sz8, nr8, gs8);
      ASSERT_TRUE(one8.size() == 64 && res8.size() == 64);
      for (size_t k = 0; k < one8.size() && k < res8.size(); ++k)
        ASSERT_TRUE(std::memcmp(&one8[k], &res8[k], sizeof(srrg2_batch_result)) == 0);
      const std::vector<double> table8 = all8.recordTable(res8, SRRG2_SE3_QUAT_RIGHT);
      ASSERT_TRUE(table8.size() == (size_t) 64 * SRRG2_RECORD_FLOATS);
      for (size_t k = 0; k < one8.size(); ++k) {
        double rec[SRRG2_RECORD_FLOATS];
        ASSERT_TRUE(srrg2_multi_gpu_pack_record((int) k, SRRG2_SE3_QUAT_RIGHT, &one8[k], rec) == 0);
        ASSERT_TRUE(std::memcmp(rec, table8.data() + k * SRRG2_RECORD_FLOATS, sizeof(rec)) == 0);
      }
      for (int g = 0; g < 8; ++g) {  // handle g took k = g, g + 8, ...: 8 candidates each
        int n = 0;
        ASSERT_TRUE((n = srrg2_multi_gpu_shard_count(64, 8, g)) == 8);
      }
      // ... and the same table between PROCESSES, natively on RCCL (RcclRecordExchange: ncclCommInitRank from a unique id,
      // ONE ncclAllReduce(sum) over the int64 patterns of the rows).  A one-GPU box can form a one-rank communicator: the
      // rank owns every row, the collective runs for real (RCCL's kernel on the device) and must hand the table back bit for
      // bit -- the table of the one-process path above and, through srrg2_multi_gpu_pack_record, of distributed.py
      // (tests/test_multi_gpu_gloo.py compares the two packers).  A rank of a larger world contributes zeros elsewhere:
      // x + 0 = x in int64 whatever the pattern, which the second call below checks on a table with -0.0, NaN and inf rows.
      try {
        const RcclRecordExchange::UniqueId id = RcclRecordExchange::createId();
        RcclRecordExchange ex(id, 1, 0, 0);
        std::vector<double> t = ex.localTable(one8, 64, SRRG2_SE3_QUAT_RIGHT);
        ASSERT_TRUE(t.size() == table8.size() && std::memcmp(t.data(), table8.data(), t.size() * sizeof(double)) == 0);
        ex.allReduce(t);
        ASSERT_TRUE(std::memcmp(t.data(), table8.data(), t.size() * sizeof(double)) == 0);
        std::vector<double> odd = {-0.0, std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::infinity(), 1e-310, -1.5};
        const std::vector<double> odd0 = odd;
        ex.allReduce(odd);
        ASSERT_TRUE(std::memcmp(odd.data(), odd0.data(), odd.size() * sizeof(double)) == 0);
        std::printf("RCCL record exchange: one-rank communicator, 64 x %d table ok\n", SRRG2_RECORD_FLOATS);
      } catch (const std::runtime_error& e) {
        std::fprintf(stderr, "RCCL record exchange: %s\n", e.what());
        ++g_failures;
      }
    }
    detector.param_relocalize_aligners.clear();
    detector.compute(source, hints);  // (back to the one-handle state the checks below read)

    // ---- ONE alignment sharded by moving points over two handles of this process (srrg2_aligner_set_point_shard): the
    // reduction hook adds the handles' integer sums through host memory before every control step; estimate and
    // statistics are bit for bit those of the unsharded alignment, whatever the deal (here: even / odd points)
    const int target = hints[0].local_map_id;
    aligner.setFixed(0, pts[(size_t) source].data(), 12, nrm[(size_t) source].data(), 12, N);
    aligner.setMoving(0, pts[(size_t) target].data(), 12, nrm[(size_t) target].data(), 12, N);
    aligner.setMovingInFixed(hints[0].initial_guess);
    aligner.compute();
    const Isometry3f X_whole = aligner.movingInFixed();
    const IterationStatsVector st_whole = aligner.iterationStats();
    std::vector<float> half_p[2], half_n[2];
    for (int i = 0; i < N; ++i)
      for (int a = 0; a < 3; ++a) {
        half_p[i & 1].push_back(pts[(size_t) target][(size_t) i * 3 + a]);
        half_n[i & 1].push_back(nrm[(size_t) target][(size_t) i * 3 + a]);
      }
    MultiAligner3DQR* hs[2] = {&aligner, &aligner_b};
    auto run_sharded = [&](srrg2_reduce_fn hook, void* user0, void* user1) {
      void* users[2] = {user0, user1};
      std::vector<std::thread> th;
      for (int g = 0; g < 2; ++g)
        th.emplace_back([&, g]() {
          hs[g]->setFixed(0, pts[(size_t) source].data(), 12, nrm[(size_t) source].data(), 12, N);
          hs[g]->setMoving(0, half_p[g].data(), 12, half_n[g].data(), 12, (int) half_p[g].size() / 3);
          hs[g]->setPointShard(hook, users[g], N);
          hs[g]->setMovingInFixed(hints[0].initial_guess);
          hs[g]->compute();
        });
      for (std::thread& t : th) t.join();
    };
    {
      HostPointShardReducer red(2);
      run_sharded(&HostPointShardReducer::hook, red.participant(0), red.participant(1));
      for (int g = 0; g < 2; ++g) {
        ASSERT_TRUE(std::memcmp(hs[g]->movingInFixed().data(), X_whole.data(), sizeof(float) * 12) == 0);
        const IterationStatsVector& sg = hs[g]->iterationStats();
        ASSERT_TRUE(sg.size() == st_whole.size());
        for (size_t i = 0; i < sg.size() && i < st_whole.size(); ++i) ASSERT_TRUE(std::memcmp(&sg[i], &st_whole[i], sizeof(IterationStats)) == 0);
        hs[g]->setPointShard(nullptr, nullptr, 0);
      }
    }
    // ... and the RCCL hook (librccl.so opened at run time): a one-rank communicator on this box -- the hook, its
    // collective and the stream ordering with nothing to add -- must leave the plain alignment's bits
    try {
      RcclPointShardReducer rccl({0});
      aligner.setMoving(0, pts[(size_t) target].data(), 12, nrm[(size_t) target].data(), 12, N);
      aligner.setPointShard(&RcclPointShardReducer::hook, rccl.participant(0), N);
      aligner.setMovingInFixed(hints[0].initial_guess);
      aligner.compute();
      ASSERT_TRUE(std::memcmp(aligner.movingInFixed().data(), X_whole.data(), sizeof(float) * 12) == 0);
      aligner.setPointShard(nullptr, nullptr, 0);
      std::printf("RCCL point-shard hook: one-rank communicator ok\n");
    } catch (const std::runtime_error& e) {
      std::fprintf(stderr, "RCCL hook: %s\n", e.what());
      ++g_failures;
    }
  }

  // MultiRelocalizer_ without an aligner: the best closure on the detector's statistics (multi_relocalizer_impl.cpp:27-66)
  std::vector<MultiRelocalizer<MultiAligner3DQR>::Candidate> cands;
  for (const LoopClosure<3>& cl : closures) {
    MultiRelocalizer<MultiAligner3DQR>::Candidate cd;
    cd.closure = cl;
    cd.moving = pts[(size_t) cl.target_graph_id].data();
    cd.moving_normals = nrm[(size_t) cl.target_graph_id].data();
    cd.size = N;
    cands.push_back(cd);
  }
  MultiRelocalizer<MultiAligner3DQR> relocalizer;
  ASSERT_TRUE(relocalizer.compute(cands) >= 0 && relocalizer.relocalized());
  // ... and with one: every candidate is re-aligned against the current measurement, lowest chi per inlier wins (:74-137)
  relocalizer.param_aligner = &aligner;
  relocalizer.setFixed(pts[(size_t) source].data(), nrm[(size_t) source].data(), N);
  const int reloc = relocalizer.compute(cands);
  ASSERT_TRUE(reloc >= 0);
  {
    const Isometry3f expect = gt[(size_t) reloc].inverse() * gt[(size_t) source];  // robot (current map) in the relocalization map
    ASSERT_TRUE(maxAbsDiff(relocalizer.robotInLocalMap(), expect) < 1.5e-2);
  }
  relocalizer.param_max_translation = 1e-4f;  // every candidate is farther than that: MAX_TRANSITION DROP (:38-42)
  ASSERT_TRUE(relocalizer.compute(cands) < 0 && relocalizer.drops().size() == cands.size());

  // MultiGraphSLAM_::loopValidate + optimize: closures enter disabled, the validator rejects one, the solver runs
  int nv = 0, nf = 0, ne = 0;
  graph.size(nv, nf, ne);
  ASSERT_TRUE(nv == M && nf == M - 1 && ne == M - 1);
  ASSERT_TRUE(slam.optimize().empty());  // no valid closure yet: optimize() is a no-op (:302-304)
  const size_t n_closures = closures.size();
  auto validator = [&](const std::vector<LoopClosure<3>>& cs) {
    std::vector<GraphSLAMLifecycle<PoseGraph3D>::Verdict> v(cs.size(), GraphSLAMLifecycle<PoseGraph3D>::Accepted);
    if (cs.size() > 1) v.back() = GraphSLAMLifecycle<PoseGraph3D>::Rejected;
    return v;
  };
  const std::vector<LoopClosure<3>> accepted = slam.loopValidate(closures, validator);
  ASSERT_TRUE(accepted.size() == (n_closures > 1 ? n_closures - 1 : 1));
  graph.size(nv, nf, ne);
  ASSERT_TRUE(nf == M - 1 + (int) accepted.size() && ne == nf);  // the rejected closure left the graph
  const std::vector<srrg2_posegraph_stats> st = slam.optimize();
  ASSERT_TRUE(!st.empty() && st.back().solver_status == 0);
  ASSERT_TRUE(st.back().chi < 0.2f * st.front().chi);
  const std::vector<Isometry3f> opt = graph.estimates();
  ASSERT_TRUE(maxAbsDiff(opt[0], gt[0]) == 0.0);  // the first local map is Fixed (:86)
  const double drift_after = maxAbsDiff(opt[(size_t) source], gt[(size_t) source]);
  ASSERT_TRUE(drift_after < 0.5 * drift_before);
  std::printf("closures %zu (dropped %zu), relocalization map %d, drift %.4f -> %.4f, chi %.5f -> %.5f\n", n_closures,
              detector.drops().size(), reloc, drift_before, drift_after, st.front().chi, st.back().chi);
  std::printf("%s (%d failed checks)\n", g_failures ? "FAILED" : "PASSED", g_failures);
  return g_failures ? 1 : 0;
}
