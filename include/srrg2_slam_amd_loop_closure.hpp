// srrg2_slam_amd_loop_closure.hpp -- header-only C++17 mirror of the reference's loop-closure drivers and of the
// pose-graph lifecycle over the C ABI.  Same vocabulary as the reference: class
// names, PARAM names and defaults, gate order and drop reasons; srrg2_core / srrg2_solver types are replaced by the
// PODs of srrg2_slam_amd.h.  All arithmetic of the hot path stays behind the C ABI; what lives here is the host logic
// of the callers of MultiAlignerBase_::compute() and of global_solver->compute().
//
//   LoopClosure                      LoopClosure_<LocalMap, Factor>             registration/loop_closure.h:21-79
//   ClosureHint                      LocalMapSelector_::ClosureHint             registration/local_map_selectors/local_map_selector.h
//   LocalMapSelectorBreadthFirst     LocalMapSelectorBreadthFirst_              local_map_selector_breadth_first.h:24-48, _impl.cpp:12-101
//   MultiLoopDetectorBruteForce      MultiLoopDetectorBruteForce_               multi_loop_detector_brute_force.h:20-41, _impl.cpp:12-132
//   MultiRelocalizer                 MultiRelocalizer_                          multi_relocalizer.h:29-43, _impl.cpp:12-145
//   PoseGraph                        the global Solver + FactorGraph of MultiGraphSLAM_   system/multi_graph_slam.h:50-54
//   GraphSLAMLifecycle               MultiGraphSLAM_::makeNewMap / loopValidate / optimize  system/multi_graph_slam_impl.cpp:52-90,227-317
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <queue>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "srrg2_slam_amd.hpp"
#include "srrg2_slam_amd_multi_device.hpp"

namespace srrg2_slam_amd {

template <int DIM>
inline float translationNorm(const Isometry<DIM>& T) {
  if (DIM == 2) return std::sqrt(T.m[2] * T.m[2] + T.m[5] * T.m[5]);
  return std::sqrt(T.m[3] * T.m[3] + T.m[7] * T.m[7] + T.m[11] * T.m[11]);
}
template <int DIM>
inline void zeroTranslation(Isometry<DIM>& T) {
  if (DIM == 2) {
    T.m[2] = T.m[5] = 0.f;
  } else {
    T.m[3] = T.m[7] = T.m[11] = 0.f;
  }
}

// ---- LoopClosure_ (loop_closure.h:21-79): the factor's payload + the detector's statistics -------------------------------
template <int DIM>
struct LoopClosure {
  static constexpr int D = DIM == 2 ? 3 : 6;
  int source_graph_id = -1;  // the moving local map
  int target_graph_id = -1;  // the fixed local map
  Isometry<DIM> measurement    = Isometry<DIM>::Identity();  // setMeasurement(moving_in_fixed)
  float information[D * D]     = {};                         // setInformationMatrix, row-major
  Isometry<DIM> pose_in_target = Isometry<DIM>::Identity();
  float chi_inliers            = 1e9f;
  size_t num_inliers           = 0;
  size_t num_correspondences   = 0;
  bool enabled                 = false;  // closures are created disabled (:71)
  int factor_id                = -1;     // id in the pose graph once added
  // not in the reference: H of the alignment's last Gauss-Newton iteration (the information the aligner itself
  // measured); the reference's detectors emit Omega = I (multi_loop_detector_brute_force_impl.cpp:120-131)
  float aligner_information[D * D] = {};
};

// ---- ClosureHint: a candidate local map with the initial guess of target-in-source -----------------------------------------
template <int DIM>
struct ClosureHint {
  int local_map_id = -1;
  Isometry<DIM> initial_guess = Isometry<DIM>::Identity();
  float cost = 0.f;  // graph distance of the visit
  // the candidate's cloud in its own frame (the slice the aligner's moving side binds): packed DIM-float records
  const float* moving         = nullptr;
  const float* moving_normals = nullptr;
  int size = 0;
};

// ---- LocalMapSelectorBreadthFirst_ ----------------------------------------------------------------------------------------
template <int DIM>
class LocalMapSelectorBreadthFirst {
public:
  using EstimateType = Isometry<DIM>;
  // PARAMs, local_map_selector_breadth_first.h:24-48
  int param_relocalize_range_scale                          = 2;
  int param_aggressive_relocalize_graph_distance            = 10;
  int param_aggressive_relocalize_graph_max_range           = 20;
  float param_aggressive_relocalize_range_increase_per_edge = 0.1f;
  float param_max_local_map_distance                        = 1.0f;

  struct Factor { int from, to; bool enabled; };
  // estimates: local map id -> pose (graph->variables(), ordered by id); the visit is srrg2_solver's FactorGraphVisit
  // with FactorGraphVisitCostUniform (_impl.cpp:52-59, un-vendored): a breadth-first search over the enabled factors
  const std::vector<ClosureHint<DIM>>& compute(const std::map<int, EstimateType>& estimates, const std::vector<Factor>& factors,
                                              int source_id, const EstimateType& robot_in_world) {
    auto src = estimates.find(source_id);
    if (src == estimates.end()) throw std::runtime_error("LocalMapSelectorBreadthFirst_::compute| _current_local_map is NULL");
    std::map<int, std::vector<int>> adj;
    for (const Factor& f : factors) {
      if (!f.enabled) continue;
      adj[f.from].push_back(f.to);
      adj[f.to].push_back(f.from);
    }
    _costs.clear();
    _costs[source_id] = 0;
    std::queue<int> frontier;
    frontier.push(source_id);
    while (!frontier.empty()) {
      const int v = frontier.front();
      frontier.pop();
      for (int w : adj[v])
        if (!_costs.count(w)) {
          _costs[w] = _costs[v] + 1;
          frontier.push(w);
        }
    }
    const EstimateType world_in_robot = robot_in_world.inverse();
    const EstimateType source_inv     = src->second.inverse();
    _hints.clear();
    for (const auto& kv : estimates) {
      const int vid = kv.first;
      if (vid == source_id || !_costs.count(vid)) continue;
      const EstimateType target_in_robot = world_in_robot * kv.second;  // :69
      EstimateType guess                 = source_inv * kv.second;      // :70-71
      const float c = (float) _costs[vid];
      float range_scale = (float) param_relocalize_range_scale * c * param_aggressive_relocalize_range_increase_per_edge + 1.f;  // :78-80
      range_scale       = std::min(range_scale, (float) param_aggressive_relocalize_graph_max_range);
      if (translationNorm(target_in_robot) > param_max_local_map_distance * range_scale) continue;  // :82-85
      if (c > (float) param_aggressive_relocalize_graph_distance) zeroTranslation(guess);            // :88-90
      ClosureHint<DIM> h;
      h.local_map_id  = vid;
      h.initial_guess = guess;
      h.cost          = c;
      _hints.push_back(h);
    }
    return _hints;
  }
  const std::vector<ClosureHint<DIM>>& hints() const { return _hints; }
  const std::map<int, int>& costs() const { return _costs; }

private:
  std::vector<ClosureHint<DIM>> _hints;
  std::map<int, int> _costs;
};

// ---- MultiLoopDetectorBruteForce_ -----------------------------------------------------------------------------------------
template <typename AlignerType>
class MultiLoopDetectorBruteForce {
public:
  static constexpr int DIM = AlignerType::Dim;
  using EstimateType       = Isometry<DIM>;
  using LoopClosureType    = LoopClosure<DIM>;
  // PARAMs, multi_loop_detector_brute_force.h:20-41
  AlignerType* param_relocalize_aligner    = nullptr;
  // not in the reference (one process, one aligner): more handles with the same slices and PARAMs, one per device.  The
  // candidates are then spread k -> handle k mod G over host threads (srrg2_slam_amd_multi_device.hpp); closures, drops and
  // their order are those of the one-handle run.
  std::vector<AlignerType*> param_relocalize_aligners;
  unsigned param_relocalize_min_inliers    = 500;
  float param_relocalize_max_chi_inliers   = 0.005f;
  float param_relocalize_min_inliers_ratio = 0.7f;

  // the fixed side: the current local map's slice (aligner->setFixed once, _impl.cpp:63)
  void setFixed(const float* coords, const float* normals, int n) {
    _fixed = coords; _fixed_normals = normals; _nfixed = n;
  }
  // one independent alignment per hint (:64-79) -- ONE compute_batch call -- then the accept gates (:80-112) and the
  // closure record (:120-131)
  const std::vector<LoopClosureType>& compute(int source_local_map_id, const std::vector<ClosureHint<DIM>>& hints,
                                              const EstimateType& pose_in_current = EstimateType::Identity()) {
    if (!param_relocalize_aligner) throw std::runtime_error("MultiLoopDetectorBruteForce_::compute| no aligner");  // :52-54
    _attempted_closures.clear();
    _detected_closures.clear();
    _drops.clear();
    std::vector<const float*> clouds, normals;
    std::vector<int> sizes, ids;
    std::vector<EstimateType> guesses;
    bool all_normals = true;
    for (const ClosureHint<DIM>& h : hints) {
      if (!h.moving) continue;  // :71-75: a hint without the slice is skipped
      _attempted_closures.push_back(h.local_map_id);
      clouds.push_back(h.moving);
      normals.push_back(h.moving_normals);
      all_normals = all_normals && h.moving_normals != nullptr;
      sizes.push_back(h.size);
      ids.push_back(h.local_map_id);
      guesses.push_back(h.initial_guess);
    }
    if (clouds.empty()) return _detected_closures;
    if (!all_normals) {
      for (const float* nm : normals)
        if (nm) throw std::runtime_error("MultiLoopDetectorBruteForce_::compute| some hints carry normals and some do not");
      normals.clear();
    }
    std::vector<AlignerType*> handles{param_relocalize_aligner};
    handles.insert(handles.end(), param_relocalize_aligners.begin(), param_relocalize_aligners.end());
    ShardedAligners<AlignerType> sharded(handles);
    sharded.setFixed(0, _fixed, DIM * 4, _fixed_normals, DIM * 4, _nfixed);  // aligner->setFixed, once per handle (:63)
    const std::vector<srrg2_batch_result> results = sharded.computeBatch(clouds, sizes, normals, guesses);
    for (size_t k = 0; k < results.size(); ++k) {
      const srrg2_batch_result& r = results[k];
      if (r.status != AlignerBase::Success) {  // :80-84
        _drops.emplace_back(ids[k], "ALIGNER DROP [code: " + std::to_string(r.status) + "]");
        continue;
      }
      const int num_correspondences = r.num_correspondences;  // aligner->numCorrespondences() after compute(), :89
      const int num_inliers         = r.last.num_inliers;
      const float chi_inliers       = r.last.chi_inliers / (float) num_inliers;  // :91
      if (num_inliers < (int) param_relocalize_min_inliers) {  // :94-97
        _drops.emplace_back(ids[k], "NUM_INLIERS DROP");
        continue;
      }
      if (chi_inliers > param_relocalize_max_chi_inliers) {  // :99-103
        _drops.emplace_back(ids[k], "MAX_CHI_INLIERS DROP");
        continue;
      }
      const float inlier_ratio = (float) num_inliers / (float) num_correspondences;  // :105
      if (inlier_ratio < param_relocalize_min_inliers_ratio) {  // :107-111
        _drops.emplace_back(ids[k], "MIN_INLIERS_RATIO DROP");
        continue;
      }
      LoopClosureType c;  // :120-131
      c.source_graph_id = source_local_map_id;
      c.target_graph_id = ids[k];
      std::memcpy(c.measurement.data(), r.moving_in_fixed, sizeof(float) * EstimateType::N);
      for (int a = 0; a < LoopClosureType::D; ++a) c.information[a * LoopClosureType::D + a] = 1.f;  // Identity
      c.pose_in_target      = c.measurement.inverse() * pose_in_current;  // :120
      c.chi_inliers         = chi_inliers;
      c.num_inliers         = (size_t) num_inliers;
      c.num_correspondences = (size_t) num_correspondences;
      std::memcpy(c.aligner_information, r.information, sizeof(c.aligner_information));
      _detected_closures.push_back(c);
    }
    return _detected_closures;
  }
  const std::vector<int>& attemptedClosures() const { return _attempted_closures; }
  const std::vector<LoopClosureType>& detectedClosures() const { return _detected_closures; }
  const std::vector<std::pair<int, std::string>>& drops() const { return _drops; }

private:
  const float* _fixed         = nullptr;
  const float* _fixed_normals = nullptr;
  int _nfixed                 = 0;
  std::vector<int> _attempted_closures;
  std::vector<LoopClosureType> _detected_closures;
  std::vector<std::pair<int, std::string>> _drops;
};

// ---- MultiRelocalizer_ ----------------------------------------------------------------------------------------------------
template <typename AlignerType>
class MultiRelocalizer {
public:
  static constexpr int DIM = AlignerType::Dim;
  using EstimateType       = Isometry<DIM>;
  using LoopClosureType    = LoopClosure<DIM>;
  // PARAMs, multi_relocalizer.h:29-43, relocalizer.h:22
  AlignerType* param_aligner               = nullptr;
  std::vector<AlignerType*> param_aligners;  // more handles, one per device (see MultiLoopDetectorBruteForce)
  float param_max_translation              = 3.0f;
  int param_relocalize_min_inliers         = 500;
  float param_relocalize_max_chi_inliers   = 0.005f;
  float param_relocalize_min_inliers_ratio = 0.7f;

  struct Candidate {  // a closure of the detector + (aligner branch) the target local map's cloud
    LoopClosureType closure;
    const float* moving         = nullptr;
    const float* moving_normals = nullptr;
    int size                    = 0;
  };
  // the current measurement (aligner->setFixed(&tracker->measurementContainer()), _impl.cpp:78)
  void setFixed(const float* coords, const float* normals, int n) {
    _fixed = coords; _fixed_normals = normals; _nfixed = n;
  }
  // returns the id of the relocalization map or -1
  int compute(const std::vector<Candidate>& candidates) {
    _relocalization_map = -1;
    _relocalized        = false;
    _robot_in_local_map = EstimateType::Identity();
    _drops.clear();
    std::vector<const Candidate*> near;
    for (const Candidate& c : candidates) {
      if (translationNorm(c.closure.pose_in_target) > param_max_translation) {  // :38-42, :84-88
        _drops.emplace_back(c.closure.target_graph_id, "MAX_TRANSITION DROP");
        continue;
      }
      near.push_back(&c);
    }
    if (!param_aligner) {  // :27-66: the best closure on the detector's statistics
      const Candidate* best = nullptr;
      for (const Candidate* c : near) {
        if (best) {
          if (c->closure.chi_inliers > best->closure.chi_inliers) {  // :45-49
            _drops.emplace_back(c->closure.target_graph_id, "HIGH_CHI_INLIERS DROP");
            continue;
          }
          if (c->closure.num_correspondences < best->closure.num_correspondences) {  // :50-54
            _drops.emplace_back(c->closure.target_graph_id, "LOW_MIN_CORRESPONDENCE DROP");
            continue;
          }
        }
        best = c;
      }
      if (best) {  // :62-66
        _relocalized_closure = best->closure;
        _relocalized         = true;
        _relocalization_map  = best->closure.target_graph_id;
        _robot_in_local_map  = best->closure.pose_in_target;
      }
      return _relocalization_map;
    }
    {  // a candidate whose local map does not carry the slice cannot be aligned: dropped like a hint without one
      std::vector<const Candidate*> with_cloud;
      for (const Candidate* c : near) {
        if (c->moving && c->size > 0)
          with_cloud.push_back(c);
        else
          _drops.emplace_back(c->closure.target_graph_id, "NO_SLICE DROP");
      }
      near.swap(with_cloud);
    }
    if (near.empty()) return -1;
    std::vector<const float*> clouds, normals;
    std::vector<int> sizes;
    std::vector<EstimateType> guesses;
    bool all_normals = true, any_normals = false;
    for (const Candidate* c : near) {
      clouds.push_back(c->moving);
      normals.push_back(c->moving_normals);
      all_normals = all_normals && c->moving_normals != nullptr;
      any_normals = any_normals || c->moving_normals != nullptr;
      sizes.push_back(c->size);
      guesses.push_back(c->closure.pose_in_target.inverse());  // :91
    }
    if (!all_normals) {
      if (any_normals) throw std::runtime_error("MultiRelocalizer_::compute| some candidates carry normals and some do not");
      normals.clear();
    }
    std::vector<AlignerType*> handles{param_aligner};
    handles.insert(handles.end(), param_aligners.begin(), param_aligners.end());
    ShardedAligners<AlignerType> sharded(handles);
    sharded.setFixed(0, _fixed, DIM * 4, _fixed_normals, DIM * 4, _nfixed);
    const std::vector<srrg2_batch_result> results = sharded.computeBatch(clouds, sizes, normals, guesses);
    float best_chi_average = std::numeric_limits<float>::max();
    for (size_t k = 0; k < results.size(); ++k) {
      const srrg2_batch_result& r = results[k];
      const int target            = near[k]->closure.target_graph_id;
      if (r.status != AlignerBase::Success) {  // :93-97
        _drops.emplace_back(target, "ALIGNER DROP [code: " + std::to_string(r.status) + "]");
        continue;
      }
      const int num_inliers = r.last.num_inliers, num_correspondences = r.num_correspondences;  // :100-101
      const float chi_inliers = r.last.chi_inliers / (float) num_inliers;                       // :102
      if (num_inliers < param_relocalize_min_inliers) {  // :108-111
        _drops.emplace_back(target, "NUM_INLIERS DROP");
        continue;
      }
      if (chi_inliers > param_relocalize_max_chi_inliers) {  // :113-117
        _drops.emplace_back(target, "MAX_CHI_INLIERS DROP");
        continue;
      }
      if ((float) num_inliers / (float) num_correspondences < param_relocalize_min_inliers_ratio) {  // :119-125
        _drops.emplace_back(target, "MIN_INLIERS_RATIO DROP");
        continue;
      }
      if (chi_inliers < best_chi_average) {  // :131-137
        _relocalization_map = target;
        EstimateType X;
        std::memcpy(X.data(), r.moving_in_fixed, sizeof(float) * EstimateType::N);
        _robot_in_local_map  = X.inverse();
        best_chi_average     = chi_inliers;
        _relocalized_closure = near[k]->closure;
        _relocalized         = true;
      }
    }
    return _relocalization_map;
  }
  int relocalizationMap() const { return _relocalization_map; }
  bool relocalized() const { return _relocalized; }
  const LoopClosureType& relocalizedClosure() const { return _relocalized_closure; }
  const EstimateType& robotInLocalMap() const { return _robot_in_local_map; }
  const std::vector<std::pair<int, std::string>>& drops() const { return _drops; }

private:
  const float* _fixed         = nullptr;
  const float* _fixed_normals = nullptr;
  int _nfixed                 = 0;
  int _relocalization_map     = -1;
  bool _relocalized           = false;
  LoopClosureType _relocalized_closure;
  EstimateType _robot_in_local_map = EstimateType::Identity();
  std::vector<std::pair<int, std::string>> _drops;
};

// ---- the pose graph + global solver of MultiGraphSLAM_ ---------------------------------------------------------------------
template <int VARIABLE_KIND>
class PoseGraph_ {
public:
  static constexpr int DIM = VARIABLE_KIND == SRRG2_SE2_RIGHT ? 2 : 3;
  static constexpr int D   = DIM == 2 ? 3 : 6;
  using EstimateType       = Isometry<DIM>;
  // PARAMs of the global solver (srrg2_posegraph_params)
  int param_max_iterations     = 10;
  int param_pcg_max_iterations = 600;
  float param_pcg_tolerance    = 1e-6f;
  float param_damping          = 0.f;

  explicit PoseGraph_(int device = 0) { check(srrg2_posegraph_create(VARIABLE_KIND, device, &_h)); }
  ~PoseGraph_() { srrg2_posegraph_destroy(_h); }
  PoseGraph_(const PoseGraph_&)            = delete;
  PoseGraph_& operator=(const PoseGraph_&) = delete;
  int addVariable(const EstimateType& pose, bool fixed) {  // _graph->addVariable(local map)
    int id = -1;
    check(srrg2_posegraph_add_variable(_h, pose.data(), fixed ? 1 : 0, &id));
    return id;
  }
  int addFactor(int from, int to, const EstimateType& Z, const float* information /* D x D or null */, bool enabled) {
    int id = -1;
    check(srrg2_posegraph_add_factor(_h, from, to, Z.data(), information, enabled ? 1 : 0, &id));
    return id;
  }
  void setFactorEnabled(int factor_id, bool enabled) { check(srrg2_posegraph_set_factor_enabled(_h, factor_id, enabled ? 1 : 0)); }
  void removeFactor(int factor_id) { check(srrg2_posegraph_remove_factor(_h, factor_id)); }
  void size(int& variables, int& factors, int& enabled_factors) const { check(srrg2_posegraph_size(_h, &variables, &factors, &enabled_factors)); }
  std::vector<srrg2_posegraph_stats> compute() {  // global_solver->compute(), blocking
    srrg2_posegraph_params p{param_max_iterations, param_pcg_max_iterations, param_pcg_tolerance, param_damping};
    std::vector<srrg2_posegraph_stats> st((size_t) std::max(param_max_iterations, 1));
    int n = (int) st.size();
    check(srrg2_posegraph_solve(_h, &p, st.data(), &n));
    st.resize((size_t) std::min<int>(n, (int) st.size()));
    return st;
  }
  std::vector<EstimateType> estimates() const {
    int v = 0, f = 0, e = 0;
    size(v, f, e);
    std::vector<float> raw((size_t) std::max(v, 1) * EstimateType::N);
    check(srrg2_posegraph_get_poses(_h, raw.data()));
    std::vector<EstimateType> out((size_t) v);
    for (int k = 0; k < v; ++k) std::memcpy(out[(size_t) k].data(), raw.data() + (size_t) k * EstimateType::N, sizeof(float) * EstimateType::N);
    return out;
  }

private:
  srrg2_posegraph_h _h = nullptr;
};
using PoseGraph2D = PoseGraph_<SRRG2_SE2_RIGHT>;
using PoseGraph3D = PoseGraph_<SRRG2_SE3_QUAT_RIGHT>;

// ---- MultiGraphSLAM_'s graph bookkeeping: makeNewMap (:52-90), loopValidate (:227-297), optimize (:300-317) ---------------
template <typename GraphType>
class GraphSLAMLifecycle {
public:
  static constexpr int DIM = GraphType::DIM;
  static constexpr int D   = GraphType::D;
  using EstimateType       = Isometry<DIM>;
  using LoopClosureType    = LoopClosure<DIM>;
  enum Verdict { Rejected = 0, Accepted = 1, Pending = 2 };  // FactorGraphClosureValidator::ClosureStatus (:262-277)
  // the closure validator is a parameter of the system (param_closure_validator); null = every closure accepted (:245-251)
  using Validator = std::function<std::vector<Verdict>(const std::vector<LoopClosureType>&)>;

  explicit GraphSLAMLifecycle(GraphType& graph) : _graph(graph) {
    for (int a = 0; a < D; ++a) _default_info[a * D + a] = 1.f;
  }
  // a variable with the current robot pose; an odometry factor from the previous local map with measurement
  // robot_in_local_map and information default_info * info_scale; the very first local map is Fixed (:86)
  int makeNewMap(const EstimateType& robot_in_world, const EstimateType& robot_in_local_map, float info_scale = 1.f) {
    const int previous = _current_local_map;
    const int vid      = _graph.addVariable(robot_in_world, previous < 0);
    if (previous >= 0) {
      float info[D * D];
      for (int k = 0; k < D * D; ++k) info[k] = _default_info[k] * info_scale;
      _graph.addFactor(previous, vid, robot_in_local_map, info, true);
    }
    _current_local_map = vid;
    return vid;
  }
  // closures enter the graph disabled (:238-241); rejected ones are removed (:279-281), accepted ones enabled (:283-286)
  std::vector<LoopClosureType> loopValidate(std::vector<LoopClosureType> detected, const Validator& validator = nullptr) {
    _num_valid_closures = 0;
    for (LoopClosureType& c : detected)
      c.factor_id = _graph.addFactor(c.source_graph_id, c.target_graph_id, c.measurement, c.information, false);
    std::vector<LoopClosureType> accepted;
    if (detected.empty()) return accepted;
    std::vector<Verdict> verdicts(detected.size(), Accepted);
    if (validator) verdicts = validator(detected);
    for (size_t k = 0; k < detected.size(); ++k) {
      if (verdicts[k] == Rejected) {
        _graph.removeFactor(detected[k].factor_id);
      } else if (verdicts[k] == Accepted) {
        _graph.setFactorEnabled(detected[k].factor_id, true);
        detected[k].enabled = true;
        ++_num_valid_closures;
        accepted.push_back(detected[k]);
      }
    }
    return accepted;
  }
  // nothing to do without a valid closure (:302-304); else bindFactors + global_solver->compute()
  std::vector<srrg2_posegraph_stats> optimize() {
    if (!_num_valid_closures) return {};
    return _graph.compute();
  }
  int currentLocalMap() const { return _current_local_map; }
  int numValidClosures() const { return _num_valid_closures; }

private:
  GraphType& _graph;
  float _default_info[D * D] = {};
  int _current_local_map     = -1;
  int _num_valid_closures    = 0;
};

}  // namespace srrg2_slam_amd
