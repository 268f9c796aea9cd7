// srrg2_slam_amd.hpp -- header-only C++17 mirror of the reference's aligner interface over the C ABI.
//
// The reference is C++ (S/registration/aligners/*.h); this header gives its callers the same vocabulary --
// class and method names, argument meaning, exceptions for misuse, AlignerBase::Status for outcomes -- without
// srrg2_core / Eigen, so it compiles in this repository.  INTEGRATION.md shows how the same calls sit inside real
// srrg2_core types.  Everything forwards to include/srrg2_slam_amd.h; no arithmetic of the hot path lives here.
//
//   MultiAligner                    MultiAlignerBase_<Variable>            multi_aligner.h:19-150, multi_aligner_impl.cpp
//   AlignerTerminationCriteria      AlignerTerminationCriteriaStandard_    aligner_termination_criteria.h:33-71
//   MotionModelConstantVelocity     MotionModelConstantVelocity<Estimate>  motion_models/motion_model_constant_velocity.hpp
//   AlignerSliceMotionModel         AlignerSliceMotionModel_               aligner_slice_motion_model.hpp:13-92
//   AlignerSliceOdomPrior           AlignerSliceOdom{2,3}DPrior            aligner_slice_odometry_prior.{h,cpp}
//   Scene / SceneClipperBall / MergerCorrespondenceHomo                    mapping/scene_clipper.h, merger_correspondence_homo.h
//   (loop-closure drivers and the pose graph: srrg2_slam_amd_loop_closure.hpp)
#pragma once
#include <array>
#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "srrg2_slam_amd.h"

namespace srrg2_slam_amd {

// ---- transforms: row-major 3x4 (SE3) / 3x3 (SE2) float, like Isometry3f / Isometry2f ---------------------------
template <int DIM>
struct Isometry {
  static constexpr int N = DIM == 2 ? 9 : 12;
  std::array<float, N> m{};
  static Isometry Identity() {
    Isometry T;
    if (DIM == 2) {
      T.m = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    } else {
      const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      std::memcpy(T.m.data(), I, sizeof(I));
    }
    return T;
  }
  const float* data() const { return m.data(); }
  float* data() { return m.data(); }
  Isometry inverse() const {
    Isometry R;
    if (DIM == 2) {
      R.m = {m[0], m[3], -(m[0] * m[2] + m[3] * m[5]), m[1], m[4], -(m[1] * m[2] + m[4] * m[5]), 0, 0, 1};
    } else {
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R.m[i * 4 + j] = m[j * 4 + i];
        R.m[i * 4 + 3] = -(m[0 * 4 + i] * m[3] + m[1 * 4 + i] * m[7] + m[2 * 4 + i] * m[11]);
      }
    }
    return R;
  }
  Isometry operator*(const Isometry& B) const {
    Isometry C;
    if (DIM == 2) {
      for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 2; ++j) C.m[i * 3 + j] = m[i * 3] * B.m[j] + m[i * 3 + 1] * B.m[3 + j];
        C.m[i * 3 + 2] = m[i * 3] * B.m[2] + m[i * 3 + 1] * B.m[5] + m[i * 3 + 2];
      }
      C.m[6] = 0; C.m[7] = 0; C.m[8] = 1;
    } else {
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
          C.m[i * 4 + j] = m[i * 4] * B.m[j] + m[i * 4 + 1] * B.m[4 + j] + m[i * 4 + 2] * B.m[8 + j];
        C.m[i * 4 + 3] = m[i * 4] * B.m[3] + m[i * 4 + 1] * B.m[7] + m[i * 4 + 2] * B.m[11] + m[i * 4 + 3];
      }
    }
    return C;
  }
};
using Isometry2f = Isometry<2>;
using Isometry3f = Isometry<3>;

struct AlignerBase {
  enum Status { Success = 0, NotEnoughCorrespondences = 1, NotEnoughInliers = 2, Fail = 3 };  // aligner.h:23-28
};

inline void check(int rc) {
  if (rc != 0) throw std::runtime_error(srrg2_amd_last_error());  // the reference throws std::runtime_error too
}

using Correspondence       = srrg2_correspondence;
using CorrespondenceVector = std::vector<Correspondence>;
using IterationStats       = srrg2_iteration_stats;
using IterationStatsVector = std::vector<IterationStats>;

// ---- MultiAlignerBase_<Variable> -------------------------------------------------------------------------------
template <int VARIABLE_KIND>
class MultiAligner_ : public AlignerBase {
public:
  static constexpr int Dim = VARIABLE_KIND == SRRG2_SE2_RIGHT ? 2 : 3;
  using EstimateType       = Isometry<Dim>;

  // PARAMs (aligner.h:30; multi_aligner.h:45-57)
  int param_max_iterations                    = 10;
  int param_min_num_inliers                   = 10;
  bool param_enable_inlier_only_runs          = false;
  bool param_keep_only_inlier_correspondences = false;

  explicit MultiAligner_(int device = 0) { check(srrg2_aligner_create(VARIABLE_KIND, device, &_h)); }
  ~MultiAligner_() { srrg2_aligner_destroy(_h); }
  MultiAligner_(const MultiAligner_&)            = delete;
  MultiAligner_& operator=(const MultiAligner_&) = delete;

  // param_slice_processors.pushBack(slice)
  int addSlice(const srrg2_slice_config& c) {
    int idx = -1;
    check(srrg2_aligner_add_slice(_h, &c, &idx));
    return idx;
  }
  static srrg2_slice_config defaultSliceConfig() {
    srrg2_slice_config c;
    srrg2_slice_default_config(&c, VARIABLE_KIND);
    return c;
  }
  // param_termination_criteria.setValue(...) / nullptr
  void setTerminationCriteria(const srrg2_termination_params* p) { check(srrg2_aligner_set_termination(_h, p)); }

  // slice->setFixed / setMoving of the named cloud (multi_aligner_impl.cpp:8-24): raw strided arrays
  void setFixed(int slice, const float* coords, int stride_bytes, const float* normals, int normal_stride_bytes, int n,
                int mem = SRRG2_MEM_HOST) {
    check(srrg2_aligner_set_fixed(_h, slice, coords, stride_bytes, normals, normal_stride_bytes, n, mem));
  }
  void setMoving(int slice, const float* coords, int stride_bytes, const float* normals, int normal_stride_bytes, int n,
                 int mem = SRRG2_MEM_HOST) {
    check(srrg2_aligner_set_moving(_h, slice, coords, stride_bytes, normals, normal_stride_bytes, n, mem));
  }
  // factor->setCorrespondences(corrs) of a SRRG2_FINDER_CORRESPONDENCES slice: locked during compute()
  // (MultiLoopDetectorHBST_::_computeAlignments, multi_loop_detector_hbst_impl.cpp:330,343)
  void setCorrespondences(int slice, const std::vector<srrg2_correspondence>& c) {
    check(srrg2_aligner_set_correspondences(_h, slice, c.data(), (int) c.size()));
  }
  // slice->setSensorInRobot(); the reference re-reads the platform TF on every setMovingInFixed
  // (aligner_slice_processor_impl.cpp:20-36): call it whenever the sensor pose changed
  void setSensorInRobot(int slice, const EstimateType& T) { check(srrg2_aligner_set_sensor_in_robot(_h, slice, T.data())); }
  void setPriorMeasurement(int slice, const EstimateType& Z) { check(srrg2_aligner_set_prior_measurement(_h, slice, Z.data())); }
  // ONE alignment over the ranks of a communicator, sharded by moving points (srrg2_aligner_set_point_shard):
  // `reduce` adds / maximises the device buffer in place over all ranks on the given stream (e.g. ncclAllReduce);
  // total_moving_points counts the moving points of all ranks.  reduce == nullptr switches the mode off.
  void setPointShard(srrg2_reduce_fn reduce, void* user, int64_t total_moving_points) {
    check(srrg2_aligner_set_point_shard(_h, reduce, user, total_moving_points));
  }

  void setMovingInFixed(const EstimateType& X) { check(srrg2_aligner_set_moving_in_fixed(_h, X.data())); }
  const EstimateType& movingInFixed() const {
    check(srrg2_aligner_get_moving_in_fixed(_h, _X.data()));
    return _X;
  }

  void compute() {  // multi_aligner_impl.cpp:47-95, blocking
    srrg2_aligner_params p{param_max_iterations, param_min_num_inliers, param_enable_inlier_only_runs ? 1 : 0,
                           param_keep_only_inlier_correspondences ? 1 : 0};
    check(srrg2_aligner_set_params(_h, &p));
    int st = Fail;
    check(srrg2_aligner_compute(_h, &st));
    _status = static_cast<Status>(st);
    int n   = 0;
    check(srrg2_aligner_get_iteration_stats(_h, nullptr, &n));
    _iteration_stats.resize((size_t) n);
    if (n) check(srrg2_aligner_get_iteration_stats(_h, _iteration_stats.data(), &n));
  }
  // K independent alignments against the fixed scene already set: the loop body of
  // MultiLoopDetectorBruteForce_::compute (multi_loop_detector_brute_force_impl.cpp:64-91) for all candidates at once.
  // clouds: K pointers to packed Dim-float records with their sizes; normals may be empty (none) or K pointers.
  std::vector<srrg2_batch_result> computeBatch(const std::vector<const float*>& clouds, const std::vector<int>& sizes,
                                               const std::vector<const float*>& normals,
                                               const std::vector<EstimateType>& guesses) {
    const int K = (int) clouds.size();
    if ((int) sizes.size() != K || (int) guesses.size() != K || (!normals.empty() && (int) normals.size() != K))
      throw std::runtime_error("MultiAligner_::computeBatch|inconsistent argument sizes");
    std::vector<int32_t> offsets((size_t) K + 1, 0);
    for (int k = 0; k < K; ++k) offsets[(size_t) k + 1] = offsets[(size_t) k] + sizes[(size_t) k];
    std::vector<float> coords((size_t) offsets[(size_t) K] * Dim), nrm;
    if (!normals.empty()) nrm.resize(coords.size());
    std::vector<float> g((size_t) K * EstimateType::N);
    for (int k = 0; k < K; ++k) {
      std::memcpy(coords.data() + (size_t) offsets[(size_t) k] * Dim, clouds[(size_t) k], sizeof(float) * (size_t) sizes[(size_t) k] * Dim);
      if (!normals.empty())
        std::memcpy(nrm.data() + (size_t) offsets[(size_t) k] * Dim, normals[(size_t) k], sizeof(float) * (size_t) sizes[(size_t) k] * Dim);
      std::memcpy(g.data() + (size_t) k * EstimateType::N, guesses[(size_t) k].data(), sizeof(float) * EstimateType::N);
    }
    srrg2_aligner_params p{param_max_iterations, param_min_num_inliers, param_enable_inlier_only_runs ? 1 : 0,
                           param_keep_only_inlier_correspondences ? 1 : 0};
    check(srrg2_aligner_set_params(_h, &p));
    std::vector<srrg2_batch_result> results((size_t) K);
    if (K)
      check(srrg2_aligner_compute_batch(_h, K, coords.data(), Dim * 4, normals.empty() ? nullptr : nrm.data(), Dim * 4,
                                        offsets.data(), SRRG2_MEM_HOST, g.data(), results.data()));
    return results;
  }
  // slice `slice` reads the clouds of slice `source` (two slices with the same fixed_slice_name / moving_slice_name bind to
  // the same clouds of the scene: aligner_slice_processor_base_impl.cpp:27-50); -1: clouds of its own again
  void shareClouds(int slice, int source) { check(srrg2_aligner_share_clouds(_h, slice, source)); }
  Status status() const { return _status; }
  const IterationStatsVector& iterationStats() const { return _iteration_stats; }
  // H = sum w J^T J of the last Gauss-Newton iteration of the last compute() (the solver's system after
  // multi_aligner_impl.cpp:112-116), D x D row-major, D = 3 (SE2) or 6 (SE3)
  std::vector<float> information() {
    std::vector<float> H(36, 0.f);
    check(srrg2_aligner_get_information(_h, H.data()));
    return H;
  }
  int numCorrespondences() {
    int n = 0;
    check(srrg2_aligner_num_correspondences(_h, &n));
    return n;
  }
  // SRRG2_PATH_* bits of the launch path the last compute() took (strategy only: results never depend on it)
  int lastComputePath() {
    int32_t f = 0;
    check(srrg2_aligner_last_compute_path(_h, &f));
    return (int) f;
  }
  CorrespondenceVector correspondences(int slice) {
    int n = 0;
    check(srrg2_aligner_get_correspondences(_h, slice, nullptr, &n));
    CorrespondenceVector v((size_t) n);
    if (n) check(srrg2_aligner_get_correspondences(_h, slice, v.data(), &n));
    return v;
  }
  srrg2_aligner_h handle() { return _h; }

private:
  srrg2_aligner_h _h = nullptr;
  Status _status     = Fail;  // aligner.h:56
  mutable EstimateType _X = EstimateType::Identity();
  IterationStatsVector _iteration_stats;
};
using MultiAligner2D   = MultiAligner_<SRRG2_SE2_RIGHT>;        // multi_aligner.h:152-158
using MultiAligner3D   = MultiAligner_<SRRG2_SE3_EULER_RIGHT>;
using MultiAligner3DQR = MultiAligner_<SRRG2_SE3_QUAT_RIGHT>;

// ---- MotionModelConstantVelocity (motion_model_constant_velocity.hpp:17-46) -----------------------------------------
template <int DIM>
class MotionModelConstantVelocity {
public:
  using EstimateType = Isometry<DIM>;
  const EstimateType& estimate() const { return _motion; }
  void compute() { _motion = _robot_in_local_map_previous.inverse() * _robot_in_local_map; }
  void setRobotInLocalMap(const EstimateType& T) {
    _robot_in_local_map_previous = _robot_in_local_map;
    _robot_in_local_map          = T;
  }
  void shiftTrackerEstimate(const EstimateType& estimate) {
    _robot_in_local_map_previous = estimate * (_robot_in_local_map_previous.inverse() * _robot_in_local_map);
    _robot_in_local_map          = estimate;
  }
  void setRobotInLocalMapPrevious(const EstimateType& T) { _robot_in_local_map_previous = T; }
  void clear() {
    _motion = _robot_in_local_map = _robot_in_local_map_previous = EstimateType::Identity();
  }

private:
  EstimateType _motion                      = EstimateType::Identity();
  EstimateType _robot_in_local_map          = EstimateType::Identity();
  EstimateType _robot_in_local_map_previous = EstimateType::Identity();
};
using MotionModelConstantVelocity2D = MotionModelConstantVelocity<2>;
using MotionModelConstantVelocity3D = MotionModelConstantVelocity<3>;

// ---- AlignerSliceMotionModel_ (aligner_slice_motion_model.hpp:44-79) -----------------------------------------------------
template <typename AlignerType>
class AlignerSliceMotionModel {
public:
  using EstimateType = typename AlignerType::EstimateType;
  using PoseBuffer   = std::deque<EstimateType>;  // StdDequeEigenIsometry3f
  MotionModelConstantVelocity<AlignerType::Dim>* param_motion_model = nullptr;

  explicit AlignerSliceMotionModel(AlignerType& aligner) : _aligner(aligner) {
    srrg2_slice_config c       = AlignerType::defaultSliceConfig();
    c.kind                     = SRRG2_SLICE_PRIOR;
    c.finder                   = SRRG2_FINDER_NONE;
    c.prior_sets_initial_guess = 1;  // init() calls aligner->setMovingInFixed(_motion_inverse), :69-70
    _slice                     = aligner.addSlice(c);
  }
  void setFixed(const PoseBuffer* fixed_slice) { _fixed_slice = fixed_slice; }
  void init() {
    if (!param_motion_model) throw std::runtime_error("AlignerSliceMotionModel_::init|ERROR: no motion model is set");
    if (!_fixed_slice) throw std::runtime_error("AlignerSliceMotionModel_::init|ERROR: no fixed pose set");
    if (!_fixed_slice->empty()) {
      param_motion_model->setRobotInLocalMap(_fixed_slice->back());
      if (_fixed_slice->size() > 1) param_motion_model->setRobotInLocalMapPrevious(*(_fixed_slice->end() - 2));
      param_motion_model->compute();
    }
    _motion_inverse = param_motion_model->estimate().inverse();
    _aligner.setPriorMeasurement(_slice, _motion_inverse);
  }

private:
  AlignerType& _aligner;
  int _slice                     = -1;
  const PoseBuffer* _fixed_slice = nullptr;
  EstimateType _motion_inverse   = EstimateType::Identity();
};

// ---- AlignerSliceOdom{2,3}DPrior (aligner_slice_odometry_prior.cpp:6-37) ------------------------------------------------
template <typename AlignerType>
class AlignerSliceOdomPrior {
public:
  using EstimateType = typename AlignerType::EstimateType;
  explicit AlignerSliceOdomPrior(AlignerType& aligner) : _aligner(aligner) {
    srrg2_slice_config c = AlignerType::defaultSliceConfig();
    c.kind               = SRRG2_SLICE_PRIOR;
    c.finder             = SRRG2_FINDER_NONE;
    _slice               = aligner.addSlice(c);
  }
  void setFixed(const EstimateType* T) { _fixed_slice = T; }
  void setMoving(const EstimateType* T) { _moving_slice = T; }
  void init() {
    if (!_fixed_slice) throw std::runtime_error("AlignerSliceProcessor_::factor| no fixed");
    if (!_moving_slice) throw std::runtime_error("AlignerSliceProcessor_::factor| no moving");
    EstimateType delta = EstimateType::Identity();
    if (_count > 1) delta = _fixed_slice->inverse() * (*_moving_slice);
    _aligner.setPriorMeasurement(_slice, delta);
    ++_count;
  }

private:
  AlignerType& _aligner;
  int _slice                        = -1;
  int _count                        = 0;
  const EstimateType* _fixed_slice  = nullptr;
  const EstimateType* _moving_slice = nullptr;
};

// ---- scene slices in device memory: SceneClipper_ / MergerCorrespondenceHomo_ ------
//   Scene                       a PointNormal{2,3}f cloud living in HBM (LocalMap scene slice / measurement)
//   SceneClipperBall            SceneClipper_<Estimate, Scene>              S/mapping/scene_clipper.h:17-122
//   MergerCorrespondenceHomo    MergerCorrespondenceHomo_<Estimate, Scene>  S/mapping/merger_correspondence_homo.h
template <int DIM>
class Scene {
public:
  explicit Scene(int device = 0) { check(srrg2_scene_create(DIM, device, &_h)); }
  ~Scene() { srrg2_scene_destroy(_h); }
  Scene(const Scene&)            = delete;
  Scene& operator=(const Scene&) = delete;
  void set(const float* coords, int stride_bytes, const float* normals, int normal_stride_bytes, int n,
           int mem = SRRG2_MEM_HOST) {
    check(srrg2_scene_set(_h, coords, stride_bytes, normals, normal_stride_bytes, n, mem));
  }
  int size() const {
    int n = 0;
    check(srrg2_scene_size(_h, &n));
    return n;
  }
  // packed DIM-float records
  void get(std::vector<float>& coords, std::vector<float>& normals) const {
    int n = size();
    coords.assign((size_t) n * DIM, 0.f);
    normals.assign((size_t) n * DIM, 0.f);
    check(srrg2_scene_get(_h, coords.data(), normals.data(), n, &n));
  }
  // device float4 arrays (stride 16 bytes), e.g. for MultiAligner_::setMoving(..., SRRG2_MEM_DEVICE)
  void deviceArrays(const float*& coords, const float*& normals, int& n) const {
    check(srrg2_scene_device_arrays(_h, &coords, &normals, &n));
  }
  srrg2_scene_h handle() const { return _h; }

private:
  srrg2_scene_h _h = nullptr;
};

template <int DIM>
class SceneClipperBall {
public:
  enum Status { Error = 0, Successful = 1, Ready = 2 };  // scene_clipper.h:24-28
  using EstimateType = Isometry<DIM>;
  using SceneType    = Scene<DIM>;
  float param_range  = 10.f;
  void setFullScene(SceneType* s) { _full = s; }
  void setClippedSceneInRobot(SceneType* s) { _clipped = s; }
  void setRobotInLocalMap(const EstimateType& T) { _robot_in_local_map = T; }
  void compute() {
    if (!_full || !_clipped) throw std::runtime_error("SceneClipperBall::compute|scene not set");
    int st = Error;
    check(srrg2_scene_clip_ball(_full->handle(), _robot_in_local_map.data(), param_range, _clipped->handle(), &st));
    _status = static_cast<Status>(st);
  }
  Status status() const { return _status; }
  std::vector<int> globalIndices() const {  // :98-101
    int n = 0;
    check(srrg2_scene_global_indices(_clipped->handle(), nullptr, &n));
    std::vector<int> v((size_t) n);
    if (n) check(srrg2_scene_global_indices(_clipped->handle(), v.data(), &n));
    return v;
  }

private:
  SceneType* _full    = nullptr;
  SceneType* _clipped = nullptr;
  EstimateType _robot_in_local_map = EstimateType::Identity();
  Status _status = Error;
};

template <int DIM>
class MergerCorrespondenceHomo {
public:
  enum Status { Error = 0, Initializing = 1, Success = 2 };  // merger.h:19-23
  using EstimateType = Isometry<DIM>;
  using SceneType    = Scene<DIM>;
  // PARAMs: merger_correspondence_homo.h:22-31, merger.h:126-131
  float param_maximum_response                  = 50.f;
  float param_maximum_distance_geometry_squared = 0.25f;
  unsigned param_target_number_of_merges        = 200;
  void setScene(SceneType* s) { _scene = s; }
  void setMeasurement(const SceneType* m) { _meas = m; }
  void setMeasurementInScene(const EstimateType& T) { _T = T; }
  void setCorrespondences(const CorrespondenceVector* c) { _corr = c; }  // nullptr: none set (impl.cpp:30)
  void compute() {
    ready();
    srrg2_merger_params p = params();
    check(srrg2_scene_merge(_scene->handle(), _meas->handle(), _T.data(), _corr ? _corr->data() : nullptr,
                            _corr ? (int) _corr->size() : -1, &p, &_last));
    _status = static_cast<Status>(_last.status);
  }
  // correspondences taken on the device from the aligner run that matched `clipped` (moving) against the measurement
  // (fixed): TrackerSliceProcessor_::merge(), tracker_slice_processor_impl.cpp:160-186
  template <typename AlignerType>
  void computeFromAligner(AlignerType& aligner, int slice, const SceneType& clipped) {
    ready();
    srrg2_merger_params p = params();
    check(srrg2_scene_merge_from_aligner(_scene->handle(), _meas->handle(), _T.data(), aligner.handle(), slice,
                                         clipped.handle(), &p, &_last));
    _status = static_cast<Status>(_last.status);
  }
  Status status() const { return _status; }
  const srrg2_merge_result& last() const { return _last; }

private:
  void ready() const {
    if (!_scene || !_meas) throw std::runtime_error("MergerCorrespondenceHomo::compute|scene or measurement not set");
  }
  srrg2_merger_params params() const {
    return srrg2_merger_params{param_maximum_response, param_maximum_distance_geometry_squared,
                               (int32_t) param_target_number_of_merges};
  }
  SceneType* _scene       = nullptr;
  const SceneType* _meas  = nullptr;
  const CorrespondenceVector* _corr = nullptr;
  EstimateType _T = EstimateType::Identity();
  srrg2_merge_result _last{};
  Status _status = Error;
};

}  // namespace srrg2_slam_amd
