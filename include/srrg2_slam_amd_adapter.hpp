// srrg2_slam_amd_adapter.hpp -- the reference-side adapter: classes with the reference's
// registry names and PARAMs that forward to the C ABI of srrg2_slam_amd.h, so that BOSS configurations written for
// srrg2_laser_slam_2d / srrg2_proslam deserialise into them unchanged.
//
// This header needs the reference's own dependencies -- srrg2_core (srrg_config, srrg_property, srrg_pcl,
// srrg_geometry) and srrg2_solver (IterationStats, VariableSE{2,3}...AD) -- which are NOT part of this repository and
// not present in the build image.  It is therefore guarded: without <srrg_config/configurable.h> on the include path it
// compiles to nothing (tests/test_abi_exports.py compiles it that way); inside a catkin workspace that has srrg2_core it
// is a drop-in next to S/registration/aligners/multi_aligner.h.  Nothing in this repository's tests depends on it; the
// same forwarding logic, on types this repository can compile, is include/srrg2_slam_amd.hpp (tested on the GPU).
//
// What is mirrored, PARAM by PARAM:
//   MultiAlignerAMD_<Variable>             MultiAlignerBase_<Variable>             multi_aligner.h:19-150 (PARAMs :34-57)
//     + AlignerBase PARAMs max_iterations, termination_criteria                  aligner.h:30-35
//   AlignerSliceProcessorAMD_<Variable>    AlignerSliceProcessor_<Factor,Fixed,Moving>  aligner_slice_processor.h:56-66,133-150
//     + base PARAMs robustifier, fixed_slice_name, moving_slice_name, frame_id, base_frame_id   aligner_slice_processor_base.h:34-53
//   AlignerTerminationCriteriaStandard_    kept as is (its PARAMs travel through srrg2_termination_params) aligner_termination_criteria.h:40-56
//   registration                           BOSS_REGISTER_CLASS names of instances.cpp:21-23,28-84 / instances.h:36-38,77
#pragma once

#if defined(__has_include)
#if __has_include(<srrg_config/configurable.h>) && __has_include(<srrg_solver/solver_core/iteration_stats.h>)
#define SRRG2_SLAM_AMD_HAVE_SRRG2_CORE 1
#endif
#endif

#ifdef SRRG2_SLAM_AMD_HAVE_SRRG2_CORE

#include <srrg_config/configurable.h>
#include <srrg_config/property_configurable.h>
#include <srrg_config/property_configurable_vector.h>
#include <srrg_data_structures/correspondence.h>
#include <srrg_data_structures/platform.h>
#include <srrg_pcl/point_types.h>
#include <srrg_property/property_container.h>
#include <srrg_solver/solver_core/iteration_stats.h>
#include <srrg_solver/solver_core/robustifier.h>
#include <srrg_solver/variables_and_factors/types_2d/variable_se2_ad.h>
#include <srrg_solver/variables_and_factors/types_3d/variable_se3_ad.h>

#include <srrg2_slam_interfaces/registration/aligners/aligner.h>
#include <srrg2_slam_interfaces/registration/aligners/aligner_termination_criteria.h>

#include <stdexcept>
#include <type_traits>
#include <vector>

#include "srrg2_slam_amd.h"

namespace srrg2_slam_interfaces {
  using namespace srrg2_core;
  using namespace srrg2_solver;

  namespace amd_detail {
    inline void check(int rc) {
      if (rc) throw std::runtime_error(srrg2_amd_last_error());  // misuse stays an exception (multi_aligner_impl.cpp:30,40,49)
    }
    template <typename VariableType_>
    constexpr int variableKind() {
      return std::is_base_of<VariableSE2Right, VariableType_>::value
               ? SRRG2_SE2_RIGHT
               : (std::is_base_of<VariableSE3QuaternionRight, VariableType_>::value ? SRRG2_SE3_QUAT_RIGHT : SRRG2_SE3_EULER_RIGHT);
    }
    template <typename IsometryType_>
    inline void toRowMajor(const IsometryType_& T, float* out) {  // 3x4 (SE3) / 3x3 homogeneous (SE2), row-major
      constexpr int Dim = IsometryType_::Dim;
      if (Dim == 3) {
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 4; ++c) out[r * 4 + c] = T.matrix()(r, c);
      } else {
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) out[r * 3 + c] = T.matrix()(r, c);
      }
    }
    template <typename IsometryType_>
    inline void fromRowMajor(const float* in, IsometryType_& T) {
      constexpr int Dim = IsometryType_::Dim;
      T.setIdentity();
      for (int r = 0; r < Dim; ++r)
        for (int c = 0; c <= Dim; ++c) T.matrix()(r, c) = in[r * (Dim + 1) + c];
    }
  }  // namespace amd_detail

  // One cue slice: the PARAMs of AlignerSliceProcessor_ and of its base, plus what the C ABI needs to know about the
  // factor and the finder that the reference expresses through template arguments and sub-configurables.
  template <typename VariableType_>
  class AlignerSliceProcessorAMD_ : public Configurable {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using VariableType = VariableType_;
    using EstimateType = typename VariableType::EstimateType;
    // aligner_slice_processor_base.h:34-53
    PARAM(PropertyConfigurable_<RobustifierBase>, robustifier, "robustifier used on this slice", 0, 0);
    PARAM(PropertyString, fixed_slice_name, "name of the slice in the fixed scene", "", 0);
    PARAM(PropertyString, moving_slice_name, "name of the slice in the moving scene", "", 0);
    PARAM(PropertyString, frame_id, "name of the sensor's frame in the tf tree", "", nullptr);
    PARAM(PropertyString, base_frame_id, "name of the base frame in the tf tree", "", nullptr);
    // aligner_slice_processor.h:56-66 (the finder is a sub-configurable there; its PARAMs are flattened here)
    PARAM(PropertyInt, min_num_correspondences, "minimum number of correspondences in this slice", 0, 0);
    PARAM(PropertyInt, slice_kind, "srrg2_slice_kind: 0 point-to-point, 1 point-to-plane, 2 reprojection", SRRG2_SLICE_P2P, 0);
    PARAM(PropertyInt, finder_kind, "srrg2_finder_kind: 1 gated nearest neighbour, 2 projective", SRRG2_FINDER_NN_GATED, 0);
    PARAM(PropertyFloat, finder_max_distance, "finder gate [m]", 1.0f, 0);
    PARAM(PropertyFloat, finder_normal_cos, "accept only if n_f . (R n_m) > this; <= -1 disables", -2.0f, 0);

    // setSensorInRobot (aligner_slice_processor.h:149); looked up on every setMovingInFixed (_impl.cpp:20-36)
    void setSensorInRobot(const EstimateType& sensor_in_robot_) { _sensor_in_robot = sensor_in_robot_; }
    const EstimateType& sensorInRobot() const { return _sensor_in_robot; }
    void setPlatform(PlatformPtr platform_) { _platform = platform_; }
    // the per-compute lookup of aligner_slice_processor_impl.cpp:24-33
    void updateSensorInRobot() {
      if (!_platform || param_frame_id.value().empty() || param_base_frame_id.value().empty()) return;
      EstimateType T;
      if (_platform->getTransform(T, param_frame_id.value(), param_base_frame_id.value())) _sensor_in_robot = T;
    }
    srrg2_slice_config config() const {
      srrg2_slice_config c;
      srrg2_slice_default_config(&c, amd_detail::variableKind<VariableType>());
      c.kind                    = param_slice_kind.value();
      c.finder                  = param_finder_kind.value();
      c.finder_max_distance     = param_finder_max_distance.value();
      c.finder_normal_cos       = param_finder_normal_cos.value();
      c.min_num_correspondences = param_min_num_correspondences.value();
      amd_detail::toRowMajor(_sensor_in_robot, c.sensor_in_robot);
      if (auto rob = param_robustifier.value()) {  // RobustifierCauchy / Saturated / Clamp -> kind + chi_threshold
        c.robustifier_chi_threshold = rob->param_chi_threshold.value();
        c.robustifier               = dynamic_cast<RobustifierCauchy*>(rob.get())      ? SRRG2_ROBUST_CAUCHY
                                      : dynamic_cast<RobustifierSaturated*>(rob.get()) ? SRRG2_ROBUST_SATURATED
                                      : dynamic_cast<RobustifierClamp*>(rob.get())     ? SRRG2_ROBUST_CLAMP
                                                                                       : SRRG2_ROBUST_NONE;
      }
      return c;
    }

  protected:
    EstimateType _sensor_in_robot = EstimateType::Identity();
    PlatformPtr _platform         = nullptr;
  };

  template <typename VariableType_>
  class MultiAlignerAMD_
    : public Aligner_<typename VariableType_::EstimateType, PropertyContainerBase, PropertyContainerBase> {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using VariableType = VariableType_;
    using EstimateType = typename VariableType::EstimateType;
    using BaseType     = Aligner_<EstimateType, PropertyContainerBase, PropertyContainerBase>;
    using SliceType    = AlignerSliceProcessorAMD_<VariableType>;
    static constexpr int Dim = EstimateType::Dim;
    using PointCloudType =
      typename std::conditional<Dim == 2, PointNormal2fVectorCloud, PointNormal3fVectorCloud>::type;

    // multi_aligner.h:34-57 (the solver PARAM has no counterpart: the one-variable Gauss-Newton step runs on the device)
    PARAM_VECTOR(PropertyConfigurableVector_<SliceType>, slice_processors, "slices", &(this->_slices_changed_flag));
    PARAM(PropertyInt, min_num_inliers, "minimum number ofinliers", 10, nullptr);
    PARAM(PropertyBool,
          enable_inlier_only_runs,
          "toggles additional inlier only runs if sufficient inliers are available",
          false,
          nullptr);
    PARAM(PropertyBool,
          keep_only_inlier_correspondences,
          "toggles removal of correspondences which factors are not inliers in the last iteration",
          false,
          nullptr);
    PARAM(PropertyInt, device, "HIP device ordinal", 0, nullptr);

    MultiAlignerAMD_() = default;
    virtual ~MultiAlignerAMD_() {
      if (_h) srrg2_aligner_destroy(_h);
    }

    // multi_aligner_impl.cpp:8-24: every slice finds its cloud by name in the scene and binds it
    void setFixed(PropertyContainerBase* fixed_) override {
      BaseType::setFixed(fixed_);
      bindSlices();
      for (size_t i = 0; i < param_slice_processors.size(); ++i)
        bindCloud(fixed_, param_slice_processors.value(i)->param_fixed_slice_name.value(), (int) i, true);
    }
    void setMoving(PropertyContainerBase* moving_) override {
      BaseType::setMoving(moving_);
      bindSlices();
      for (size_t i = 0; i < param_slice_processors.size(); ++i)
        bindCloud(moving_, param_slice_processors.value(i)->param_moving_slice_name.value(), (int) i, false);
    }
    // multi_aligner_impl.cpp:27-44
    void setMovingInFixed(const EstimateType& moving_in_fixed_) override {
      bindSlices();
      float T[12];
      amd_detail::toRowMajor(moving_in_fixed_, T);
      amd_detail::check(srrg2_aligner_set_moving_in_fixed(_h, T));
      this->_moving_in_fixed = moving_in_fixed_;
    }
    // multi_aligner_impl.cpp:47-95: blocking; status, estimate and iteration statistics are host visible on return
    void compute() override {
      bindSlices();
      for (size_t i = 0; i < param_slice_processors.size(); ++i) {  // the TF lookup of aligner_slice_processor_impl.cpp:24-33
        float S[12];
        param_slice_processors.value(i)->updateSensorInRobot();
        amd_detail::toRowMajor(param_slice_processors.value(i)->sensorInRobot(), S);
        amd_detail::check(srrg2_aligner_set_sensor_in_robot(_h, (int) i, S));
      }
      srrg2_aligner_params p{this->param_max_iterations.value(), param_min_num_inliers.value(),
                             param_enable_inlier_only_runs.value() ? 1 : 0,
                             param_keep_only_inlier_correspondences.value() ? 1 : 0};
      amd_detail::check(srrg2_aligner_set_params(_h, &p));
      if (auto tc = std::dynamic_pointer_cast<AlignerTerminationCriteriaStandard_<BaseType>>(this->param_termination_criteria.value())) {
        srrg2_termination_params t{tc->param_window_size.value(), tc->param_num_correspondences_range.value(),
                                   tc->param_num_inliers_range.value(), tc->param_num_outliers_range.value(),
                                   tc->param_chi_epsilon.value()};
        amd_detail::check(srrg2_aligner_set_termination(_h, &t));
      } else {
        amd_detail::check(srrg2_aligner_set_termination(_h, nullptr));
      }
      int st = AlignerBase::Fail;
      amd_detail::check(srrg2_aligner_compute(_h, &st));
      this->_status = static_cast<AlignerBase::Status>(st);  // identical values, aligner.h:23-28
      float T[12];
      amd_detail::check(srrg2_aligner_get_moving_in_fixed(_h, T));
      amd_detail::fromRowMajor(T, this->_moving_in_fixed);
      int n = 0;
      amd_detail::check(srrg2_aligner_get_iteration_stats(_h, nullptr, &n));
      std::vector<srrg2_iteration_stats> s((size_t) n);
      if (n) amd_detail::check(srrg2_aligner_get_iteration_stats(_h, s.data(), &n));
      this->_iteration_stats.clear();
      for (const srrg2_iteration_stats& e : s) {  // the fields the reference reads (multi_aligner_impl.cpp:81-82)
        IterationStats is;
        is.iteration    = e.iteration;
        is.num_inliers  = e.num_inliers;
        is.num_outliers = e.num_outliers;
        is.chi_inliers  = e.chi_inliers;
        is.chi_outliers = e.chi_outliers;
        this->_iteration_stats.push_back(is);
      }
    }
    // multi_aligner_impl.cpp:275-285
    int numCorrespondences() const override {
      int n = 0;
      amd_detail::check(srrg2_aligner_num_correspondences(_h, &n));
      return n;
    }
    // aligner_slice_processor_impl.cpp:51-74: the correspondences go back into the MOVING scene as the property
    // "<fixed_slice_name>_correspondences" (suffix: correspondence_finder.h:25)
    void storeCorrespondences() override {
      if (!this->_moving) return;
      for (size_t i = 0; i < param_slice_processors.size(); ++i) {
        int n = 0;
        amd_detail::check(srrg2_aligner_get_correspondences(_h, (int) i, nullptr, &n));
        std::vector<srrg2_correspondence> buf((size_t) n);
        if (n) amd_detail::check(srrg2_aligner_get_correspondences(_h, (int) i, buf.data(), &n));
        const std::string name = param_slice_processors.value(i)->param_fixed_slice_name.value() + "_correspondences";
        auto* prop = this->_moving->template property<Property_<CorrespondenceVector>>(name);
        if (!prop) {
          prop = new Property_<CorrespondenceVector>(name, "", this->_moving, CorrespondenceVector(), nullptr);
        }
        CorrespondenceVector& out = prop->value();
        out.clear();
        out.reserve(buf.size());
        for (const srrg2_correspondence& c : buf) out.emplace_back(Correspondence(c.fixed_idx, c.moving_idx, c.response));
      }
    }

  protected:
    // (re)creates the device-side aligner when the slice list changed (_slices_changed_flag, multi_aligner.h:37)
    void bindSlices() {
      if (_h && !this->_slices_changed_flag) return;
      if (_h) srrg2_aligner_destroy(_h);
      _h = nullptr;
      amd_detail::check(srrg2_aligner_create(amd_detail::variableKind<VariableType>(), param_device.value(), &_h));
      for (size_t i = 0; i < param_slice_processors.size(); ++i) {
        srrg2_slice_config c = param_slice_processors.value(i)->config();
        int idx              = -1;
        amd_detail::check(srrg2_aligner_add_slice(_h, &c, &idx));
      }
      this->_slices_changed_flag = false;
    }
    // the three storage forms of aligner_slice_processor_base_impl.cpp:27-50
    void bindCloud(PropertyContainerBase* scene, const std::string& name, int slice, bool fixed) {
      if (!scene) throw std::runtime_error("MultiAlignerAMD_::bindCloud|no scene");
      const PointCloudType* cloud = nullptr;
      if (auto* p = scene->template property<Property_<PointCloudType*>>(name)) cloud = p->value();
      else if (auto* v = scene->template property<Property_<PointCloudType>>(name)) cloud = &v->value();
      else if (auto* s = scene->template property<Property_<std::shared_ptr<PointCloudType>>>(name)) cloud = s->value().get();
      if (!cloud) throw std::runtime_error("MultiAlignerAMD_::bindCloud|slice [" + name + "] not found in the scene");
      using PointType = typename PointCloudType::value_type;
      const float* coords  = cloud->empty() ? nullptr : cloud->front().coordinates().data();
      const float* normals = cloud->empty() ? nullptr : cloud->front().normal().data();
      amd_detail::check((fixed ? srrg2_aligner_set_fixed : srrg2_aligner_set_moving)(
        _h, slice, coords, (int) sizeof(PointType), normals, (int) sizeof(PointType), (int) cloud->size(), SRRG2_MEM_HOST));
    }
    srrg2_aligner_h _h = nullptr;
    bool _slices_changed_flag = true;
  };

  // the registry names of the reference (multi_aligner.h:152-159, instances.cpp:21-23): a configuration that names
  // "MultiAligner2D" loads this class when the adapter library is linked instead of the reference's aligner
  using MultiAligner2D   = MultiAlignerAMD_<VariableSE2RightAD>;
  using MultiAligner3D   = MultiAlignerAMD_<VariableSE3EulerRightAD>;
  using MultiAligner3DQR = MultiAlignerAMD_<VariableSE3QuaternionRightAD>;
  using AlignerSliceProcessorAMD2D   = AlignerSliceProcessorAMD_<VariableSE2RightAD>;
  using AlignerSliceProcessorAMD3D   = AlignerSliceProcessorAMD_<VariableSE3EulerRightAD>;
  using AlignerSliceProcessorAMD3DQR = AlignerSliceProcessorAMD_<VariableSE3QuaternionRightAD>;

  // call from the adapter library's registerTypes() (the counterpart of srrg2_slam_interfaces_registerTypes,
  // instances.cpp:28-84; BOSS_REGISTER_CLASS_LINKER_FRIENDLY as defined at instances.cpp:21-23)
#define SRRG2_SLAM_AMD_REGISTER_TYPES()                                    \
  do {                                                                     \
    BOSS_REGISTER_CLASS(MultiAligner2D);                                   \
    BOSS_REGISTER_CLASS(MultiAligner3D);                                   \
    BOSS_REGISTER_CLASS(MultiAligner3DQR);                                 \
    BOSS_REGISTER_CLASS(AlignerSliceProcessorAMD2D);                       \
    BOSS_REGISTER_CLASS(AlignerSliceProcessorAMD3D);                       \
    BOSS_REGISTER_CLASS(AlignerSliceProcessorAMD3DQR);                     \
  } while (0)

}  // namespace srrg2_slam_interfaces

#endif  // SRRG2_SLAM_AMD_HAVE_SRRG2_CORE
