// srrg2_slam_amd_adapter.hpp -- the reference-side adapter.
//
// WHAT IT IS.  A replacement of ONE class of the reference, under the reference's own name:
//     srrg2_slam_interfaces::MultiAlignerBase_<VariableType_>      (S/registration/aligners/multi_aligner.h:19-150)
// with the same base class (Aligner_<EstimateType, PropertyContainerBase, PropertyContainerBase>), the same PARAMs under
// the same names (slice_processors, solver, min_num_inliers, enable_inlier_only_runs, keep_only_inlier_correspondences,
// multi_aligner.h:34-57; max_iterations and termination_criteria come from AlignerBase, aligner.h:30-35) and the same
// type aliases MultiAligner2D / MultiAligner3D / MultiAligner3DQR (multi_aligner.h:152-159).  In a workspace it takes the
// place of multi_aligner.h + multi_aligner_impl.cpp and nothing else: S/instances.cpp:35-40 keeps registering
// "MultiAligner2D", the prior slices "AlignerSliceOdom2DPrior" / "AlignerSliceOdom3DPrior" / "AlignerSliceMotionModel2D" /
// "AlignerSliceMotionModel3D" and the termination criteria "AlignerTerminationCriteriaStandard2D" / "...3DQR" exactly as
// before -- they are the reference's OWN classes and stay untouched -- so a BOSS configuration written for the reference
// (or for srrg2_laser_slam_2d / srrg2_proslam on top of it) deserialises unchanged:
//   * `slice_processors` holds the reference's slice BASE type, AlignerSliceProcessorBase_<BaseVariableType>
//     (aligner_slice_processor_base.h:19): any registered slice -- prior slice, motion-model slice, a downstream cue
//     slice (a subclass of AlignerSliceProcessor_, aligner_slice_processor.h:26-66, with its `finder` sub-configurable,
//     `min_num_correspondences`, `robustifier`, `fixed_slice_name`, ...) -- goes into it;
//   * prior slices keep THEIR logic: this class is MultiAlignerBase_, which the slice classes befriend
//     (aligner_slice_processor_base.h:30-31, aligner_slice_processor_prior.h:31-32), so it calls slice->init(this) as the
//     reference does in _preCompute (multi_aligner_impl.cpp:131-141) -- the slice computes its measurement
//     (aligner_slice_odometry_prior.cpp:6-37: delta = fixed^-1 * moving from the second call on;
//     aligner_slice_motion_model.hpp:44-79: the motion model's inverse estimate) and overrides the initial guess through
//     aligner->setMovingInFixed() -- and then reads measurement and information matrix off the slice's own factor
//     (SE2PriorErrorFactor / SE3PriorErrorFactorAD) into a SRRG2_SLICE_PRIOR slice of the C ABI;
//   * cue slices are translated by a small registry (CueSliceRegistry below): the factor type says which rows the device
//     evaluates (srrg2_slice_kind), the slice's `finder` sub-configurable gives the gate.  The reference ships NO concrete
//     cue slice or finder (they live in srrg2_laser_slam_2d / srrg2_proslam, not a build dependency: SURVEY.md 8c), so the
//     translators for those are one registration line each in the downstream package's registerTypes(), next to its
//     BOSS_REGISTER_CLASS line; the PARAM names of the configuration files do not change;
//   * termination criteria: param_termination_criteria holds the reference's AlignerTerminationCriteriaStandard_
//     (aligner_termination_criteria.h:40-56); its five PARAMs travel through srrg2_termination_params, the windows run on
//     the device with the reference's quirks (aligner_termination_criteria_impl.cpp:46,53).
//
// WHAT IT NEEDS.  srrg2_core and srrg2_solver, which are NOT part of this repository and not present in the build image:
// the header is guarded and compiles to nothing without them (tests/test_abi_exports.py compiles it that way).  No
// stand-in headers are provided.  Nothing in this repository's tests depends on it; the same forwarding logic on types this
// repository can compile is include/srrg2_slam_amd.hpp (tested on the GPU).
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
#include <srrg_solver/solver_core/solver.h>
#include <srrg_solver/variables_and_factors/types_2d/se2_prior_error_factor.h>
#include <srrg_solver/variables_and_factors/types_2d/variable_se2_ad.h>
#include <srrg_solver/variables_and_factors/types_3d/se3_prior_error_factor_ad.h>
#include <srrg_solver/variables_and_factors/types_3d/variable_se3_ad.h>

// the reference's own headers that stay in place (this header replaces multi_aligner.h only)
#include <srrg2_slam_interfaces/registration/aligners/aligner.h>
#include <srrg2_slam_interfaces/registration/aligners/aligner_slice_processor.h>
#include <srrg2_slam_interfaces/registration/aligners/aligner_slice_processor_base.h>
#include <srrg2_slam_interfaces/registration/aligners/aligner_termination_criteria.h>

#include <cstring>
#include <functional>
#include <iostream>
#include <typeinfo>
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
    inline void robustifierOf(RobustifierBase* rob, srrg2_slice_config& c) {  // RobustifierCauchy / Saturated / Clamp -> kind + threshold
      c.robustifier = SRRG2_ROBUST_NONE;
      if (!rob) return;
      c.robustifier_chi_threshold = rob->param_chi_threshold.value();
      c.robustifier               = dynamic_cast<RobustifierCauchy*>(rob)      ? SRRG2_ROBUST_CAUCHY
                                    : dynamic_cast<RobustifierSaturated*>(rob) ? SRRG2_ROBUST_SATURATED
                                    : dynamic_cast<RobustifierClamp*>(rob)     ? SRRG2_ROBUST_CLAMP
                                                                               : -1;
      // (a robustifier the device does not evaluate must not silently become "none" -- the alignment would
      // run without the kernel the configuration asked for)
      if (c.robustifier < 0)
        throw std::runtime_error("MultiAlignerBase_|robustifier type [" + rob->className() +
                                 "] is not evaluated on the device (Cauchy, Saturated, Clamp or none)");
    }
    // PARAM `solver` (multi_aligner.h:39-43).  The registration solver of the reference runs ONE Gauss-Newton iteration per
    // ICP iteration on a one-variable graph (its constructor forces max_iterations = [1], multi_aligner.h:61-62); that step
    // is fixed on the device (dense L D L^T, no damping).  A configuration that swaps the solver for a subclass or asks for
    // another iteration count would behave differently there: refuse it instead of ignoring it.  Everything else the Solver
    // exposes (linear solver, termination criteria, robustifier policies: [EXT] srrg2_solver, not under the reference tree)
    // cannot change a single dense 3 x 3 / 6 x 6 solve and is not consulted -- said once on stderr when the object is not
    // the one the aligner constructed.
    inline void checkSolver(const std::shared_ptr<Solver>& solver, const Solver* constructed) {
      if (!solver) throw std::runtime_error("MultiAlignerBase_|no solver");  // multi_aligner_impl.cpp:29,39,48
      if (typeid(*solver) != typeid(Solver))
        throw std::runtime_error("MultiAlignerBase_|PARAM solver holds a [" + solver->className() +
                                 "]: the registration step runs on the device and cannot host a Solver subclass");
      const auto& its = solver->param_max_iterations.value();
      if (its.size() != 1 || its[0] != 1)
        throw std::runtime_error("MultiAlignerBase_|PARAM solver: max_iterations must stay [1] (multi_aligner.h:61-62): one "
                                 "Gauss-Newton step per ICP iteration is what the device runs");
      static bool told = false;
      if (solver.get() != constructed && !told) {
        told = true;
        std::cerr << "MultiAlignerBase_|PARAM solver was replaced by the configuration: only its class and max_iterations are "
                     "checked; its other PARAMs are not consulted (the Gauss-Newton step of the registration runs on the device)\n";
      }
    }
  }  // namespace amd_detail

  // ---- cue slices: what the device needs to know about a downstream slice type ------------------------------------------
  // A cue slice is an AlignerSliceProcessor_<FactorType, FixedCloud, MovingCloud> of a downstream package.  Its
  // translator fills, from the LIVE slice object (so every PARAM of the configuration is honoured):
  //   * the factor rows the device evaluates (srrg2_slice_kind) and the finder (srrg2_finder_kind + gate, normal gate,
  //     camera matrix for projective finders) into `config`;
  //   * the bound clouds as strided float arrays (coordinates / normals of the slice's _fixed_slice / _moving_slice).
  // registerCueSlice<SliceType>(slice_kind, finder_translator) covers the common case: clouds of points with
  // coordinates() and normal() (PointNormal2f / PointNormal3f ...), the finder read through `finder_translator`.
  struct CueBinding {
    srrg2_slice_config config;
    const float* fixed_coords = nullptr;   const float* fixed_normals = nullptr;   int fixed_stride = 0;   int fixed_size = 0;
    const float* moving_coords = nullptr;  const float* moving_normals = nullptr;  int moving_stride = 0;  int moving_size = 0;
  };
  template <typename BaseVariableType_>
  class CueSliceRegistry {
  public:
    using SliceBase  = AlignerSliceProcessorBase_<BaseVariableType_>;
    using Translator = std::function<bool(SliceBase*, CueBinding&)>;  // false: not my slice type
    static std::vector<Translator>& translators() {
      static std::vector<Translator> t;
      return t;
    }
    static bool translate(SliceBase* slice, CueBinding& out) {
      for (const Translator& t : translators())
        if (t(slice, out)) return true;
      return false;
    }
  };
  // SliceType: the downstream AlignerSliceProcessor_ subclass; FinderFn: void(FinderType&, srrg2_slice_config&) reads the
  // finder's PARAMs (e.g. c.finder = SRRG2_FINDER_NN_GATED; c.finder_max_distance = finder.param_max_distance_m.value();)
  template <typename SliceType, typename FinderFn>
  inline void registerCueSlice(int slice_kind, FinderFn finder_fn) {
    using BaseVariableType = typename SliceType::BaseVariableType;
    using Registry         = CueSliceRegistry<BaseVariableType>;
    Registry::translators().push_back([slice_kind, finder_fn](typename Registry::SliceBase* base, CueBinding& b) -> bool {
      SliceType* s = dynamic_cast<SliceType*>(base);
      if (!s) return false;
      srrg2_slice_default_config(&b.config, amd_detail::variableKind<typename SliceType::VariableType>());
      b.config.kind                    = slice_kind;
      b.config.min_num_correspondences = s->param_min_num_correspondences.value();  // aligner_slice_processor.h:62-66
      amd_detail::robustifierOf(s->param_robustifier.value().get(), b.config);     // aligner_slice_processor_base.h:34-38
      if (!s->param_finder.value())
        throw std::runtime_error("MultiAlignerBase_|cue slice without a finder (aligner_slice_processor_impl.cpp:8-17)");
      finder_fn(*s->param_finder.value(), b.config);                                // aligner_slice_processor.h:56-60
      using FixedPoint  = typename SliceType::FixedPointType;
      using MovingPoint = typename SliceType::MovingPointType;
      if (auto* f = s->fixed()) {
        b.fixed_size = (int) f->size();  b.fixed_stride = (int) sizeof(FixedPoint);
        if (!f->empty()) { b.fixed_coords = f->front().coordinates().data();  b.fixed_normals = f->front().normal().data(); }
      }
      if (auto* m = s->moving()) {
        b.moving_size = (int) m->size();  b.moving_stride = (int) sizeof(MovingPoint);
        if (!m->empty()) { b.moving_coords = m->front().coordinates().data();  b.moving_normals = m->front().normal().data(); }
      }
      return true;
    });
  }

  // ---- MultiAlignerBase_ : the reference's class name, PARAMs and base class; the registration runs on the device -------
  template <typename VariableType_>
  class MultiAlignerBase_
    : public Aligner_<typename VariableType_::EstimateType, PropertyContainerBase, PropertyContainerBase> {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using VariableType     = VariableType_;
    using BaseVariableType = typename VariableType::BaseVariableType;
    using EstimateType     = typename VariableType::EstimateType;
    using BaseType         = Aligner_<EstimateType, PropertyContainerBase, PropertyContainerBase>;
    using AlignerSliceProcessorType    = AlignerSliceProcessorBase_<BaseVariableType>;
    using AlignerSliceProcessorTypePtr = std::shared_ptr<AlignerSliceProcessorType>;
    static constexpr int Dim = EstimateType::Dim;
    static constexpr int D   = Dim == 2 ? 3 : 6;

    // multi_aligner.h:34-57, names and defaults unchanged (`solver` is accepted and ignored: the one-variable Gauss-Newton
    // step of multi_aligner_impl.cpp:112 runs on the device; a configuration that sets it still loads)
    PARAM_VECTOR(PropertyConfigurableVector_<AlignerSliceProcessorType>, slice_processors, "slices", &(this->_slices_changed_flag));
    PARAM(PropertyConfigurable_<Solver>, solver, "this solver", std::shared_ptr<Solver>(new Solver), nullptr);
    PARAM(PropertyInt, min_num_inliers, "minimum number ofinliers", 10, nullptr);
    PARAM(PropertyBool, enable_inlier_only_runs, "toggles additional inlier only runs if sufficient inliers are available", false, nullptr);
    PARAM(PropertyBool,
          keep_only_inlier_correspondences,
          "toggles removal of correspondences which factors are not inliers in the last iteration",
          false,
          nullptr);

    MultiAlignerBase_() {
      // multi_aligner.h:61-62: the registration solver runs ONE iteration per call
      param_solver->param_max_iterations.value().clear();
      param_solver->param_max_iterations.pushBack(1);
      _constructed_solver = param_solver.value().get();
    }
    virtual ~MultiAlignerBase_() {
      if (_h) srrg2_aligner_destroy(_h);
    }
    // HIP device ordinal of this aligner (not a PARAM: configurations do not know about devices)
    void setDevice(int device_) {
      _device              = device_;
      _slices_changed_flag = true;
    }

    // multi_aligner_impl.cpp:8-24: every slice demuxes the scene itself (bindFixed / bindMoving find the cloud by name in
    // its three storage forms, aligner_slice_processor_base_impl.cpp:27-50)
    void setFixed(PropertyContainerBase* fixed_scene_) override {
      BaseType::setFixed(fixed_scene_);
      for (size_t i = 0; i < param_slice_processors.size(); ++i) param_slice_processors.value(i)->setFixed(fixed_scene_);
      _clouds_changed = true;
    }
    void setMoving(PropertyContainerBase* moving_scene_) override {
      BaseType::setMoving(moving_scene_);
      for (size_t i = 0; i < param_slice_processors.size(); ++i) param_slice_processors.value(i)->setMoving(moving_scene_);
      _clouds_changed = true;
    }
    // multi_aligner_impl.cpp:27-44 (also what a prior slice's init() calls to override the guess)
    void setMovingInFixed(const EstimateType& moving_in_fixed_) override {
      if (!param_slice_processors.size()) throw std::runtime_error("MultiAlignerBase_::setMovingInFixed|no slices");  // :30
      this->_moving_in_fixed = moving_in_fixed_;
    }
    const EstimateType& movingInFixed() const override { return this->_moving_in_fixed; }

    // multi_aligner_impl.cpp:47-95: blocking; status, estimate and iteration statistics are host visible on return
    void compute() override {
      if (!param_slice_processors.size()) throw std::runtime_error("MultiAlignerBase_::compute|no slices");  // :49
      amd_detail::checkSolver(param_solver.value(), _constructed_solver);
      // (an override of _preCompute / _postCompute written like the reference's default calls _runSolver; that
      // must run the iterations, not compute() with its hooks again)
      if (_in_compute) throw std::runtime_error("MultiAlignerBase_::compute|re-entered from one of its own hooks: call _runSolver there");
      struct Guard { bool& f; explicit Guard(bool& f_) : f(f_) { f = true; } ~Guard() { f = false; } } guard(_in_compute);
      // the virtual hook of multi_aligner.h:141: a downstream subclass that overrides it still compiles and still runs
      // before the registration; the default does what the reference's does (:131-141)
      _preCompute();
      runOnDevice();
      // the virtual hook of multi_aligner.h:143.  The reference reaches it only when the first run left statistics and
      // enough inliers (multi_aligner_impl.cpp:73-88: it returns before it on Fail and on NotEnoughInliers), and its default
      // starts the inlier-only run (:163-181) -- that run has already happened on the device (enable_inlier_only_runs
      // travels in srrg2_aligner_params), so the default here is empty; an override sees the final status, estimate,
      // statistics and correspondences.
      if (this->_status == AlignerBase::Success && !this->_iteration_stats.empty()) _postCompute();
    }

  protected:
    // one blocking registration on the device with the CURRENT PARAMs, slices, clouds and guess: no hooks (compute() and
    // _runSolver() both end up here)
    void runOnDevice() {
      this->_status = AlignerBase::Fail;
      this->_iteration_stats.clear();
      bindSlices();
      float T[12];
      amd_detail::toRowMajor(this->_moving_in_fixed, T);
      amd_detail::check(srrg2_aligner_set_moving_in_fixed(_h, T));
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
        amd_detail::check(srrg2_aligner_set_termination(_h, nullptr));  // null criterion: run max_iterations (aligner.h:31-35)
      }
      int st = AlignerBase::Fail;
      amd_detail::check(srrg2_aligner_compute(_h, &st));
      this->_status = static_cast<AlignerBase::Status>(st);  // identical values, aligner.h:23-28
      amd_detail::check(srrg2_aligner_get_moving_in_fixed(_h, T));
      amd_detail::fromRowMajor(T, this->_moving_in_fixed);
      int n = 0;
      amd_detail::check(srrg2_aligner_get_iteration_stats(_h, nullptr, &n));
      std::vector<srrg2_iteration_stats> s((size_t) n);
      if (n) amd_detail::check(srrg2_aligner_get_iteration_stats(_h, s.data(), &n));
      for (const srrg2_iteration_stats& e : s) {  // the fields the reference reads (multi_aligner_impl.cpp:81-82)
        IterationStats is;
        is.iteration    = e.iteration;
        is.num_inliers  = e.num_inliers;
        is.num_outliers = e.num_outliers;
        is.chi_inliers  = e.chi_inliers;
        is.chi_outliers = e.chi_outliers;
        this->_iteration_stats.push_back(is);
      }
      // slice->correspondences() of every cue slice, as the reference leaves them (after pruning if enabled, :214-263)
      for (size_t i = 0; i < param_slice_processors.size(); ++i) {
        AlignerSliceProcessorTypePtr slice = param_slice_processors.value(i);
        if (slice->isPrior()) continue;
        int nc = 0;
        amd_detail::check(srrg2_aligner_get_correspondences(_h, (int) i, nullptr, &nc));
        std::vector<srrg2_correspondence> buf((size_t) nc);
        if (nc) amd_detail::check(srrg2_aligner_get_correspondences(_h, (int) i, buf.data(), &nc));
        CorrespondenceVector& out = slice->correspondences();
        out.clear();
        out.reserve(buf.size());
        for (const srrg2_correspondence& c : buf) out.emplace_back(Correspondence(c.fixed_idx, c.moving_idx, c.response));
      }
    }

  public:
    // multi_aligner_impl.cpp:275-285
    int numCorrespondences() override {
      int n = 0;
      if (_h) amd_detail::check(srrg2_aligner_num_correspondences(_h, &n));
      return n;
    }
    // multi_aligner_impl.cpp:266-272: every slice exports ITS correspondences into the moving scene as the property
    // "<fixed_slice_name>_correspondences" (aligner_slice_processor_impl.cpp:51-74; suffix correspondence_finder.h:25) --
    // the slices' own code, fed by the vectors compute() filled
    void storeCorrespondences() override {
      for (size_t i = 0; i < param_slice_processors.size(); ++i) param_slice_processors.value(i)->storeCorrespondences();
    }
    void setPlatform(PlatformPtr platform_) override {  // multi_aligner_impl.cpp:288-295
      BaseType::setPlatform(platform_);
      for (size_t i = 0; i < param_slice_processors.size(); ++i) param_slice_processors.value(i)->setPlatform(platform_);
    }
    void draw(srrg2_core::ViewerCanvasPtr canvas_) const override {
      for (size_t i = 0; i < param_slice_processors.size(); ++i) param_slice_processors.value(i)->draw(canvas_);
    }

  protected:
    // ---- the protected surface of multi_aligner.h:108-149, kept so that a downstream subclass written against the
    // reference compiles and behaves.  What they do in the reference, and here:
    //   _preCompute()  (virtual, :141 / impl :131-141)  keeps the slices' robustifiers for the inlier-only run and calls
    //       slice->init(this) on every slice: prior slices compute their measurement there and override the initial guess
    //       through setMovingInFixed().  Same here (the robustifiers are read by value when the slices are bound).
    //   _postCompute() (virtual, :143 / impl :163-181)  see compute(): called last, default empty.
    //   _setupAligner() (:117 / impl :130-161)  rebuilds the one-variable factor graph from the slices: here it (re)binds
    //       the slices to the device handle, which is what "the graph" is on this side.
    //   _runSolver(n, criterion) (:121 / impl :98-128)  the iteration loop: on the device, inside srrg2_aligner_compute; a
    //       subclass cannot interleave host code with its iterations -- calling it runs a whole compute() of n iterations.
    //   _pruneCorrespondences / _setClampRobustifiers / _restoreRobustifiers (:111-115)  device-side steps of compute()
    //       (keep_only_inlier_correspondences, enable_inlier_only_runs): nothing left to do on the host, kept as no-ops.
    //   _graph (:147)  stays null: there is no host-side FactorGraph; _robustifiers_original_per_slice (:149) is filled by
    //       _preCompute as in the reference.
    virtual void _preCompute() {
      const size_t ns = param_slice_processors.size();
      _robustifiers_original_per_slice.clear();
      _robustifiers_original_per_slice.reserve(ns);
      for (size_t i = 0; i < ns; ++i) {
        AlignerSliceProcessorTypePtr slice = param_slice_processors.value(i);
        _robustifiers_original_per_slice.push_back(slice->param_robustifier.value());
        slice->init(this);
      }
    }
    virtual void _postCompute() {}
    void _setupAligner() { bindSlices(); }
    void _runSolver(const size_t& number_of_iterations_, const std::shared_ptr<AlignerTerminationCriteriaBase> termination_criterion_) {
      const int keep_it = this->param_max_iterations.value();
      auto keep_tc      = this->param_termination_criteria.value();
      this->param_max_iterations.setValue((int) number_of_iterations_);
      this->param_termination_criteria.setValue(termination_criterion_);
      runOnDevice();  // (the iterations only: no _preCompute / _postCompute, as in the reference)
      this->param_max_iterations.setValue(keep_it);
      this->param_termination_criteria.setValue(keep_tc);
    }
    void _pruneCorrespondences() {}
    void _setClampRobustifiers() {}
    void _restoreRobustifiers() {}

    // (re)creates the device-side aligner from the CURRENT slice objects: slice list, PARAMs, bound clouds, prior
    // measurements.  Cheap when nothing changed (the clouds are re-bound only after setFixed / setMoving).
    void bindSlices() {
      const size_t ns = param_slice_processors.size();
      std::vector<srrg2_slice_config> configs(ns);
      std::vector<CueBinding> cues(ns);
      std::vector<EstimateType, Eigen::aligned_allocator<EstimateType>> prior_Z(ns, EstimateType::Identity());
      for (size_t i = 0; i < ns; ++i) {
        AlignerSliceProcessorTypePtr slice = param_slice_processors.value(i);
        if (slice->isPrior()) {
          // the slice's own factor after its own setupFactor() (computeCorrespondences() of a prior slice is exactly that,
          // aligner_slice_processor_prior.h:52-55): measurement, information, robustifier
          slice->computeCorrespondences();
          srrg2_slice_default_config(&configs[i], amd_detail::variableKind<VariableType>());
          configs[i].kind                     = SRRG2_SLICE_PRIOR;
          configs[i].finder                   = SRRG2_FINDER_NONE;
          configs[i].prior_sets_initial_guess = 0;  // the slice's init() has already set the guess through setMovingInFixed
          amd_detail::robustifierOf(slice->param_robustifier.value().get(), configs[i]);
          FactorBasePtr f = slice->factor();  // (throws "no fixed" / "no moving" like the reference, prior_impl.cpp:11-22)
          if (auto* f2 = dynamic_cast<SE2PriorErrorFactor*>(f.get())) {
            readPrior(*f2, configs[i], prior_Z[i]);
          } else if (auto* f3 = dynamic_cast<SE3PriorErrorFactorAD*>(f.get())) {
            readPrior(*f3, configs[i], prior_Z[i]);
          } else {
            throw std::runtime_error("MultiAlignerBase_|prior slice with a factor type the device does not evaluate");
          }
        } else {
          if (!CueSliceRegistry<BaseVariableType>::translate(slice.get(), cues[i]))
            throw std::runtime_error("MultiAlignerBase_|cue slice type [" + slice->className() +
                                     "] has no translator: registerCueSlice<...>() it next to its BOSS_REGISTER_CLASS");
          configs[i] = cues[i].config;
          // the sensor pose the reference looks up on every setMovingInFixed (aligner_slice_processor_impl.cpp:20-36)
          EstimateType sensor_in_robot = EstimateType::Identity();
          if (this->_platform && !slice->param_frame_id.value().empty() && !slice->param_base_frame_id.value().empty())
            this->_platform->getTransform(sensor_in_robot, slice->param_frame_id.value(), slice->param_base_frame_id.value());
          amd_detail::toRowMajor(sensor_in_robot, configs[i].sensor_in_robot);
        }
      }
      const bool rebuild = !_h || _slices_changed_flag || configs.size() != _bound_configs.size() ||
                           (ns && std::memcmp(configs.data(), _bound_configs.data(), sizeof(srrg2_slice_config) * ns) != 0);
      if (rebuild) {
        if (_h) srrg2_aligner_destroy(_h);
        _h = nullptr;
        amd_detail::check(srrg2_aligner_create(amd_detail::variableKind<VariableType>(), _device, &_h));
        for (size_t i = 0; i < ns; ++i) {
          int idx = -1;
          amd_detail::check(srrg2_aligner_add_slice(_h, &configs[i], &idx));
        }
        _bound_configs       = configs;
        _slices_changed_flag = false;
        _clouds_changed      = true;
      }
      for (size_t i = 0; i < ns; ++i) {
        if (configs[i].kind == SRRG2_SLICE_PRIOR) {
          float Z[12];
          amd_detail::toRowMajor(prior_Z[i], Z);
          amd_detail::check(srrg2_aligner_set_prior_measurement(_h, (int) i, Z));
        } else if (_clouds_changed) {
          const CueBinding& b = cues[i];
          // two slices bound to the SAME cloud objects of the scene (same fixed_slice_name / moving_slice_name,
          // aligner_slice_processor_base_impl.cpp:27-50): the later one reads the earlier one's device copy (projective
          // finders; with equal finder parameters they then share one association pass per iteration)
          int shares = -1;
          for (size_t j = 0; j < i && shares < 0; ++j)
            if (configs[j].kind != SRRG2_SLICE_PRIOR && configs[j].finder == SRRG2_FINDER_PROJECTIVE &&
                configs[i].finder == SRRG2_FINDER_PROJECTIVE && cues[j].fixed_coords == b.fixed_coords &&
                cues[j].moving_coords == b.moving_coords && cues[j].fixed_size == b.fixed_size && cues[j].moving_size == b.moving_size)
              shares = (int) j;
          if (shares >= 0) {
            amd_detail::check(srrg2_aligner_share_clouds(_h, (int) i, shares));
            continue;
          }
          amd_detail::check(srrg2_aligner_share_clouds(_h, (int) i, -1));
          amd_detail::check(srrg2_aligner_set_fixed(_h, (int) i, b.fixed_coords, b.fixed_stride, b.fixed_normals, b.fixed_stride,
                                                    b.fixed_size, SRRG2_MEM_HOST));
          amd_detail::check(srrg2_aligner_set_moving(_h, (int) i, b.moving_coords, b.moving_stride, b.moving_normals,
                                                     b.moving_stride, b.moving_size, SRRG2_MEM_HOST));
        }
      }
      _clouds_changed = false;
    }
    template <typename PriorFactorType>
    static void readPrior(const PriorFactorType& f, srrg2_slice_config& c, EstimateType& Z) {
      Z = f.measurement();
      const auto& Om = f.informationMatrix();  // diagonal by construction (param_diagonal_info_matrix.asDiagonal())
      for (int k = 0; k < D; ++k) c.prior_information_diag[k] = Om(k, k);
    }

    // multi_aligner.h:146-149
    FactorGraphPtr _graph     = nullptr;  // (never built: the registration graph lives on the device)
    bool _slices_changed_flag = true;     // set by PARAM_VECTOR slice_processors (:34-37) and setDevice: the handle is rebuilt
    std::vector<RobustifierBasePtr> _robustifiers_original_per_slice;
    srrg2_aligner_h _h   = nullptr;
    int _device          = 0;
    bool _clouds_changed = true;
    bool _in_compute     = false;               // compute() is running its hooks (re-entrancy guard)
    const Solver* _constructed_solver = nullptr;  // the solver this aligner built itself (checkSolver)
    std::vector<srrg2_slice_config> _bound_configs;
  };

  // multi_aligner.h:152-159 -- the names instances.cpp:35 and downstream code use
  using MultiAligner2D      = MultiAlignerBase_<VariableSE2RightAD>;
  using MultiAligner3D      = MultiAlignerBase_<VariableSE3EulerRightAD>;
  using MultiAligner2DPtr   = std::shared_ptr<MultiAligner2D>;
  using MultiAligner3DPtr   = std::shared_ptr<MultiAligner3D>;
  using MultiAligner3DQR    = MultiAlignerBase_<VariableSE3QuaternionRightAD>;
  using MultiAligner3DQRPtr = std::shared_ptr<MultiAligner3DQR>;

}  // namespace srrg2_slam_interfaces

#endif  // SRRG2_SLAM_AMD_HAVE_SRRG2_CORE
