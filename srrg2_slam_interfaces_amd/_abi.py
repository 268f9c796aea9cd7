"""ctypes mirror of the PODs and enums in include/srrg2_slam_amd.h.

Shared by the product binding (``_capi.py``) and by the test-side oracle binding
(``oracle/pyoracle.py``); it contains declarations only, no behaviour.
"""
import ctypes as C

ABI_VERSION = 4
MAX_SLICES = 8

# enum srrg2_variable_kind  (S/registration/aligners/multi_aligner.h:152-158)
SE2_RIGHT, SE3_EULER_RIGHT, SE3_QUAT_RIGHT = 0, 1, 2
# enum srrg2_status  (S/registration/aligners/aligner.h:23-28)
SUCCESS, NOT_ENOUGH_CORRESPONDENCES, NOT_ENOUGH_INLIERS, FAIL = 0, 1, 2, 3
# enum srrg2_slice_kind
SLICE_P2P, SLICE_P2PLANE, SLICE_REPROJECTION, SLICE_PRIOR = 0, 1, 2, 3
# enum srrg2_finder_kind
FINDER_NONE, FINDER_NN_GATED, FINDER_PROJECTIVE, FINDER_CORRESPONDENCES = 0, 1, 2, 3
# enum srrg2_robustifier_kind
ROBUST_NONE, ROBUST_CLAMP, ROBUST_SATURATED, ROBUST_CAUCHY = 0, 1, 2, 3
# enum srrg2_factor_status
FACTOR_INLIER, FACTOR_KERNELIZED, FACTOR_SUPPRESSED = 0, 1, 2
# enum srrg2_mem
MEM_HOST, MEM_DEVICE, MEM_DEVICE_KEPT = 0, 1, 2


class Correspondence(C.Structure):
    _fields_ = [("fixed_idx", C.c_int32), ("moving_idx", C.c_int32), ("response", C.c_float)]


class IterationStats(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32),
        ("num_inliers", C.c_int32),
        ("num_outliers", C.c_int32),
        ("num_suppressed", C.c_int32),
        ("num_correspondences", C.c_int32),
        ("solver_status", C.c_int32),
        ("chi_inliers", C.c_float),
        ("chi_outliers", C.c_float),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class AlignerParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("min_num_inliers", C.c_int32),
        ("enable_inlier_only_runs", C.c_int32),
        ("keep_only_inlier_correspondences", C.c_int32),
    ]


class TerminationParams(C.Structure):
    _fields_ = [
        ("window_size", C.c_int32),
        ("num_correspondences_range", C.c_int32),
        ("num_inliers_range", C.c_int32),
        ("num_outliers_range", C.c_int32),
        ("chi_epsilon", C.c_float),
    ]


class SliceConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("finder", C.c_int32),
        ("robustifier", C.c_int32),
        ("robustifier_chi_threshold", C.c_float),
        ("min_num_correspondences", C.c_int32),
        ("finder_max_distance", C.c_float),
        ("finder_normal_cos", C.c_float),
        ("finder_cell_size", C.c_float),
        ("sensor_in_robot", C.c_float * 12),
        ("camera_matrix", C.c_float * 9),
        ("image_rows", C.c_int32),
        ("image_cols", C.c_int32),
        ("depth_min", C.c_float),
        ("depth_max", C.c_float),
        ("prior_information_diag", C.c_float * 6),
        ("prior_sets_initial_guess", C.c_int32),
    ]


class BatchResult(C.Structure):
    _fields_ = [
        ("moving_in_fixed", C.c_float * 12),
        ("status", C.c_int32),
        ("num_iterations", C.c_int32),
        ("last", IterationStats),
        ("num_correspondences", C.c_int32),
        ("reserved_", C.c_int32),
        ("information", C.c_float * 36),
    ]


class AlignerTuning(C.Structure):
    """srrg2_aligner_tuning: strategy knobs, every setting gives the same results (include/srrg2_slam_amd.h)"""
    _fields_ = [
        ("strategy_mask", C.c_int32),
        ("queue_probe_iteration", C.c_int32),
        ("small_max_points", C.c_int32),
        ("fast_from_iteration", C.c_int32),
        ("fast_points_per_thread", C.c_int32),
        ("fast_min_points", C.c_int32),
        ("fast_gather", C.c_int32),
        ("fast_batch_queue", C.c_int32),
        ("queue_min_points", C.c_int32),
        ("msort_segments", C.c_int32),
        ("msort_key_bits", C.c_int32),
        ("lds_tile", C.c_int32),
        ("cell_target", C.c_float),
        ("rmax_cap", C.c_float),
        ("search_lists", C.c_int32),
        ("search_team", C.c_int32),
        ("batch_pipeline", C.c_int32),
        ("fused_control", C.c_int32),
        ("reserved_", C.c_int32 * 6),
    ]


def transform_size(variable_kind):
    return 9 if variable_kind == SE2_RIGHT else 12


def point_dim(variable_kind):
    return 2 if variable_kind == SE2_RIGHT else 3


def default_aligner_params():
    # aligner.h:30; multi_aligner.h:45-57
    return AlignerParams(10, 10, 0, 0)


def default_termination_params():
    # aligner_termination_criteria.h:40-56
    return TerminationParams(5, 20, 20, 20, 0.2)


def default_slice_config(variable_kind):
    c = SliceConfig()
    c.kind = SLICE_P2P
    c.finder = FINDER_NN_GATED
    c.robustifier = ROBUST_NONE
    c.robustifier_chi_threshold = 1.0
    c.min_num_correspondences = 0  # aligner_slice_processor.h:62-66
    c.finder_max_distance = 1.0
    c.finder_normal_cos = -2.0
    c.finder_cell_size = 0.0
    ident = [1, 0, 0, 0, 1, 0, 0, 0, 1] if variable_kind == SE2_RIGHT else [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    for i, v in enumerate(ident):
        c.sensor_in_robot[i] = v
    for i in range(6):
        # aligner_slice_odometry_prior.h:17-21 (2D: 1e2) and :41-45 (3D: 1)
        c.prior_information_diag[i] = 100.0 if variable_kind == SE2_RIGHT else 1.0
    c.prior_sets_initial_guess = 1
    return c

# srrg2_aligner_last_compute_path (strategy introspection: what the path tests assert)
PATH_FUSED_CONTROL, PATH_ALL_PASSES_FUSED, PATH_FINAL_WAVE, PATH_PROLOGUE_IN_PASS, PATH_ONE_WORKGROUP, PATH_PRIORS_FUSED = 1, 2, 4, 8, 16, 32
