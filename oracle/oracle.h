/*
 * oracle.h -- CPU oracle of the multi-cue aligner hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (srrg2_slam_interfaces_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED for the arithmetic: the reference's ICP arithmetic lives in un-vendored,
 * un-pinned dependencies (srrg2_solver, srrg2_core, concrete finders of srrg2_laser_slam_2d /
 * srrg2_proslam; srrg2_slam_interfaces/CMakeLists.txt:4-9, package.xml:9-13) that are absent
 * from /root/reference and the reference's own tests hold no point cloud, correspondence
 * or ICP golden vector.  What IS pinned: the control flow restated
 * from S/registration/aligners/multi_aligner_impl.cpp:47-303 and the prior-slice scenarios of
 * T/test_motion_model_slice.cpp:44-227 / T/test_motion_model.cpp:14-315 (tests/test_oracle_*).
 *
 * The oracle exports the same call surface as include/srrg2_slam_amd.h with the prefix
 * `oracle_`, so parity tests drive both sides with identical calls.
 */
#ifndef SRRG2_ORACLE_H
#define SRRG2_ORACLE_H

#include "../include/srrg2_slam_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct o_aligner o_aligner;

typedef struct o_scene o_scene;

/* ---- scene clipping / merging (o_scene.c; same surface as srrg2_scene_* without devices) ---- */
int oracle_scene_create(int dim, o_scene** out);
int oracle_scene_destroy(o_scene* s);
int oracle_scene_set(o_scene* s, const float* coords, int coord_stride_bytes, const float* normals, int normal_stride_bytes,
                     int n);
int oracle_scene_size(o_scene* s, int* n);
int oracle_scene_get(o_scene* s, float* coords_out, float* normals_out, int capacity, int* n);
int oracle_scene_clip_ball(o_scene* full, const float* robot_in_local_map, float range, o_scene* clipped, int* status);
int oracle_scene_global_indices(o_scene* clipped, int32_t* buf, int* n_inout);
int oracle_scene_merge(o_scene* scene, o_scene* measurement, const float* measurement_in_scene,
                       const srrg2_correspondence* correspondences, int n_correspondences, const srrg2_merger_params* p,
                       srrg2_merge_result* out);

/* ---- deterministic math (o_math.c) --------------------------------------- */
void   o_sincos(double x, double* s, double* c);
double o_atan2(double y, double x);
/* SE3 3x4 / SE2 3x3 row-major float; products evaluated in double, rounded once to float */
void o_se3_compose(const float* A, const float* B, float* C);
void o_se3_inverse(const float* A, float* Ainv);
void o_se2_compose(const float* A, const float* B, float* C);
void o_se2_inverse(const float* A, float* Ainv);
void o_se3_v2t(int variable_kind, const double* v, double* R9, double* t3);
void o_se3_t2v_quat(const float* T, double* v6);
void o_se2_t2v(const float* T, double* v3);
void o_se3_fix_transform(float* T);
void o_se2_fix_transform(float* T);
/* X <- X * v2t(dx) */
void o_box_plus(int variable_kind, float* X, const double* dx);
/* L D L^T solve of H dx = -b (D = 3 or 6, H full row-major DxD). returns 0 ok, 1 not PD */
int o_solve(int D, const double* H, const double* b, double* dx);
/* exponent k of the fixed-point accumulators (DESIGN.md "fixed-point reduction") */
int o_fixed_point_exponent(int n_terms, double term_bound);

/* ---- aligner (o_aligner.c), mirrors include/srrg2_slam_amd.h -------------- */
int oracle_abi_version(void);
const char* oracle_last_error(void);

int oracle_aligner_create(int variable_kind, o_aligner** out);
int oracle_aligner_destroy(o_aligner* h);
int oracle_aligner_set_params(o_aligner* h, const srrg2_aligner_params* p);
int oracle_aligner_set_termination(o_aligner* h, const srrg2_termination_params* p);
int oracle_aligner_add_slice(o_aligner* h, const srrg2_slice_config* c, int* slice_idx_out);
int oracle_aligner_clear_slices(o_aligner* h);
int oracle_aligner_set_robustifier(o_aligner* h, int slice_idx, int kind, float chi_threshold);
int oracle_aligner_share_clouds(o_aligner* h, int slice_idx, int source_slice_idx);
int oracle_aligner_set_fixed(o_aligner* h, int slice_idx, const float* coords, int coord_stride_bytes,
                             const float* normals, int normal_stride_bytes, int n, int mem);
int oracle_aligner_set_moving(o_aligner* h, int slice_idx, const float* coords, int coord_stride_bytes,
                              const float* normals, int normal_stride_bytes, int n, int mem);
int oracle_aligner_set_sensor_in_robot(o_aligner* h, int slice_idx, const float* T);
int oracle_aligner_set_prior_measurement(o_aligner* h, int slice_idx, const float* T);
int oracle_aligner_set_moving_in_fixed(o_aligner* h, const float* T);
int oracle_aligner_get_moving_in_fixed(o_aligner* h, float* T_out);
int oracle_aligner_compute(o_aligner* h, int* status_out);
int oracle_aligner_status(o_aligner* h, int* status_out);
int oracle_aligner_get_iteration_stats(o_aligner* h, srrg2_iteration_stats* buf, int* n_inout);
int oracle_aligner_num_correspondences(o_aligner* h, int* n_out);
int oracle_aligner_get_correspondences(o_aligner* h, int slice_idx, srrg2_correspondence* buf, int* n_inout);
int oracle_aligner_get_factor_status(o_aligner* h, int slice_idx, uint8_t* buf, int* n_inout);
int oracle_aligner_set_correspondences(o_aligner* h, int slice_idx, const srrg2_correspondence* correspondences, int n);
int oracle_aligner_compute_batch_correspondences(o_aligner* h, int K, const float* coords, int coord_stride_bytes,
                                                 const float* normals, int normal_stride_bytes, const int32_t* offsets,
                                                 int mem, const srrg2_correspondence* correspondences,
                                                 const int32_t* corr_offsets, const float* guesses,
                                                 srrg2_batch_result* results);
int oracle_aligner_compute_batch(o_aligner* h, int K, const float* coords, int coord_stride_bytes,
                                 const float* normals, int normal_stride_bytes, const int32_t* offsets,
                                 int mem, const float* guesses, srrg2_batch_result* results);

/* ---- pose graph (o_posegraph.c), mirrors srrg2_posegraph_* ------------------------------- */
typedef struct o_posegraph o_posegraph;
int oracle_posegraph_create(int variable_kind, o_posegraph** out);
int oracle_posegraph_destroy(o_posegraph* g);
int oracle_posegraph_set(o_posegraph* g, int V, const float* poses, const uint8_t* fixed_mask, int E, const int32_t* ij,
                         const float* Z, const float* omega, const uint8_t* enabled);
int oracle_posegraph_set_enabled(o_posegraph* g, const uint8_t* enabled);
int oracle_posegraph_solve(o_posegraph* g, const srrg2_posegraph_params* p, srrg2_posegraph_stats* stats, int* n_inout);
int oracle_posegraph_get_poses(o_posegraph* g, float* out);
/* oracle-only: dense Cholesky instead of PCG; chi2 at the current poses; one edge's e/Ji/Jj */
int oracle_posegraph_set_direct(o_posegraph* g, int enable);
double oracle_posegraph_chi(o_posegraph* g);
int oracle_posegraph_edge(o_posegraph* g, int e, double* err, double* Ji, double* Jj);

/* ---- oracle-only hooks used by the tests ---------------------------------- */
/* 1 = brute-force O(Nf*Nm) gated NN (ground truth for indices), 0 = voxel-grid finder */
int oracle_aligner_set_bruteforce(o_aligner* h, int enable);
/* one finder pass + one linearisation at the current estimate, no solve: fills the
 * correspondences and returns the raw fixed-point accumulators (32 x int64) and exponent k */
int oracle_aligner_linearize_once(o_aligner* h, int slice_idx, int64_t* acc32, int* k_out);
/* H (DxD row-major), b (D) of the last linearisation as doubles, summed over slices */
int oracle_aligner_get_last_system(o_aligner* h, double* H, double* b, double* dx);
/* mirror of srrg2_aligner_get_information: that H as float32 */
int oracle_aligner_get_information(o_aligner* h, float* H);

#ifdef __cplusplus
}
#endif
#endif
