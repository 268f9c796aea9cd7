// device_types.h -- PODs shared by the HIP kernels and the host driver of the aligner.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srrg2_slam_amd.h"

#define ACC_N 32          // int64 accumulators per (problem, slice)
#define PARTIAL_SLOTS 32  // slot sets per problem that the step kernels add into (atomics), summed by the control kernel
#define ACC_B 21          // b starts after the 21 upper-triangular entries of H
#define ACC_CHI_IN 27
#define ACC_CHI_OUT 28
#define ACC_N_IN 29
#define ACC_N_OUT 30
#define ACC_N_CORR 31
#define TERM_WINDOW_MAX 64
#define NO_MATCH 0x7fffffff

// upper-triangular index of H(a,b), a <= b, in a 6x6 layout (3-dof variables use the same table)
__host__ __device__ __forceinline__ constexpr int hidx(int a, int b) {
  return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

// Search structure over the fixed cloud (CorrespondenceFinder_ state, rebuilt when
// _fixed_changed_flag is set: S/registration/correspondence_finder.h:80-83,117).
// Points are stored sorted by cell; pts[j] = {x, y, z, bits(original index)}.
struct GridDev {
  float ox, oy, oz;
  float h, inv_h;
  int nx, ny, nz;
  int rmax;        // cube radius that covers the extended gate
  int rfar_gate;   // cube radius that covers the ball of radius max_distance (<= rmax; computed by the host)
  float gate2;     // max_distance^2
  float gate2_ext; // (1.25 max_distance)^2: how far scans of unmatched points reach (rmax covers it)
  int n;           // number of fixed points (entries of pts)
  const int* cell_start;  // ncell + 1
  const float4* pts;
  const float4* nrm;      // sorted like pts (w unused); null when the cloud has no normals
  const int* pos_of;      // [n] original index -> position in pts (the searches track keys, not positions)
  // Cell neighbour lists (cnl_search; built on demand, kernels_prep.hip: k_cnl_count / k_cnl_fill): GridLists, in device
  // memory behind a pointer -- carried inline in the kernel arguments their 100 bytes are preloaded into scalar registers by
  // EVERY kernel that takes a SliceDev, and the converged-pass kernel then spills 30 of them in its hot path.
  int list_R;                     // 0: no lists
  const struct GridLists* lists;  // (device)
};
// For every cell c of the grid EXTENDED by R cells on every side: the OCCUPIED cells whose points can lie within the
// extended gate of a query in c, nearest class first.  An entry = {position of its first point in pts,
// byte 0: count - 1 | class << 4, bytes 1 .. 3: dx + R, dy + R, dz + R; the tight box of its points, lower and upper corner:
// one byte per axis, (d + R) * 16 + sixteenths of the cell (k_cnl_boxes)}; class m = sum_i max(|d_i| - 1, 0)^2 = squared
// cell-to-cell separation in cells (entries of a list are sorted by class, then by centre distance); a cell of more than
// CNL_ENTRY_MAX points takes several entries (the entries are the work items the lanes of a wave share).
// cls_b2[m] = ((sqrt(m) - 0.01) h)^2 * 0.9999: every point of a class-m cell is farther than that from every query of c
// (bound2_of generalised to non-integer separations).
struct GridLists {
  int R;
  int lnx, lny, lnz;   // extended grid: n + 2 R per axis (2-D: lnz = 1)
  const int* start;    // [lnx * lny * lnz + 1]
  const uint4* ent;
  float cls_b2[14];    // classes 0 .. 12 (R <= 3) + sentinel
};
#define CNL_MAX_R 3
#ifndef CNL_ENTRY_MAX
#define CNL_ENTRY_MAX 16  // (<= 16: the count travels in four bits)
#endif
#define CNL_MAX_CLASS 12

// Fused control steps (round 5; tuning.fused_control).  The control step of iteration i -- sum the slot sets, one
// Gauss-Newton step, X <- X [+] dx, statistics, termination: k_icp_control's work -- runs in the PROLOGUE of the first pass
// kernel of iteration i + 1, on ONE wave (wave 0 of workgroup (0, problem), which the dispatcher starts before its
// siblings): no launch and no kernel boundary between the passes of two iterations.  What the passes need of the state
// (finder transforms, exponent, flags) travels in a per-(problem, slice) RECORD of 8-byte granules {value, epoch tag}, each
// written by one store and read by one load per lane: a wave that finds every tag at the epoch of its launch proceeds; a
// stale record means the control step has not been applied yet -- the designated wave applies it, the others poll
// replicated epoch words.  pub == nullptr: legacy path (the passes read ProblemState, every control step is a launch).
#define PUB_SLICE_GRANULES 64  // per (problem, slice), one granule per lane of a wave: [0, 12) Tf, [12, 24) Tfprev, ...
#define PUB_G_KEXP 24
#define PUB_G_FLAGS 25
#define PUB_G_NSTATS 26
#define PUB_G_WCOUNT 27
#define PUB_G_NPASSES 28
#define PUB_G_X 32             //   ... [32, 44) X: with these the control step reads nothing of ProblemState
#define PUB_FLAG_STOP 1u       //   flags: done | finished
#define PUB_FLAG_PHASE1 2u     //          inlier-only run (Clamp robustifiers)
#define PUB_FLAG_PRIOR 4u      //          nstats > 0 (neighbours of a previous pass exist)
#define PUB_EPOCH_REPLICAS 8   // poll words per problem, 128 bytes apart
#define PUB_EPOCH_STRIDE 32    // (in 4-byte words)
struct CtlParams;
struct FusedCtl {
  unsigned long long* pub;     // [K][SRRG2_MAX_SLICES][PUB_SLICE_GRANULES]
  unsigned* pub_epoch;         // [K][PUB_EPOCH_REPLICAS][PUB_EPOCH_STRIDE]
  const CtlParams* ctl;        // device copy of this compute()'s control parameters (written by k_icp_init): termination criterion
  srrg2_iteration_stats* stats;
  long long* prev_partials;    // the slot sets the control step of this launch's epoch consumes, [K][32][32]: one of THREE buffers --
                               // the passes of round k add into buffer k % 3, the control step behind them reads it and zeroes
                               // buffer (k + 2) % 3 (`zero_partials`: read by the step before, added to again by the round after
                               // next), so the buffer a fused step reads is READ-ONLY for the whole launch that carries it: a
                               // polling wave may apply the step itself from the same inputs (pass_view_fused)
  long long* zero_partials;
  int epoch;                   // control steps the passes of this launch build on
  int min_num_correspondences; // of the (one) cue slice
  int max_stats;
  int has_term;
  int first_round;             // workgroups (linear index, problem fastest) that may start before the control steps have published
  int prior;                   // neighbours of a previous pass of this compute() exist (every pass but the first of the first run)
  int prior_mask;              // bit s: slice s is a PRIOR slice (its factor is linearised by the control wave from ctl->slices[s]:
                               // wave_prior); 0: cue slices only
  int nslices;                 // slices of the aligner (cue + prior)
};

// One cue slice (AlignerSliceProcessor_) as the step kernel sees it.
struct SliceDev {
  GridDev grid;
  const float4* mpts;   // moving points of all problems of the batch, concatenated
  const float4* mnrm;   // moving normals or null
  // Correspondence records per moving point.  Projective / given-correspondences passes write them; the nearest-
  // neighbour passes do not (9 bytes per point and iteration that nobody reads before compute() returns): k_icp_outputs
  // derives them on demand from the neighbour state and the transform of the last pass (ProblemState::Tlast).
  int* corr_fixed;      // matched fixed index or -1
  float* corr_resp;     // response (squared distance)
  uint8_t* corr_stat;   // srrg2_factor_status of the last linearisation
  float4* prev_f;       // per moving point (sorted order): the nearest neighbour {x, y, z, bits(index)} found by the
                        // previous iteration of this compute() (.w = NO_MATCH: none); an upper bound for the next search
  float4* prev_n;       // ... and its normal (plane factors / normal gate): a kept neighbour needs no dependent load
  int* prev_pos;        // ... and its position in grid.pts (-1: none): batches gather the neighbour from the cache-resident
                        // fixed cloud instead of streaming 32 bytes per moving point (k_icp_step_fast<.., GATHER>)
  float* prev_m;        // ... and its exclusion radius: no other fixed point within prev_m of the previous query
  long long* partials;  // [problem][PARTIAL_SLOTS][ACC_N]: fixed-point partial sums, added with 64-bit atomics
  void* queue;          // deferred searches: QEntry[total moving points] (per problem at its moving offset), or null
  int* qcount;          // [problem] entries in the queue
  int slice_idx;
  int robust_kind;
  float robust_thr;
  float normal_cos;
  int use_normal_gate;
  int gather_prev;  // the previous neighbour and its normal are re-read from the fixed cloud through prev_pos (batches: the cloud
                    // is shared by all alignments and stays in L2) instead of kept per moving point in prev_f / prev_n
  int variable_kind;
  // projective finder (organised fixed cloud in ingest order + per-problem z-buffer)
  int finder;           // srrg2_finder_kind
  int factor;           // srrg2_slice_kind
  float K[9];
  int rows, cols;
  float depth_min, depth_max;
  float gate;           // finder_max_distance
  const float4* fixed_org;      // rows*cols points, NaN = invalid pixel
  const float4* fixed_org_nrm;  // or null
  unsigned long long* zbuf;     // [2][problem][rows*cols] keys (depth bits << 32 | caller index), ping-pong
  int zbuf_parity;              // which of the two this pass uses
  // given correspondences (SRRG2_FINDER_CORRESPONDENCES): pairs of all problems, per-problem offsets, factor status
  const srrg2_correspondence* gcorr;
  const int* gcorr_off;         // [K + 1]
  uint8_t* gcorr_stat;
  const float4* moving_raw;     // moving clouds in ingest order (moving_idx is the caller's index)
  unsigned long long* dbg;  // -DSRRG2_TIMELINE builds only: [iteration < 32][wave][16] shader-clock stamps, or null
  int tune;             // strategy switches (env SRRG2_AMD_TUNE; all exact); the result-changing timing knobs exist only
                        // in -DSRRG2_TIMING_KNOBS builds (kernels.hip: KNOB)
  int prob0;            // first problem of the launch (blockIdx.y = 0): a launch may cover a sub-range of the batch (the two
                        // halves of a pipelined batch run on two streams)
  float Sinv[12];       // robot_in_sensor = sensor_in_robot^-1
  FusedCtl fc;          // fused control steps (fc.pub == nullptr: off)
};

struct ProblemDev {
  int moff;  // offset of this problem's moving points in the concatenated arrays
  int nm;
};

// device-resident state of one alignment (one MultiAlignerBase_::compute())
struct ProblemState {
  float X[12];      // moving_in_fixed; equals the `moving_in_fixed` backup of _runSolver (:102)
  float Xprev[12];  // X of the previous iteration's finder passes (temporal coherence of the search)
  int status;       // srrg2_status
  int done;         // current _runSolver loop left (break at :110 or :125)
  int finished;     // compute() returned early (:75-78, :81-85)
  int nstats;       // IterationStats appended so far
  int phase;        // 0 = main run, 1 = inlier-only run (_postCompute :165-175)
  int kexp[SRRG2_MAX_SLICES];   // fixed-point exponent per slice
  int ncorr[SRRG2_MAX_SLICES];  // correspondences of the last finder pass per slice
  int ninl[SRRG2_MAX_SLICES];   // inliers of the last linearisation per slice
  int qmode[SRRG2_MAX_SLICES];  // 1: open points are deferred to the queue; 0: finished inside the step kernel
  // finder transform robot_in_sensor * X of every cue slice (rows [r0 r1 r2 t], SE(2) spread into the same slots) for
  // the coming passes and for the previous ones: computed once by the init / control kernels instead of by every wave
  float Tf[SRRG2_MAX_SLICES][12], Tfprev[SRRG2_MAX_SLICES][12];
  float Tlast[SRRG2_MAX_SLICES][12];  // finder transforms of the last EXECUTED finder passes (k_icp_outputs)
  int npasses;                          // finder passes executed by this compute()
  int w_count;
  double w_corr[TERM_WINDOW_MAX], w_inl[TERM_WINDOW_MAX], w_out[TERM_WINDOW_MAX], w_chi[TERM_WINDOW_MAX];
  double last_H[36], last_b[6], last_dx[6];
};

// what the host reads back after compute()
struct ProblemOut {
  float X[12];
  int status;
  int nstats;
  int ncorr[SRRG2_MAX_SLICES];
  float H[36];  // H of the last Gauss-Newton iteration, D x D row-major
  int seq;  // CtlParams::seq of the compute() that wrote this record: written last, behind a system-scope fence
};

struct SliceCtl {
  int kind;                 // srrg2_slice_kind
  int min_num_correspondences;
  int robust_kind;
  float robust_thr;
  float gate;               // finder_max_distance
  int has_prior;
  int prior_sets_initial_guess;
  float prior_Z[12];
  float prior_info[6];
  int finder;               // srrg2_finder_kind
  float K0, K4;             // focal lengths (projective / reprojection bounds)
  int rows, cols;
  float depth_min;
  int* qcount;              // deferred-search queue counters of the slice (reset by the control kernel), or null
  int* qprobe_host;         // pinned host copy of the counters of iteration probe_it ([problem][near, far]), or null
  const ProblemDev* probs;  // [problem] table of the slice (device): point counts for the decision threshold
  int nm_global;            // > 0: moving points of the alignment over ALL ranks (point-sharded alignment: sizes the exponent)
  const long long* partials;  // [3][problem][PARTIAL_SLOTS][ACC_N] (null for priors; buffers 1 and 2 are used by fused control steps)
  const unsigned* pinf_bits;      // [problem] max |coord| of the finite moving points (float bits)
  const unsigned* ninf_bits;      // [1] max |component| of the fixed normals
  const unsigned* finf_bits;      // [1] max |coordinate| of the fixed cloud (given-correspondences slices)
  const int* gcorr_off;           // [K + 1] offsets of the given correspondences (or null)
  float Sinv[12];                 // robot_in_sensor = sensor_in_robot^-1 (SliceDev::Sinv)
};

// a single alignment's initial guess and per-slice problem table, carried in k_icp_init's arguments
struct InitInline {
  int use;
  float guess[12];
  ProblemDev pd[SRRG2_MAX_SLICES];
};

// the guesses and problem-table rows of up to 16 alignments of ONE launch (a part of a pipelined batch), carried in the arguments of
// a first pass that has the prologue inside (k_icp_step_cnl_init_batch): no table is read from pinned memory by the pass
#define INIT_BATCH_MAX 16
struct InitBatch {
  float guess[INIT_BATCH_MAX][12];  // [problem - prob0]
  ProblemDev pd[INIT_BATCH_MAX];    // ... of the cue slice (prior slices have no cloud: {0, 0})
  int cue_slice;
};

struct CtlParams {
  int variable_kind;
  int nslices;
  int K;
  srrg2_aligner_params params;
  int has_term;
  srrg2_termination_params term;
  int max_stats;  // capacity of the per-problem stats array
  int tune;       // SRRG2_AMD_TUNE (see SliceDev::tune)
  int probe_it;   // iteration after which the use of the deferred-search queue is decided (-1: never)
  int seq;        // sequence number of this compute() (completion flag in ProblemOut)
  int prob0, nprob;  // the problems [prob0, prob0 + nprob) of the batch are this launch's (nprob = 0: all K)
  // fused control steps (FusedCtl): the records, the epoch THIS control step produces (0: k_icp_init), which of the three
  // slot-set buffers it consumes (`parity`: the passes of round k add into buffer k % 3; SliceCtl::partials is the base of all:
  // [3][K][32][32]) and which one it zeroes (`zero_parity` = (k + 2) % 3; legacy, pub == nullptr: one buffer, zeroed when read)
  unsigned long long* pub;  // null: legacy
  unsigned* pub_epoch;
  CtlParams* ctl_dev;       // k_icp_init stores this record there for the fused control steps
  int epoch;
  int parity;
  int zero_parity;
  SliceCtl slices[SRRG2_MAX_SLICES];
};
