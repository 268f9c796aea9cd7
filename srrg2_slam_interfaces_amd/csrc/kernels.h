// kernels.h -- launchers of the HIP kernels in kernels.hip (C++ linkage, internal to the library)
#pragma once
#include "device_types.h"

namespace srrg2amd {

void launch_ingest(const float* src, int stride_floats, int n, int dim, float4* dst, unsigned* maxabs_bits,
                   int finite_per_point, hipStream_t s);
void launch_ingest_batch(const float* src, int stride_floats, const ProblemDev* probs, int K, int max_nm, int dim,
                         float4* dst, unsigned* maxabs_bits, int finite_per_point, hipStream_t s);
void launch_bbox(const float4* pts, int n, unsigned* mn, unsigned* mx, int* nvalid, hipStream_t s);
void launch_ingest_bbox(const float* src, int stride_floats, int n, int dim, float4* dst, unsigned* maxabs_bits, unsigned* mn,
                        unsigned* mx, int* nvalid, hipStream_t s, unsigned* ticket = nullptr, unsigned* host_out = nullptr,
                        unsigned seq = 0, unsigned* block_out = nullptr /* [INGEST_BBOX_MAX_BLOCKS][8] */,
                        unsigned* clear_after = nullptr);
#define INGEST_BBOX_MAX_BLOCKS 512
void launch_grid_count(const GridDev& g, const float4* pts, int n, int* counts, hipStream_t s);
void launch_count_nonzero(const int* counts, int n, int* out, hipStream_t s);
int scan_num_blocks(int n);
void launch_exclusive_scan(int* data, int n, int* block_sums, int* grand_total, hipStream_t s, int* copy = nullptr);
void launch_grid_scatter(const GridDev& g, const float4* pts, const float4* nrm, int n, int* cursor, float4* out_pts,
                         float4* out_nrm, int* pos_of, hipStream_t s);
// cell neighbour lists of the grid (GridDev::list_*): ent == null counts the entries per cell into list_start, else fills
// (box_at: [fixed points] scratch of the fill pass -- the tight box of every entry, by the position of its first point)
void launch_cnl_build(const GridDev& g, const GridLists& L, const int4* offs, int noffs, int* list_start, uint2* box_at,
                      uint4* ent, hipStream_t s);
// search pass over the cell neighbour lists (any number of alignments per launch; no deferred-search queue)
// (init_C / init_inl != null: the FIRST pass of a single alignment's compute() with the prologue inside -- no k_icp_init launch;
// K == 1, fused control steps, the slot sets left zeroed by the previous compute()'s final step: run_compute decides)
void launch_icp_step_cnl(int dim, bool plane, const SliceDev& S, const GridLists& GL, const ProblemDev* probs,
                         ProblemState* states, int K, int max_nm, int team, hipStream_t s, const CtlParams* init_C = nullptr,
                         const InitInline* init_inl = nullptr, const InitBatch* init_bat = nullptr);
// Morton sort of K moving clouds (counts/cursor: (K << kbits) + 1 ints; bb: K*6 keys initialised to
// {0xffffffff x3, 0 x3}; counts zeroed)
// kbits = total key bits (2^kbits cells per cloud); aniso != 0: bits dealt to the axes by extent (kernels_prep.hip: KeySpec);
// segments = workgroups per cloud (0: automatic)
bool launch_msort_local(const float* src, int sf, const float* nsrc, int nsf, const ProblemDev* probs, int K, int dim, int kbits,
                        int aniso, int segments, int max_nm, float4* out_pts, float4* out_nrm, unsigned* maxabs_bits,
                        hipStream_t s);
void launch_msort(const float4* pts, const float4* nrm, const ProblemDev* probs, int K, int max_nm, int kbits, int aniso,
                  unsigned* bb, int* counts, int* cursor, int* scan_sums, int* scan_total, float4* out_pts,
                  float4* out_nrm, hipStream_t s);
void launch_icp_step(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                     int max_nm, hipStream_t s, const CtlParams* init_C = nullptr, const InitInline* init_inl = nullptr);
// search pass of a batch (no deferred-search queue) with every wave's neighbourhood of the fixed cloud staged in LDS
void launch_icp_step_tile(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                          int max_nm, int cap, hipStream_t s);
// iterations >= 1 of a compute(): the converged pass, ppt moving points per thread, + the deferred-search kernel if S.queue
void launch_icp_step_fast(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                          int max_nm, int ppt, bool gather, hipStream_t s);
// correspondence records of a nearest-neighbour slice, derived on demand from the state the last pass left behind
void launch_icp_outputs(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, const ProblemState* states, int K,
                        int max_nm, hipStream_t s);
int icp_step_blocks(int max_nm);
int icp_queue_blocks(int max_nm, int K);
// projective slices: z-buffer reset + z-buffer kernel + step kernel (one ICP iteration of the slice)
void launch_proj_step_pack(const SliceDev* slices, const ProblemDev* const* probs, int nslices, ProblemState* states, int K,
                           int max_nm, hipStream_t s);
// ... sharing clouds and finder parameters: one z-buffer pass and one step launch for all of them
void launch_proj_records(const SliceDev& S0, const SliceDev& S, const ProblemDev* probs0, const ProblemDev* probs,
                         ProblemState* states, int K, int max_nm, bool rebuild, hipStream_t s);
void launch_proj_step_fused(const SliceDev* slices, const ProblemDev* const* probs, int nslices, ProblemState* states, int K,
                            int max_nm, hipStream_t s,
                            const CtlParams* init_C = nullptr, const InitInline* init_inl = nullptr, ProblemDev* probs_base = nullptr);
void launch_corr_step(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                      int max_ncorr, hipStream_t s);
void launch_proj_step(bool repro, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K, int max_nm,
                      hipStream_t s);
void launch_icp_init(const CtlParams& C, const ProblemDev* probs_host, ProblemDev* probs, ProblemState* states,
                     const float* guesses_host, int tsize, hipStream_t s);
// what k_icp_init gets of a single alignment in its arguments (guess, problem table); false: a batch (read from pinned memory)
bool make_init_inline(const CtlParams& C, const ProblemDev* probs_host, const float* guesses_host, int tsize, InitInline* inl);
void launch_icp_control(const CtlParams& C, ProblemState* states, srrg2_iteration_stats* stats, hipStream_t s);
// the last control step + post + finalize of a pack of projective slices on one wave (K == 1; the control parameters come from
// their device copy, FusedCtl::ctl)
void launch_icp_final_wave_pack(const SliceDev* slices, const ProblemDev* const* probs, int nslices, ProblemState* states,
                                srrg2_iteration_stats* stats, ProblemOut* outs_host, srrg2_iteration_stats* stats_host,
                                bool with_post, hipStream_t s);
void launch_icp_small(int dim, bool plane, const SliceDev& S, const CtlParams& C, const ProblemDev* probs, ProblemState* states,
                      srrg2_iteration_stats* stats, ProblemOut* outs_host, srrg2_iteration_stats* stats_host, hipStream_t s);
// the last control step of a compute() with fused control steps (one nearest-neighbour cue slice): one wave per problem
void launch_icp_final_wave(const CtlParams& C, const SliceDev& S, ProblemState* states, srrg2_iteration_stats* stats,
                           ProblemOut* outs_host, srrg2_iteration_stats* stats_host, bool with_post, hipStream_t s);
void launch_icp_control_final(const CtlParams& C, ProblemState* states, srrg2_iteration_stats* stats, ProblemOut* outs_host,
                               srrg2_iteration_stats* stats_host, bool with_post, hipStream_t s);
void launch_icp_post(const CtlParams& C, ProblemState* states, const srrg2_iteration_stats* stats, hipStream_t s);
void launch_icp_finalize(const CtlParams& C, ProblemState* states, const srrg2_iteration_stats* stats,
                         ProblemOut* outs_host, srrg2_iteration_stats* stats_host, bool with_post, hipStream_t s);

}  // namespace srrg2amd
