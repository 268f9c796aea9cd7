// aligner_host.hip -- host driver + C ABI of the aligner (include/srrg2_slam_amd.h).
//
// The host side owns configuration, HBM buffers and the launch sequence; every per-iteration
// decision of MultiAlignerBase_::compute() (multi_aligner_impl.cpp:47-128) is taken on the device
// by the control kernels, so one compute() is one stream of launches followed by ONE small
// device->host copy.  There is no CPU fallback: without a HIP device every entry point that needs
// one fails with SRRG2_E_NO_DEVICE / SRRG2_E_HIP.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "det_math.h"
#include "kernels.h"

#include "host_util.h"

namespace srrg2amd {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace srrg2amd

namespace {

using srrg2amd::DevBuf;
using srrg2amd::fail;

struct Slice {
  srrg2_slice_config cfg;
  // fixed cloud + search grid
  DevBuf<float4> fixed_raw, fixed_nrm_raw;   // ingest order
  DevBuf<float4> fixed_sorted, fixed_nrm_sorted;
  DevBuf<int> cell_start, cursor, scan_sums, pos_of;
  DevBuf<unsigned> scalars;  // [0..2] bbox min keys, [3..5] bbox max keys, [6] nvalid, [7] ninf bits, [8] scan total
  GridDev grid{};
  // cell neighbour lists of the grid (GridDev::list_*), built by the first compute() that wants them after a set_fixed
  DevBuf<int> list_start, list_sums;
  DevBuf<uint4> list_ent;
  DevBuf<uint2> list_box;  // scratch of the build: tight box per chunk of 16 points, by the position of its first point
  DevBuf<int4> list_offs;
  DevBuf<GridLists> list_hdr;  // the GridLists record the grid points to
  GridLists lists_host{};      // ... and its host copy (the search-pass kernel takes it by value)
  bool lists_tried = false;  // built, or found not to pay / fit (grid.list_R says which)
  int grid_computes = 0;     // compute() calls on this grid so far (a grid searched a second time gets its lists)
  int nf          = 0;
  float probe_h = 0.f, probe_ext = 0.f, probe_gate = 0.f, probe_target = 0.f;  // last automatic cell size and the cloud it was probed on
  int probe_n   = 0;
  bool has_fixed  = false;
  bool fixed_has_normals = false;
  int alias_of = -1;  // this slice reads the clouds of that slice (srrg2_aligner_share_clouds); -1: its own
  // moving cloud(s)
  DevBuf<float4> moving, moving_nrm;          // Morton-sorted per problem, .w = caller's index
  DevBuf<float4> moving_raw, moving_nrm_raw;  // ingest order
  DevBuf<int> ms_counts, ms_cursor, ms_sums;
  DevBuf<unsigned> ms_bb;
  DevBuf<ProblemDev> ms_probs;
  DevBuf<unsigned> pinf;  // per problem
  int nm_total            = 0;
  bool has_moving         = false;
  bool moving_is_batch    = false;  // the bound moving cloud is the concatenation of a compute_batch with K > 1
  bool moving_has_normals = false;
  // outputs
  DevBuf<int> corr_fixed;
  // given correspondences (SRRG2_FINDER_CORRESPONDENCES)
  DevBuf<srrg2_correspondence> gcorr;
  DevBuf<int> gcorr_off;
  DevBuf<uint8_t> gcorr_stat;
  std::vector<srrg2_correspondence> h_gcorr;
  std::vector<int> h_gcorr_off;  // [K + 1]; empty: none set
  int* qprobe_host = nullptr; size_t qprobe_cap = 0;  // pinned: deferred counts of the last finished iteration
  ProblemDev* ms_probs_host = nullptr; size_t ms_probs_host_cap = 0;  // pinned: problem table of the last moving batch
  bool ms_pending = false;  // a sort that reads ms_probs_host may still be in flight
  bool fast_queue_only = false;  // the queue exists for the converged pass of a batch only (k_icp_step does not use it)
  DevBuf<float4> prev_f;
  DevBuf<float> prev_m;
  DevBuf<unsigned long long> dbg;   // SRRG2_AMD_TIMELINE (debug builds): per-wave stamps of the step kernel
  DevBuf<float4> prev_n;            // normal of the previous nearest neighbour per moving point
  DevBuf<int> prev_pos;             // its position in the sorted fixed cloud (batches gather through it)
  DevBuf<float> corr_resp;
  DevBuf<uint8_t> corr_stat;
  DevBuf<long long> partials;
  // the slot sets as the handle's last compute() left them: `slots_zeroed` of them, from `slots_zeroed_at` on, are zero
  // (k_icp_final_wave zeroes the sets of its problems; run_compute clears this before it launches anything)
  const void* slots_zeroed_at = nullptr;
  int slots_zeroed = 0;
  // set_fixed's bounding box: written into pinned memory by the last block of k_ingest_bbox, word 8 = the sequence number polled for
  unsigned* bbox_host = nullptr; size_t bbox_host_cap = 0;
  unsigned bbox_seq = 0;
  bool bbox_polled = false;        // the last k_ingest_bbox writes its result into bbox_host
  bool scalars_self_init = false;  // the last set_fixed's k_ingest_bbox left the scalars as the next one needs them
  DevBuf<unsigned> bbox_rows;  // [INGEST_BBOX_MAX_BLOCKS][8]: the blocks' partial results of k_ingest_bbox
  DevBuf<unsigned long long> zbuf;  // projective finder: [problem][rows*cols]
  DevBuf<int> queue;                // deferred searches: one 32-byte QEntry per moving point
  DevBuf<int> qcount;               // [problem]
  // prior
  float prior_Z[12]{};
  bool has_prior = false;
  void release() {
    if (qprobe_host) (void) hipHostFree(qprobe_host);
    qprobe_host = nullptr;
    qprobe_cap  = 0;
    if (bbox_host) (void) hipHostFree(bbox_host);
    bbox_host     = nullptr;
    bbox_host_cap = 0;
    if (ms_probs_host) (void) hipHostFree(ms_probs_host);
    ms_probs_host     = nullptr;
    ms_probs_host_cap = 0;
    fixed_raw.release(); fixed_nrm_raw.release(); fixed_sorted.release(); fixed_nrm_sorted.release();
    cell_start.release(); cursor.release(); scan_sums.release(); scalars.release(); pos_of.release();
    list_start.release(); list_sums.release(); list_ent.release(); list_box.release(); list_offs.release(); list_hdr.release();
    moving.release(); moving_nrm.release(); pinf.release();
    moving_raw.release(); moving_nrm_raw.release(); ms_counts.release(); ms_cursor.release(); ms_sums.release();
    ms_bb.release(); ms_probs.release();
    corr_fixed.release(); gcorr.release(); gcorr_off.release(); gcorr_stat.release(); prev_n.release(); prev_pos.release(); prev_f.release(); prev_m.release(); corr_resp.release(); corr_stat.release(); partials.release(); zbuf.release(); queue.release(); qcount.release(); bbox_rows.release();
  }
};

}  // namespace

struct srrg2_aligner_s {
  int kind = 0, dim = 3, dof = 6, tsize = 12, device = 0;
  hipStream_t stream = nullptr;
  // Batches run as up to MAX_PARTS sub-batches on as many streams (run_compute, "pipelined"): while one part's control
  // step -- one workgroup per alignment, ~10 us of an otherwise idle chip per iteration -- and kernel boundary pass, the
  // other parts' pass kernels have the chip.  The parts share nothing but read-only data (fixed cloud, grid, lists).
  static constexpr int MAX_PARTS = 8;
  hipStream_t pstream[MAX_PARTS]{};  // pstream[0] == stream
  int cu_count          = 256;       // compute units of the device (FusedCtl::first_round)
  int fused_grid_max    = 130000;    // SRRG2_AMD_FUSED_GRID_MAX (read at create): largest cloud whose grid search passes drop the
                                     // deferred-search queue for the fused kernel (run_compute)
  hipEvent_t ev_staged  = nullptr;   // the staging copies of a batch upload (stream) before the other parts' sorts
  hipEvent_t ev_records = nullptr;   // behind the kernels that materialise the correspondence records (aligner_slice_view)
  bool records_unsynced = false;     // ... which the host has not waited for yet
  int parts_dirty       = 0;         // pstream[1 .. parts_dirty) may still be running the tail of the last pipelined batch
  int batch_parts       = 0;         // upload_moving -> run_compute: the batch at hand runs as this many parts (0: one)
  int part_begin[MAX_PARTS + 1]{};   // ... part p = alignments [part_begin[p], part_begin[p + 1])
  srrg2_aligner_params params{10, 10, 0, 0};
  // point-sharded alignment (srrg2_aligner_set_point_shard): the host side's reduction and the global point count
  srrg2_reduce_fn reduce_fn = nullptr;
  void* reduce_user         = nullptr;
  long long shard_total     = 0;
  bool has_term = false;
  srrg2_termination_params term{5, 20, 20, 20, 0.2f};
  std::vector<Slice*> slices;
  float X[12]{};
  int status = SRRG2_FAIL;
  // problems of the current batch (K = 1 for compute())
  int K = 1;
  DevBuf<ProblemDev> probs;
  DevBuf<ProblemState> states;
  DevBuf<ProblemOut> outs;
  DevBuf<srrg2_iteration_stats> stats;
  DevBuf<float> guesses;
  // fused control steps (FusedCtl, device_types.h): the records the passes read, their poll words, the device copies of the
  // control parameters (one per part of a pipelined batch)
  DevBuf<unsigned long long> pub;
  DevBuf<unsigned> pub_epoch;
  DevBuf<CtlParams> ctl_dev;
  DevBuf<char> staging;  // raw strided input staged on the device
  // pinned host mirrors
  ProblemOut* outs_host = nullptr; size_t outs_host_cap = 0;
  int seq = 0;  // sequence number of the last compute() (completion flags in outs_host)
  srrg2_iteration_stats* stats_host = nullptr; size_t stats_host_cap = 0;
  float* guesses_host = nullptr; size_t guesses_host_cap = 0;
  ProblemDev* probs_host = nullptr; size_t probs_host_cap = 0;
  int max_stats = 0;
  std::vector<srrg2_iteration_stats> last_stats;  // of problem K-1 (== the only one for compute())
  int last_ncorr[SRRG2_MAX_SLICES]{};
  float last_H[36]{};  // H of the last Gauss-Newton iteration of the last compute() (problem K - 1)
  bool computed = false;
  int last_path = 0;  // SRRG2_PATH_* of the last compute() (srrg2_aligner_last_compute_path)
  // The nearest-neighbour passes do not store correspondence records; they are derived on demand (k_icp_outputs) from
  // the state of the last compute(): 0 = the arrays are current, 1 = to be derived, 2 = lost (the clouds changed since)
  int records_state = 0;
  std::vector<SliceDev> last_sdev;
  std::vector<int> last_proj_owner;  // projective slices of the last compute(): the slice whose z-buffer held their association (-1: their own)
  std::vector<int> last_nm_max;
  // strategy knobs: defaults overridden by the SRRG2_AMD_* environment ONCE, at create; srrg2_aligner_set_tuning replaces them
  srrg2_aligner_tuning tuning{};
  std::string timeline_path;  // SRRG2_AMD_TIMELINE (-DSRRG2_TIMELINE builds), read at create
  bool hosttime = false;      // SRRG2_AMD_HOSTTIME, read at create
  // profiling
  bool profile = false;
  double prof_ms = 0.0;
  int64_t prof_launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
};

namespace {

using srrg2_aligner = srrg2_aligner_s;

void identity(int kind, float* T) {
  if (kind == SRRG2_SE2_RIGHT) {
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(T, I, sizeof(I));
  } else {
    const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::memcpy(T, I, sizeof(I));
  }
}

int set_device(srrg2_aligner* a) {
  HIP_TRY(hipSetDevice(a->device));
  return 0;
}

// Everything outside a pipelined batch runs on `stream` alone: before it touches state the second half of the last pipelined
// batch wrote, that half's stream must have retired (its results were seen by the host long ago: the wait is a formality).
int quiesce_stream2(srrg2_aligner* a) {
  const int n    = a->parts_dirty;
  a->parts_dirty = 0;
  for (int p = 1; p < n; ++p) HIP_TRY(hipStreamSynchronize(a->pstream[p]));
  return 0;
}

template <typename T>
int ensure_pinned(T*& p, size_t& cap, size_t n) {
  if (n <= cap) return 0;
  if (p) (void) hipHostFree(p);
  p = nullptr;
  cap = 0;
  HIP_TRY(hipHostMalloc((void**) &p, (n + 16) * sizeof(T), hipHostMallocDefault));
  cap = n + 16;
  return 0;
}

// stage `n` records of `dim` floats, `stride_bytes` apart, from host or device memory; returns a device
// pointer to floats and the stride in floats
int stage_input(srrg2_aligner* a, const float* src, int stride_bytes, int n, int dim, int mem, const float** dev_ptr,
                int* stride_floats, size_t staging_offset) {
  if (n <= 0) {
    *dev_ptr       = (const float*) a->staging.p;
    *stride_floats = dim;
    return 0;
  }
  if (stride_bytes % 4 != 0 || stride_bytes < dim * 4) return fail(SRRG2_E_INVALID, "stride must be a multiple of 4 and >= dim*4");
  if (mem == SRRG2_MEM_DEVICE || mem == SRRG2_MEM_DEVICE_KEPT) {
    *dev_ptr       = src;
    *stride_floats = stride_bytes / 4;
    return 0;
  }
  if (mem != SRRG2_MEM_HOST) return fail(SRRG2_E_INVALID, "bad mem kind");
  char* dst = a->staging.p + staging_offset;
  if (stride_bytes == dim * 4) {
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t) n * dim * 4, hipMemcpyHostToDevice, a->stream));
  } else {
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t) dim * 4, src, (size_t) stride_bytes, (size_t) dim * 4, (size_t) n,
                             hipMemcpyHostToDevice, a->stream));
  }
  *dev_ptr       = (const float*) dst;
  *stride_floats = dim;
  return 0;
}

float key_to_float(unsigned k) {
  unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

float bound2_of_host(int r, float h) {
  float b = ((float) r - 0.01f) * h;
  return (b * b) * 0.9999f;
}

// (re)build the finder's search grid of one slice; called from set_fixed
// force_h > 0: this cell size (grown until the grid fits) instead of the automatic one (ensure_lists: a grid coarse enough for
// the neighbour lists)
// (have_bbox: set_fixed's ingest has left the bounding box and the count of valid points in the scalars already -- k_ingest_bbox)
int build_grid(srrg2_aligner* a, Slice* s, float force_h = 0.f, bool have_bbox = false) {
  const int n = s->nf;
  int rc;
  if ((rc = s->scalars.reserve(16))) return rc;
  if (!have_bbox) {
    unsigned init[16];
    for (int i = 0; i < 3; ++i) { init[i] = 0xffffffffu; init[3 + i] = 0u; }
    init[6] = 0; init[7] = 0; init[8] = 0;
    for (int i = 9; i < 16; ++i) init[i] = 0;
    // ninf (init[7]) is accumulated by the normal ingest which already ran: keep it
    HIP_TRY(hipMemcpyAsync(s->scalars.p, init, 7 * sizeof(unsigned), hipMemcpyHostToDevice, a->stream));
    srrg2amd::launch_bbox(s->fixed_raw.p, n, s->scalars.p, s->scalars.p + 3, (int*) (s->scalars.p + 6), a->stream);
  }
  unsigned back[8];
  bool polled = false;
  if (have_bbox && s->bbox_host && s->bbox_polled) {  // (set_fixed: the last block of k_ingest_bbox has left box and count in pinned memory)
    volatile unsigned* flag = s->bbox_host + 8;
    int spins = 0;
    polled    = true;
    while (*flag != s->bbox_seq)
      if ((++spins & 4095) == 0 && hipStreamQuery(a->stream) != hipErrorNotReady) {
        polled = *flag == s->bbox_seq;  // (drained: the word has just arrived, or the launch failed -- the copy below says which)
        break;
      }
    if (polled)
      for (int i = 0; i < 7; ++i) back[i] = s->bbox_host[i];
  }
  if (!polled) {
    HIP_TRY(hipMemcpyAsync(back, s->scalars.p, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));
  }
  const int nvalid = (int) back[6];
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  if (nvalid > 0) {
    for (int d = 0; d < 3; ++d) {
      mn[d] = key_to_float(back[d]);
      mx[d] = key_to_float(back[3 + d]);
    }
  }
  const float gate = s->cfg.finder_max_distance;
  const int dim    = a->dim;
  // grow h until the dense grid fits: <= 2048 cells per axis and <= max(4 Mi, min(16 n, 128 Mi)) cells (big clouds get
  // big grids: a 5 M-point cloud at 4 Mi cells had ~50 points per cell and ran 30x slower than at 80 Mi cells)
  const double cell_cap = std::max(4194304.0, std::min(16.0 * (double) n, 134217728.0));
  auto fit_cell = [&](float h0) {
    float h = h0 > 0.f ? h0 : 1.f;
    for (;;) {
      double cells = 1.0;
      bool ok      = true;
      for (int d = 0; d < dim; ++d) {
        double nd = std::floor(((double) mx[d] - (double) mn[d]) / (double) h) + 1.0;
        if (nd > 2048.0) ok = false;
        cells *= nd;
      }
      if (ok && cells <= cell_cap) return h;
      h *= 1.25f;
    }
  };
  auto grid_dims = [&](float h, GridDev& g) {
    g.ox = mn[0]; g.oy = mn[1]; g.oz = dim == 3 ? mn[2] : 0.f;
    g.h     = h;
    g.inv_h = 1.0f / h;
    auto ccoord = [&](float x, float o) {
      float u = (x - o) * g.inv_h;
      u       = std::fmin(std::fmax(u, -2048.f), 4096.f);
      return (int) std::floor(u);
    };
    g.nx = nvalid > 0 ? ccoord(mx[0], g.ox) + 1 : 1;
    g.ny = nvalid > 0 ? ccoord(mx[1], g.oy) + 1 : 1;
    g.nz = (nvalid > 0 && dim == 3) ? ccoord(mx[2], g.oz) + 1 : 1;
  };
  float h = fit_cell(force_h > 0.f ? force_h : (s->cfg.finder_cell_size > 0.f ? s->cfg.finder_cell_size : gate * 0.25f));
  if (!(force_h > 0.f) && !(s->cfg.finder_cell_size > 0.f) && nvalid > 0) {
    // Automatic cell size: probe the density with a histogram at gate/4 and rescale so that an occupied cell holds
    // ~8 points on average (surface-like scaling: occupancy ~ h^2).  Too small a cell leaves many lanes unsettled
    // after the 3^DIM block (misaligned first iterations, sparser regions), too large a cell inflates the candidate
    // lists.  Measured on C2/C4: optimum at 8 (4: -5% / -17%, 16: -5% / -8%); it was 4 before converged iterations
    // learnt to skip their searches.  Any h gives the same exact results.
    // (a tracker sets a similar cloud every frame: when the point count, the extent and the gate are within 10% of
    // the last probed cloud, its cell size is reused and the probe with its host round trip is skipped)
    float ext = 0.f;
    for (int d = 0; d < dim; ++d) ext = std::fmax(ext, mx[d] - mn[d]);
    const bool similar = s->probe_h > 0.f && s->probe_gate == gate && std::fabs((float) nvalid - (float) s->probe_n) <= 0.1f * (float) s->probe_n &&
                         std::fabs(ext - s->probe_ext) <= 0.1f * s->probe_ext && s->probe_target == a->tuning.cell_target;
    if (similar) {
      h = fit_cell(s->probe_h);
    } else {
    GridDev probe{};
    grid_dims(h, probe);
    const int pcell = probe.nx * probe.ny * probe.nz;
    if ((rc = s->cell_start.reserve((size_t) pcell + 1))) return rc;
    HIP_TRY(hipMemsetAsync(s->cell_start.p, 0, ((size_t) pcell + 1) * sizeof(int), a->stream));
    HIP_TRY(hipMemsetAsync(s->scalars.p + 9, 0, sizeof(unsigned), a->stream));
    srrg2amd::launch_grid_count(probe, s->fixed_raw.p, n, s->cell_start.p, a->stream);
    srrg2amd::launch_count_nonzero(s->cell_start.p, pcell, (int*) (s->scalars.p + 9), a->stream);
    unsigned nocc = 0;
    HIP_TRY(hipMemcpyAsync(&nocc, s->scalars.p + 9, sizeof(unsigned), hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));
    if (nocc > 0) {
      const float occupancy = (float) nvalid / (float) nocc;
      const float target    = a->tuning.cell_target > 0.f ? a->tuning.cell_target : 8.0f;
      float scale           = std::sqrt(target / occupancy);
      scale                 = std::fmin(std::fmax(scale, 0.5f), 4.0f);
      h                     = std::fmin(h * scale, gate);
      // 2-D scans are points along lines: their occupancy grows like h, not h^2, and the density rule above shrinks
      // the cells until the gate spans 10-20 of them, so that every misaligned point becomes a far (whole-wave)
      // search.  Keep the cube that covers the extended gate at radius <= 3 there (measured, tools/bench_small.py:
      // 2000 beams 0.48 -> 0.26 ms, 4000 beams 0.67 -> 0.39 ms per compute(); 3-D clouds are better off with the
      // density rule: 1 M points 4.5 k it/s without a cap, 4.0 / 3.0 / 2.8 k with radius <= 3 / 4 / 6).
      float rcap = dim == 2 ? 3.f : 0.f;
      if (a->tuning.rmax_cap > 0.f) rcap = a->tuning.rmax_cap;
      if (rcap > 1.f) h = std::fmax(h, gate * 1.25f / (rcap - 0.011f));
      h                     = fit_cell(h);
      s->probe_h = h; s->probe_n = nvalid; s->probe_ext = ext; s->probe_gate = gate; s->probe_target = a->tuning.cell_target;
    }
    }
  }
  GridDev& g = s->grid;
  g.list_R         = 0;      // (the neighbour lists belong to the previous grid)
  s->lists_tried   = false;
  s->grid_computes = 0;
  grid_dims(h, g);
  g.gate2     = gate * gate;
  g.gate2_ext = (gate * 1.25f) * (gate * 1.25f);
  g.n         = n;
  int r       = 1;
  while (bound2_of_host(r, h) < g.gate2_ext && r < 4096) ++r;
  g.rmax          = r;
  int rg          = 1;
  while (rg < r && bound2_of_host(rg, h) < g.gate2) ++rg;
  g.rfar_gate     = rg;
  const int ncell = g.nx * g.ny * g.nz;
  if ((rc = s->cell_start.reserve((size_t) ncell + 1))) return rc;
  if ((rc = s->cursor.reserve((size_t) ncell + 1))) return rc;
  if ((rc = s->scan_sums.reserve((size_t) srrg2amd::scan_num_blocks(ncell) + 1))) return rc;
  if ((rc = s->fixed_sorted.reserve((size_t) std::max(n, 1) + 8))) return rc;  // scan_range over-reads <= 3 entries
  if (s->fixed_has_normals && (rc = s->fixed_nrm_sorted.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->pos_of.reserve((size_t) std::max(n, 1)))) return rc;
  HIP_TRY(hipMemsetAsync(s->cell_start.p, 0, ((size_t) ncell + 1) * sizeof(int), a->stream));
  srrg2amd::launch_grid_count(g, s->fixed_raw.p, n, s->cell_start.p, a->stream);
  // (the scan's last phase leaves the scatter's cursors beside the cell starts)
  srrg2amd::launch_exclusive_scan(s->cell_start.p, ncell, s->scan_sums.p, (int*) (s->scalars.p + 8), a->stream, s->cursor.p);
  srrg2amd::launch_grid_scatter(g, s->fixed_raw.p, s->fixed_has_normals ? s->fixed_nrm_raw.p : nullptr, n, s->cursor.p,
                                s->fixed_sorted.p, s->fixed_has_normals ? s->fixed_nrm_sorted.p : nullptr, s->pos_of.p, a->stream);
  g.cell_start = s->cell_start.p;
  g.pts        = s->fixed_sorted.p;
  g.nrm        = s->fixed_has_normals ? s->fixed_nrm_sorted.p : nullptr;
  g.pos_of     = s->pos_of.p;
  HIP_TRY(hipGetLastError());
  return 0;
}

// Cell neighbour lists of a slice's grid (k_icp_step_cnl; GridDev::list_*): once per grid, on the first compute() that wants
// them.  Leaves grid.list_R = 0 when the lists are not available: a cube radius above CNL_MAX_R (dense clouds: the lists
// would hold thousands of cells each) or more entries than `max_entries`.
int ensure_lists_of_grid(srrg2_aligner* a, Slice* s, long long max_entries);
int ensure_lists(srrg2_aligner* a, Slice* s, long long max_entries) {
  if (s->lists_tried) return 0;
  s->lists_tried = true;
  GridDev& g     = s->grid;
  g.list_R       = 0;
  int rc;
  // A dense cloud gets cells so small (~8 points each) that the extended gate spans more than CNL_MAX_R of them: no lists, every
  // compute() on the grid kernels (200 k points: 0.34 ms where 100 k take 0.18).  The grid kernels do like the fine grid --
  // but this is the SECOND compute() on the cloud (or a batch), and with lists the coarser grid wins by far: 200 k points
  // 0.34 -> 0.23 ms, 400 k 0.63 -> 0.36 ms (profiles/r7h).  The grid is rebuilt once with the smallest cells that keep the
  // radius at CNL_MAX_R; if the lists do not fit even then, the fine grid comes back.  (Any grid gives the same results.)
  if (g.rmax > CNL_MAX_R && s->nf > 0 && s->nf <= 2000000 && !(s->cfg.finder_cell_size > 0.f) &&
      !(a->tuning.strategy_mask & (1 << 28))) {
    const float h_fine = g.h;
    const int computes = s->grid_computes;
    const float h_list = s->cfg.finder_max_distance * 1.25f / ((float) CNL_MAX_R - 0.011f);
    if (h_list > h_fine) {
      if ((rc = build_grid(a, s, h_list))) return rc;
      s->lists_tried   = true;
      s->grid_computes = computes;
      if (g.rmax <= CNL_MAX_R && (rc = ensure_lists_of_grid(a, s, max_entries))) return rc;
      if (g.list_R == 0) {  // (no lists after all: back to the grid the grid kernels prefer)
        if ((rc = build_grid(a, s, h_fine))) return rc;
        s->lists_tried   = true;
        s->grid_computes = computes;
        // (complete before anybody reads the grid: the other parts of a pipelined batch run on other streams)
        HIP_TRY(hipStreamSynchronize(a->stream));
      }
      return 0;
    }
  }
  return ensure_lists_of_grid(a, s, max_entries);
}
int ensure_lists_of_grid(srrg2_aligner* a, Slice* s, long long max_entries) {
  GridDev& g     = s->grid;
  g.list_R       = 0;
  const int R    = g.rmax;
  if (R < 1 || R > CNL_MAX_R || s->nf <= 0) return 0;
  const int dim = a->dim;
  // the offsets whose cells can hold a point within the extended gate of a query of the centre cell, by (class, centre
  // distance); class m = sum_i max(|d_i| - 1, 0)^2 = squared separation of the two cells in cells
  auto class_bound2 = [&](int m) {
    if (m <= 0) return -1.0f;
    const float b = (std::sqrt((float) m) - 0.01f) * g.h;
    return (b * b) * 0.9999f;
  };
  struct Off { int x, y, z, cls, c2; };
  std::vector<Off> offs;
  const int Rz = dim == 3 ? R : 0;
  for (int z = -Rz; z <= Rz; ++z)
    for (int y = -R; y <= R; ++y)
      for (int x = -R; x <= R; ++x) {
        auto sep = [](int d) { const int v = std::abs(d) - 1; return v > 0 ? v * v : 0; };
        const int m = sep(x) + sep(y) + sep(z);
        if (m > CNL_MAX_CLASS || class_bound2(m) > g.gate2_ext) continue;
        offs.push_back(Off{x, y, z, m, x * x + y * y + z * z});
      }
  // (the entry counts are scanned in 32 bits.  A fixed cell of c points puts ceil(c / 16) entries into at most
  // |offs| lists, so the total is below |offs| (nf + nf / 16): refuse the lists where that bound does not fit an int --
  // a wrapped sum could land back inside [0, max_entries] and under-allocate the entry array)
  if ((long long) offs.size() * ((long long) s->nf + s->nf / CNL_ENTRY_MAX + 1) >= (1LL << 31)) return 0;
  std::stable_sort(offs.begin(), offs.end(), [](const Off& p, const Off& q) { return p.cls != q.cls ? p.cls < q.cls : p.c2 < q.c2; });
  std::vector<int4> offs4(offs.size());
  for (size_t k = 0; k < offs.size(); ++k) offs4[k] = make_int4(offs[k].x, offs[k].y, offs[k].z, offs[k].cls);
  GridLists L{};
  L.R   = R;
  L.lnx = g.nx + 2 * R;
  L.lny = g.ny + 2 * R;
  L.lnz = dim == 3 ? g.nz + 2 * R : 1;
  const long long ncell_ll = (long long) L.lnx * L.lny * L.lnz;
  if (ncell_ll > (1LL << 28)) return 0;
  const int ncell = (int) ncell_ll;
  for (int m = 0; m <= CNL_MAX_CLASS; ++m) L.cls_b2[m] = class_bound2(m);
  L.cls_b2[CNL_MAX_CLASS + 1] = 3.0e38f;
  int rc;
  if ((rc = s->list_offs.reserve(offs4.size()))) return rc;
  if ((rc = s->list_start.reserve((size_t) ncell + 1))) return rc;
  if ((rc = s->list_sums.reserve((size_t) srrg2amd::scan_num_blocks(ncell) + 2))) return rc;
  if ((rc = s->list_hdr.reserve(1))) return rc;
  HIP_TRY(hipMemcpyAsync(s->list_offs.p, offs4.data(), offs4.size() * sizeof(int4), hipMemcpyHostToDevice, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));  // (offs4 is a stack-lifetime host buffer)
  srrg2amd::launch_cnl_build(g, L, s->list_offs.p, (int) offs4.size(), s->list_start.p, nullptr, nullptr, a->stream);
  int* total_dev = s->list_sums.p + s->list_sums.cap - 1;
  srrg2amd::launch_exclusive_scan(s->list_start.p, ncell, s->list_sums.p, total_dev, a->stream);
  int total = 0;
  HIP_TRY(hipMemcpyAsync(&total, total_dev, sizeof(int), hipMemcpyDeviceToHost, a->stream));
  HIP_TRY(hipStreamSynchronize(a->stream));
  if (total < 0 || (long long) total > max_entries) return 0;
  // (the lists are an optional accelerator -- up to 1.5 GB of them, allocated implicitly by the second compute() on a cloud:
  // a device that cannot hold them keeps the grid kernels instead of failing compute())
  if (s->list_ent.reserve((size_t) std::max(total, 1) + 8) || s->list_box.reserve((size_t) std::max(s->nf, 1) + 8)) {
    (void) hipGetLastError();
    s->list_ent.release();
    s->list_box.release();
    return 0;
  }
  L.start = s->list_start.p;
  L.ent   = s->list_ent.p;
  srrg2amd::launch_cnl_build(g, L, s->list_offs.p, (int) offs4.size(), s->list_start.p, s->list_box.p, s->list_ent.p, a->stream);
  HIP_TRY(hipMemcpyAsync(s->list_hdr.p, &L, sizeof(L), hipMemcpyHostToDevice, a->stream));
  HIP_TRY(hipGetLastError());
  // (complete before anybody reads them: the second half of a pipelined batch runs on another stream; L is a local)
  HIP_TRY(hipStreamSynchronize(a->stream));
  g.lists       = s->list_hdr.p;
  g.list_R      = R;
  s->lists_host = L;
  return 0;
}

// defaults overridden by the SRRG2_AMD_* variables; called once per handle, from srrg2_aligner_create
void tuning_from_environment(srrg2_aligner_tuning* t) {
  srrg2_aligner_default_tuning(t);
  auto geti = [](const char* name, int32_t& v) {
    if (const char* e = std::getenv(name)) v = (int32_t) std::atoi(e);
  };
  auto getf = [](const char* name, float& v) {
    if (const char* e = std::getenv(name)) v = (float) std::atof(e);
  };
  geti("SRRG2_AMD_TUNE", t->strategy_mask);
  geti("SRRG2_AMD_QPROBE", t->queue_probe_iteration);
  geti("SRRG2_AMD_SMALL_MAX", t->small_max_points);
  geti("SRRG2_AMD_FAST_FROM", t->fast_from_iteration);
  geti("SRRG2_AMD_FAST_PPT", t->fast_points_per_thread);
  geti("SRRG2_AMD_FAST_MIN", t->fast_min_points);
  geti("SRRG2_AMD_FAST_GATHER", t->fast_gather);
  geti("SRRG2_AMD_FAST_QUEUE", t->fast_batch_queue);
  geti("SRRG2_AMD_QUEUE_MIN", t->queue_min_points);
  geti("SRRG2_AMD_MSORT_SEGMENTS", t->msort_segments);
  geti("SRRG2_AMD_MSORT_BITS", t->msort_key_bits);
  geti("SRRG2_AMD_LDS_TILE", t->lds_tile);
  geti("SRRG2_AMD_SEARCH_LISTS", t->search_lists);
  geti("SRRG2_AMD_SEARCH_TEAM", t->search_team);
  geti("SRRG2_AMD_BATCH_PIPELINE", t->batch_pipeline);
  geti("SRRG2_AMD_FUSED_CONTROL", t->fused_control);
  getf("SRRG2_AMD_CELL_TARGET", t->cell_target);
  getf("SRRG2_AMD_RMAX_CAP", t->rmax_cap);
  // (the environment bypasses srrg2_aligner_set_tuning's range check: clamp to what that check accepts)
  if (t->fast_points_per_thread < 0) t->fast_points_per_thread = 0;
  if (t->fast_from_iteration < 1) t->fast_from_iteration = 1;  // (iteration 0 has no previous neighbours to certify)
  if (t->msort_key_bits > 18) t->msort_key_bits = 18;
  if (t->msort_key_bits < -1) t->msort_key_bits = -1;
  if (t->msort_segments < 0) t->msort_segments = 0;
  if (t->search_team < 0) t->search_team = 0;
  if (!(t->cell_target > 0.f)) t->cell_target = 8.0f;
}

int check_slice(srrg2_aligner* a, int si, const char* what) {
  if (!a) return fail(SRRG2_E_INVALID, std::string(what) + ": null handle");
  if (si < 0 || si >= (int) a->slices.size()) return fail(SRRG2_E_INVALID, std::string(what) + ": bad slice index");
  return 0;
}

// upload the moving clouds of K problems (concatenated, offsets[K+1]) into slice `si`
// `wait`: return only when the ingest has finished reading the caller's buffer (set_moving: the caller may reuse it on
// return).  compute_batch passes false: it returns after the compute() that follows on the same stream has delivered
// its results, and the launches of that compute() overlap the sort instead of waiting behind it.
int upload_moving(srrg2_aligner* a, int si, const float* coords, int cs, const float* normals, int ns,
                  const int32_t* offsets, int K, int mem, bool wait = true, int nparts = 0, const int* pbegin = nullptr) {
  Slice* s    = a->slices[si];
  a->batch_parts = 0;
  {
    int rcq = quiesce_stream2(a);
    if (rcq) return rcq;
  }
  if (a->computed) a->records_state = 2;  // (the records of the last compute() can no longer be derived)
  const int n = offsets[K] - offsets[0];
  int rc;
  if ((rc = s->moving.reserve((size_t) std::max(n, 1)))) return rc;
  if (normals && (rc = s->moving_nrm.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->pinf.reserve((size_t) K))) return rc;
  if ((rc = s->corr_fixed.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->prev_n.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->prev_pos.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->prev_f.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->prev_m.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->corr_resp.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->corr_stat.reserve((size_t) std::max(n, 1)))) return rc;
  const size_t bytes_c = (size_t) n * a->dim * 4;
  if (mem == SRRG2_MEM_HOST && (rc = a->staging.reserve(2 * bytes_c + 64))) return rc;
  const float* base_c = (const float*) ((const char*) coords + (size_t) offsets[0] * cs);
  const float* dsrc;
  int sf;
  if ((rc = stage_input(a, base_c, cs, n, a->dim, mem, &dsrc, &sf, 0))) return rc;
  const float* nsrc = nullptr;
  int nsf = 0;
  if (normals) {
    const float* base_n = (const float*) ((const char*) normals + (size_t) offsets[0] * ns);
    if ((rc = stage_input(a, base_n, ns, n, a->dim, mem, &nsrc, &nsf, (bytes_c + 63) / 64 * 64))) return rc;
  }
  if ((rc = s->moving_raw.reserve((size_t) std::max(n, 1)))) return rc;
  if (normals && (rc = s->moving_nrm_raw.reserve((size_t) std::max(n, 1)))) return rc;
  std::vector<ProblemDev> pd((size_t) K);
  int max_nm = 0;
  for (int k = 0; k < K; ++k) {
    const int off = offsets[k] - offsets[0], cnt = offsets[k + 1] - offsets[k];
    pd[k]  = ProblemDev{off, cnt};
    max_nm = std::max(max_nm, cnt);
  }
  // Morton sort per problem.  Key space: 2^18 cells for one cloud (global histogram), 2^15 / 2^12 for batches (histogram
  // in LDS), dealt to the axes by extent (tuning.msort_key_bits: -1 = the isotropic keys of round 2, 6 / 5 / 4 bits per axis)
  const int mk    = a->tuning.msort_key_bits;
  const int aniso = mk >= 0 ? 1 : 0;
  const int bits  = K <= 4 ? 6 : (K <= 32 ? 5 : 4);
  // (automatic: 15 bits = the one-kernel sort in LDS for every call.  Round 5 kept 18 bits -- the eight launches of the
  // global-histogram sort -- for calls of up to four clouds: the passes of ONE 100 k-point cloud are ~1 % faster on the finer
  // order, but its sort is 70-100 us slower, and a caller that binds a new moving cloud per alignment -- a tracker, a
  // compute_batch of 1 .. 4 -- pays that every time: 0.28 -> 0.20 ms per single 50 k-point compute_batch, 0.32 -> 0.25 ms per
  // 4-batch, set_moving of a tracker frame 0.105 -> 0.086 ms (profiles/r6v, r6y).  msort_key_bits = 18 brings it back.)
  int kbits       = aniso ? (mk > 0 ? mk : 15) : 3 * bits;
  kbits           = std::min(kbits, 18);  // (more than 6 bits per axis would collide in the key's bit spreading)
  // (the global-histogram sort indexes K << kbits cells with an int: coarser keys when a huge batch asks for fine ones)
  while (kbits > 3 && ((long long) K << kbits) > 0x3fffffffLL) --kbits;
  // small key spaces: one workgroup per problem sorts straight from the caller's (staged) layout, with the problem
  // table read from pinned host memory (no copies, memsets or waits on the stream); the ingest-order copy of the
  // clouds, which this path does not produce, is only read by given-correspondences slices
  const bool local_sort = !(a->tuning.strategy_mask & (1 << 22));
  if (local_sort && kbits <= 15 && s->cfg.finder != SRRG2_FINDER_CORRESPONDENCES) {
    if ((rc = ensure_pinned(s->ms_probs_host, s->ms_probs_host_cap, (size_t) K))) return rc;
    if (s->ms_pending) HIP_TRY(hipStreamSynchronize(a->stream));  // (set_moving twice without a compute() in between)
    s->ms_pending = false;
    std::memcpy(s->ms_probs_host, pd.data(), (size_t) K * sizeof(ProblemDev));
    bool sorted = false;
    if (nparts >= 2 && nparts <= srrg2_aligner::MAX_PARTS && pbegin) {
      // pipelined batch: every part is sorted on its own stream (behind the staging copies of this call), the later parts
      // under the first part's first pass
      if (mem == SRRG2_MEM_HOST) {
        HIP_TRY(hipEventRecord(a->ev_staged, a->stream));
        for (int p = 1; p < nparts; ++p) HIP_TRY(hipStreamWaitEvent(a->pstream[p], a->ev_staged, 0));
      }
      sorted = true;
      for (int p = 0; p < nparts && sorted; ++p) {
        int nmp = 0;
        for (int k = pbegin[p]; k < pbegin[p + 1]; ++k) nmp = std::max(nmp, pd[k].nm);
        sorted = srrg2amd::launch_msort_local(dsrc, sf, nsrc, nsf, s->ms_probs_host + pbegin[p], pbegin[p + 1] - pbegin[p], a->dim,
                                              kbits, aniso, a->tuning.msort_segments, nmp, s->moving.p,
                                              normals ? s->moving_nrm.p : nullptr, s->pinf.p + pbegin[p], a->pstream[p]);
        if (p > 0) a->parts_dirty = std::max(a->parts_dirty, p + 1);
      }
      if (sorted) {
        a->batch_parts = nparts;
        for (int p = 0; p <= nparts; ++p) a->part_begin[p] = pbegin[p];
      }
    } else {
      sorted = srrg2amd::launch_msort_local(dsrc, sf, nsrc, nsf, s->ms_probs_host, K, a->dim, kbits, aniso, a->tuning.msort_segments,
                                            max_nm, s->moving.p, normals ? s->moving_nrm.p : nullptr, s->pinf.p, a->stream);
    }
    if (sorted) {
      HIP_TRY(hipGetLastError());
      // the caller may reuse its buffer on return (host or device memory: the ingest has finished reading it)
      if (wait) {
        HIP_TRY(hipStreamSynchronize(a->stream));
        for (int p = 1; p < a->batch_parts; ++p) HIP_TRY(hipStreamSynchronize(a->pstream[p]));
      }
      s->ms_pending         = !wait;  // (the sort reads the pinned problem table: the compute() that follows drains the stream)
      s->nm_total           = n;
      s->has_moving         = true;
      s->moving_is_batch    = K > 1;
      s->moving_has_normals = normals != nullptr;
      return 0;
    }
  }
  HIP_TRY(hipMemsetAsync(s->pinf.p, 0, (size_t) K * sizeof(unsigned), a->stream));
  const size_t ncell = (size_t) K << kbits;
  if ((rc = s->ms_counts.reserve(ncell + 1))) return rc;
  if ((rc = s->ms_cursor.reserve(ncell + 1))) return rc;
  if ((rc = s->ms_sums.reserve((size_t) srrg2amd::scan_num_blocks((int) ncell) + 2))) return rc;
  if ((rc = s->ms_bb.reserve((size_t) K * 6))) return rc;
  if ((rc = s->ms_probs.reserve((size_t) K))) return rc;
  {
    std::vector<unsigned> bb((size_t) K * 6);
    for (int k = 0; k < K; ++k)
      for (int d = 0; d < 3; ++d) {
        bb[(size_t) k * 6 + d]     = 0xffffffffu;
        bb[(size_t) k * 6 + 3 + d] = 0u;
      }
    HIP_TRY(hipMemcpyAsync(s->ms_bb.p, bb.data(), bb.size() * sizeof(unsigned), hipMemcpyHostToDevice, a->stream));
    HIP_TRY(hipMemcpyAsync(s->ms_probs.p, pd.data(), pd.size() * sizeof(ProblemDev), hipMemcpyHostToDevice, a->stream));
    HIP_TRY(hipMemsetAsync(s->ms_counts.p, 0, (ncell + 1) * sizeof(int), a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));  // bb / pd are stack-lifetime host buffers
  }
  // all clouds of the batch in one launch each (points, normals)
  srrg2amd::launch_ingest_batch(dsrc, sf, s->ms_probs.p, K, max_nm, a->dim, s->moving_raw.p, s->pinf.p, 1, a->stream);
  if (normals)
    srrg2amd::launch_ingest_batch(nsrc, nsf, s->ms_probs.p, K, max_nm, a->dim, s->moving_nrm_raw.p, nullptr, 0, a->stream);
  srrg2amd::launch_msort(s->moving_raw.p, normals ? s->moving_nrm_raw.p : nullptr, s->ms_probs.p, K, max_nm, kbits, aniso,
                         s->ms_bb.p, s->ms_counts.p, s->ms_cursor.p, s->ms_sums.p, s->ms_sums.p + s->ms_sums.cap - 1,
                         s->moving.p, normals ? s->moving_nrm.p : nullptr, a->stream);
  HIP_TRY(hipGetLastError());
  if (wait) HIP_TRY(hipStreamSynchronize(a->stream));  // the caller may reuse its buffer on return (host or device memory)
  s->nm_total           = n;
  s->has_moving         = true;
  s->moving_is_batch    = K > 1;
  s->moving_has_normals = normals != nullptr;
  return 0;
}

int run_compute(srrg2_aligner* a, int K, const int32_t* offsets /* K+1 or null for K == 1 */, const float* guesses) {
  auto t_begin = std::chrono::steady_clock::now();
  int rc;
  if ((rc = set_device(a))) return rc;
  // (a pipelined batch: upload_moving has already put the second half's sort on the second stream)
  const int split = (K > 1 && a->batch_parts >= 2 && a->part_begin[a->batch_parts] == K) ? a->batch_parts : 0;  // parts (0: one)
  a->batch_parts  = 0;
  if (!split && (rc = quiesce_stream2(a))) return rc;
  const int nslices = (int) a->slices.size();
  // sanity checks (reference: sanityCheck throws, aligner_slice_processor_impl.cpp:8-17;
  // aligner_slice_processor_prior_impl.cpp:11-22)
  int max_nm = 0;
  std::vector<ProblemDev> probs((size_t) K);
  for (int si = 0; si < nslices; ++si) {  // slices that share another slice's clouds: views of its buffers, its sizes
    Slice* s = a->slices[si];
    if (s->alias_of < 0) continue;
    if (s->alias_of >= nslices || a->slices[s->alias_of]->alias_of >= 0 || a->slices[s->alias_of]->cfg.kind == SRRG2_SLICE_PRIOR)
      return fail(SRRG2_E_STATE, "compute: a slice shares the clouds of a slice that is gone");
    Slice* o = a->slices[s->alias_of];
    if (s->cfg.finder != SRRG2_FINDER_PROJECTIVE || o->cfg.finder != SRRG2_FINDER_PROJECTIVE)
      return fail(SRRG2_E_UNSUPPORTED, "compute: shared clouds are for slices with the projective finder");
    s->fixed_raw.borrow(o->fixed_raw);   s->fixed_nrm_raw.borrow(o->fixed_nrm_raw);
    s->moving.borrow(o->moving);         s->moving_nrm.borrow(o->moving_nrm);
    s->moving_raw.borrow(o->moving_raw); s->moving_nrm_raw.borrow(o->moving_nrm_raw);
    s->pinf.borrow(o->pinf);             s->scalars.borrow(o->scalars);
    s->nf = o->nf; s->nm_total = o->nm_total; s->has_fixed = o->has_fixed; s->has_moving = o->has_moving;
    s->fixed_has_normals = o->fixed_has_normals; s->moving_has_normals = o->moving_has_normals;
    s->moving_is_batch = o->moving_is_batch;
    const size_t n = (size_t) std::max(s->nm_total, 1);
    if ((rc = s->corr_fixed.reserve(n)) || (rc = s->corr_resp.reserve(n)) || (rc = s->corr_stat.reserve(n))) return rc;
  }
  for (int si = 0; si < nslices; ++si) {
    Slice* s = a->slices[si];
    if (s->cfg.kind == SRRG2_SLICE_PRIOR) {
      if (!s->has_prior) return fail(SRRG2_E_STATE, "compute: prior slice without measurement");
      continue;
    }
    if (!s->has_fixed) return fail(SRRG2_E_STATE, "compute: cue slice| no fixed");
    if (!s->has_moving) return fail(SRRG2_E_STATE, "compute: cue slice| no moving");
    if (K == 1 && s->moving_is_batch)
      return fail(SRRG2_E_STATE, "compute: the moving cloud bound to the cue slice is the concatenation of the last "
                                 "compute_batch (K > 1); bind one with set_moving first");
    if (s->cfg.kind == SRRG2_SLICE_P2PLANE && !s->fixed_has_normals)
      return fail(SRRG2_E_STATE, "compute: point-to-plane slice needs fixed normals");
    if (s->cfg.finder == SRRG2_FINDER_PROJECTIVE && s->nf != s->cfg.image_rows * s->cfg.image_cols)
      return fail(SRRG2_E_STATE, "compute: projective finder needs an organised fixed cloud of rows x cols points");
  }
  // problems: all cue slices share the problem layout of slice 0's batch; for K == 1 each slice has its own nm
  const int slots = 2 * std::max(a->params.max_iterations, 1);
  a->max_stats    = slots;
  if ((rc = a->probs.reserve((size_t) K * std::max(nslices, 1)))) return rc;
  if ((rc = a->states.reserve((size_t) K))) return rc;
  if ((rc = a->outs.reserve((size_t) K))) return rc;
  if ((rc = a->stats.reserve((size_t) K * slots))) return rc;
  if ((rc = a->guesses.reserve((size_t) K * a->tsize))) return rc;
  if ((rc = ensure_pinned(a->outs_host, a->outs_host_cap, (size_t) K))) return rc;
  if ((rc = ensure_pinned(a->stats_host, a->stats_host_cap, (size_t) K * slots))) return rc;
  if ((rc = ensure_pinned(a->guesses_host, a->guesses_host_cap, (size_t) K * a->tsize))) return rc;
  // guesses and problem tables go through pinned host memory that k_icp_init reads directly (no copies on the stream;
  // the previous compute() has been waited for, so the buffers are free)
  std::memcpy(a->guesses_host, guesses, (size_t) K * a->tsize * sizeof(float));
  if ((rc = ensure_pinned(a->probs_host, a->probs_host_cap, (size_t) K * std::max(nslices, 1)))) return rc;
  // per-slice problem tables live back to back in a->probs: [slice][K]
  std::vector<ProblemDev> all((size_t) K * std::max(nslices, 1));
  for (int si = 0; si < nslices; ++si) {
    Slice* s = a->slices[si];
    for (int k = 0; k < K; ++k) {
      ProblemDev pd{0, 0};
      if (s->cfg.kind != SRRG2_SLICE_PRIOR) {
        if (K == 1) {
          pd.moff = 0;
          pd.nm   = s->nm_total;
        } else {
          pd.moff = offsets[k] - offsets[0];
          pd.nm   = offsets[k + 1] - offsets[k];
        }
        max_nm = std::max(max_nm, pd.nm);
      }
      all[(size_t) si * K + k] = pd;
    }
  }
  std::memcpy(a->probs_host, all.data(), all.size() * sizeof(ProblemDev));

  CtlParams C{};
  C.variable_kind = a->kind;
  C.nslices       = nslices;
  C.K             = K;
  C.params        = a->params;
  C.has_term      = a->has_term ? 1 : 0;
  C.term          = a->term;
  C.max_stats     = slots;
  C.seq           = ++a->seq;
  if (C.seq <= 0) C.seq = a->seq = 1;
  for (int k = 0; k < K; ++k) a->outs_host[k].seq = 0;  // (never a sequence number: fresh pinned memory is not zeroed)
  const srrg2_aligner_tuning& tn = a->tuning;  // (read once at create / set_tuning: no environment look-ups in compute())
  C.tune          = tn.strategy_mask;
  C.probe_it      = tn.queue_probe_iteration;
  if (a->params.max_iterations <= C.probe_it + 3 || K > 4) C.probe_it = -1;
  // Small problems (laser scans, landmark maps) with one nearest-neighbour cue slice (plus priors): one workgroup per
  // problem runs the whole compute() (k_icp_small) instead of ~24 launches of a few microseconds of work each.
  bool small = false;
  {
    int ncue = 0, fc = -1;
    for (int si = 0; si < nslices; ++si)
      if (a->slices[si]->cfg.kind != SRRG2_SLICE_PRIOR) {
        if (fc < 0) fc = si;
        ++ncue;
      }
    // (measured, tools/bench_small.py: ahead of one launch per pass up to ~1000 points -- 1000-beam scan 0.33 -> 0.23 ms,
    // 360 beams 0.18 -> 0.12 ms, 500 3-D points 0.34 -> 0.25 ms -- and behind it from ~2000 points on)
    const int small_max = tn.small_max_points;
    small = ncue == 1 && a->slices[fc]->cfg.finder == SRRG2_FINDER_NN_GATED && max_nm <= small_max &&
            a->timeline_path.empty() && !a->reduce_fn;  // (the one-workgroup kernel has no place for the reduction)
    // (round 5, late: ... and only where the launches cannot carry their control steps -- prior slices next to the cue slice, or
    // fused_control = 0.  With fused control steps and lists the launch-per-pass path has overtaken it at every size: a
    // 1000-beam scan 0.226 -> 0.134 ms, 1000 3-D points 0.42 -> 0.135 ms, 256 x 1000 points per call 0.56 -> 0.31 ms,
    // first compute() on a new fixed cloud 0.42 -> 0.26 ms; 300 points: equal there, 0.24 -> 0.12 ms afterwards; profiles/r6z)
    // ... except on a NEW fixed cloud (no lists yet: a laser tracker's every frame), where the search passes of the
    // launch-per-pass path run on the grid kernels with a control launch each: the smallest clouds stay on the one-workgroup
    // kernel there (360 beams: 0.113 against 0.124 ms, 384 3-D points 0.245 against 0.255 ms; from ~700 points on the launches
    // win: 1000 beams 0.154 against 0.232 ms, 1000 3-D points 0.26 against 0.42 ms), profiles/r7a, r7e.
    // (aligners whose control steps stay launches -- a prior slice next to the cue slice: a laser tracker with odometry --:
    // the one-workgroup kernel up to ~640 points; 360 beams 0.126 against 0.165 ms, 1000 beams 0.246 against 0.19 ms, r7f)
    // (round 6, late: the control wave linearises prior slices too -- wave_prior --, such aligners follow the rule of the cue-only
    // ones; fused_control = 2 / SRRG2_AMD_TUNE bit 24: prior + cue aligners on control launches, as before)
    const bool priors_launch = nslices > ncue && (tn.fused_control == 2 || (tn.strategy_mask & (1 << 24)) || nslices - ncue > 2);
    if (small && (priors_launch || tn.fused_control == 0 || a->params.max_iterations < 2)) small = max_nm <= 640;
    if (small && !(priors_launch || tn.fused_control == 0 || a->params.max_iterations < 2)) {
      const Slice* sfc = a->slices[fc];
      const int sl     = tn.search_lists;
      const bool lists = sl >= 2 || (sl == 1 && K > 4) || (sl < 0 && (K > 4 || sfc->grid_computes >= 1 || sfc->lists_tried));
      small            = !lists && max_nm <= 384;
    }
  }
  // The converged pass kernel takes over from iteration `fast_from` of the first run (all of the inlier-only run): by
  // then nearly every point keeps its neighbour.  Batches give it `fast_ppt` points per thread and a queue.
  const int fast_from = tn.fast_from_iteration;
  // (measured on C4, 32 x 50k, profiles/archive/r2c: one point per thread with the failed certificates searched by their own wave
  // 37.7 us per pass / 268.8 k it/s; two points per thread 45 us -- the accumulators stay live across the search, 181
  // registers -- ; with a queue the nearly idle deferred-search launch costs 15 us per iteration: 31 + 15 us, 254 k it/s)
  // (0 = automatic: two points per thread share one reduction once a launch holds 64 alignments or more -- C4-256 575 -> 595 k it/s;
  // smaller launches and single alignments lose the waves they need to fill the chip: C4-32 unchanged, C2 45.1 -> 42.6 k it/s)
  // (decided per LAUNCH: a part of a pipelined batch is its own launch)
  // (round 5, launches with fused control steps: two points per thread from 16 alignments per launch on -- C4-32 556 -> 567 k it/s,
  // C4-48 587 -> 597 k, profiles/r5s: half as many workgroups read the record and share a reduction)
  bool fuse = false;  // (decided below, before the first launch)
  auto fast_ppt_of = [&](int k_launch) {
    return tn.fast_points_per_thread > 0 ? tn.fast_points_per_thread : (k_launch >= (fuse ? 16 : 64) ? 2 : 1);
  };
  // batches gather the kept neighbour from the cache-resident fixed cloud (36 -> 8 streamed bytes per point); single
  // alignments read it from per-point arrays (no dependent load on the chain of a latency-bound launch)
  // (smallest moving cloud that uses the converged-pass kernel.  Sparse clouds of a few thousand points leave a larger
  // share of their certificates behind; since those searches run four at a time with 16 lanes each the kernel is ahead
  // at every size -- 2 000 points: 0.26 ms per compute() either way, 0.42 ms with one search per wave at a time,
  // profiles/archive/r2m_bench_small.json / r2n_bench_small*.json -- so the threshold is 0; kept as a switch)
  const int fast_min = tn.fast_min_points;
  const bool fast_gather = tn.fast_gather >= 0 ? tn.fast_gather != 0 : K > 4;
  const bool fast_batch_queue = tn.fast_batch_queue != 0;
  // search passes of batches: every wave's neighbourhood of the fixed cloud staged in LDS (k_icp_step_tile; 1: tiles of
  // 416 candidates, four workgroups per CU; 2: 504 candidates, three workgroups per CU)
  // (automatic: batches of more than four alignments -- C4-256 374 -> 406 k it/s, C4-32 305 -> 322 k, C4-8 167 -> 171 k;
  // single alignments are launch / latency bound and neutral to 1.5 % slower with it: 30 k points 41.7 -> 41.1 k it/s,
  // profiles/archive/r3i_ab_tile_default.txt)
  const int lds_tile = tn.lds_tile >= 0 ? tn.lds_tile : (K > 4 ? 1 : 0);
  // Search passes over the cell neighbour lists of the grid (k_icp_step_cnl, round 4): 0 = never, 1 = batches of more than
  // four alignments, 2 = every alignment (no deferred-search queue then).  -1 = automatic: batches always (the build --
  // two kernels, ~0.2 ms at 100 k points, two host waits -- is shared by all their alignments); single alignments from the
  // SECOND compute() on a fixed cloud on: a relocalizer or a loop detector aligns many clouds against one map and gains
  // ~15 % per compute(), a tracker sets a new fixed cloud every frame and would pay the build for one alignment
  // (tools/bench_tracker.py: compute 0.26 -> 0.54 ms per frame with the lists built every frame).
  const int search_lists = tn.search_lists;
  // lanes per moving point of that kernel: 1 = throughput, 4 = latency (a single alignment leaves the chip half empty and
  // its first pass, where every point searches without a bound, is a chain of dependent round trips: four lanes share a
  // point's headers and candidates: C2 55.7 -> 29.7 us; the passes with priors, where a fraction of the points search, are
  // better off with one lane: 14.3 vs 18.3 us on the third pass); 0 = automatic: 4 on the first pass of up to four
  // alignments per launch, 1 otherwise
  const int search_team_knob = tn.search_team;
  std::vector<char> cnl((size_t) std::max(nslices, 1), 0);
  std::vector<SliceDev> sdev((size_t) nslices);
  int first_cue = -1;
  for (int si = 0; si < nslices; ++si) {
    Slice* s     = a->slices[si];
    SliceCtl& sc = C.slices[si];
    sc.kind                     = s->cfg.kind;
    sc.min_num_correspondences  = s->cfg.min_num_correspondences;
    sc.robust_kind              = s->cfg.robustifier;
    sc.robust_thr               = s->cfg.robustifier_chi_threshold;
    sc.gate                     = s->cfg.finder_max_distance;
    sc.has_prior                = s->has_prior ? 1 : 0;
    sc.prior_sets_initial_guess = s->cfg.prior_sets_initial_guess;
    std::memcpy(sc.prior_Z, s->prior_Z, sizeof(sc.prior_Z));
    std::memcpy(sc.prior_info, s->cfg.prior_information_diag, sizeof(sc.prior_info));
    sc.partials  = nullptr;
    sc.finder    = s->cfg.finder;
    sc.K0        = s->cfg.camera_matrix[0];
    sc.K4        = s->cfg.camera_matrix[4];
    sc.rows      = s->cfg.image_rows;
    sc.cols      = s->cfg.image_cols;
    sc.depth_min = s->cfg.depth_min;
    sc.pinf_bits = nullptr;
    sc.ninf_bits = nullptr;
    if (s->cfg.kind == SRRG2_SLICE_PRIOR) continue;
    if (first_cue < 0) first_cue = si;
    int nm_max_s = 0;
    for (int k = 0; k < K; ++k) nm_max_s = std::max(nm_max_s, all[(size_t) si * K + k].nm);
    // deferred-search queue: a win in the latency regime (few alignments per launch: C2 0.78 -> 0.63 ms); with many
    // alignments per launch the in-kernel path has more throughput (C4: 2.46 vs 2.74 ms per 32 x 50k batch)
    // ... and with few points per launch the extra launches cost more than they save (measured, tools/loop_compute.py:
    // without the queue 3 k / 10 k / 30 k / 60 k points take 0.286 / 0.249 / 0.254 / 0.272 ms per compute() instead of
    // 0.307 / 0.271 / 0.264 / 0.277 ms; equal at 80-100 k; 150 k: 0.356 ms with the queue, 0.462 ms without)
    const int queue_min = tn.queue_min_points;
    const bool want_lists = search_lists >= 2 || (search_lists == 1 && K > 4) ||
                            (search_lists < 0 && (K > 4 || s->grid_computes >= 1 || s->lists_tried));
    s->grid_computes++;
    if (s->cfg.finder == SRRG2_FINDER_NN_GATED && !small && want_lists) {
      // (the lists are built once per grid; the guard: 96 Mi entries = 1.5 GB -- C2: 3 M entries, a 2 M-point cloud: ~60 M)
      if ((rc = ensure_lists(a, s, 96LL << 20))) return rc;
      cnl[(size_t) si] = s->grid.list_R > 0 ? 1 : 0;
    }
    const bool use_queue = s->cfg.finder == SRRG2_FINDER_NN_GATED && !(C.tune & 512) && nm_max_s >= queue_min && K <= 4 && !small &&
                           !cnl[(size_t) si];
    // batches: the converged pass (k_icp_step_fast) hands the points whose certificate failed to the deferred-search
    // kernel; the first iterations (k_icp_step) finish their open points themselves
    const bool fast_queue = s->cfg.finder == SRRG2_FINDER_NN_GATED && K > 4 && !small && fast_batch_queue;
    s->fast_queue_only = fast_queue && !use_queue;
    const int nblocks    = PARTIAL_SLOTS;
    {
      const size_t cap_before = s->partials.cap;
      if ((rc = s->partials.reserve((size_t) 3 * K * nblocks * ACC_N))) return rc;  // (three buffers: fused control steps, FusedCtl)
      if (s->partials.cap != cap_before) s->slots_zeroed = 0;  // (a new allocation -- possibly at the old address -- holds anything)
    }
    if (use_queue || fast_queue) {
      if ((rc = s->queue.reserve((size_t) std::max(s->nm_total, 1) * 10))) return rc;  // QEntry = 10 x 4 bytes
      if ((rc = s->qcount.reserve((size_t) 2 * K))) return rc;  // [problem][near, far]
    }
    sc.qcount = (use_queue || fast_queue) ? s->qcount.p : nullptr;
    sc.qprobe_host = nullptr;
    sc.probs       = nullptr;
    if (use_queue) {
      if ((rc = ensure_pinned(s->qprobe_host, s->qprobe_cap, (size_t) 2 * K))) return rc;
      std::memset(s->qprobe_host, 0x7f, sizeof(int) * (size_t) 2 * K);  // "large" until the device reports
      sc.qprobe_host = s->qprobe_host;
      sc.probs       = a->probs.p + (size_t) si * K;
    }
    // (k_icp_init zeroes the slot sets and the queue counters; afterwards the control kernel resets them each iteration)
    sc.partials  = s->partials.p;
    sc.pinf_bits = s->pinf.p;
    sc.ninf_bits = s->scalars.p + 7;
    sc.finf_bits = s->scalars.p + 10;
    sc.gcorr_off = nullptr;
    sc.nm_global = (a->reduce_fn && s->cfg.kind != SRRG2_SLICE_PRIOR) ? (int) a->shard_total : 0;
    SliceDev& d       = sdev[si];
    d.grid            = s->grid;
    d.mpts            = s->moving.p;
    d.mnrm            = s->moving_has_normals ? s->moving_nrm.p : nullptr;
    d.corr_fixed      = s->corr_fixed.p;
    d.corr_resp       = s->corr_resp.p;
    d.corr_stat       = s->corr_stat.p;
    d.prev_n          = s->prev_n.p;
    d.prev_pos        = s->prev_pos.p;
    d.prev_f          = s->prev_f.p;
    d.prev_m          = s->prev_m.p;
    d.dbg             = nullptr;
    if (!a->timeline_path.empty()) {
      const size_t nw = (size_t) K * ((nm_max_s + 255) / 256) * 4;
      if ((rc = s->dbg.reserve(32 * nw * 16))) return rc;
      HIP_TRY(hipMemsetAsync(s->dbg.p, 0, 32 * nw * 16 * sizeof(unsigned long long), a->stream));
      d.dbg = s->dbg.p;
    }
    d.partials        = s->partials.p;
    d.queue           = (use_queue || fast_queue) ? (void*) s->queue.p : nullptr;
    d.qcount          = (use_queue || fast_queue) ? s->qcount.p : nullptr;
    d.slice_idx       = si;
    d.robust_kind     = s->cfg.robustifier;
    d.robust_thr      = s->cfg.robustifier_chi_threshold;
    d.normal_cos      = s->cfg.finder_normal_cos;
    d.use_normal_gate = (s->cfg.finder_normal_cos > -1.f && s->fixed_has_normals && s->moving_has_normals) ? 1 : 0;
    // (batches: every pass re-reads the previous neighbour through its position in the shared, L2-resident fixed cloud; the
    // search passes then stream 48 instead of 112 bytes per point -- the algorithmic figure)
    d.gather_prev     = (s->cfg.finder == SRRG2_FINDER_NN_GATED && fast_gather) ? 1 : 0;
    d.variable_kind   = a->kind;
    d.finder          = s->cfg.finder;
    d.factor          = s->cfg.kind;
    std::memcpy(d.K, s->cfg.camera_matrix, sizeof(d.K));
    d.rows            = s->cfg.image_rows;
    d.cols            = s->cfg.image_cols;
    d.depth_min       = s->cfg.depth_min;
    d.depth_max       = s->cfg.depth_max;
    d.gate            = s->cfg.finder_max_distance;
    d.fixed_org       = s->fixed_raw.p;
    d.fixed_org_nrm   = s->fixed_has_normals ? s->fixed_nrm_raw.p : nullptr;
    d.zbuf            = nullptr;
    d.zbuf_parity     = 0;
    d.gcorr = nullptr; d.gcorr_off = nullptr; d.gcorr_stat = nullptr;
    d.moving_raw      = s->moving_raw.p;
    if (s->cfg.finder == SRRG2_FINDER_CORRESPONDENCES) {
      if ((int) s->h_gcorr_off.size() != K + 1)
        return fail(SRRG2_E_STATE, "compute: given-correspondences slice without correspondences for every alignment");
      if (s->cfg.kind == SRRG2_SLICE_P2PLANE && !s->fixed_has_normals)
        return fail(SRRG2_E_STATE, "compute: point-to-plane slice without fixed normals");
      const int total = s->h_gcorr_off[(size_t) K];
      for (int k = 0; k < K; ++k) {
        const int nmk = all[(size_t) si * K + k].nm;
        for (int c = s->h_gcorr_off[(size_t) k]; c < s->h_gcorr_off[(size_t) k + 1]; ++c)
          if (s->h_gcorr[(size_t) c].fixed_idx < 0 || s->h_gcorr[(size_t) c].fixed_idx >= s->nf ||
              s->h_gcorr[(size_t) c].moving_idx < 0 || s->h_gcorr[(size_t) c].moving_idx >= nmk)
            return fail(SRRG2_E_INVALID, "compute: correspondence index out of range");
      }
      if ((rc = s->gcorr.reserve((size_t) std::max(total, 1)))) return rc;
      if ((rc = s->gcorr_off.reserve((size_t) K + 1))) return rc;
      if ((rc = s->gcorr_stat.reserve((size_t) std::max(total, 1)))) return rc;
      if (total > 0)
        HIP_TRY(hipMemcpyAsync(s->gcorr.p, s->h_gcorr.data(), sizeof(srrg2_correspondence) * (size_t) total,
                               hipMemcpyHostToDevice, a->stream));
      HIP_TRY(hipMemcpyAsync(s->gcorr_off.p, s->h_gcorr_off.data(), sizeof(int) * ((size_t) K + 1), hipMemcpyHostToDevice,
                             a->stream));
      HIP_TRY(hipStreamSynchronize(a->stream));  // (pageable host vectors)
      d.gcorr      = s->gcorr.p;
      d.gcorr_off  = s->gcorr_off.p;
      d.gcorr_stat = s->gcorr_stat.p;
      sc.gcorr_off = s->gcorr_off.p;
    }
    if (s->cfg.finder == SRRG2_FINDER_PROJECTIVE) {
      if ((rc = s->zbuf.reserve((size_t) 2 * K * d.rows * d.cols))) return rc;
      d.zbuf = s->zbuf.p;
      // both z-buffers start clean; afterwards every pass resets the buffer of the next one
      HIP_TRY(hipMemsetAsync(s->zbuf.p, 0xff, (size_t) 2 * K * d.rows * d.cols * sizeof(unsigned long long), a->stream));
    }
    d.tune            = tn.strategy_mask;
    if (a->dim == 3)
      dm::se3_inverse(s->cfg.sensor_in_robot, d.Sinv);
    else
      dm::se2_inverse(s->cfg.sensor_in_robot, d.Sinv);
    std::memcpy(sc.Sinv, d.Sinv, sizeof(sc.Sinv));
  }
  if (K > 1 && first_cue >= 0) {
    for (int si = 0; si < nslices; ++si)
      if (a->slices[si]->cfg.kind != SRRG2_SLICE_PRIOR && si != first_cue)
        return fail(SRRG2_E_UNSUPPORTED, "compute_batch supports one cue slice (plus prior slices)");
  }
  // all cue slices projective (RGB-D: point-to-plane + reprojection): one launch pair per iteration for all of them
  std::vector<int> proj_group;
  {
    bool all_proj = true;
    for (int si = 0; si < nslices; ++si) {
      if (a->slices[si]->cfg.kind == SRRG2_SLICE_PRIOR) continue;
      if (a->slices[si]->cfg.finder != SRRG2_FINDER_PROJECTIVE) all_proj = false;
      proj_group.push_back(si);
    }
    if (!all_proj || proj_group.size() < 2 || proj_group.size() > 4 || (C.tune & 131072)) proj_group.clear();
  }
  // ... and when they all read the SAME clouds (srrg2_aligner_share_clouds) through the same finder parameters, their
  // associations are identical: one z-buffer pass and one step launch serve all of them (k_icp_step_proj_fused)
  bool proj_fused = !proj_group.empty() && K >= 1;
  for (size_t z = 1; z < proj_group.size() && proj_fused; ++z) {
    const Slice* s0 = a->slices[proj_group[0]];
    const Slice* sz = a->slices[proj_group[z]];
    const srrg2_slice_config &c0 = s0->cfg, &cz = sz->cfg;
    proj_fused = sz->alias_of == proj_group[0] && s0->alias_of < 0 && cz.image_rows == c0.image_rows && cz.image_cols == c0.image_cols &&
                 cz.depth_min == c0.depth_min && cz.depth_max == c0.depth_max && cz.finder_max_distance == c0.finder_max_distance &&
                 std::memcmp(cz.camera_matrix, c0.camera_matrix, sizeof(c0.camera_matrix)) == 0 &&
                 std::memcmp(cz.sensor_in_robot, c0.sensor_in_robot, sizeof(c0.sensor_in_robot)) == 0;
  }
  // Fused control steps (FusedCtl): the control step of iteration i runs in the prologue of the first pass kernel of
  // iteration i + 1 -- no control launch between the passes of a run.  For aligners whose slices are all nearest-neighbour
  // cue slices on the list / converged-pass kernels; everything else keeps one control launch per iteration.
  // tuning.fused_control: 0 = never, 1 / -1 = whenever the configuration allows it.  Measured (profiles/r5m): C2 45.5 -> 52.4 k
  // it/s, C4 with 8 / 16 / 24 / 32 / 64 / 256 alignments per call +17 / +12 / +11 / +8 / +3 / +0 %.  (A first version read the record
  // with an agent-scope load in EVERY workgroup and lost 8 - 13 % from 24 alignments on, profiles/r5e, r5f: the launch now
  // walks the problems first, so the control steps of all problems run in its first workgroups, and the workgroups of the
  // later rounds read the published records through the L2.)
  fuse = tn.fused_control != 0 && !small && !a->reduce_fn && a->timeline_path.empty() && first_cue >= 0 &&
         a->params.max_iterations >= 2;
  // (one nearest-neighbour cue slice, or projective slices that share one association -- k_proj_zbuf_fz --; no prior slices:
  // the step finds everything in the slices' records)
  // (round 6, late: prior slices next to them -- an odometry prior or a motion model: what a tracker configures, S/instances.cpp:35-38
  // -- are linearised by the control wave itself, wave_prior; they may stand before or behind the cue slices, not between two)
  int ncue_all = 0, last_cue = -1, prior_mask = 0;
  for (int si = 0; si < nslices; ++si) {
    if (a->slices[si]->cfg.kind == SRRG2_SLICE_PRIOR) {
      prior_mask |= 1 << si;
    } else {
      ++ncue_all;
      last_cue = si;
    }
  }
  const bool cues_consecutive = first_cue >= 0 && last_cue - first_cue + 1 == ncue_all;
  // (at most two prior slices: the control wave keeps their linearisations in registers until the sums are in)
  if (prior_mask && (tn.fused_control == 2 /* cue slices only, as before: A/B */ || (C.tune & (1 << 24)) || nslices - ncue_all > 2)) fuse = false;
  const bool fuse_proj = proj_fused && (int) proj_group.size() == ncue_all && K == 1;
  fuse = fuse && cues_consecutive && (ncue_all == 1 || fuse_proj);
  // A nearest-neighbour slice WITHOUT lists (the first compute() on a fixed cloud: a tracker's frame) or with the deferred-search
  // queue runs its search passes on the grid kernels, which read ProblemState and have no prologue: those iterations keep their
  // control launch (it publishes the record too), and the control steps are fused from the first converged pass on --
  // `fused_all` = every pass kernel of this compute() can carry a control step, else only k_icp_step_fast (fast_at below).
  bool fused_all = true;
  int nm_max_cue = 0;
  if (first_cue >= 0)
    for (int k = 0; k < K; ++k) nm_max_cue = std::max(nm_max_cue, all[(size_t) first_cue * K + k].nm);
  auto fast_at = [&](int slot0, int it) {  // (the choice of run_phase below)
    return (slot0 > 0 || (it >= fast_from && it >= 1)) && !(C.tune & 4) && nm_max_cue >= fast_min;
  };
  for (int si = 0; si < nslices && fuse; ++si) {
    const Slice* s = a->slices[si];
    if (s->cfg.kind == SRRG2_SLICE_PRIOR) continue;
    fuse = fuse_proj ? s->cfg.finder == SRRG2_FINDER_PROJECTIVE : s->cfg.finder == SRRG2_FINDER_NN_GATED;
    // Round 6: the grid kernel has a fused instantiation too (k_icp_step_fused) -- without the deferred-search queue, whose kernel and
    // counters belong to the control LAUNCH.  Whether the queue is worth its launches depends on the frame: a tracker's 100 k-point
    // frame against its clipped local map (everything overlaps, small motion) leaves it nearly empty -- compute() 0.225 -> 0.199 ms
    // on the fused kernels --, a 60 % overlap fills it with hundreds of far searches per pass (0.405 ms with the queue, 0.498 without:
    // profiles/r9/r9q_*).  The fused kernels have no counters to probe, so the evidence is the handle's PREVIOUS compute(): it matched
    // at least 90 % of its moving points (a tracker aligns similar frames one after the other); a handle's first compute(), a frame
    // after a partial overlap, clouds beyond `fused_grid_max` points and the LDS-tile kernels stay on the launches.
    const int fused_grid_max = a->fused_grid_max;
    const bool last_overlapped = a->computed && K == 1 && (size_t) si < a->last_nm_max.size() && a->last_nm_max[(size_t) si] > 0 &&
                                 a->last_ncorr[si] >= (int) (0.9 * a->last_nm_max[(size_t) si]);
    if (fuse && !fuse_proj && !cnl[(size_t) si] && !(lds_tile > 0) && !split && !(C.tune & (1 << 27)) &&
        (!sdev[si].queue || (nm_max_cue <= fused_grid_max && last_overlapped) || fused_grid_max < 0 /* forced: tests */)) {
      sdev[si].queue  = nullptr;  // (finished inside the step kernel)
      sdev[si].qcount = nullptr;
      C.slices[si].qcount = nullptr;
      C.slices[si].qprobe_host = nullptr;
      C.slices[si].probs = nullptr;
      continue;
    }
    if (fuse && !fuse_proj && !(cnl[(size_t) si] && !sdev[si].queue)) {
      fused_all = false;
      // (worth it when converged passes follow; SRRG2_AMD_TUNE bit 27: lists or nothing, as before)
      fuse = !split && !(C.tune & (1 << 27)) &&
             (fast_at(0, a->params.max_iterations - 1) || (a->params.enable_inlier_only_runs && fast_at(a->params.max_iterations, 0)));
    }
  }
  // (the fused launches carry the tile index in gridDim.y -- x = problem, so that the control steps run in the launch's first
  // workgroups --, and y stops at 65 535: clouds beyond that many tiles of the four-lanes-per-point first pass, 4.19 M points,
  // keep the control launches and the x = tile grids of the legacy kernels)
  if (((long long) nm_max_cue * 4 + 255) / 256 > 65535) fuse = false;
  // (no moving point at all: no pass is launched, so nothing would carry the control steps -- without prior slices the run ends at its
  // first control step anyway, the final one; WITH a prior slice the prior keeps it alive, iteration after iteration: launches)
  if (prior_mask && nm_max_cue <= 0) fuse = false;
  if (fuse) {
    if ((rc = a->pub.reserve((size_t) K * SRRG2_MAX_SLICES * PUB_SLICE_GRANULES))) return rc;
    if ((rc = a->pub_epoch.reserve((size_t) K * PUB_EPOCH_REPLICAS * PUB_EPOCH_STRIDE))) return rc;
    if ((rc = a->ctl_dev.reserve((size_t) srrg2_aligner::MAX_PARTS))) return rc;
    C.pub       = a->pub.p;
    C.pub_epoch = a->pub_epoch.p;
  }
  bool hook_failed = false;
  if (a->reduce_fn) {
    // point-sharded alignment: one nearest-neighbour cue slice, K = 1; max |coordinate| over all ranks before the
    // exponent is sized
    int ncue_nn = 0, ncue = 0;
    for (int si = 0; si < nslices; ++si) {
      if (a->slices[si]->cfg.kind == SRRG2_SLICE_PRIOR) continue;
      ++ncue;
      if (a->slices[si]->cfg.finder == SRRG2_FINDER_NN_GATED) ++ncue_nn;
    }
    if (K != 1 || ncue != 1 || ncue_nn != 1)
      return fail(SRRG2_E_UNSUPPORTED, "point-sharded alignments: compute() with one nearest-neighbour cue slice");
    if (a->reduce_fn(a->reduce_user, SRRG2_REDUCE_MAX_U32, a->slices[first_cue]->pinf.p, (size_t) K, (void*) a->stream))
      return fail(SRRG2_E_INVALID, "the reduction hook of the point-sharded alignment failed");
  }
  // k_icp_init sizes the fixed-point exponents from the per-slice problem tables ([slice][K])
  auto t_prep = std::chrono::steady_clock::now();
  // the parts of a pipelined batch: problems [part_begin[p], part_begin[p + 1]) on pstream[p]; otherwise one range
  constexpr int MP  = srrg2_aligner::MAX_PARTS;
  const int nhalves = split ? split : 1;
  int h0[MP], hn[MP];
  hipStream_t hstream[MP];
  std::vector<CtlParams> Ch((size_t) nhalves, C);
  for (int h = 0; h < nhalves; ++h) {
    h0[h]       = split ? a->part_begin[h] : 0;
    hn[h]       = split ? a->part_begin[h + 1] - a->part_begin[h] : K;
    hstream[h]  = a->pstream[h];
    Ch[h].prob0 = h0[h];
    Ch[h].nprob = hn[h];
    if (fuse) Ch[h].ctl_dev = a->ctl_dev.p + h;
  }
  int epoch = 0;  // control steps applied (or scheduled inside a pass) so far: what the next passes build on
  int pround = 0; // rounds of passes launched so far: round k adds into slot-set buffer k % 3 (FusedCtl::prev_partials)
  auto slot_buffer = [&](Slice* sl, int k) { return sl->partials.p + (size_t) (k % 3) * K * PARTIAL_SLOTS * ACC_N; };
  if (fuse)
    for (int si = 0; si < nslices; ++si) {
      sdev[si].fc.pub       = a->pub.p;
      sdev[si].fc.pub_epoch = a->pub_epoch.p;
      sdev[si].fc.stats     = a->stats.p;
      sdev[si].fc.min_num_correspondences = a->slices[si]->cfg.min_num_correspondences;
      sdev[si].fc.max_stats = slots;
      sdev[si].fc.has_term  = a->has_term ? 1 : 0;
      sdev[si].fc.first_round = a->cu_count * 6;  // (CUs x 6 workgroups: nothing beyond them is resident when a launch starts)
      sdev[si].fc.prior_mask  = prior_mask;
      sdev[si].fc.nslices     = nslices;
    }
  if (split && (small || a->reduce_fn || a->profile || !a->timeline_path.empty()))
    return fail(SRRG2_E_STATE, "internal: a pipelined batch on a path that cannot be split");
  // compute()'s prologue INSIDE the first pass (k_icp_step_cnl_init / k_icp_step_fused_init; kernels.hip: pass_view_init) instead of
  // a k_icp_init launch in front of it (9 us of a 100 k-point compute()'s 164): a single alignment whose every pass carries its
  // control step, on the list or the fused grid kernel, whose last step is k_icp_final_wave (it leaves the slot sets zeroed: the
  // one thing of the prologue the first pass cannot do for itself), on a handle whose previous compute() ended that way.
  // (SRRG2_AMD_TUNE bit 23: always the launch)
  const bool final_wave = fuse && !(C.tune & (1 << 25));
  bool fold_init = false;
  InitInline fold_inl{};
  std::vector<InitBatch> fold_bat;  // (one per part of a batch)
  // (batches too, while every launch -- a part of a pipelined batch -- holds at most INIT_BATCH_MAX alignments: their guesses and table
  // rows ride in the arguments of the part's first pass, InitBatch.  Read from the PINNED tables by the pass itself -- every wave of
  // it, over PCIe -- 32 alignments lost 50 us where the launch costs 12: 0.545 -> 0.60 ms, profiles/r10/r10f)
  int part_max = 0;
  for (int h = 0; h < nhalves; ++h) part_max = std::max(part_max, hn[h]);
  if (fuse && !fuse_proj && fused_all && final_wave && (K == 1 || (part_max <= INIT_BATCH_MAX && cnl[(size_t) first_cue])) && first_cue >= 0 && nm_max_cue > 0 && !(C.tune & (1 << 23)) &&
      (cnl[(size_t) first_cue] || !(lds_tile > 0)) && !a->profile) {
    const Slice* s = a->slices[first_cue];
    fold_init = s->slots_zeroed >= 3 * K && s->slots_zeroed_at == (const void*) s->partials.p;
    if (fold_init) {
      const bool single = srrg2amd::make_init_inline(Ch[0], a->probs_host, a->guesses_host, a->tsize, &fold_inl);
      for (int si = 0; si < nslices && fold_init; ++si)  // (a prior slice that sets the initial guess: k_icp_init's override, here)
        if (C.slices[si].kind == SRRG2_SLICE_PRIOR && C.slices[si].prior_sets_initial_guess) {
          if (single)
            for (int i = 0; i < a->tsize; ++i) fold_inl.guess[i] = C.slices[si].prior_Z[i];
          else
            fold_init = false;  // (a batch: the override per problem stays k_icp_init's)
        }
      if (fold_init && !single) {
        fold_bat.assign((size_t) nhalves, InitBatch{});
        for (int h = 0; h < nhalves; ++h) {
          InitBatch& B = fold_bat[(size_t) h];
          B.cue_slice  = first_cue;
          for (int k = 0; k < hn[h]; ++k) {
            for (int i = 0; i < 12; ++i) B.guess[k][i] = i < a->tsize ? a->guesses_host[(size_t) (h0[h] + k) * a->tsize + i] : 0.f;
            B.pd[k] = a->probs_host[(size_t) first_cue * K + h0[h] + k];
          }
        }
      }
    }
  }
  // (a pack of projective slices: the prologue rides in the z-buffer pass of the first iteration, k_proj_zbuf_fz_init)
  if (fuse && fuse_proj && final_wave && K == 1 && first_cue >= 0 && nm_max_cue > 0 && !(C.tune & (1 << 23)) && !a->profile) {
    fold_init = true;
    for (int si : proj_group) {
      const Slice* s = a->slices[si];
      fold_init = fold_init && s->slots_zeroed >= 3 * K && s->slots_zeroed_at == (const void*) s->partials.p;
    }
    if (fold_init) {
      fold_init = srrg2amd::make_init_inline(Ch[0], a->probs_host, a->guesses_host, a->tsize, &fold_inl);
      for (int si = 0; si < nslices && fold_init; ++si)
        if (C.slices[si].kind == SRRG2_SLICE_PRIOR && C.slices[si].prior_sets_initial_guess)
          for (int i = 0; i < a->tsize; ++i) fold_inl.guess[i] = C.slices[si].prior_Z[i];
    }
  }
  for (Slice* sl : a->slices) sl->slots_zeroed = 0;  // (until this compute() has ended the same way)
  if (!fold_init)
    for (int h = 0; h < nhalves; ++h)
      srrg2amd::launch_icp_init(Ch[h], a->probs_host, a->probs.p, a->states.p, a->guesses_host, a->tsize, hstream[h]);
  auto t_init = std::chrono::steady_clock::now();

  if (small) {
    Slice* s = a->slices[first_cue];
    srrg2amd::launch_icp_small(a->dim, s->cfg.kind == SRRG2_SLICE_P2PLANE, sdev[first_cue], C,
                               a->probs.p + (size_t) first_cue * K, a->states.p, a->stats.p, a->outs_host, a->stats_host,
                               a->stream);
  }
  // Adaptive use of the deferred-search kernel.  After iteration `probe_it` the control kernel looks at what that
  // iteration deferred: few entries, none of them far (a far entry is a whole-wave scan: expensive when a wave has to do
  // several in a row) => it clears st->qmode and the step kernels finish their open points themselves from then on; it
  // also mirrors the counters into pinned memory.  The host enqueues one more full iteration, then reads the mirror
  // (it has arrived by then: no idle GPU) and applies the same rule to drop the deferred-search launch for the
  // remaining iterations (C2: 8 us each).  Partial overlaps keep hundreds of far entries per iteration (near-ties along
  // the border of the fixed cloud) and keep the queue.
  const int probe_it = C.probe_it;
  std::vector<char> queue_on((size_t) std::max(nslices, 1), 1);
  bool probed = false;
  bool final_launched = false;  // the last control step of compute() carried the post / finalize steps
  // (the probe: the queue counters of iteration `probe_it`, reported by that iteration's control launch)
  auto read_probe = [&]() {
    probed = true;
    for (int si = 0; si < nslices; ++si) {
      if (!sdev[si].queue) continue;
      volatile int* mirror = a->slices[si]->qprobe_host;
      bool small           = true;
      for (int k = 0; k < K; ++k) {
        int spins = 0;
        while (mirror[2 * k] == 0x7f7f7f7f || mirror[2 * k + 1] == 0x7f7f7f7f) {  // not reported yet
          // (a problem that stopped before the probe iteration never reports: once the stream has drained, give up
          // and keep the queue)
          if ((++spins & 1023) == 0 && hipStreamQuery(a->stream) != hipErrorNotReady) break;  // drained or failed
        }
        const int near = mirror[2 * k], far = mirror[2 * k + 1];
        if (near == 0x7f7f7f7f || far > 32 || near > std::max(1024, all[(size_t) si * K + k].nm / 64)) small = false;
      }
      queue_on[(size_t) si] = small ? 0 : 1;
    }
  };
  // Does the pass of iteration `it` carry the control step of the one before it (a FUSED kernel instantiation)?  Without lists
  // only the converged-pass kernel can, and only while the deferred-search queue is not needed: with the queue on (a partial
  // overlap: hundreds of far searches per pass, C2 at 60 % overlap) the in-wave searches of the fused kernel cost more than the
  // control launches they save (0.39 -> 0.57 ms, profiles/r6r) -- those alignments stay on the launches.
  auto fused_pass = [&](int slot0, int it) {
    if (!fuse) return false;
    if (fused_all) return true;
    const bool queue_needed = sdev[first_cue].queue && queue_on[(size_t) first_cue];
    return fast_at(slot0, it) && !queue_needed && (slot0 > 0 || probed || probe_it < 0 || !sdev[first_cue].queue);
  };
  auto control = [&](int it, bool last_phase, int h) {
    if (a->reduce_fn) {  // the ranks' partial sums, added in place, before anybody looks at them
      // (after a failure the hook is still called for the remaining control steps: the collectives of the ranks stay
      // matched -- a rank that stopped calling would leave its healthy peers hanging in theirs -- and the error is
      // reported when compute() returns)
      Slice* s = a->slices[first_cue];
      if (a->reduce_fn(a->reduce_user, SRRG2_REDUCE_SUM_I64, s->partials.p, (size_t) K * PARTIAL_SLOTS * ACC_N, (void*) a->stream))
        hook_failed = true;
    }
    const bool last_it = it == a->params.max_iterations - 1;
    if (fuse) {
      // (the passes of this iteration added into slot-set buffer pround % 3; its control step produces epoch + 1: inside the
      // first pass of the next iteration, or -- the last iteration of a run -- as a launch)
      Ch[h].parity      = pround % 3;
      Ch[h].zero_parity = (pround + 2) % 3;
      Ch[h].epoch       = epoch + 1;
      // (the next iteration's first kernel applies this step -- if it can: the grid kernels cannot, and the launch after the
      // probe iteration also reports the queue counters to the host)
      const int slot0_now = last_phase && a->params.enable_inlier_only_runs ? a->params.max_iterations : 0;
      // (the verdict on the queue is needed one control step earlier than the loop below asks for it)
      if (!fused_all && !probed && slot0_now == 0 && probe_it >= 0 && it == probe_it + 1) read_probe();
      if (!last_it && fused_pass(slot0_now, it + 1) && (fused_all || it != probe_it)) return;
    }
    if (last_phase && last_it) {
      // (the last step on one wave too, k_icp_final_wave.  Round 5's version let the finalize part read the state back from memory
      // and measured SLOWER than the 256-thread kernel, which has it staged in LDS (C2 0.191 against 0.188 ms, profiles/r6a);
      // round 6, late: post and finalize run from the step's REGISTERS and the record for the host is written by the lanes of the
      // wave, one word each, behind one system-scope fence.  SRRG2_AMD_TUNE bit 25 switches back to the 256-thread kernel)
      if (final_wave && fuse_proj) {  // (a pack of projective slices: the same wave, the parameters from their device copy)
        SliceDev pack[4];
        const ProblemDev* pp[4];
        for (size_t z = 0; z < proj_group.size(); ++z) {
          const int si = proj_group[z];
          Slice* sz    = a->slices[si];
          pack[z]      = sdev[si];
          pp[z]        = a->probs.p + (size_t) si * K;
          pack[z].fc.ctl           = a->ctl_dev.p;
          pack[z].fc.epoch         = epoch + 1;
          pack[z].fc.prev_partials = slot_buffer(sz, pround);
          pack[z].fc.zero_partials = slot_buffer(sz, pround + 2);
        }
        srrg2amd::launch_icp_final_wave_pack(pack, pp, (int) proj_group.size(), a->states.p, a->stats.p, a->outs_host, a->stats_host,
                                             !a->params.enable_inlier_only_runs, hstream[h]);
      } else if (final_wave) {
        SliceDev sd          = sdev[first_cue];
        sd.prob0             = h0[h];
        sd.fc.ctl            = a->ctl_dev.p + h;
        sd.fc.epoch          = epoch + 1;
        sd.fc.prev_partials  = slot_buffer(a->slices[first_cue], pround);
        sd.fc.zero_partials  = slot_buffer(a->slices[first_cue], pround + 2);
        srrg2amd::launch_icp_final_wave(Ch[h], sd, a->states.p, a->stats.p, a->outs_host, a->stats_host,
                                        !a->params.enable_inlier_only_runs, hstream[h]);
      } else {
        srrg2amd::launch_icp_control_final(Ch[h], a->states.p, a->stats.p, a->outs_host, a->stats_host,
                                           !a->params.enable_inlier_only_runs /* post step inside */, hstream[h]);
      }
      final_launched = true;
    } else {
      srrg2amd::launch_icp_control(Ch[h], a->states.p, a->stats.p, hstream[h]);
    }
  };
  auto run_phase = [&](int slot0, bool last_phase) -> int {
    for (int it = 0; it < a->params.max_iterations; ++it) {
      if (!probed && slot0 == 0 && probe_it >= 0 && it == probe_it + 2) read_probe();
      if (!proj_group.empty()) {
        SliceDev pack[4];
        const ProblemDev* pp[4];
        int nm_max = 0;
        for (size_t z = 0; z < proj_group.size(); ++z) {
          const int si = proj_group[z];
          pack[z]      = sdev[si];
          pp[z]        = a->probs.p + (size_t) si * K;
          for (int k = 0; k < K; ++k) nm_max = std::max(nm_max, all[(size_t) si * K + k].nm);
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (a->profile) {
          if (a->prof_used == a->prof_events.size()) {
            hipEvent_t x, y;
            HIP_TRY(hipEventCreate(&x));
            HIP_TRY(hipEventCreate(&y));
            a->prof_events.emplace_back(x, y);
          }
          e0 = a->prof_events[a->prof_used].first;
          e1 = a->prof_events[a->prof_used].second;
          a->prof_used++;
          HIP_TRY(hipEventRecord(e0, a->stream));
        }
        if (fuse)
          for (size_t z = 0; z < proj_group.size(); ++z) {
            Slice* sz              = a->slices[proj_group[z]];
            pack[z].fc.ctl         = a->ctl_dev.p;
            pack[z].fc.epoch       = epoch;
            pack[z].partials       = slot_buffer(sz, pround);
            pack[z].fc.prev_partials = slot_buffer(sz, pround + 2);  // (round pround - 1; the step zeroes the one after this round's)
            pack[z].fc.zero_partials = slot_buffer(sz, pround + 1);
            pack[z].fc.prior       = (slot0 > 0 || it > 0) ? 1 : 0;
            if (it == 0) pack[z].fc.prior_mask = 0;  // (as for the nearest-neighbour passes below)
          }
        if (proj_fused) {
          const bool first_pass_init = fold_init && fuse_proj && slot0 == 0 && it == 0;
          srrg2amd::launch_proj_step_fused(pack, pp, (int) proj_group.size(), a->states.p, K, nm_max, a->stream,
                                           first_pass_init ? &Ch[0] : nullptr, first_pass_init ? &fold_inl : nullptr,
                                           first_pass_init ? a->probs.p : nullptr);
          sdev[proj_group[0]].zbuf_parity ^= 1;  // (the one z-buffer that is used)
        } else {
          srrg2amd::launch_proj_step_pack(pack, pp, (int) proj_group.size(), a->states.p, K, nm_max, a->stream);
          for (int si : proj_group) sdev[si].zbuf_parity ^= 1;
        }
        if (a->profile) HIP_TRY(hipEventRecord(e1, a->stream));
        control(it, last_phase, 0);
        ++epoch;
        ++pround;
        continue;
      }
      for (int si = 0; si < nslices; ++si) {
        Slice* s = a->slices[si];
        if (s->cfg.kind == SRRG2_SLICE_PRIOR) continue;
        const bool plane = s->cfg.kind == SRRG2_SLICE_P2PLANE;
        int nm_max       = 0;
        for (int k = 0; k < K; ++k) nm_max = std::max(nm_max, all[(size_t) si * K + k].nm);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (a->profile) {
          if (a->prof_used == a->prof_events.size()) {
            hipEvent_t x, y;
            HIP_TRY(hipEventCreate(&x));
            HIP_TRY(hipEventCreate(&y));
            a->prof_events.emplace_back(x, y);
          }
          e0 = a->prof_events[a->prof_used].first;
          e1 = a->prof_events[a->prof_used].second;
          a->prof_used++;
          HIP_TRY(hipEventRecord(e0, a->stream));
        }
        if (s->cfg.finder == SRRG2_FINDER_CORRESPONDENCES) {
          int max_nc = 0;
          for (int k = 0; k < K; ++k) max_nc = std::max(max_nc, s->h_gcorr_off[(size_t) k + 1] - s->h_gcorr_off[(size_t) k]);
          srrg2amd::launch_corr_step(a->dim, plane, sdev[si], a->probs.p + (size_t) si * K, a->states.p, K, max_nc, a->stream);
        } else if (s->cfg.finder == SRRG2_FINDER_PROJECTIVE) {
          srrg2amd::launch_proj_step(s->cfg.kind == SRRG2_SLICE_REPROJECTION, sdev[si], a->probs.p + (size_t) si * K,
                                     a->states.p, K, nm_max, a->stream);
          sdev[si].zbuf_parity ^= 1;  // the pass just launched reset the other buffer
        } else {
          // The deferred-search kernel pays off while many points are open; once the searches are mostly skipped
          // its launch costs more than finishing a few near points inside the step kernel (queue_on, decided below).
          SliceDev sd = sdev[si];
          // (it >= 1 in the first run whatever the knob says: iteration 0 has no previous neighbours)
          const bool fast = (slot0 > 0 || (it >= fast_from && it >= 1)) && !(C.tune & 4) && nm_max >= fast_min;
          if (!queue_on[si] || (s->fast_queue_only && !fast)) {
            sd.queue  = nullptr;
            sd.qcount = nullptr;
          }
          for (int h = 0; h < nhalves; ++h) {
            // (a pipelined batch has one cue slice: each half's pass is followed by that half's control step on its stream)
            sd.prob0 = h0[h];
            if (fuse) {
              if (!fused_pass(slot0, it)) sd.fc.pub = nullptr;  // (this pass on the kernels that read ProblemState)
              sd.fc.ctl   = a->ctl_dev.p + h;
              sd.fc.epoch = epoch;
              sd.partials = slot_buffer(s, pround);
              sd.fc.prev_partials = slot_buffer(s, pround + 2);  // (round pround - 1; the step zeroes the one after this round's)
              sd.fc.zero_partials = slot_buffer(s, pround + 1);
              sd.fc.prior         = (slot0 > 0 || it > 0) ? 1 : 0;
              // (the first pass of a run starts behind the init / post LAUNCH: its record is current, no control step is due, and it
              // runs on the instantiation without the prior factors' code -- the four-lanes-per-point search pass of a 100 k-point
              // cloud took 40.6 us at the 128 registers of that code against 26.7 at its own 74)
              if (it == 0) sd.fc.prior_mask = 0;
            }
            const ProblemDev* pt = a->probs.p + (size_t) si * K;
            hipStream_t hs = hstream[h];
            const int Kh   = hn[h];
            const bool first_pass_init = fold_init && slot0 == 0 && it == 0 && si == first_cue;  // (the prologue rides in this pass)
            if (fast)
              srrg2amd::launch_icp_step_fast(a->dim, plane, sd, pt, a->states.p, Kh, nm_max, fast_ppt_of(Kh), fast_gather, hs);
            else if (cnl[(size_t) si] && !sd.queue)
              srrg2amd::launch_icp_step_cnl(a->dim, plane, sd, s->lists_host, pt, a->states.p, Kh, nm_max,
                                            search_team_knob > 0 ? search_team_knob : ((K <= 4 && slot0 == 0 && it == 0) ? 4 : 1), hs,
                                            first_pass_init ? &Ch[h] : nullptr, first_pass_init ? &fold_inl : nullptr,
                                            first_pass_init && !fold_bat.empty() ? &fold_bat[(size_t) h] : nullptr);
            else if (!sd.queue && lds_tile > 0 && !small)
              srrg2amd::launch_icp_step_tile(a->dim, plane, sd, pt, a->states.p, Kh, nm_max, lds_tile == 2 ? 504 : 416, hs);
            else
              srrg2amd::launch_icp_step(a->dim, plane, sd, pt, a->states.p, Kh, nm_max, hs, first_pass_init ? &Ch[h] : nullptr,
                                        first_pass_init ? &fold_inl : nullptr);
            if (split) control(it, last_phase, h);
          }
        }
        if (a->profile) HIP_TRY(hipEventRecord(e1, a->stream));
      }
      if (!split) control(it, last_phase, 0);
      ++epoch;
      ++pround;
    }
    return 0;
  };
  if (small) {
    final_launched = true;  // (k_icp_small did everything)
  } else {
    if ((rc = run_phase(0, !a->params.enable_inlier_only_runs))) return rc;
    if (a->params.enable_inlier_only_runs) {
      for (int h = 0; h < nhalves; ++h) {
        Ch[h].epoch = epoch + 1;  // (the post step changes flags the passes read: it republishes the records)
        srrg2amd::launch_icp_post(Ch[h], a->states.p, a->stats.p, hstream[h]);
      }
      ++epoch;
      if ((rc = run_phase(a->params.max_iterations, true))) return rc;
    }
  }
  // results land in pinned host memory (written by k_icp_finalize): the only host-device interaction of compute() after
  // the launches is this wait
  if (!final_launched)  // (max_iterations < 1)
    for (int h = 0; h < nhalves; ++h)
      srrg2amd::launch_icp_finalize(Ch[h], a->states.p, a->stats.p, a->outs_host, a->stats_host,
                                    !a->params.enable_inlier_only_runs /* post step inside */, hstream[h]);
  if (split) a->parts_dirty = std::max(a->parts_dirty, split);
  HIP_TRY(hipGetLastError());
  const bool hosttime = a->hosttime;
  auto t_enq = std::chrono::steady_clock::now();
  // The last kernel of compute() writes every result into pinned host memory and, behind a system-scope fence, the
  // sequence number of this compute() into ProblemOut::seq: the host polls those words instead of waiting for the
  // stream to retire (hipStreamSynchronize returns ~5 us after the results are visible).  The stream is only waited for
  // when something else needs it idle (event timing, the timeline dump) or when it reports an error.
  bool seen = true;
  for (int k = 0; k < K && seen; ++k) {
    volatile int* flag = &a->outs_host[k].seq;
    int spins          = 0;
    hipStream_t qs = a->stream;
    for (int h = 1; h < nhalves; ++h)
      if (k >= h0[h]) qs = hstream[h];
    while (*flag != C.seq)
      if ((++spins & 4095) == 0 && hipStreamQuery(qs) != hipErrorNotReady) {
        seen = *flag == C.seq;  // (drained: either the flag has just arrived or a launch failed)
        break;
      }
  }
  if (!seen || a->profile || !a->timeline_path.empty()) {
    HIP_TRY(hipStreamSynchronize(a->stream));
    for (int h = 1; h < nhalves; ++h) HIP_TRY(hipStreamSynchronize(hstream[h]));
  }
  if (hook_failed) return fail(SRRG2_E_INVALID, "the reduction hook of the point-sharded alignment failed");
  if (hosttime) {
    auto t_end = std::chrono::steady_clock::now();
    std::fprintf(stderr, "compute: prep %.1f us, first launch %.1f us, enqueue %.1f us, wait %.1f us\n",
                 std::chrono::duration<double, std::micro>(t_prep - t_begin).count(),
                 std::chrono::duration<double, std::micro>(t_init - t_prep).count(),
                 std::chrono::duration<double, std::micro>(t_enq - t_begin).count(),
                 std::chrono::duration<double, std::micro>(t_end - t_enq).count());
  }
  if (const char* tl_path = a->timeline_path.empty() ? nullptr : a->timeline_path.c_str()) {  // dump of the last compute(): u64 nwaves, then stamps
    for (int si = 0; si < nslices; ++si) {
      Slice* s = a->slices[si];
      if (!sdev[si].dbg) continue;
      int nm_max = 0;
      for (int k = 0; k < K; ++k) nm_max = std::max(nm_max, all[(size_t) si * K + k].nm);
      const unsigned long long nw = (unsigned long long) K * ((nm_max + 255) / 256) * 4;
      std::vector<unsigned long long> host(32 * nw * 16);
      HIP_TRY(hipMemcpy(host.data(), s->dbg.p, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (FILE* f = std::fopen(tl_path, "wb")) {
        std::fwrite(&nw, sizeof(nw), 1, f);
        std::fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
        std::fclose(f);
      }
    }
  }
  if (a->profile) {
    for (size_t i = 0; i < a->prof_used; ++i) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, a->prof_events[i].first, a->prof_events[i].second));
      a->prof_ms += ms;
      a->prof_launches++;
    }
    a->prof_used = 0;
  }
  for (Slice* sl : a->slices) sl->ms_pending = false;  // (the stream has drained)
  a->K = K;
  // the handle's observable state is that of the last alignment (sequential semantics)
  const ProblemOut& o = a->outs_host[K - 1];
  std::memcpy(a->X, o.X, sizeof(float) * a->tsize);
  a->status = o.status;
  int ns    = std::min(o.nstats, slots);
  a->last_stats.assign(a->stats_host + (size_t) (K - 1) * slots, a->stats_host + (size_t) (K - 1) * slots + ns);
  for (int si = 0; si < SRRG2_MAX_SLICES; ++si) a->last_ncorr[si] = o.ncorr[si];
  std::memcpy(a->last_H, o.H, sizeof(a->last_H));
  a->computed = true;
  a->last_sdev = sdev;
  a->last_proj_owner.assign((size_t) nslices, -1);
  if (proj_fused)
    for (int si : proj_group) a->last_proj_owner[(size_t) si] = proj_group[0];
  a->last_nm_max.assign((size_t) nslices, 0);
  for (int si = 0; si < nslices; ++si)
    for (int k = 0; k < K; ++k) a->last_nm_max[(size_t) si] = std::max(a->last_nm_max[(size_t) si], all[(size_t) si * K + k].nm);
  a->records_state = 1;
  a->last_path = (fuse ? SRRG2_PATH_FUSED_CONTROL : 0) | (fuse && fused_all ? SRRG2_PATH_ALL_PASSES_FUSED : 0) |
                 (final_wave && final_launched && !small ? SRRG2_PATH_FINAL_WAVE : 0) | (fold_init ? SRRG2_PATH_PROLOGUE_IN_PASS : 0) |
                 (small ? SRRG2_PATH_ONE_WORKGROUP : 0) | (fuse && prior_mask ? SRRG2_PATH_PRIORS_FUSED : 0);
  // (k_icp_final_wave has left the slot sets of its problems zeroed: the next compute() of the handle may skip the k_icp_init launch)
  if (final_wave && final_launched && first_cue >= 0 && !small) {
    for (int si = 0; si < nslices; ++si) {  // (the one nearest-neighbour cue slice, or every slice of the projective pack)
      if (si != first_cue && !(fuse_proj && std::find(proj_group.begin(), proj_group.end(), si) != proj_group.end())) continue;
      Slice* s           = a->slices[si];
      s->slots_zeroed    = 3 * K;
      s->slots_zeroed_at = (const void*) s->partials.p;
    }
  }
  return 0;
}

// correspondence records of the nearest-neighbour slices of the last compute(), derived on demand
// (host_wait = false: a consumer on the DEVICE -- the scene merger on its own stream, aligner_slice_view -- waits for the event
// recorded behind the launches instead of the host waiting for the stream; a host reader that comes later still waits)
int materialize_records(srrg2_aligner* a, bool host_wait = true) {
  if (a->records_state != 1) {
    if (host_wait && a->records_unsynced) {
      HIP_TRY(hipStreamSynchronize(a->stream));
      a->records_unsynced = false;
    }
    return 0;
  }
  int rc;
  if ((rc = set_device(a))) return rc;
  if ((rc = quiesce_stream2(a))) return rc;
  std::vector<char> rebuilt(a->slices.size(), 0);
  for (size_t si = 0; si < a->slices.size() && si < a->last_sdev.size(); ++si) {
    Slice* s = a->slices[si];
    if (s->cfg.kind == SRRG2_SLICE_PRIOR) continue;
    if (s->cfg.finder == SRRG2_FINDER_PROJECTIVE) {
      // (the association belongs to the slice itself, or to the first slice of a group that shared clouds and finder)
      const int owner = si < a->last_proj_owner.size() && a->last_proj_owner[si] >= 0 ? a->last_proj_owner[si] : (int) si;
      srrg2amd::launch_proj_records(a->last_sdev[(size_t) owner], a->last_sdev[si], a->probs.p + (size_t) owner * a->K,
                                    a->probs.p + si * (size_t) a->K, a->states.p, a->K, a->last_nm_max[si],
                                    !rebuilt[(size_t) owner], a->stream);
      rebuilt[(size_t) owner] = 1;
      continue;
    }
    if (s->cfg.finder != SRRG2_FINDER_NN_GATED) continue;
    srrg2amd::launch_icp_outputs(a->dim, s->cfg.kind == SRRG2_SLICE_P2PLANE, a->last_sdev[si], a->probs.p + si * (size_t) a->K,
                                 a->states.p, a->K, a->last_nm_max[si], a->stream);
  }
  HIP_TRY(hipGetLastError());
  if (host_wait) {
    HIP_TRY(hipStreamSynchronize(a->stream));
    a->records_unsynced = false;
  } else {
    a->records_unsynced = true;
  }
  a->records_state = 0;
  return 0;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int srrg2_amd_abi_version(void) {
  return SRRG2_AMD_ABI_VERSION;
}

const char* srrg2_amd_last_error(void) {
  return srrg2amd::g_err.c_str();
}

int srrg2_amd_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(SRRG2_E_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

void srrg2_aligner_default_params(srrg2_aligner_params* p) {
  if (!p) return;
  p->max_iterations = 10;  // aligner.h:30
  p->min_num_inliers = 10; // multi_aligner.h:45
  p->enable_inlier_only_runs = 0;
  p->keep_only_inlier_correspondences = 0;
}

void srrg2_aligner_default_tuning(srrg2_aligner_tuning* t) {
  if (!t) return;
  std::memset(t, 0, sizeof(*t));
  t->strategy_mask          = 0;
  t->queue_probe_iteration  = 1;
  t->small_max_points       = 1024;
  t->fast_from_iteration    = 3;
  t->fast_points_per_thread = 0;
  t->fast_min_points        = 0;
  t->fast_gather            = -1;
  t->fast_batch_queue       = 0;
  t->queue_min_points       = 90000;
  t->msort_segments         = 0;
  t->msort_key_bits         = 0;
  t->lds_tile               = -1;
  t->search_lists           = -1;
  t->search_team            = 0;
  t->batch_pipeline         = -1;
  t->fused_control          = -1;
  t->cell_target            = 8.0f;
  t->rmax_cap               = 0.f;
}

int srrg2_aligner_get_tuning(srrg2_aligner_h a, srrg2_aligner_tuning* t) {
  if (!a || !t) return fail(SRRG2_E_INVALID, "get_tuning: null argument");
  *t = a->tuning;
  return 0;
}

int srrg2_aligner_set_tuning(srrg2_aligner_h a, const srrg2_aligner_tuning* t) {
  if (!a || !t) return fail(SRRG2_E_INVALID, "set_tuning: null argument");
  // (fast_from_iteration >= 1: the converged-pass kernel certifies against the neighbours the PREVIOUS pass of this
  // compute() left behind; at iteration 0 there are none -- ADVICE r3)
  if (t->fast_points_per_thread < 0 || t->fast_from_iteration < 1 || !(t->cell_target > 0.f) || t->msort_key_bits > 18 ||
      t->msort_key_bits < -1 || t->msort_segments < 0 || t->search_team < 0)
    return fail(SRRG2_E_INVALID, "set_tuning: value out of range");
  a->tuning = *t;
  return 0;
}

void srrg2_termination_default_params(srrg2_termination_params* p) {
  if (!p) return;
  p->window_size = 5;  // aligner_termination_criteria.h:40-56
  p->num_correspondences_range = 20;
  p->num_inliers_range = 20;
  p->num_outliers_range = 20;
  p->chi_epsilon = 0.2f;
}

void srrg2_slice_default_config(srrg2_slice_config* c, int variable_kind) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->kind = SRRG2_SLICE_P2P;
  c->finder = SRRG2_FINDER_NN_GATED;
  c->robustifier = SRRG2_ROBUST_NONE;
  c->robustifier_chi_threshold = 1.f;
  c->min_num_correspondences = 0;
  c->finder_max_distance = 1.f;
  c->finder_normal_cos = -2.f;
  c->finder_cell_size = 0.f;
  identity(variable_kind, c->sensor_in_robot);
  for (int i = 0; i < 6; ++i) c->prior_information_diag[i] = variable_kind == SRRG2_SE2_RIGHT ? 100.f : 1.f;
  c->prior_sets_initial_guess = 1;
}

int srrg2_aligner_create(int variable_kind, int device, srrg2_aligner_h* out) {
  if (!out || variable_kind < 0 || variable_kind > 2) return fail(SRRG2_E_INVALID, "create: bad arguments");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(SRRG2_E_NO_DEVICE, "create: no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(SRRG2_E_INVALID, "create: bad device ordinal");
  srrg2_aligner* a = new srrg2_aligner();
  a->kind   = variable_kind;
  a->dim    = variable_kind == SRRG2_SE2_RIGHT ? 2 : 3;
  a->dof    = variable_kind == SRRG2_SE2_RIGHT ? 3 : 6;
  a->tsize  = variable_kind == SRRG2_SE2_RIGHT ? 9 : 12;
  a->device = device;
  identity(variable_kind, a->X);
  tuning_from_environment(&a->tuning);
  if (const char* tl = std::getenv("SRRG2_AMD_TIMELINE")) a->timeline_path = tl;
  a->hosttime = std::getenv("SRRG2_AMD_HOSTTIME") != nullptr;
  if (const char* fg = std::getenv("SRRG2_AMD_FUSED_GRID_MAX")) a->fused_grid_max = std::atoi(fg);
  bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) == hipSuccess;
  a->pstream[0] = a->stream;  // (pstream[1 ..]: created by the first pipelined batch that wants them, compute_batch)
  ok = ok && hipEventCreateWithFlags(&a->ev_staged, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&a->ev_records, hipEventDisableTiming) == hipSuccess;
  if (ok && (hipDeviceGetAttribute(&a->cu_count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || a->cu_count <= 0)) {
    (void) hipGetLastError();
    a->cu_count = 256;
  }
  if (!ok) {
    for (int p = 0; p < srrg2_aligner::MAX_PARTS; ++p)
      if (a->pstream[p]) (void) hipStreamDestroy(a->pstream[p]);
    delete a;
    return fail(SRRG2_E_HIP, "create: cannot create stream");
  }
  *out = a;
  return 0;
}

int srrg2_aligner_destroy(srrg2_aligner_h a) {
  if (!a) return 0;
  (void) hipSetDevice(a->device);
  for (int p = 0; p < srrg2_aligner::MAX_PARTS; ++p)
    if (a->pstream[p]) (void) hipStreamSynchronize(a->pstream[p]);
  for (Slice* s : a->slices) {
    s->release();
    delete s;
  }
  a->probs.release(); a->states.release(); a->outs.release(); a->stats.release(); a->guesses.release();
  a->pub.release(); a->pub_epoch.release(); a->ctl_dev.release();
  a->staging.release();
  if (a->outs_host) (void) hipHostFree(a->outs_host);
  if (a->stats_host) (void) hipHostFree(a->stats_host);
  if (a->guesses_host) (void) hipHostFree(a->guesses_host);
  if (a->probs_host) (void) hipHostFree(a->probs_host);
  for (auto& ev : a->prof_events) {
    (void) hipEventDestroy(ev.first);
    (void) hipEventDestroy(ev.second);
  }
  if (a->ev_staged) (void) hipEventDestroy(a->ev_staged);
  if (a->ev_records) (void) hipEventDestroy(a->ev_records);
  for (int p = srrg2_aligner::MAX_PARTS - 1; p >= 0; --p)
    if (a->pstream[p]) (void) hipStreamDestroy(a->pstream[p]);
  delete a;
  return 0;
}

int srrg2_aligner_set_params(srrg2_aligner_h a, const srrg2_aligner_params* p) {
  if (!a || !p || p->max_iterations < 0) return fail(SRRG2_E_INVALID, "set_params: bad arguments");
  a->params = *p;
  return 0;
}

int srrg2_aligner_set_termination(srrg2_aligner_h a, const srrg2_termination_params* p) {
  if (!a) return fail(SRRG2_E_INVALID, "set_termination: null handle");
  if (!p) {
    a->has_term = false;
    return 0;
  }
  if (p->window_size < 1 || p->window_size > TERM_WINDOW_MAX) return fail(SRRG2_E_INVALID, "set_termination: window_size");
  a->has_term = true;
  a->term     = *p;
  return 0;
}

int srrg2_aligner_add_slice(srrg2_aligner_h a, const srrg2_slice_config* c, int* idx) {
  if (!a || !c) return fail(SRRG2_E_INVALID, "add_slice: bad arguments");
  if ((int) a->slices.size() >= SRRG2_MAX_SLICES) return fail(SRRG2_E_INVALID, "add_slice: too many slices");
  if (c->kind != SRRG2_SLICE_PRIOR && c->finder != SRRG2_FINDER_NN_GATED && c->finder != SRRG2_FINDER_PROJECTIVE &&
      c->finder != SRRG2_FINDER_CORRESPONDENCES)
    return fail(SRRG2_E_INVALID, "add_slice| no finder");  // aligner_slice_processor_impl.cpp:13-16
  if (c->finder == SRRG2_FINDER_CORRESPONDENCES && c->kind != SRRG2_SLICE_P2P && c->kind != SRRG2_SLICE_P2PLANE)
    return fail(SRRG2_E_INVALID, "add_slice: given correspondences drive point-to-point / point-to-plane factors");
  if (c->finder == SRRG2_FINDER_PROJECTIVE || c->kind == SRRG2_SLICE_REPROJECTION) {
    if (a->dim != 3) return fail(SRRG2_E_UNSUPPORTED, "add_slice: projective finder / reprojection factor are SE(3) only");
    if (c->finder != SRRG2_FINDER_PROJECTIVE)
      return fail(SRRG2_E_INVALID, "add_slice: a reprojection slice needs the projective finder");
    if (c->image_rows <= 0 || c->image_cols <= 0 || !(c->depth_min > 0.f) || !(c->depth_max >= c->depth_min) ||
        !(c->camera_matrix[0] > 0.f) || !(c->camera_matrix[4] > 0.f))
      return fail(SRRG2_E_INVALID, "add_slice: projective slice with bad camera / image / depth range");
  }
  if (c->kind == SRRG2_SLICE_PRIOR && a->kind == SRRG2_SE3_EULER_RIGHT)
    return fail(SRRG2_E_UNSUPPORTED, "add_slice: SE3 prior factors exist for the quaternion variable only "
                                     "(SE3PriorErrorFactorAD, aligner_slice_odometry_prior.h:33)");
  Slice* s = new Slice();
  s->cfg   = *c;
  if (idx) *idx = (int) a->slices.size();
  a->slices.push_back(s);
  return 0;
}

int srrg2_aligner_clear_slices(srrg2_aligner_h a) {
  if (!a) return fail(SRRG2_E_INVALID, "clear_slices: null handle");
  (void) hipSetDevice(a->device);
  (void) hipStreamSynchronize(a->stream);
  for (Slice* s : a->slices) {
    s->release();
    delete s;
  }
  a->slices.clear();
  return 0;
}

int srrg2_aligner_set_robustifier(srrg2_aligner_h a, int si, int kind, float thr) {
  int rc = check_slice(a, si, "set_robustifier");
  if (rc) return rc;
  a->slices[si]->cfg.robustifier               = kind;
  a->slices[si]->cfg.robustifier_chi_threshold = thr;
  return 0;
}

int srrg2_aligner_set_fixed(srrg2_aligner_h a, int si, const float* coords, int cs, const float* normals, int ns, int n,
                            int mem) {
  int rc = check_slice(a, si, "set_fixed");
  if (rc) return rc;
  if (n < 0 || (n > 0 && !coords)) return fail(SRRG2_E_INVALID, "set_fixed: bad cloud");
  Slice* s = a->slices[si];
  if (s->cfg.kind == SRRG2_SLICE_PRIOR) return fail(SRRG2_E_INVALID, "set_fixed on a prior slice: use set_prior_measurement");
  if (s->alias_of >= 0) return fail(SRRG2_E_STATE, "set_fixed: this slice shares the clouds of another slice (share_clouds): set them there");
  if ((rc = set_device(a))) return rc;
  if ((rc = quiesce_stream2(a))) return rc;
  if (a->computed) a->records_state = 2;
  if ((rc = s->fixed_raw.reserve((size_t) std::max(n, 1)))) return rc;
  if (normals && (rc = s->fixed_nrm_raw.reserve((size_t) std::max(n, 1)))) return rc;
  if ((rc = s->scalars.reserve(16))) return rc;
  const size_t bytes_c = (size_t) n * a->dim * 4;
  if (mem == SRRG2_MEM_HOST && (rc = a->staging.reserve(2 * bytes_c + 64))) return rc;
  // (the scalars: bounding box minima start at all ones, everything else at zero; a nearest-neighbour slice's ingest leaves the box
  // and the count of valid points there on its way -- one pass over the cloud and one launch instead of two, round 6)
  const bool nn = s->cfg.finder == SRRG2_FINDER_NN_GATED;
  // (a nearest-neighbour slice whose previous set_fixed went through k_ingest_bbox's last block needs no initialising copy: that block
  // wrote box, count and norm itself and left the ticket and the normals' norm cleared -- one stream operation less per frame)
  if (!(nn && n > 0 && s->scalars_self_init)) {
    unsigned init[16] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(s->scalars.p, init, sizeof(init), hipMemcpyHostToDevice, a->stream));
  }
  s->scalars_self_init = false;
  const float* dsrc;
  int sf;
  if ((rc = stage_input(a, coords, cs, n, a->dim, mem, &dsrc, &sf, 0))) return rc;
  if (nn) {
    // (the box comes back through pinned memory: word 8 = this call's sequence number, never 0; scalars[11] = the blocks' ticket)
    if ((rc = ensure_pinned(s->bbox_host, s->bbox_host_cap, (size_t) 16))) return rc;
    if ((rc = s->bbox_rows.reserve((size_t) INGEST_BBOX_MAX_BLOCKS * 8))) return rc;
    if (++s->bbox_seq == 0) s->bbox_seq = 1;
    s->bbox_host[8] = 0;
    const bool rows = !(a->tuning.strategy_mask & (1 << 21));  // (SRRG2_AMD_TUNE bit 21: atomics + a copy + a wait, as before)
    srrg2amd::launch_ingest_bbox(dsrc, sf, n, a->dim, s->fixed_raw.p, s->scalars.p + 10, s->scalars.p, s->scalars.p + 3,
                                 (int*) (s->scalars.p + 6), a->stream, rows ? s->scalars.p + 11 : nullptr, rows ? s->bbox_host : nullptr,
                                 s->bbox_seq, rows ? s->bbox_rows.p : nullptr, s->scalars.p + 7);
    s->scalars_self_init = rows && n > 0;
    s->bbox_polled       = rows;
  }
  else
    srrg2amd::launch_ingest(dsrc, sf, n, a->dim, s->fixed_raw.p, s->scalars.p + 10, 1, a->stream);  // [10] = |fixed|inf
  if (normals) {
    const float* nsrc;
    int nsf;
    if ((rc = stage_input(a, normals, ns, n, a->dim, mem, &nsrc, &nsf, (bytes_c + 63) / 64 * 64))) return rc;
    srrg2amd::launch_ingest(nsrc, nsf, n, a->dim, s->fixed_nrm_raw.p, s->scalars.p + 7, 0, a->stream);
  }
  s->nf                = n;
  s->fixed_has_normals = normals != nullptr;
  if (nn && (rc = build_grid(a, s, 0.f, /*have_bbox=*/n > 0))) return rc;  // projective: organised cloud as is
  // (SRRG2_MEM_DEVICE_KEPT: the caller keeps the buffer as it is until the next compute() has returned -- the reference's own
  // contract, its aligner holds raw pointers to the clouds -- so nothing waits here: the ingest is ordered in front of everything
  // compute() launches)
  if (mem != SRRG2_MEM_DEVICE_KEPT) HIP_TRY(hipStreamSynchronize(a->stream));
  s->has_fixed = true;
  return 0;
}

int srrg2_aligner_set_moving(srrg2_aligner_h a, int si, const float* coords, int cs, const float* normals, int ns, int n,
                             int mem) {
  int rc = check_slice(a, si, "set_moving");
  if (rc) return rc;
  if (n < 0 || (n > 0 && !coords)) return fail(SRRG2_E_INVALID, "set_moving: bad cloud");
  if (a->slices[si]->cfg.kind == SRRG2_SLICE_PRIOR)
    return fail(SRRG2_E_INVALID, "set_moving on a prior slice: use set_prior_measurement");
  if (a->slices[si]->alias_of >= 0)
    return fail(SRRG2_E_STATE, "set_moving: this slice shares the clouds of another slice (share_clouds): set them there");
  if ((rc = set_device(a))) return rc;
  const int32_t offsets[2] = {0, n};
  return upload_moving(a, si, coords, cs, normals, ns, offsets, 1, mem, /*wait=*/mem != SRRG2_MEM_DEVICE_KEPT);
}

int srrg2_aligner_share_clouds(srrg2_aligner_h a, int si, int source) {
  int rc = check_slice(a, si, "share_clouds");
  if (rc) return rc;
  if (source == -1) {  // back to clouds of its own (to be set again)
    Slice* s = a->slices[si];
    if (s->alias_of >= 0) {
      s->alias_of = -1;
      s->fixed_raw.release(); s->fixed_nrm_raw.release(); s->moving.release(); s->moving_nrm.release();
      s->moving_raw.release(); s->moving_nrm_raw.release(); s->pinf.release(); s->scalars.release();
      s->has_fixed = s->has_moving = false;
      s->nf = s->nm_total = 0;
    }
    return 0;
  }
  if ((rc = check_slice(a, source, "share_clouds"))) return rc;
  Slice *s = a->slices[si], *o = a->slices[source];
  if (si == source || s->cfg.kind == SRRG2_SLICE_PRIOR || o->cfg.kind == SRRG2_SLICE_PRIOR)
    return fail(SRRG2_E_INVALID, "share_clouds: two different cue slices");
  if (o->alias_of >= 0) return fail(SRRG2_E_INVALID, "share_clouds: the source slice shares another slice's clouds itself");
  for (Slice* t : a->slices)
    if (t->alias_of == si) return fail(SRRG2_E_INVALID, "share_clouds: other slices share this slice's clouds");
  if (s->cfg.finder != SRRG2_FINDER_PROJECTIVE || o->cfg.finder != SRRG2_FINDER_PROJECTIVE)
    return fail(SRRG2_E_UNSUPPORTED, "share_clouds: both slices must use the projective finder");
  if ((rc = set_device(a)) || (rc = quiesce_stream2(a))) return rc;
  HIP_TRY(hipStreamSynchronize(a->stream));
  s->alias_of = source;
  // (its own copies, if any, are dropped: from now on it views the source's buffers, re-borrowed at every compute())
  s->fixed_raw.release(); s->fixed_nrm_raw.release(); s->moving.release(); s->moving_nrm.release();
  s->moving_raw.release(); s->moving_nrm_raw.release(); s->pinf.release(); s->scalars.release();
  if (a->computed) a->records_state = 2;
  return 0;
}

int srrg2_aligner_set_point_shard(srrg2_aligner_h a, srrg2_reduce_fn fn, void* user, int64_t total_moving_points) {
  if (!a) return fail(SRRG2_E_INVALID, "null handle");
  if (fn && (total_moving_points <= 0 || total_moving_points > 0x7fffffffLL))
    return fail(SRRG2_E_INVALID, "total_moving_points must be in [1, 2^31)");
  a->reduce_fn   = fn;
  a->reduce_user = fn ? user : nullptr;
  a->shard_total = fn ? (long long) total_moving_points : 0;
  return 0;
}

int srrg2_amd_memcpy(void* dst, const void* src, size_t bytes, int kind, void* stream) {
  if (!dst || !src || kind < 0 || kind > 2) return fail(SRRG2_E_INVALID, "srrg2_amd_memcpy: bad arguments");
  const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToHost : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
  if (stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t) stream));
  } else {
    HIP_TRY(hipMemcpy(dst, src, bytes, k));
  }
  return 0;
}

int srrg2_amd_stream_synchronize(void* stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
  return 0;
}

int srrg2_amd_device_malloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return fail(SRRG2_E_INVALID, "srrg2_amd_device_malloc: bad arguments");
  *out = nullptr;
  HIP_TRY(hipMalloc(out, bytes));
  return 0;
}

int srrg2_amd_device_free(void* p) {
  if (p) HIP_TRY(hipFree(p));
  return 0;
}

int srrg2_aligner_set_sensor_in_robot(srrg2_aligner_h a, int si, const float* T) {
  int rc = check_slice(a, si, "set_sensor_in_robot");
  if (rc) return rc;
  if (!T) return fail(SRRG2_E_INVALID, "set_sensor_in_robot: null transform");
  for (int i = 0; i < a->tsize; ++i)
    if (!std::isfinite(T[i])) return fail(SRRG2_E_INVALID, "set_sensor_in_robot: non-finite transform");
  // (run_compute inverts it into robot_in_sensor for every compute(): aligner_slice_processor_impl.cpp:24-35)
  std::memcpy(a->slices[si]->cfg.sensor_in_robot, T, sizeof(float) * a->tsize);
  return 0;
}

int srrg2_aligner_set_prior_measurement(srrg2_aligner_h a, int si, const float* T) {
  int rc = check_slice(a, si, "set_prior_measurement");
  if (rc) return rc;
  if (!T) return fail(SRRG2_E_INVALID, "set_prior_measurement: null transform");
  Slice* s = a->slices[si];
  if (s->cfg.kind != SRRG2_SLICE_PRIOR) return fail(SRRG2_E_INVALID, "set_prior_measurement: not a prior slice");
  std::memcpy(s->prior_Z, T, sizeof(float) * a->tsize);
  s->has_prior = true;
  return 0;
}

int srrg2_aligner_set_moving_in_fixed(srrg2_aligner_h a, const float* T) {
  if (!a || !T) return fail(SRRG2_E_INVALID, "set_moving_in_fixed: bad arguments");
  std::memcpy(a->X, T, sizeof(float) * a->tsize);
  return 0;
}

int srrg2_aligner_get_moving_in_fixed(srrg2_aligner_h a, float* T) {
  if (!a || !T) return fail(SRRG2_E_INVALID, "get_moving_in_fixed: bad arguments");
  std::memcpy(T, a->X, sizeof(float) * a->tsize);
  return 0;
}

int srrg2_aligner_compute(srrg2_aligner_h a, int* status_out) {
  if (!a) return fail(SRRG2_E_INVALID, "compute: null handle");
  int rc = run_compute(a, 1, nullptr, a->X);
  if (rc) return rc;
  if (status_out) *status_out = a->status;
  return 0;
}

int srrg2_aligner_status(srrg2_aligner_h a, int* s) {
  if (!a || !s) return fail(SRRG2_E_INVALID, "status: bad arguments");
  *s = a->status;
  return 0;
}

int srrg2_aligner_get_iteration_stats(srrg2_aligner_h a, srrg2_iteration_stats* buf, int* n) {
  if (!a || !n) return fail(SRRG2_E_INVALID, "get_iteration_stats: bad arguments");
  const int have = (int) a->last_stats.size();
  if (buf) std::memcpy(buf, a->last_stats.data(), sizeof(srrg2_iteration_stats) * (size_t) std::min(*n, have));
  *n = have;
  return 0;
}

int srrg2_aligner_get_information(srrg2_aligner_h a, float* H) {
  if (!a || !H) return fail(SRRG2_E_INVALID, "get_information: null argument");
  if (!a->computed) return fail(SRRG2_E_STATE, "get_information: no compute() yet");
  std::memcpy(H, a->last_H, sizeof(float) * (size_t) a->dof * a->dof);
  return 0;
}

int srrg2_aligner_num_correspondences(srrg2_aligner_h a, int* n) {
  if (!a || !n) return fail(SRRG2_E_INVALID, "num_correspondences: bad arguments");
  int tot = 0;  // multi_aligner_impl.cpp:275-285
  for (size_t si = 0; si < a->slices.size(); ++si) {
    int c = a->slices[si]->cfg.kind == SRRG2_SLICE_PRIOR ? 1 : (a->computed ? a->last_ncorr[si] : 0);
    if (c >= 0) tot += c;
  }
  *n = tot;
  return 0;
}

// dense per-point arrays of the LAST problem -> host, compacted in ascending moving index
static int fetch_dense(srrg2_aligner* a, int si, std::vector<int>& cf, std::vector<float>& cr, std::vector<uint8_t>& cst,
                       int* moff_out) {
  Slice* s = a->slices[si];
  int rc;
  if ((rc = set_device(a))) return rc;
  if (!a->computed || s->cfg.kind == SRRG2_SLICE_PRIOR || !s->has_moving ||
      (a->records_state == 2 && (s->cfg.finder == SRRG2_FINDER_NN_GATED || s->cfg.finder == SRRG2_FINDER_PROJECTIVE))) {
    cf.clear(); cr.clear(); cst.clear();
    *moff_out = 0;
    return 0;
  }
  if ((rc = materialize_records(a))) return rc;
  if (s->alias_of >= 0 && s->alias_of < (int) a->slices.size()) {  // (the owner may have re-allocated its clouds since)
    Slice* o = a->slices[s->alias_of];
    s->moving.borrow(o->moving);
    s->nm_total = std::min(s->nm_total, o->nm_total);
  }
  // problem K-1 of the last run
  int moff = 0, nm = s->nm_total;
  if (a->K > 1) {
    std::vector<ProblemDev> pd((size_t) a->K);
    HIP_TRY(hipMemcpy(pd.data(), a->probs.p + (size_t) si * a->K, sizeof(ProblemDev) * (size_t) a->K, hipMemcpyDeviceToHost));
    moff = pd[a->K - 1].moff;
    nm   = pd[a->K - 1].nm;
  }
  cf.resize((size_t) nm); cr.resize((size_t) nm); cst.resize((size_t) nm);
  if (nm > 0) {
    // the kernels write these arrays in the (Morton) sorted order of the moving cloud; moving[g].w = caller's index
    std::vector<int> sf((size_t) nm);
    std::vector<float> sr((size_t) nm);
    std::vector<uint8_t> sst((size_t) nm);
    std::vector<float4> mp((size_t) nm);
    HIP_TRY(hipMemcpy(sf.data(), s->corr_fixed.p + moff, sizeof(int) * (size_t) nm, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sr.data(), s->corr_resp.p + moff, sizeof(float) * (size_t) nm, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sst.data(), s->corr_stat.p + moff, (size_t) nm, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mp.data(), s->moving.p + moff, sizeof(float4) * (size_t) nm, hipMemcpyDeviceToHost));
    for (int g = 0; g < nm; ++g) {
      int ci;
      std::memcpy(&ci, &mp[(size_t) g].w, sizeof(int));
      if (ci < 0 || ci >= nm) return fail(SRRG2_E_STATE, "correspondences: corrupt moving index");
      cf[(size_t) ci]  = sf[(size_t) g];
      cr[(size_t) ci]  = sr[(size_t) g];
      cst[(size_t) ci] = sst[(size_t) g];
    }
  }
  *moff_out = moff;
  return 0;
}

}  // extern "C" (reopened below)
namespace srrg2amd {
int aligner_slice_view(srrg2_aligner_s* a, int si, AlignerSliceView* v) {
  int rc = check_slice(a, si, "aligner_slice_view");
  if (rc) return rc;
  Slice* s = a->slices[si];
  if (!a->computed || a->K != 1 || s->cfg.kind == SRRG2_SLICE_PRIOR || !s->has_moving || !s->has_fixed)
    return fail(SRRG2_E_STATE, "aligner_slice_view: needs a cue slice after a single-problem compute()");
  if (a->records_state == 2) return fail(SRRG2_E_STATE, "aligner_slice_view: the clouds changed since the last compute()");
  if ((rc = materialize_records(a, /*host_wait=*/false))) return rc;
  // (the consumer's stream waits for this event: everything the view points to is complete behind it)
  HIP_TRY(hipEventRecord(a->ev_records, a->stream));
  v->ready_event   = (void*) a->ev_records;
  v->moving_sorted = s->moving.p;
  v->corr_fixed    = s->corr_fixed.p;
  v->corr_resp     = s->corr_resp.p;
  v->corr_stat     = s->corr_stat.p;
  v->nm            = s->nm_total;
  v->nf            = s->nf;
  v->prune         = a->params.keep_only_inlier_correspondences && a->status == SRRG2_SUCCESS;
  v->device        = a->device;
  return 0;
}
}  // namespace srrg2amd
extern "C" {

// given-correspondences slices: the pairs of the LAST problem and their factor status
static int fetch_given(srrg2_aligner* a, int si, std::vector<srrg2_correspondence>& pairs, std::vector<uint8_t>& st) {
  Slice* s = a->slices[si];
  pairs.clear();
  st.clear();
  int rc;
  if ((rc = set_device(a))) return rc;
  if (!a->computed || (int) s->h_gcorr_off.size() != a->K + 1) return 0;
  const int c0 = s->h_gcorr_off[(size_t) a->K - 1], c1 = s->h_gcorr_off[(size_t) a->K];
  pairs.assign(s->h_gcorr.begin() + c0, s->h_gcorr.begin() + c1);
  st.resize((size_t) (c1 - c0));
  if (c1 > c0) HIP_TRY(hipMemcpy(st.data(), s->gcorr_stat.p + c0, (size_t) (c1 - c0), hipMemcpyDeviceToHost));
  return 0;
}

int srrg2_aligner_get_correspondences(srrg2_aligner_h a, int si, srrg2_correspondence* buf, int* n) {
  int rc = check_slice(a, si, "get_correspondences");
  if (rc) return rc;
  if (!n) return fail(SRRG2_E_INVALID, "get_correspondences: null count");
  if (a->slices[si]->cfg.finder == SRRG2_FINDER_CORRESPONDENCES && a->slices[si]->cfg.kind != SRRG2_SLICE_PRIOR) {
    std::vector<srrg2_correspondence> pairs;
    std::vector<uint8_t> st;
    if ((rc = fetch_given(a, si, pairs, st))) return rc;
    const bool prune = a->params.keep_only_inlier_correspondences && a->status == SRRG2_SUCCESS;
    int cnt = 0;
    for (size_t i = 0; i < pairs.size(); ++i) {
      if (prune && st[i] != SRRG2_FACTOR_INLIER) continue;
      if (buf && cnt < *n) buf[cnt] = pairs[i];
      ++cnt;
    }
    *n = cnt;
    return 0;
  }
  std::vector<int> cf;
  std::vector<float> cr;
  std::vector<uint8_t> cst;
  int moff;
  if ((rc = fetch_dense(a, si, cf, cr, cst, &moff))) return rc;
  const bool prune = a->params.keep_only_inlier_correspondences && a->status == SRRG2_SUCCESS;
  int cnt = 0;
  for (size_t i = 0; i < cf.size(); ++i) {
    if (cf[i] < 0) continue;
    if (prune && cst[i] != SRRG2_FACTOR_INLIER) continue;  // _pruneCorrespondences, multi_aligner_impl.cpp:243-250
    if (buf && cnt < *n) {
      buf[cnt].fixed_idx  = cf[i];
      buf[cnt].moving_idx = (int) i;
      buf[cnt].response   = cr[i];
    }
    ++cnt;
  }
  *n = cnt;
  return 0;
}

int srrg2_aligner_get_factor_status(srrg2_aligner_h a, int si, uint8_t* buf, int* n) {
  int rc = check_slice(a, si, "get_factor_status");
  if (rc) return rc;
  if (!n) return fail(SRRG2_E_INVALID, "get_factor_status: null count");
  if (a->slices[si]->cfg.finder == SRRG2_FINDER_CORRESPONDENCES && a->slices[si]->cfg.kind != SRRG2_SLICE_PRIOR) {
    std::vector<srrg2_correspondence> pairs;
    std::vector<uint8_t> st;
    if ((rc = fetch_given(a, si, pairs, st))) return rc;
    const bool prune = a->params.keep_only_inlier_correspondences && a->status == SRRG2_SUCCESS;
    int cnt = 0;
    for (size_t i = 0; i < st.size(); ++i) {
      if (prune && st[i] != SRRG2_FACTOR_INLIER) continue;
      if (buf && cnt < *n) buf[cnt] = st[i];
      ++cnt;
    }
    *n = cnt;
    return 0;
  }
  std::vector<int> cf;
  std::vector<float> cr;
  std::vector<uint8_t> cst;
  int moff;
  if ((rc = fetch_dense(a, si, cf, cr, cst, &moff))) return rc;
  const bool prune = a->params.keep_only_inlier_correspondences && a->status == SRRG2_SUCCESS;
  int cnt = 0;
  for (size_t i = 0; i < cf.size(); ++i) {
    if (cf[i] < 0) continue;
    if (prune && cst[i] != SRRG2_FACTOR_INLIER) continue;
    if (buf && cnt < *n) buf[cnt] = cst[i];
    ++cnt;
  }
  *n = cnt;
  return 0;
}

int srrg2_aligner_compute_batch(srrg2_aligner_h a, int K, const float* coords, int cs, const float* normals, int ns,
                                const int32_t* offsets, int mem, const float* guesses, srrg2_batch_result* results) {
  if (!a || K < 0 || !offsets || !guesses || !results) return fail(SRRG2_E_INVALID, "compute_batch: bad arguments");
  if (K == 0) return 0;
  if (a->slices.empty()) return fail(SRRG2_E_STATE, "compute_batch: no slices");
  int cue = -1;
  for (size_t si = 0; si < a->slices.size(); ++si)
    if (a->slices[si]->cfg.kind != SRRG2_SLICE_PRIOR) {
      cue = (int) si;
      break;
    }
  if (cue != 0) return fail(SRRG2_E_UNSUPPORTED, "compute_batch: slice 0 must be the cue slice");
  if (K > 65535) return fail(SRRG2_E_INVALID, "compute_batch: at most 65535 alignments per call (grid.y)");
  if (offsets[0] < 0) return fail(SRRG2_E_INVALID, "compute_batch: negative offset");
  for (int k = 0; k < K; ++k)
    if (offsets[k + 1] < offsets[k]) return fail(SRRG2_E_INVALID, "compute_batch: offsets must not decrease");
  if (offsets[K] > offsets[0] && !coords) return fail(SRRG2_E_INVALID, "compute_batch: null cloud");
  int rc;
  if ((rc = set_device(a))) return rc;
  const auto t_up0 = std::chrono::steady_clock::now();
  // Pipelined batches: the alignments are independent, so the batch runs as P parts on P streams -- while one part's
  // control step (one workgroup per alignment on an otherwise idle chip, ~10 us per iteration, + the kernel boundaries) and
  // Morton sort pass, the other parts' pass kernels have the chip.  Same kernels on the same data: the same bits.
  // tuning.batch_pipeline: 0 = never, 1 = two halves for every batch of >= 2, P >= 2: P parts (batches of >= P), -1 = automatic
  int nparts = 0, pbegin[srrg2_aligner::MAX_PARTS + 1] = {0};
  {
    const srrg2_aligner_tuning& tn = a->tuning;
    int max_nm = 0;
    for (int k = 0; k < K; ++k) max_nm = std::max(max_nm, offsets[k + 1] - offsets[k]);
    bool one_cue = true;
    for (size_t si = 1; si < a->slices.size(); ++si)
      if (a->slices[si]->cfg.kind != SRRG2_SLICE_PRIOR) one_cue = false;
    int want = tn.batch_pipeline == 0 ? 1 : (tn.batch_pipeline == 1 ? 2 : (tn.batch_pipeline > 1 ? tn.batch_pipeline : (K >= 8 ? 2 : 1)));
    want     = std::min(std::min(want, (int) srrg2_aligner::MAX_PARTS), K);
    if (want >= 2 && one_cue && a->slices[0]->cfg.finder == SRRG2_FINDER_NN_GATED && max_nm > tn.small_max_points &&
        !tn.fast_batch_queue && !a->reduce_fn && !a->profile && a->timeline_path.empty()) {
      nparts = want;
      // (the parts' streams are created when a batch first wants them: an application with hundreds of aligners -- per-thread
      // relocalisers -- that never runs pipelined batches owns one stream per aligner, not eight)
      for (int p = 1; p < nparts; ++p)
        if (!a->pstream[p] && hipStreamCreateWithFlags(&a->pstream[p], hipStreamNonBlocking) != hipSuccess) {
          (void) hipGetLastError();
          a->pstream[p] = nullptr;
          nparts        = p >= 2 ? p : 0;  // (fewer parts, or none)
          break;
        }
      for (int p = 0; p <= nparts; ++p) pbegin[p] = (int) ((long long) K * p / nparts);  // (contiguous, sizes differ by <= 1)
    }
  }
  if ((rc = upload_moving(a, 0, coords, cs, normals, ns, offsets, K, mem, /*wait=*/false, nparts, pbegin))) return rc;
  if (a->hosttime)
    std::fprintf(stderr, "compute_batch: upload_moving (enqueue) %.1f us\n",
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_up0).count());
  if ((rc = run_compute(a, K, offsets, guesses))) {
    // (run_compute may have failed before anything drained the stream: the ingest must have finished reading the
    // caller's buffer before this call returns, error or not)
    (void) hipStreamSynchronize(a->stream);
    for (Slice* sl : a->slices) sl->ms_pending = false;
    return rc;
  }
  const int slots = a->max_stats;
  for (int k = 0; k < K; ++k) {
    const ProblemOut& o = a->outs_host[k];
    std::memset(&results[k], 0, sizeof(results[k]));
    std::memcpy(results[k].moving_in_fixed, o.X, sizeof(float) * a->tsize);
    results[k].status         = o.status;
    results[k].num_iterations = o.nstats;
    if (o.nstats > 0) results[k].last = a->stats_host[(size_t) k * slots + std::min(o.nstats, slots) - 1];
    int tot = 0;  // multi_aligner_impl.cpp:275-285, after pruning (k_icp_finalize)
    for (size_t si = 0; si < a->slices.size(); ++si) {
      const int c = a->slices[si]->cfg.kind == SRRG2_SLICE_PRIOR ? 1 : o.ncorr[si];
      if (c >= 0) tot += c;
    }
    results[k].num_correspondences = tot;
    std::memcpy(results[k].information, o.H, sizeof(o.H));
  }
  // The handle's observable state is that of the LAST alignment (K x {set_moving; set_moving_in_fixed; compute()}):
  // status, estimate, statistics, correspondences and factor status of problem K - 1 stay readable.  The bound moving
  // cloud, however, is the concatenation of the batch (Slice::moving_is_batch): a later plain compute() must bind its
  // own cloud first and says so (SRRG2_E_STATE) instead of aligning K clouds as one.
  return 0;
}

int srrg2_aligner_set_correspondences(srrg2_aligner_h a, int si, const srrg2_correspondence* corr, int n) {
  int rc = check_slice(a, si, "set_correspondences");
  if (rc) return rc;
  if (n < 0 || (n > 0 && !corr)) return fail(SRRG2_E_INVALID, "set_correspondences: bad arguments");
  Slice* s = a->slices[si];
  if (s->cfg.finder != SRRG2_FINDER_CORRESPONDENCES)
    return fail(SRRG2_E_INVALID, "set_correspondences: the slice's finder kind is not SRRG2_FINDER_CORRESPONDENCES");
  s->h_gcorr.assign(corr, corr + n);
  s->h_gcorr_off = {0, n};
  return 0;
}

int srrg2_aligner_compute_batch_correspondences(srrg2_aligner_h a, int K, const float* coords, int cs, const float* normals,
                                                int ns, const int32_t* offsets, int mem, const srrg2_correspondence* corr,
                                                const int32_t* corr_offsets, const float* guesses,
                                                srrg2_batch_result* results) {
  if (!a || K < 0 || !offsets || !corr_offsets || !guesses || !results)
    return fail(SRRG2_E_INVALID, "compute_batch_correspondences: bad arguments");
  if (K == 0) return 0;
  if (a->slices.empty() || a->slices[0]->cfg.finder != SRRG2_FINDER_CORRESPONDENCES)
    return fail(SRRG2_E_STATE, "compute_batch_correspondences: slice 0 must be a given-correspondences slice");
  for (int k = 0; k < K; ++k)
    if (corr_offsets[k + 1] < corr_offsets[k]) return fail(SRRG2_E_INVALID, "compute_batch_correspondences: bad offsets");
  if (corr_offsets[K] > corr_offsets[0] && !corr) return fail(SRRG2_E_INVALID, "compute_batch_correspondences: null pairs");
  Slice* s = a->slices[0];
  s->h_gcorr.assign(corr + corr_offsets[0], corr + corr_offsets[K]);
  s->h_gcorr_off.resize((size_t) K + 1);
  for (int k = 0; k <= K; ++k) s->h_gcorr_off[(size_t) k] = corr_offsets[k] - corr_offsets[0];
  return srrg2_aligner_compute_batch(a, K, coords, cs, normals, ns, offsets, mem, guesses, results);
}

int srrg2_multi_gpu_init(int local_rank, int* device_out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(SRRG2_E_NO_DEVICE, "multi_gpu_init: no HIP device visible");
  if (local_rank < 0) return fail(SRRG2_E_INVALID, "multi_gpu_init: negative local rank");
  const int dev = local_rank % n;
  HIP_TRY(hipSetDevice(dev));
  HIP_TRY(hipFree(nullptr));  // (creates the context: a broken device fails here, not in the first kernel)
  if (device_out) *device_out = dev;
  return 0;
}

int srrg2_multi_gpu_shard_count(int K, int world, int rank) {
  if (K < 0 || world < 1 || rank < 0 || rank >= world) return fail(SRRG2_E_INVALID, "shard_count: bad arguments");
  return K > rank ? (K - rank + world - 1) / world : 0;
}

int srrg2_multi_gpu_shard_indices(int K, int world, int rank, int32_t* out) {
  const int n = srrg2_multi_gpu_shard_count(K, world, rank);
  if (n < 0) return n;
  if (n > 0 && !out) return fail(SRRG2_E_INVALID, "shard_indices: null output");
  for (int i = 0; i < n; ++i) out[i] = rank + i * world;
  return n;
}

int srrg2_multi_gpu_pack_record(int k, int variable_kind, const srrg2_batch_result* r, double* rec) {
  if (!r || !rec || variable_kind < 0 || variable_kind > 2) return fail(SRRG2_E_INVALID, "pack_record: bad arguments");
  const int tsize = variable_kind == SRRG2_SE2_RIGHT ? 9 : 12, D = variable_kind == SRRG2_SE2_RIGHT ? 3 : 6;
  for (int i = 0; i < SRRG2_RECORD_FLOATS; ++i) rec[i] = 0.0;
  for (int i = 0; i < tsize; ++i) rec[i] = (double) r->moving_in_fixed[i];
  rec[12] = r->status;
  rec[13] = r->num_iterations;
  rec[14] = r->last.num_inliers;
  rec[15] = r->last.num_outliers;
  rec[16] = r->num_correspondences;
  rec[17] = (double) r->last.chi_inliers;
  rec[18] = k;
  rec[19] = D;
  int t   = 20;
  for (int a_ = 0; a_ < D; ++a_)
    for (int b_ = a_; b_ < D; ++b_) rec[t++] = (double) r->information[a_ * D + b_];
  return 0;
}

int srrg2_multi_gpu_unpack_record(const double* rec, int variable_kind, int* k_out, srrg2_batch_result* r) {
  if (!r || !rec || variable_kind < 0 || variable_kind > 2) return fail(SRRG2_E_INVALID, "unpack_record: bad arguments");
  const int tsize = variable_kind == SRRG2_SE2_RIGHT ? 9 : 12, D = variable_kind == SRRG2_SE2_RIGHT ? 3 : 6;
  std::memset(r, 0, sizeof(*r));
  for (int i = 0; i < tsize; ++i) r->moving_in_fixed[i] = (float) rec[i];
  r->status                   = (int) rec[12];
  r->num_iterations           = (int) rec[13];
  r->last.num_inliers         = (int) rec[14];
  r->last.num_outliers        = (int) rec[15];
  r->num_correspondences      = (int) rec[16];
  r->last.num_correspondences = (int) rec[16];
  r->last.chi_inliers         = (float) rec[17];
  if (k_out) *k_out = (int) rec[18];
  int t = 20;
  for (int a_ = 0; a_ < D; ++a_)
    for (int b_ = a_; b_ < D; ++b_) {
      r->information[a_ * D + b_] = (float) rec[t];
      r->information[b_ * D + a_] = (float) rec[t++];
    }
  return 0;
}

int srrg2_aligner_profile_enable(srrg2_aligner_h a, int enable) {
  if (!a) return fail(SRRG2_E_INVALID, "profile_enable: null handle");
  a->profile = enable != 0;
  return 0;
}

int srrg2_aligner_profile_get(srrg2_aligner_h a, double* ms, int64_t* launches, int reset) {
  if (!a) return fail(SRRG2_E_INVALID, "profile_get: null handle");
  if (ms) *ms = a->prof_ms;
  if (launches) *launches = a->prof_launches;
  if (reset) {
    a->prof_ms       = 0.0;
    a->prof_launches = 0;
  }
  return 0;
}

int srrg2_aligner_last_compute_path(srrg2_aligner_h a, int32_t* flags_out) {
  if (!a || !flags_out) return fail(SRRG2_E_INVALID, "last_compute_path: null argument");
  *flags_out = a->computed ? a->last_path : 0;
  return 0;
}

}  // extern "C"
