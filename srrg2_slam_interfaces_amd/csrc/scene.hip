// scene.hip -- scene slices kept in HBM between frames: ball clipping and correspondence-based merging
// (SURVEY.md section 8f row 2).  Replaces, behind the C ABI of include/srrg2_slam_amd.h:
//   MergerCorrespondenceHomo_::compute()   S/mapping/merger_correspondence_homo_impl.cpp:11-125
//   SceneClipper_::compute() (interface)   S/mapping/scene_clipper.h:17-122
//   the index flip + local->global mapping of TrackerSliceProcessor_::merge()
//                                          S/trackers/tracker_slice_processor_impl.cpp:160-186
// A scene is two float4 arrays (points {x, y, z, 0}, normals) with spare capacity.  All kernels are one thread per
// point or correspondence, coalesced, HBM bound: clip = 2 passes over the scene (flag+count, scatter) around an
// exclusive scan; merge = one pass over the correspondences + (if the merge target was not reached) flag/scan/scatter
// of the measurement.  The reference merges sequentially; results are identical because
//   - a scene point hit by ONE correspondence is independent of all others (the common case: the tracker's
//     correspondences come through an injective local->global map),
//   - scene points hit by several correspondences are replayed in correspondence order by one thread per scene point
//     (k_merge_dups: the duplicates are grouped by a radix sort),
//   - appends keep measurement order (stable compaction by exclusive scan).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "det_math.h"
#include "device_types.h"
#include "host_util.h"
#include "kernels.h"

using srrg2amd::DevBuf;
using srrg2amd::fail;

struct srrg2_scene {
  int dim = 3, device = 0;
  hipStream_t stream = nullptr;
  DevBuf<float4> pts, nrm;
  int n            = 0;
  bool has_normals = false;
  DevBuf<int> gidx;  // local -> global indices of the last clip into this scene
  int ng = 0;
  // scratch
  DevBuf<int> flags, scan_sums, counts, dup_list;
  DevBuf<unsigned char> merged;
  DevBuf<srrg2_correspondence> corr;
  DevBuf<unsigned long long> dup_keys;  // (scene index << 32 | correspondence index) of the duplicates: unsorted, sorted
  DevBuf<char> sort_tmp;
  DevBuf<char> staging;
  int* scalars = nullptr;  // pinned host mirror of dscalars
  DevBuf<int> dscalars;    // device: [0] scan total, [1] num_merged, [2] error flag, [3] duplicates seen, [4] ncorr
};

namespace {

struct Xf {  // rows of [R|t]; SE(2) spread into the same slots
  float m[12];
};

Xf load_transform(int dim, const float* T) {
  Xf x;
  if (dim == 3) {
    std::memcpy(x.m, T, sizeof(x.m));
  } else {
    const float v[12] = {T[0], T[1], 0.f, T[2], T[3], T[4], 0.f, T[5], 0.f, 0.f, 1.f, 0.f};
    std::memcpy(x.m, v, sizeof(v));
  }
  return x;
}

__device__ __forceinline__ bool valid_point(int dim, const float4 p) {
  return isfinite(p.x) && isfinite(p.y) && (dim == 2 || isfinite(p.z));
}

__device__ __forceinline__ float4 xform_point(int dim, const Xf& M, const float4 p) {
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (dim == 3) {
    q.x = ((M.m[0] * p.x + M.m[1] * p.y) + M.m[2] * p.z) + M.m[3];
    q.y = ((M.m[4] * p.x + M.m[5] * p.y) + M.m[6] * p.z) + M.m[7];
    q.z = ((M.m[8] * p.x + M.m[9] * p.y) + M.m[10] * p.z) + M.m[11];
  } else {
    q.x = (M.m[0] * p.x + M.m[1] * p.y) + M.m[3];
    q.y = (M.m[4] * p.x + M.m[5] * p.y) + M.m[7];
  }
  return q;
}

__device__ __forceinline__ float4 rotate_normal(int dim, const Xf& M, const float4 n) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (dim == 3) {
    r.x = (M.m[0] * n.x + M.m[1] * n.y) + M.m[2] * n.z;
    r.y = (M.m[4] * n.x + M.m[5] * n.y) + M.m[6] * n.z;
    r.z = (M.m[8] * n.x + M.m[9] * n.y) + M.m[10] * n.z;
  } else {
    r.x = M.m[0] * n.x + M.m[1] * n.y;
    r.y = M.m[4] * n.x + M.m[5] * n.y;
  }
  return r;
}

// ---- clip ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool clip_keep(int dim, const Xf& L, float range2, const float4 p, float4& q) {
  if (!valid_point(dim, p)) return false;
  q              = xform_point(dim, L, p);
  const float d2 = (q.x * q.x + q.y * q.y) + q.z * q.z;
  return d2 <= range2;
}

__global__ void k_clip_flag(int dim, Xf L, float range2, const float4* __restrict__ pts, int n, int* __restrict__ flags) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 q;
    flags[i] = clip_keep(dim, L, range2, pts[i], q) ? 1 : 0;
  }
}

__global__ void k_clip_scatter(int dim, Xf L, float range2, const float4* __restrict__ pts, const float4* __restrict__ nrm,
                               int n, const int* __restrict__ offset, float4* __restrict__ out_pts,
                               float4* __restrict__ out_nrm, int* __restrict__ gidx, int cap) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 q;
    if (!clip_keep(dim, L, range2, pts[i], q)) continue;
    const int k = offset[i];
    if (k >= cap) continue;  // (a scatter launched before the host knew the total: the caller repeats it with room for all)
    out_pts[k]  = q;
    out_nrm[k]  = nrm ? rotate_normal(dim, L, nrm[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    gidx[k]     = i;
  }
}

// block sum -> ONE atomic per block (same-address atomics serialise at tens of ns each; blockDim = 256)
__device__ __forceinline__ void block_add(int v, int* target) {
  __shared__ int red[4];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = (red[0] + red[1]) + (red[2] + red[3]);
    if (t) atomicAdd(target, t);
  }
}

// ---- merge --------------------------------------------------------------------------------------------------
// one correspondence: merger_correspondence_homo_impl.cpp:55-76.  Returns true if merged.
__device__ __forceinline__ bool merge_one(int dim, const Xf& M, float max_response, float max_d2, float4* scene_pts,
                                          float4* scene_nrm, const float4* meas_pts, const float4* meas_nrm, int s, int m,
                                          float response) {
  if (!(response < max_response)) return false;  // :60
  const float4 ps = scene_pts[s];
  const float4 q  = xform_point(dim, M, meas_pts[m]);  // :62-63
  const float dx = q.x - ps.x, dy = q.y - ps.y, dz = q.z - ps.z;
  const float d2 = (dx * dx + dy * dy) + dz * dz;  // :66-67
  if (!(d2 < max_d2)) return false;                // :69
  // :71 point_scene = point_meas (all fields: the normal stays in the measurement frame), :74 mean coordinates
  if (scene_nrm) scene_nrm[s] = meas_nrm ? meas_nrm[m] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 r = make_float4((q.x + ps.x) * 0.5f, (q.y + ps.y) * 0.5f, dim == 3 ? (q.z + ps.z) * 0.5f : 0.f, 0.f);
  scene_pts[s] = r;
  return true;
}

__global__ void k_merge_count(const srrg2_correspondence* __restrict__ corr, int ncorr, int n_scene, int n_meas,
                              int* __restrict__ counts, int* __restrict__ scalars) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncorr; c += gridDim.x * blockDim.x) {
    const srrg2_correspondence k = corr[c];
    if (k.fixed_idx < 0 || k.fixed_idx >= n_scene || k.moving_idx < 0 || k.moving_idx >= n_meas) {
      scalars[2] = 1;  // the reference asserts (:53-54); here: error
      continue;
    }
    if (atomicAdd(&counts[k.fixed_idx], 1) >= 1) scalars[3] = 1;  // some scene point is hit more than once
  }
}

// scene points hit exactly once: independent of every other correspondence
__global__ void k_merge_apply(int dim, Xf M, float max_response, float max_d2, const srrg2_correspondence* __restrict__ corr,
                              int ncorr, const int* __restrict__ counts, float4* scene_pts, float4* scene_nrm,
                              const float4* __restrict__ meas_pts, const float4* __restrict__ meas_nrm,
                              unsigned char* __restrict__ merged, int* __restrict__ dup_flags) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncorr; c += gridDim.x * blockDim.x) {
    const srrg2_correspondence k = corr[c];
    const bool dup               = counts[k.fixed_idx] > 1;
    if (dup_flags) dup_flags[c] = dup ? 1 : 0;
    if (dup) continue;
    if (merge_one(dim, M, max_response, max_d2, scene_pts, scene_nrm, meas_pts, meas_nrm, k.fixed_idx, k.moving_idx, k.response))
      merged[k.moving_idx] = 1;
  }
}

__global__ void k_compact_dups(const int* __restrict__ dup_flags_scanned, const srrg2_correspondence* __restrict__ corr,
                               int ncorr, const int* __restrict__ counts, int* __restrict__ dup_list) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncorr; c += gridDim.x * blockDim.x)
    if (counts[corr[c].fixed_idx] > 1) dup_list[dup_flags_scanned[c]] = c;
}

// scene points hit several times: replay their correspondences in order (:51 "for all correspondences").  Different
// scene points do not interact, so the duplicates are grouped by scene point -- a radix sort of the keys
// (scene index << 32 | correspondence index): groups come out contiguous, each in correspondence order -- and ONE THREAD
// PER DISTINCT SCENE POINT replays its group (a single thread walking the whole list paid one memory latency per entry).
__global__ void k_dup_keys(const srrg2_correspondence* __restrict__ corr, const int* __restrict__ dup_list, int ndup,
                           unsigned long long* __restrict__ keys) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ndup; t += gridDim.x * blockDim.x) {
    const int c = dup_list[t];
    keys[t]     = ((unsigned long long) (unsigned) corr[c].fixed_idx << 32) | (unsigned) c;
  }
}

__global__ void k_merge_dups(int dim, Xf M, float max_response, float max_d2, const srrg2_correspondence* __restrict__ corr,
                             const unsigned long long* __restrict__ keys, int ndup, float4* scene_pts, float4* scene_nrm,
                             const float4* __restrict__ meas_pts, const float4* __restrict__ meas_nrm,
                             unsigned char* __restrict__ merged) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ndup; t += gridDim.x * blockDim.x) {
    const unsigned s = (unsigned) (keys[t] >> 32);
    if (t > 0 && (unsigned) (keys[t - 1] >> 32) == s) continue;  // not the first of its group
    for (int j = t; j < ndup && (unsigned) (keys[j] >> 32) == s; ++j) {
      const srrg2_correspondence k = corr[(unsigned) keys[j]];
      if (merge_one(dim, M, max_response, max_d2, scene_pts, scene_nrm, meas_pts, meas_nrm, k.fixed_idx, k.moving_idx, k.response))
        merged[k.moving_idx] = 1;
    }
  }
}

// the tracker's path: correspondences straight from the aligner's per-point arrays (sorted order of its moving cloud =
// the clipped scene), flipped and mapped to the global scene (tracker_slice_processor_impl.cpp:175-181).  The map is
// injective, so every scene point is hit at most once: no ordering to respect.
__global__ void k_merge_from_aligner(int dim, Xf M, float max_response, float max_d2, const float4* __restrict__ moving_sorted,
                                     const int* __restrict__ corr_fixed, const float* __restrict__ corr_resp,
                                     const unsigned char* __restrict__ corr_stat, int prune, int nm,
                                     const int* __restrict__ gidx, int n_scene, int n_meas, float4* scene_pts,
                                     float4* scene_nrm, const float4* __restrict__ meas_pts,
                                     const float4* __restrict__ meas_nrm, unsigned char* __restrict__ merged,
                                     int* __restrict__ scalars) {
  int ncorr = 0, nmerged = 0;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < nm; g += gridDim.x * blockDim.x) {
    const int m = corr_fixed[g];  // aligner "fixed" = the measurement
    if (m < 0) continue;
    if (prune && corr_stat[g] != SRRG2_FACTOR_INLIER) continue;  // as srrg2_aligner_get_correspondences
    const int local = __float_as_int(moving_sorted[g].w);        // aligner "moving" = the clipped scene
    const int s     = gidx[local];
    if (m >= n_meas || s < 0 || s >= n_scene) {
      scalars[2] = 1;
      continue;
    }
    ++ncorr;
    if (merge_one(dim, M, max_response, max_d2, scene_pts, scene_nrm, meas_pts, meas_nrm, s, m, corr_resp[g])) {
      // (several scene points may merge into one measurement point: its flag byte is set through an atomic OR on the word that
      // holds it, and whoever finds it clear counts it -- the number of DISTINCT merged points without a second pass over the
      // flags, k_count_merged, and its launch, copy and wait: a tracker's merge 0.088 -> ~0.07 ms)
      unsigned* w        = reinterpret_cast<unsigned*>(merged) + (m >> 2);
      const unsigned bit = 1u << ((m & 3) * 8);
      if (!(atomicOr(w, bit) & bit)) ++nmerged;
    }
  }
  block_add(ncorr, &scalars[4]);
  block_add(nmerged, &scalars[1]);
}

__global__ void k_count_merged(const unsigned char* __restrict__ merged, int n, int* __restrict__ scalars) {
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += merged[i] ? 1 : 0;
  block_add(c, &scalars[1]);
}

// append: flags (unmerged && Valid) -> exclusive scan -> scatter in measurement order (:100-114, :33-40)
__global__ void k_append_flag(int dim, const float4* __restrict__ meas_pts, const unsigned char* __restrict__ merged, int n,
                              int* __restrict__ flags) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    flags[i] = (!(merged && merged[i]) && valid_point(dim, meas_pts[i])) ? 1 : 0;
}

__global__ void k_append_scatter(int dim, Xf M, const float4* __restrict__ meas_pts, const float4* __restrict__ meas_nrm,
                                 const unsigned char* __restrict__ merged, int n, const int* __restrict__ offset, int base,
                                 float4* __restrict__ scene_pts, float4* __restrict__ scene_nrm) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = meas_pts[i];
    if ((merged && merged[i]) || !valid_point(dim, p)) continue;
    const int k  = base + offset[i];
    scene_pts[k] = xform_point(dim, M, p);  // transformInPlace: coordinates and normal
    scene_nrm[k] = meas_nrm ? rotate_normal(dim, M, meas_nrm[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

int blocks_for(int n) {
  int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}

int scene_device(srrg2_scene* s) {
  HIP_TRY(hipSetDevice(s->device));
  return 0;
}

// grow the point arrays to hold n points, keeping the first `keep`
int scene_reserve(srrg2_scene* s, int n, int keep) {
  if ((size_t) n <= s->pts.cap && (size_t) n <= s->nrm.cap) return 0;
  DevBuf<float4> np, nn;
  int rc;
  const size_t want = (size_t) n + (size_t) n / 2 + 1024;
  if ((rc = np.reserve(want))) return rc;
  if ((rc = nn.reserve(want))) return rc;
  if (keep > 0) {
    HIP_TRY(hipMemcpyAsync(np.p, s->pts.p, sizeof(float4) * (size_t) keep, hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(nn.p, s->nrm.p, sizeof(float4) * (size_t) keep, hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  s->pts.release();
  s->nrm.release();
  s->pts = np;
  s->nrm = nn;
  return 0;
}

// flags[0..n) -> exclusive scan in place; total in scalars[0] (host value returned)
int scan_flags(srrg2_scene* s, int n, int* total) {
  int rc;
  if ((rc = s->scan_sums.reserve((size_t) srrg2amd::scan_num_blocks(n) + 2))) return rc;
  srrg2amd::launch_exclusive_scan(s->flags.p, n, s->scan_sums.p, s->scan_sums.p + s->scan_sums.cap - 1, s->stream);
  HIP_TRY(hipMemcpyAsync(&s->scalars[0], s->scan_sums.p + s->scan_sums.cap - 1, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  *total = s->scalars[0];
  return 0;
}

int read_scalars(srrg2_scene* s, const int* from = nullptr) {
  HIP_TRY(hipMemcpyAsync(s->scalars, from ? from : s->dscalars.p, 16 * sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

int check_params(const srrg2_merger_params* p) {
  if (!p) return fail(SRRG2_E_INVALID, "scene_merge: null params");
  if (p->target_number_of_merges < 0) return fail(SRRG2_E_INVALID, "scene_merge: target_number_of_merges < 0");
  return 0;
}

// the tail shared by both merge entry points: count merged points, append if the target was not reached
// (counted: scalars[1] already holds the number of merged measurement points -- k_merge_from_aligner counts them on its way)
// (merged_flags: where the flags of this call lie when not at the head of scene->merged)
int finish_merge(srrg2_scene* scene, srrg2_scene* meas, const Xf& M, bool have_corr, const srrg2_merger_params* p,
                 srrg2_merge_result* out, bool counted = false, const unsigned char* merged_flags = nullptr) {
  int rc;
  const int n_meas = meas->n;
  int num_merged   = 0;
  if (counted) {
    num_merged = scene->scalars[1];
  } else if (have_corr && n_meas > 0) {
    hipLaunchKernelGGL(k_count_merged, dim3(std::min(blocks_for(n_meas), 64)), dim3(256), 0, scene->stream, scene->merged.p, n_meas,
                       scene->dscalars.p);
    if ((rc = read_scalars(scene))) return rc;
    num_merged = scene->scalars[1];
  }
  out->num_merged = num_merged;
  if (!have_corr || (unsigned) num_merged < (unsigned) p->target_number_of_merges) {  // :92
    if (n_meas > 0) {
      if ((rc = scene->flags.reserve((size_t) n_meas + 1))) return rc;
      const unsigned char* mg = have_corr ? (merged_flags ? merged_flags : scene->merged.p) : nullptr;
      hipLaunchKernelGGL(k_append_flag, dim3(blocks_for(n_meas)), dim3(256), 0, scene->stream, scene->dim, meas->pts.p, mg,
                         n_meas, scene->flags.p);
      int total = 0;
      if ((rc = scan_flags(scene, n_meas, &total))) return rc;
      if (total > 0) {
        if ((rc = scene_reserve(scene, scene->n + total, scene->n))) return rc;
        hipLaunchKernelGGL(k_append_scatter, dim3(blocks_for(n_meas)), dim3(256), 0, scene->stream, scene->dim, M,
                           meas->pts.p, meas->has_normals ? meas->nrm.p : nullptr, mg, n_meas, scene->flags.p, scene->n,
                           scene->pts.p, scene->nrm.p);
        scene->n += total;
      }
      out->num_added = total;
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(scene->stream));
  out->scene_size = scene->n;
  out->status     = SRRG2_MERGER_SUCCESS;  // :122
  return 0;
}

}  // namespace

extern "C" {

int srrg2_merger_default_params(srrg2_merger_params* p) {
  if (!p) return fail(SRRG2_E_INVALID, "merger_default_params: null");
  p->maximum_response                  = 50.f;   // merger_correspondence_homo.h:22-26
  p->maximum_distance_geometry_squared = 0.25f;  // :27-31
  p->target_number_of_merges           = 200;    // merger.h:126-131
  return 0;
}

int srrg2_scene_create(int dim, int device, srrg2_scene_h* out) {
  if ((dim != 2 && dim != 3) || !out) return fail(SRRG2_E_INVALID, "scene_create: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SRRG2_E_NO_DEVICE, "scene_create: no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(SRRG2_E_INVALID, "scene_create: bad device index");
  srrg2_scene* s = new srrg2_scene();
  s->dim         = dim;
  s->device      = device;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_TRY(hipHostMalloc((void**) &s->scalars, 16 * sizeof(int), hipHostMallocDefault));
  std::memset(s->scalars, 0, 16 * sizeof(int));
  int rc;
  if ((rc = s->dscalars.reserve(16))) return rc;
  HIP_TRY(hipMemset(s->dscalars.p, 0, 16 * sizeof(int)));
  *out = s;
  return 0;
}

int srrg2_scene_destroy(srrg2_scene_h s) {
  if (!s) return 0;
  (void) hipSetDevice(s->device);
  if (s->stream) (void) hipStreamSynchronize(s->stream);
  s->pts.release(); s->nrm.release(); s->gidx.release(); s->flags.release(); s->scan_sums.release();
  s->counts.release(); s->dup_list.release(); s->dup_keys.release(); s->sort_tmp.release(); s->merged.release(); s->corr.release(); s->staging.release(); s->dscalars.release();
  if (s->scalars) (void) hipHostFree(s->scalars);
  if (s->stream) (void) hipStreamDestroy(s->stream);
  delete s;
  return 0;
}

int srrg2_scene_set(srrg2_scene_h s, const float* coords, int cs, const float* normals, int ns, int n, int mem) {
  if (!s || n < 0 || (n > 0 && !coords)) return fail(SRRG2_E_INVALID, "scene_set: bad arguments");
  if (mem != SRRG2_MEM_HOST && mem != SRRG2_MEM_DEVICE) return fail(SRRG2_E_INVALID, "scene_set: bad mem");
  if (n > 0 && (cs < s->dim * 4 || cs % 4 || (normals && (ns < s->dim * 4 || ns % 4))))
    return fail(SRRG2_E_INVALID, "scene_set: strides must be multiples of 4 bytes and >= dim floats");
  int rc;
  if ((rc = scene_device(s))) return rc;
  if ((rc = scene_reserve(s, n > 0 ? n : 1, 0))) return rc;
  s->n           = n;
  s->ng          = 0;
  s->has_normals = normals != nullptr;
  if (n == 0) return 0;
  const float* dc = coords;
  const float* dn = normals;
  if (mem == SRRG2_MEM_HOST) {
    const size_t bc = (size_t) (n - 1) * cs + (size_t) s->dim * 4, bn = normals ? (size_t) (n - 1) * ns + (size_t) s->dim * 4 : 0;
    const size_t off_n = (bc + 63) / 64 * 64;
    if ((rc = s->staging.reserve(off_n + bn + 64))) return rc;
    HIP_TRY(hipMemcpyAsync(s->staging.p, coords, bc, hipMemcpyHostToDevice, s->stream));
    dc = (const float*) s->staging.p;
    if (normals) {
      HIP_TRY(hipMemcpyAsync(s->staging.p + off_n, normals, bn, hipMemcpyHostToDevice, s->stream));
      dn = (const float*) (s->staging.p + off_n);
    }
  }
  srrg2amd::launch_ingest(dc, cs / 4, n, s->dim, s->pts.p, nullptr, 0, s->stream);
  if (normals) {
    srrg2amd::launch_ingest(dn, ns / 4, n, s->dim, s->nrm.p, nullptr, 0, s->stream);
  } else {
    HIP_TRY(hipMemsetAsync(s->nrm.p, 0, sizeof(float4) * (size_t) n, s->stream));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

int srrg2_scene_size(srrg2_scene_h s, int* n) {
  if (!s || !n) return fail(SRRG2_E_INVALID, "scene_size: bad arguments");
  *n = s->n;
  return 0;
}

int srrg2_scene_get(srrg2_scene_h s, float* coords_out, float* normals_out, int capacity, int* n) {
  if (!s || !n || capacity < 0) return fail(SRRG2_E_INVALID, "scene_get: bad arguments");
  int rc;
  if ((rc = scene_device(s))) return rc;
  const int m = s->n < capacity ? s->n : capacity;
  if (m > 0 && (coords_out || normals_out)) {
    std::vector<float4> h((size_t) m);
    if (coords_out) {
      HIP_TRY(hipMemcpy(h.data(), s->pts.p, sizeof(float4) * (size_t) m, hipMemcpyDeviceToHost));
      for (int i = 0; i < m; ++i) {
        const float v[3] = {h[(size_t) i].x, h[(size_t) i].y, h[(size_t) i].z};
        for (int d = 0; d < s->dim; ++d) coords_out[(size_t) i * s->dim + d] = v[d];
      }
    }
    if (normals_out) {
      HIP_TRY(hipMemcpy(h.data(), s->nrm.p, sizeof(float4) * (size_t) m, hipMemcpyDeviceToHost));
      for (int i = 0; i < m; ++i) {
        const float v[3] = {h[(size_t) i].x, h[(size_t) i].y, h[(size_t) i].z};
        for (int d = 0; d < s->dim; ++d) normals_out[(size_t) i * s->dim + d] = v[d];
      }
    }
  }
  *n = s->n;
  return 0;
}

int srrg2_scene_device_arrays(srrg2_scene_h s, const float** coords, const float** normals, int* n) {
  if (!s || !coords || !n) return fail(SRRG2_E_INVALID, "scene_device_arrays: bad arguments");
  *coords = (const float*) s->pts.p;
  if (normals) *normals = s->has_normals ? (const float*) s->nrm.p : nullptr;
  *n = s->n;
  return 0;
}

int srrg2_scene_clip_ball(srrg2_scene_h full, const float* robot_in_local_map, float range, srrg2_scene_h clipped,
                          int* status) {
  if (!full || !clipped || !robot_in_local_map || full == clipped || full->dim != clipped->dim ||
      full->device != clipped->device)
    return fail(SRRG2_E_INVALID, "scene_clip_ball: bad arguments");
  int rc;
  if ((rc = scene_device(full))) return rc;
  float Linv[12];
  if (full->dim == 3)
    dm::se3_inverse(robot_in_local_map, Linv);  // scene_clipper.h:64-68
  else
    dm::se2_inverse(robot_in_local_map, Linv);
  const Xf L         = load_transform(full->dim, Linv);
  const float range2 = range * range;
  const int n        = full->n;
  clipped->has_normals = full->has_normals;
  clipped->n = clipped->ng = 0;
  if (status) *status = n == 0 ? SRRG2_CLIPPER_READY : SRRG2_CLIPPER_SUCCESSFUL;  // :24-28
  if (n == 0) return 0;
  hipStream_t st = full->stream;
  if ((rc = full->flags.reserve((size_t) n + 1))) return rc;
  hipLaunchKernelGGL(k_clip_flag, dim3(blocks_for(n)), dim3(256), 0, st, full->dim, L, range2, full->pts.p, n, full->flags.p);
  int total = 0;
  // A clipped scene that has room from the call before (a tracker clips around a pose that moves a little per frame): the scatter
  // is launched BEHIND the scan without the host having seen the total -- one wait per clip instead of two (0.046 -> ~0.03 ms for a
  // 100 k-point map); a total beyond the room repeats the scatter the slow way.
  const int room = (int) std::min<size_t>(std::min(clipped->pts.cap, clipped->nrm.cap), clipped->gidx.cap);
  bool speculated = false;
  if (room > 0) {
    if ((rc = full->scan_sums.reserve((size_t) srrg2amd::scan_num_blocks(n) + 2))) return rc;
    int* dtotal = full->scan_sums.p + full->scan_sums.cap - 1;
    srrg2amd::launch_exclusive_scan(full->flags.p, n, full->scan_sums.p, dtotal, st);
    hipLaunchKernelGGL(k_clip_scatter, dim3(blocks_for(n)), dim3(256), 0, st, full->dim, L, range2, full->pts.p,
                       full->has_normals ? full->nrm.p : nullptr, n, full->flags.p, clipped->pts.p, clipped->nrm.p,
                       clipped->gidx.p, room);
    HIP_TRY(hipMemcpyAsync(&full->scalars[0], dtotal, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    total      = full->scalars[0];
    speculated = total <= room;
  } else if ((rc = scan_flags(full, n, &total))) {
    return rc;
  }
  if (!speculated) {
    if ((rc = scene_reserve(clipped, total > 0 ? total : 1, 0))) return rc;
    if ((rc = clipped->gidx.reserve((size_t) (total > 0 ? total : 1)))) return rc;
    if (total > 0)
      hipLaunchKernelGGL(k_clip_scatter, dim3(blocks_for(n)), dim3(256), 0, st, full->dim, L, range2, full->pts.p,
                         full->has_normals ? full->nrm.p : nullptr, n, full->flags.p, clipped->pts.p, clipped->nrm.p,
                         clipped->gidx.p, total);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
  }
  clipped->n = clipped->ng = total;
  return 0;
}

int srrg2_scene_global_indices(srrg2_scene_h s, int32_t* buf, int* n_inout) {
  if (!s || !n_inout) return fail(SRRG2_E_INVALID, "scene_global_indices: bad arguments");
  int rc;
  if ((rc = scene_device(s))) return rc;
  const int m = s->ng < *n_inout ? s->ng : *n_inout;
  if (buf && m > 0) HIP_TRY(hipMemcpy(buf, s->gidx.p, sizeof(int) * (size_t) m, hipMemcpyDeviceToHost));
  *n_inout = s->ng;
  return 0;
}

int srrg2_scene_merge(srrg2_scene_h scene, srrg2_scene_h meas, const float* measurement_in_scene,
                      const srrg2_correspondence* correspondences, int ncorr, const srrg2_merger_params* p,
                      srrg2_merge_result* out) {
  if (!scene || !meas || !measurement_in_scene || !out || scene == meas || scene->dim != meas->dim ||
      scene->device != meas->device)
    return fail(SRRG2_E_INVALID, "scene_merge: bad arguments");
  if (ncorr > 0 && !correspondences) return fail(SRRG2_E_INVALID, "scene_merge: null correspondences");
  int rc;
  if ((rc = check_params(p))) return rc;
  if ((rc = scene_device(scene))) return rc;
  std::memset(out, 0, sizeof(*out));
  out->status  = SRRG2_MERGER_INITIALIZING;  // :15
  const Xf M   = load_transform(scene->dim, measurement_in_scene);
  const int n_scene = scene->n, n_meas = meas->n;
  if (scene->n == 0 && meas->has_normals) scene->has_normals = true;  // a fresh scene takes the measurement's fields
  hipStream_t st = scene->stream;
  if (ncorr >= 0) {
    out->num_correspondences = ncorr;
    if ((rc = scene->merged.reserve((size_t) n_meas + 1))) return rc;
    HIP_TRY(hipMemsetAsync(scene->merged.p, 0, (size_t) n_meas + 1, st));
    HIP_TRY(hipMemsetAsync(scene->dscalars.p, 0, 16 * sizeof(int), st));
    if (ncorr > 0) {
      if ((rc = scene->corr.reserve((size_t) ncorr))) return rc;
      if ((rc = scene->counts.reserve((size_t) n_scene + 1))) return rc;
      HIP_TRY(hipMemcpyAsync(scene->corr.p, correspondences, sizeof(srrg2_correspondence) * (size_t) ncorr,
                             hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemsetAsync(scene->counts.p, 0, sizeof(int) * ((size_t) n_scene + 1), st));
      hipLaunchKernelGGL(k_merge_count, dim3(blocks_for(ncorr)), dim3(256), 0, st, scene->corr.p, ncorr, n_scene, n_meas,
                         scene->counts.p, scene->dscalars.p);
      if ((rc = read_scalars(scene))) return rc;
      if (scene->scalars[2]) return fail(SRRG2_E_INVALID, "scene_merge: correspondence index out of range");
      const bool dups = scene->scalars[3] != 0;
      if (dups && (rc = scene->flags.reserve((size_t) ncorr + 1))) return rc;
      float4* snrm       = scene->nrm.p;  // (always allocated; written like the oracle's)
      const float4* mnrm = meas->has_normals ? meas->nrm.p : nullptr;
      hipLaunchKernelGGL(k_merge_apply, dim3(blocks_for(ncorr)), dim3(256), 0, st, scene->dim, M, p->maximum_response,
                         p->maximum_distance_geometry_squared, scene->corr.p, ncorr, scene->counts.p, scene->pts.p, snrm,
                         meas->pts.p, mnrm, scene->merged.p, dups ? scene->flags.p : nullptr);
      if (dups) {
        int ndup = 0;
        if ((rc = scan_flags(scene, ncorr, &ndup))) return rc;
        if ((rc = scene->dup_list.reserve((size_t) ndup + 1))) return rc;
        hipLaunchKernelGGL(k_compact_dups, dim3(blocks_for(ncorr)), dim3(256), 0, st, scene->flags.p, scene->corr.p, ncorr,
                           scene->counts.p, scene->dup_list.p);
        if (ndup > 0) {
          if ((rc = scene->dup_keys.reserve(2 * (size_t) ndup))) return rc;
          unsigned long long* kin  = scene->dup_keys.p;
          unsigned long long* kout = scene->dup_keys.p + ndup;
          hipLaunchKernelGGL(k_dup_keys, dim3(blocks_for(ndup)), dim3(256), 0, st, scene->corr.p, scene->dup_list.p, ndup, kin);
          size_t tmp_bytes = 0;
          HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, kin, kout, ndup, 0, 64, st));
          if ((rc = scene->sort_tmp.reserve(std::max<size_t>(tmp_bytes, 1)))) return rc;
          HIP_TRY(hipcub::DeviceRadixSort::SortKeys(scene->sort_tmp.p, tmp_bytes, kin, kout, ndup, 0, 64, st));
          hipLaunchKernelGGL(k_merge_dups, dim3(blocks_for(ndup)), dim3(256), 0, st, scene->dim, M, p->maximum_response,
                             p->maximum_distance_geometry_squared, scene->corr.p, kout, ndup, scene->pts.p, snrm, meas->pts.p,
                             mnrm, scene->merged.p);
        }
      }
    }
  }
  return finish_merge(scene, meas, M, ncorr >= 0, p, out);
}

int srrg2_scene_merge_from_aligner(srrg2_scene_h scene, srrg2_scene_h meas, const float* measurement_in_scene,
                                   srrg2_aligner_h aligner, int slice_idx, srrg2_scene_h clipped,
                                   const srrg2_merger_params* p, srrg2_merge_result* out) {
  if (!scene || !meas || !measurement_in_scene || !aligner || !clipped || !out || scene == meas ||
      scene->dim != meas->dim || scene->device != meas->device || clipped->device != scene->device)
    return fail(SRRG2_E_INVALID, "scene_merge_from_aligner: bad arguments");
  int rc;
  if ((rc = check_params(p))) return rc;
  if ((rc = scene_device(scene))) return rc;
  srrg2amd::AlignerSliceView v;
  if ((rc = srrg2amd::aligner_slice_view(aligner, slice_idx, &v))) return rc;
  if (v.device != scene->device) return fail(SRRG2_E_INVALID, "scene_merge_from_aligner: aligner lives on another device");
  if (v.nm != clipped->ng || v.nm != clipped->n)
    return fail(SRRG2_E_STATE, "scene_merge_from_aligner: the aligner's moving cloud is not the clipped scene");
  if (v.nf != meas->n) return fail(SRRG2_E_STATE, "scene_merge_from_aligner: the aligner's fixed cloud is not the measurement");
  if (v.ready_event) HIP_TRY(hipStreamWaitEvent(scene->stream, (hipEvent_t) v.ready_event, 0));  // (no host wait in between)
  std::memset(out, 0, sizeof(*out));
  out->status = SRRG2_MERGER_INITIALIZING;
  const Xf M  = load_transform(scene->dim, measurement_in_scene);
  const int n_scene = scene->n, n_meas = meas->n;
  hipStream_t st = scene->stream;
  // (the call's sixteen counters in FRONT of the flags, in one allocation: one memset instead of two; whole words: the flags are set
  // through 32-bit atomics)
  if ((rc = scene->merged.reserve((size_t) n_meas + 8 + 64))) return rc;
  HIP_TRY(hipMemsetAsync(scene->merged.p, 0, 64 + ((size_t) n_meas + 4) / 4 * 4, st));
  int* const counters         = reinterpret_cast<int*>(scene->merged.p);
  unsigned char* const flags_ = scene->merged.p + 64;
  if (v.nm > 0) {
    hipLaunchKernelGGL(k_merge_from_aligner, dim3(std::min(blocks_for(v.nm), 256)), dim3(256), 0, st, scene->dim, M, p->maximum_response,
                       p->maximum_distance_geometry_squared, v.moving_sorted, v.corr_fixed, v.corr_resp, v.corr_stat,
                       v.prune ? 1 : 0, v.nm, clipped->gidx.p, n_scene, n_meas, scene->pts.p,
                       scene->nrm.p, meas->pts.p, meas->has_normals ? meas->nrm.p : nullptr,
                       flags_, counters);
    if ((rc = read_scalars(scene, counters))) return rc;
    if (scene->scalars[2]) return fail(SRRG2_E_STATE, "scene_merge_from_aligner: index out of range");
  }
  else if ((rc = read_scalars(scene, counters)))
    return rc;
  out->num_correspondences = scene->scalars[4];
  return finish_merge(scene, meas, M, true, p, out, /*counted=*/true, flags_);
}

}  // extern "C"
