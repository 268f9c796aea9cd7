// kernels_prep.hip -- everything that happens once per set_fixed / set_moving: ingest of the caller's strided clouds,
// bounding boxes, the dense voxel grid over the fixed cloud (counting sort: CorrespondenceFinder_ search-structure
// construction, S/registration/correspondence_finder.h:80-91), prefix scans, and the Morton sort of the moving clouds.
// Pure data movement and integer work: HBM-bound, nothing here touches the arithmetic specification of the ICP passes.
#include <cstdlib>
#include "kernels.h"

#include "device_util.h"

// ============================================================================================
// ingest: strided raw floats -> float4 (x,y,z|0,w) ; also max |coord| over finite points
// ============================================================================================
__global__ void k_ingest(const float* __restrict__ src, int stride_floats, int n, int dim, float4* __restrict__ dst,
                         unsigned* __restrict__ maxabs_bits, int finite_per_point) {
  int i      = blockIdx.x * blockDim.x + threadIdx.x;
  float amax = 0.f;
  if (i < n) {
    const float* p = src + (size_t) i * stride_floats;
    float x = p[0], y = p[1], z = dim == 3 ? p[2] : 0.f;
    dst[i] = make_float4(x, y, z, 0.f);
    bool ok = finite3(x, y, z);
    if (finite_per_point) {
      if (ok) amax = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
    } else {
      if (isfinite(x)) amax = fmaxf(amax, fabsf(x));
      if (isfinite(y)) amax = fmaxf(amax, fabsf(y));
      if (isfinite(z)) amax = fmaxf(amax, fabsf(z));
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  // One atomic per BLOCK, and only when it would raise the maximum: same-address atomics serialise at tens of ns each -- one per
  // wave, 1 564 of them for a 100 k-point cloud, made this kernel 19.7 us where the copy itself takes ~5 (a tracker's set_fixed
  // runs it for the normals of every frame).  The unconditional atomic of a block that has seen a smaller value is still exact:
  // the look is only a filter.
  if (!maxabs_bits) return;
  __shared__ float red_ing[4];
  if ((threadIdx.x & 63) == 0) red_ing[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(red_ing[0], red_ing[1]), fmaxf(red_ing[2], red_ing[3]));
    const unsigned bits = __float_as_uint(amax);
    if (amax > 0.f && bits > __hip_atomic_load(maxabs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxabs_bits, bits);
  }
}

// the clouds of a batch in ONE launch: blockIdx.y = problem (its points are [moff, moff + nm) of src and dst)
__global__ void k_ingest_batch(const float* __restrict__ src, int stride_floats, const ProblemDev* __restrict__ probs,
                               int dim, float4* __restrict__ dst, unsigned* __restrict__ maxabs_bits /* [K] or null */,
                               int finite_per_point) {
  const ProblemDev pd = probs[blockIdx.y];
  float amax          = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pd.nm; i += gridDim.x * blockDim.x) {
    const float* p = src + (size_t) (pd.moff + i) * stride_floats;
    float x = p[0], y = p[1], z = dim == 3 ? p[2] : 0.f;
    dst[pd.moff + i] = make_float4(x, y, z, 0.f);
    if (finite_per_point) {
      if (finite3(x, y, z)) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)));
    } else {
      if (isfinite(x)) amax = fmaxf(amax, fabsf(x));
      if (isfinite(y)) amax = fmaxf(amax, fabsf(y));
      if (isfinite(z)) amax = fmaxf(amax, fabsf(z));
    }
  }
  if (!maxabs_bits) return;
  // block maximum, then one atomic per block (same-address atomics serialise at tens of ns each)
  __shared__ float red[4];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (amax > 0.f) atomicMax(maxabs_bits + blockIdx.y, __float_as_uint(amax));
  }
}

// ============================================================================================
// grid build
// ============================================================================================
// SRC != nullptr (k_ingest_bbox): the points are ingested on the way -- caller layout in, float4 out, max |coordinate| of the
// finite points -- a fixed cloud's set_fixed in one pass over the cloud instead of two launches (round 6: a tracker sets a new
// fixed cloud every frame, multi_tracker_impl.cpp:97-98)
template <bool INGEST>
__device__ __forceinline__ void bbox_body(const float* __restrict__ src, int stride_floats, int dim, float4* __restrict__ dst,
                                          unsigned* __restrict__ maxabs_bits, const float4* __restrict__ pts, int n,
                                          unsigned* __restrict__ mn_out, unsigned* __restrict__ mx_out, int* __restrict__ nvalid,
                                          unsigned* __restrict__ block_out = nullptr) {
  unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  int valid = 0;
  float amax = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p;
    if constexpr (INGEST) {
      const float* q = src + (size_t) i * stride_floats;
      p      = make_float4(q[0], q[1], dim == 3 ? q[2] : 0.f, 0.f);
      dst[i] = p;
      if (finite3(p.x, p.y, p.z)) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)));
    } else {
      p = pts[i];
    }
    if (finite3(p.x, p.y, p.z)) {
      valid += 1;
      const unsigned k[3] = {fkey(p.x), fkey(p.y), fkey(p.z)};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        mn[d] = min(mn[d], k[d]);
        mx[d] = max(mx[d], k[d]);
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = min(mn[d], (unsigned) __shfl_xor((int) mn[d], off));
      mx[d] = max(mx[d], (unsigned) __shfl_xor((int) mx[d], off));
    }
    valid += __shfl_xor(valid, off);
  }
  // one atomic per block and value (a few grid-striding blocks): same-address atomics serialise at tens of ns each
  __shared__ unsigned red[16][8];  // (up to 16 waves per block: k_ingest_bbox runs 1024 threads; the atomics below assume four)
  if constexpr (INGEST) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      red[threadIdx.x >> 6][d]     = mn[d];
      red[threadIdx.x >> 6][3 + d] = mx[d];
    }
    red[threadIdx.x >> 6][6] = (unsigned) valid;
    red[threadIdx.x >> 6][7] = __float_as_uint(amax);  // (non-negative floats order like their bit patterns)
  }
  __syncthreads();
  if (block_out) {  // (k_ingest_bbox: the block's eight values for the last block to reduce -- no atomics on shared words at all)
    if (threadIdx.x < 8) {
      const int d = threadIdx.x;
      unsigned v = red[0][d];
      for (int w = 1; w < (int) (blockDim.x >> 6); ++w) v = d < 3 ? min(v, red[w][d]) : ((d < 6 || d == 7) ? max(v, red[w][d]) : v + red[w][d]);
      block_out[(size_t) blockIdx.x * 8 + d] = v;
    }
    return;
  }
  // (minima / maxima: the atomic only when it would move the value -- a look first, an exact filter: same-address atomics serialise)
  if (INGEST && threadIdx.x == 7 && maxabs_bits) {
    const unsigned v = max(max(red[0][7], red[1][7]), max(red[2][7], red[3][7]));
    if (v != 0u && v > __hip_atomic_load(maxabs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxabs_bits, v);
  }
  if (threadIdx.x < 7) {
    const int d = threadIdx.x;
    if (d < 3) {
      const unsigned v = min(min(red[0][d], red[1][d]), min(red[2][d], red[3][d]));
      if (v != 0xffffffffu && v < __hip_atomic_load(&mn_out[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mn_out[d], v);
    } else if (d < 6) {
      const unsigned v = max(max(red[0][d], red[1][d]), max(red[2][d], red[3][d]));
      if (v != 0u && v > __hip_atomic_load(&mx_out[d - 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&mx_out[d - 3], v);
    } else {
      const int v = (int) (red[0][6] + red[1][6] + red[2][6] + red[3][6]);
      if (v) atomicAdd(nvalid, v);
    }
  }
}
__global__ void k_bbox(const float4* __restrict__ pts, int n, unsigned* __restrict__ mn_out, unsigned* __restrict__ mx_out,
                       int* __restrict__ nvalid) {
  bbox_body<false>(nullptr, 0, 3, nullptr, nullptr, pts, n, mn_out, mx_out, nvalid);
}
// (ticket / host_out / seq: the LAST block to finish copies the box and the count into pinned host memory and, behind a system-scope
// fence, the caller's sequence number -- set_fixed polls that word instead of queueing a device-to-host copy and waiting for the
// stream, and does not wait for the normals' ingest queued behind this kernel)
__global__ void k_ingest_bbox(const float* __restrict__ src, int stride_floats, int n, int dim, float4* __restrict__ dst,
                              unsigned* __restrict__ maxabs_bits, unsigned* __restrict__ mn_out, unsigned* __restrict__ mx_out,
                              int* __restrict__ nvalid, unsigned* __restrict__ ticket, unsigned* __restrict__ host_out, unsigned seq,
                              unsigned* __restrict__ block_out, unsigned* __restrict__ clear_after) {
  // (block_out: [gridDim.x][8] -- every block leaves its minima, maxima, count and max |coordinate| there instead of in 8 atomics
  // on neighbouring words: 256 blocks finishing together serialised 2 048 of them, 20 of this kernel's 24 us on a 100 k-point cloud)
  bbox_body<true>(src, stride_floats, dim, dst, maxabs_bits, nullptr, n, mn_out, mx_out, nvalid, block_out);
  if (!host_out || !block_out) return;
  __shared__ int last_block;
  __shared__ unsigned fin[16][8];
  // (the row was written by threads 0 .. 7: their wave releases it, thread 0 of the same wave takes the ticket behind the fence.  A
  // fence by EVERY wave -- an L2 write-back each -- made the kernel slower the more waves it had: 12 us with 256 of them, 28 with 1 024)
  if (threadIdx.x < 64) {
    __threadfence();
    if (threadIdx.x == 0) last_block = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!last_block) return;
  // the last block: rows of all blocks (<= 256: one per thread), reduced over the block
  unsigned v[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) v[d] = d < 3 ? 0xffffffffu : 0u;
  for (int b = threadIdx.x; b < (int) gridDim.x; b += blockDim.x)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const unsigned x = __hip_atomic_load(block_out + (size_t) b * 8 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[d] = d < 3 ? min(v[d], x) : ((d < 6 || d == 7) ? max(v[d], x) : v[d] + x);
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const unsigned x = (unsigned) __shfl_xor((int) v[d], off);
      v[d] = d < 3 ? min(v[d], x) : ((d < 6 || d == 7) ? max(v[d], x) : v[d] + x);
    }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int d = 0; d < 8; ++d) fin[threadIdx.x >> 6][d] = v[d];
  __syncthreads();
  if (threadIdx.x < 8) {
    const int d = threadIdx.x;
    unsigned r = fin[0][d];
    for (int w = 1; w < (int) (blockDim.x >> 6); ++w) r = d < 3 ? min(r, fin[w][d]) : ((d < 6 || d == 7) ? max(r, fin[w][d]) : r + fin[w][d]);
    // the device's copies (what build_grid's launches and k_icp_init read) and the host's
    if (d < 3) mn_out[d] = r;
    else if (d < 6) mx_out[d - 3] = r;
    else if (d == 6) *nvalid = (int) r;
    else if (maxabs_bits) *maxabs_bits = r;
    if (d < 7) host_out[d] = r;
  }
  // (what the NEXT call and the kernels behind this one start from: the ticket back at zero, the normals' norm -- the word behind the
  // count, accumulated by the normals' ingest that follows on the stream -- cleared: set_fixed queues no initialising copy)
  if (threadIdx.x == 8) *ticket = 0u;
  if (threadIdx.x == 9 && clear_after) *clear_after = 0u;
  if (threadIdx.x < 64) {  // (the writers' wave)
    __threadfence_system();
    if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned*>(host_out + 8) = seq;
  }
}

__device__ __forceinline__ int grid_cell_of(const GridDev& g, float4 p) {
  int cx = cell_coord(p.x, g.ox, g.inv_h);
  int cy = cell_coord(p.y, g.oy, g.inv_h);
  int cz = cell_coord(p.z, g.oz, g.inv_h);
  cx     = min(max(cx, 0), g.nx - 1);
  cy     = min(max(cy, 0), g.ny - 1);
  cz     = min(max(cz, 0), g.nz - 1);
  return (cz * g.ny + cy) * g.nx + cx;
}

__global__ void k_grid_count(GridDev g, const float4* __restrict__ pts, int n, int* __restrict__ counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  if (!finite3(p.x, p.y, p.z)) return;
  atomicAdd(&counts[grid_cell_of(g, p)], 1);
}

// number of non-empty cells (density probe for the automatic cell size)
__global__ void k_count_nonzero(const int* __restrict__ counts, int n, int* __restrict__ out) {
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += counts[i] != 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
  __shared__ int red[4];  // one atomic per block (same-address atomics serialise: per-wave ones cost 47 us here)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = (red[0] + red[1]) + (red[2] + red[3]);
    if (t) atomicAdd(out, t);
  }
}

// exclusive scan, 3 kernels: per-block scan of SCAN_TILE elements, scan of block sums, add back
#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[SCAN_THREADS / 64];
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    if (w < wid) base += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void k_scan_tiles(int* __restrict__ data, int n, int* __restrict__ block_sums) {
  int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0;
    s += v[k];
  }
  int total;
  int ex = block_exclusive_scan(s, &total);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) data[base + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void k_scan_sums(int* __restrict__ block_sums, int nblocks, int* __restrict__ grand_total) {
  // single block; nblocks <= SCAN_TILE * 64 handled by looping tiles with a carry
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < nblocks; t0 += SCAN_TILE) {
    int base = t0 + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      v[k] = (base + k < nblocks) ? block_sums[base + k] : 0;
      s += v[k];
    }
    int total;
    int ex = block_exclusive_scan(s, &total) + carry;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      if (base + k < nblocks) block_sums[base + k] = ex;
      ex += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *grand_total = carry;
}

// (copy != nullptr: a second copy of the scanned values -- the scatter's cursors -- instead of a device-to-device copy behind the scan)
__global__ void k_scan_add(int* __restrict__ data, int n, const int* __restrict__ block_sums,
                           const int* __restrict__ grand_total, int* __restrict__ copy) {
  int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int add  = block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) {
      const int v    = data[base + k] + add;
      data[base + k] = v;
      if (copy) copy[base + k] = v;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) data[n] = *grand_total;  // cell_start[ncell]
}

__global__ void k_grid_scatter(GridDev g, const float4* __restrict__ pts, const float4* __restrict__ nrm, int n,
                               int* __restrict__ cursor, float4* __restrict__ out_pts, float4* __restrict__ out_nrm,
                               int* __restrict__ pos_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  if (!finite3(p.x, p.y, p.z)) {
    pos_of[i] = 0;
    return;
  }
  int pos      = atomicAdd(&cursor[grid_cell_of(g, p)], 1);
  p.w          = __int_as_float(i);
  out_pts[pos] = p;
  pos_of[i]    = pos;
  if (nrm) out_nrm[pos] = nrm[i];
}

// ============================================================================================
// Cell neighbour lists (GridDev::list_*; the search passes of k_icp_step_cnl walk them).  One thread per cell of the
// extended grid walks the offset table (host-built: the offsets d with cls_b2[class(d)] <= extended gate^2, sorted by
// (class, centre distance)) and emits one entry per CNL_ENTRY_MAX points of every occupied target cell -- k_cnl_count counts them,
// an exclusive scan places the lists, k_cnl_fill writes them.  Once per set_fixed, and only when a compute() wants them.
// ============================================================================================
__device__ __forceinline__ int cnl_target_cell(const GridDev& g, const GridLists& L, int lc, const int4 o, int& cnt) {
  const int lx = lc % L.lnx, ly = (lc / L.lnx) % L.lny, lz = lc / (L.lnx * L.lny);
  const int Rz = L.lnz > 1 ? L.R : 0;
  const int X = lx - L.R + o.x, Y = ly - L.R + o.y, Z = lz - Rz + o.z;
  cnt = 0;
  if (X < 0 || X >= g.nx || Y < 0 || Y >= g.ny || Z < 0 || Z >= g.nz) return -1;
  const int c = (Z * g.ny + Y) * g.nx + X;
  cnt         = g.cell_start[c + 1] - g.cell_start[c];
  return c;
}

__global__ __launch_bounds__(256) void k_cnl_count(GridDev g, GridLists L, const int4* __restrict__ offs, int noffs, int ncell,
                                                   int* __restrict__ counts) {
  const int lc = blockIdx.x * blockDim.x + threadIdx.x;
  if (lc >= ncell) return;
  int total = 0;
  for (int k = 0; k < noffs; ++k) {
    int cnt;
    (void) cnl_target_cell(g, L, lc, offs[k], cnt);
    total += (cnt + CNL_ENTRY_MAX - 1) / CNL_ENTRY_MAX;
  }
  counts[lc] = total;
}

// Tight boxes (round 5).  A cloud is a surface: the <= 16 points of an entry fill a thin slab of their cell, and a query
// 0.3 m off the surface is farther than the gate from all of them although the gate ball cuts the cell's cube.  Every entry
// carries the bounding box of ITS points in sixteenths of a cell (k_cnl_boxes: once per chunk of 16 points of every occupied
// cell, looked up by the position of the chunk's first point); the header test of cnl_search prunes against that box instead
// of the cube -- on iteration 0 of C4 it passes 1.5 - 2 entries per query on instead of ~5 (17 - 22 candidates instead of
// 43 - 49).  Box bytes per axis: lo = floor(16 f_min), hi = floor(16 f_max) + 1 with f = fl(fl(x - o) inv_h) - cell, the
// expression the cell assignment itself uses (a point whose cell was clamped at the grid's border gets the whole cube).
__global__ __launch_bounds__(256) void k_cnl_boxes(GridDev g, int ncell, uint2* __restrict__ box_at) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
  const int s0 = g.cell_start[c], s1 = g.cell_start[c + 1];
  for (int s = s0; s < s1; s += CNL_ENTRY_MAX) {
    int lo[3] = {16, 16, 16}, hi[3] = {0, 0, 0};
    const int e = s + CNL_ENTRY_MAX < s1 ? s + CNL_ENTRY_MAX : s1;
    for (int j = s; j < e; ++j) {
      const float4 p = g.pts[j];
      const float v[3] = {p.x, p.y, p.z}, o[3] = {g.ox, g.oy, g.oz};
      const int cc[3] = {cx, cy, cz};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float u = (v[d] - o[d]) * g.inv_h;
        const float f = u - (float) cc[d];
        int a = (int) floorf(f * 16.f), b = a + 1;
        if (!(f >= 0.f) || !(f < 1.f)) {  // (clamped at the border of the grid, or a 2-D cloud's z: the whole cube)
          a = 0;
          b = 16;
        }
        a = a < 0 ? 0 : (a > 15 ? 15 : a);
        b = b < 1 ? 1 : (b > 16 ? 16 : b);
        lo[d] = a < lo[d] ? a : lo[d];
        hi[d] = b > hi[d] ? b : hi[d];
      }
    }
    box_at[s] = make_uint2((unsigned) lo[0] | ((unsigned) lo[1] << 8) | ((unsigned) lo[2] << 16),
                           (unsigned) hi[0] | ((unsigned) hi[1] << 8) | ((unsigned) hi[2] << 16));
  }
}

__global__ __launch_bounds__(256) void k_cnl_fill(GridDev g, GridLists L, const int4* __restrict__ offs, int noffs, int ncell,
                                                  const int* __restrict__ list_start, const uint2* __restrict__ box_at,
                                                  uint4* __restrict__ ent) {
  const int lc = blockIdx.x * blockDim.x + threadIdx.x;
  if (lc >= ncell) return;
  int at       = list_start[lc];
  const int R  = L.R;
  for (int k = 0; k < noffs; ++k) {
    const int4 o = offs[k];
    int cnt;
    const int c = cnl_target_cell(g, L, lc, o, cnt);
    if (cnt <= 0) continue;
    const unsigned code = ((unsigned) o.w << 4) | ((unsigned) (o.x + R) << 8) | ((unsigned) (o.y + R) << 16) |
                          ((unsigned) (o.z + R) << 24);
    int start = g.cell_start[c];
    // box bytes of the entry: (offset + R) * 16 + sixteenths, per axis (<= 7 * 16 + 16 = 128: a byte)
    const unsigned base = ((unsigned) (o.x + R) * 16u) | (((unsigned) (o.y + R) * 16u) << 8) | (((unsigned) (o.z + R) * 16u) << 16);
    while (cnt > 0) {
      const int m    = cnt < CNL_ENTRY_MAX ? cnt : CNL_ENTRY_MAX;
      const uint2 bx = box_at[start];
      ent[at++]      = make_uint4((unsigned) start, code | (unsigned) (m - 1), base + bx.x, base + bx.y);
      start += m;
      cnt -= m;
    }
  }
}

// ============================================================================================
// spatial sort of the moving cloud(s): Morton order inside each problem's bounding box, so that the 64
// lanes of a wave query neighbouring cells of the fixed grid (L1/L2-coherent candidate loads).  The
// order has no effect on results: outputs are addressed by the caller's index (kept in .w) and every
// sum is exact.
// ============================================================================================
__device__ __forceinline__ unsigned spread3(unsigned v) {  // v < 64: insert two zero bits between bits
  v = (v | (v << 8)) & 0x0000f00fu;
  v = (v | (v << 4)) & 0x000c30c3u;
  v = (v | (v << 2)) & 0x00249249u;
  return v;
}

__device__ __forceinline__ unsigned spread2(unsigned v) {  // v < 65536: insert one zero bit between bits
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// Key layout of the moving-cloud sort.  The key has `kbits` bits in all; each axis gets as many of them as it takes to
// make the cells roughly cubic (a 10 m x 10 m x 1 m scene sorted with the same number of bits per axis has cells that
// are ten times flatter than wide: the 64 consecutive points of a wave then spread over a 0.6 m patch instead of 0.15 m).
// Bits are dealt one at a time to the axis whose cells are currently the longest (aniso == 0: kbits / 3 per axis, the
// layout of round 2).  The key interleaves the axes like a Morton code, the longer axes contributing their extra top
// bits first:  [ top bits of the longest axis | 2-way interleave of the two longest | 3-way interleave of all three ].
// Any monotone cell assignment gives a valid sort; the order only serves the coherence of neighbouring lanes.
struct KeySpec {
  float mn[3], scale[3];  // cell coordinate of axis d = (v - mn[d]) * scale[d], clamped to [0, 2^b[d])
  int b[3];               // bits per axis
  int ia, ib, ic;         // the axes by bits, descending
};

__device__ __forceinline__ void key_spec_from_bbox(const float* mn, const float* mx, bool any, int kbits, int aniso, KeySpec& k) {
  float ext[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    k.mn[d] = mn[d];
    ext[d]  = any ? mx[d] - mn[d] : 0.f;
    k.b[d]  = 0;
  }
  if (aniso) {
    float cell[3] = {ext[0], ext[1], ext[2]};
    for (int t = 0; t < kbits; ++t) {
      int best = -1;
      float bv = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (cell[d] > bv) { bv = cell[d]; best = d; }
      if (best < 0) break;  // (a single point: every extent is zero)
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (d == best) { k.b[d]++; cell[d] *= 0.5f; }
    }
  } else {
#pragma unroll
    for (int d = 0; d < 3; ++d) k.b[d] = kbits / 3;
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) k.scale[d] = (ext[d] > 0.f) ? (float) (1 << k.b[d]) / ext[d] : 0.f;
  // axes by bits, descending (ties keep x, y, z order: equal bits give the plain Morton code)
  int i0 = 0, i1 = 1, i2 = 2;
  if (k.b[i1] > k.b[i0]) { int t = i0; i0 = i1; i1 = t; }
  if (k.b[i2] > k.b[i1]) { int t = i1; i1 = i2; i2 = t; }
  if (k.b[i1] > k.b[i0]) { int t = i0; i0 = i1; i1 = t; }
  k.ia = i0; k.ib = i1; k.ic = i2;
}

__device__ __forceinline__ unsigned key_of_point(const float4 p, const KeySpec& k, int kbits) {
  if (!finite3(p.x, p.y, p.z)) return (1u << kbits) - 1u;
  const float v[3] = {p.x, p.y, p.z};
  unsigned c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int ci = (int) ((v[d] - k.mn[d]) * k.scale[d]);
    c[d]         = (unsigned) min(max(ci, 0), (1 << k.b[d]) - 1);
  }
  auto pick = [&](int i) { return i == 0 ? c[0] : (i == 1 ? c[1] : c[2]); };
  auto bits_of = [&](int i) { return i == 0 ? k.b[0] : (i == 1 ? k.b[1] : k.b[2]); };
  const unsigned ca = pick(k.ia), cb = pick(k.ib), cc = pick(k.ic);
  const int ba = bits_of(k.ia), bb = bits_of(k.ib), bc = bits_of(k.ic);
  (void) ba;
  const unsigned mc = (1u << bc) - 1u, mm = (1u << (bb - bc)) - 1u;
  const unsigned lo  = spread3(ca & mc) | (spread3(cb & mc) << 1) | (spread3(cc & mc) << 2);
  const unsigned mid = spread2((ca >> bc) & mm) | (spread2((cb >> bc) & mm) << 1);
  const unsigned hi  = ca >> bb;
  return (hi << (3 * bc + 2 * (bb - bc))) | (mid << (3 * bc)) | lo;
}

// (global-histogram path: the key layout is re-derived from the problem's bounding box by every thread -- uniform values)
__device__ __forceinline__ unsigned morton_key(const float4 p, const unsigned* bb, int kbits, int aniso) {
  float mn[3], mx[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const unsigned kmn = bb[d], kmx = bb[3 + d];
    mn[d] = __uint_as_float((kmn & 0x80000000u) ? (kmn & 0x7fffffffu) : ~kmn);
    mx[d] = __uint_as_float((kmx & 0x80000000u) ? (kmx & 0x7fffffffu) : ~kmx);
  }
  KeySpec k;
  key_spec_from_bbox(mn, mx, bb[0] != 0xffffffffu, kbits, aniso, k);
  return key_of_point(p, k, kbits);
}

__global__ void k_msort_bbox(const float4* __restrict__ pts, const ProblemDev* __restrict__ probs,
                             unsigned* __restrict__ bb /* [K][6] */) {
  const ProblemDev pd = probs[blockIdx.y];
  unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pd.nm; i += gridDim.x * blockDim.x) {
    float4 p = pts[pd.moff + i];
    if (!finite3(p.x, p.y, p.z)) continue;
    const unsigned k[3] = {fkey(p.x), fkey(p.y), fkey(p.z)};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = min(mn[d], k[d]);
      mx[d] = max(mx[d], k[d]);
    }
  }
  // wave reduction first: one atomic per wave and bound instead of one per point (same-address atomics serialise)
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mn[d] = min(mn[d], (unsigned) __shfl_xor((int) mn[d], off));
      mx[d] = max(mx[d], (unsigned) __shfl_xor((int) mx[d], off));
    }
  }
  // ... then the 4 waves of the block through LDS: one atomic per block and bound (measured: per-wave atomics on the
  // 6 addresses of a problem cost 109 us at C2 and 541 us for a 32 x 50k batch)
  __shared__ unsigned red[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      red[threadIdx.x >> 6][d]     = mn[d];
      red[threadIdx.x >> 6][3 + d] = mx[d];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int d = threadIdx.x;
    unsigned* b = bb + blockIdx.y * 6;
    if (d < 3) {
      const unsigned v = min(min(red[0][d], red[1][d]), min(red[2][d], red[3][d]));
      if (v != 0xffffffffu) atomicMin(&b[d], v);
    } else {
      const unsigned v = max(max(red[0][d], red[1][d]), max(red[2][d], red[3][d]));
      if (v != 0u) atomicMax(&b[d], v);
    }
  }
}

__global__ void k_msort_count(const float4* __restrict__ pts, const ProblemDev* __restrict__ probs,
                              const unsigned* __restrict__ bb, int kbits, int aniso, int* __restrict__ counts) {
  const ProblemDev pd = probs[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pd.nm; i += gridDim.x * blockDim.x) {
    unsigned key = morton_key(pts[pd.moff + i], bb + blockIdx.y * 6, kbits, aniso);
    atomicAdd(&counts[((size_t) blockIdx.y << kbits) + key], 1);
  }
}

__global__ void k_msort_scatter(const float4* __restrict__ pts, const float4* __restrict__ nrm,
                                const ProblemDev* __restrict__ probs, const unsigned* __restrict__ bb, int kbits, int aniso,
                                int* __restrict__ cursor, float4* __restrict__ out_pts, float4* __restrict__ out_nrm) {
  const ProblemDev pd = probs[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pd.nm; i += gridDim.x * blockDim.x) {
    float4 p     = pts[pd.moff + i];
    unsigned key = morton_key(p, bb + blockIdx.y * 6, kbits, aniso);
    int pos      = atomicAdd(&cursor[((size_t) blockIdx.y << kbits) + key], 1);
    p.w          = __int_as_float(i);
    out_pts[pos] = p;
    if (nrm) out_nrm[pos] = nrm[pd.moff + i];
  }
}

// Morton sort of a batch with a small key space (<= 32 Ki cells per problem: the histogram fits in LDS): ONE workgroup per
// problem does everything -- bounding box and max |coordinate| of the raw strided input, histogram with LDS atomics,
// exclusive scan in place, scatter of the widened points (and normals) through LDS cursors.  Replaces two ingest
// kernels, the bounding-box / count / scan x 3 / copy / scatter kernels and their 2 x nm global atomics per problem
// (a 32 x 50k batch: 68 + 188 us -> one kernel).  Each thread keeps eight points in flight per round.
__global__ __launch_bounds__(1024) void k_msort_local(const float* __restrict__ src, int sf, const float* __restrict__ nsrc,
                                                      int nsf, const ProblemDev* __restrict__ probs, int dim, int kbits, int aniso,
                                                      float4* __restrict__ out_pts, float4* __restrict__ out_nrm,
                                                      unsigned* __restrict__ maxabs_bits /* [K] */) {
#ifndef SRRG2_MSORT_NPT
#define SRRG2_MSORT_NPT 8
#endif
  constexpr int NPT = SRRG2_MSORT_NPT;  // points in flight per thread and round
  extern __shared__ int hist[];  // 1 << kbits counters, then cursors
  __shared__ unsigned red[16][6];
  __shared__ unsigned bbs[6];
  __shared__ KeySpec kspec;  // cell coordinate = (v - mn) * scale (one reciprocal per axis, not a division per point)
  __shared__ int wsum[16];
  // gridDim.x workgroups per problem (blockIdx.y): each sorts ITS contiguous share of the problem's points on its own --
  // own bounding box, own histogram -- into the same share of the output.  The cloud then is gridDim.x sorted segments
  // instead of one: the sort only serves the coherence of neighbouring lanes, the results do not depend on the order.
  // (a batch of 32 clouds is 32 workgroups otherwise: 118 us on an eighth of the chip)
  const ProblemDev whole = probs[blockIdx.y];
  const int seg0 = (int) ((long long) whole.nm * blockIdx.x / gridDim.x);
  const int seg1 = (int) ((long long) whole.nm * (blockIdx.x + 1) / gridDim.x);
  const ProblemDev pd = ProblemDev{whole.moff + seg0, seg1 - seg0};
  const int ncell     = 1 << kbits;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* base = src + (size_t) pd.moff * sf;
  auto load = [&](int i) {
    const float* p = base + (size_t) i * sf;
    return make_float4(p[0], p[1], dim == 3 ? p[2] : 0.f, 0.f);
  };
  for (int c = tid; c < ncell; c += 1024) hist[c] = 0;
  // ---- pass 1: bounding box of the finite points (order-preserving unsigned keys, like k_msort_bbox)
  unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int i0 = tid; i0 < pd.nm; i0 += NPT * 1024) {
    float4 q[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) q[j] = (i0 + j * 1024 < pd.nm) ? load(i0 + j * 1024) : make_float4(NAN, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      if (!finite3(q[j].x, q[j].y, q[j].z)) continue;
      const unsigned k[3] = {fkey(q[j].x), fkey(q[j].y), fkey(q[j].z)};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        mn[d] = min(mn[d], k[d]);
        mx[d] = max(mx[d], k[d]);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mn[d] = min(mn[d], (unsigned) __shfl_xor((int) mn[d], off));
      mx[d] = max(mx[d], (unsigned) __shfl_xor((int) mx[d], off));
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      red[wid][d]     = mn[d];
      red[wid][3 + d] = mx[d];
    }
  }
  __syncthreads();
  if (tid < 6) {
    unsigned v = red[0][tid];
    for (int w = 1; w < 16; ++w) v = tid < 3 ? min(v, red[w][tid]) : max(v, red[w][tid]);
    bbs[tid] = v;
  }
  __syncthreads();
  if (tid == 0 && maxabs_bits) {  // max |coordinate| over the finite points = the largest |bound|
    float amax = 0.f;
    if (bbs[0] != 0xffffffffu) {
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        const unsigned k = bbs[d];
        const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        amax             = fmaxf(amax, fabsf(__uint_as_float(b)));
      }
    }
    // (non-negative floats order like their bit patterns; the host zeroes the word when segments share it)
    if (gridDim.x == 1)
      maxabs_bits[blockIdx.y] = __float_as_uint(amax);
    else
      atomicMax(&maxabs_bits[blockIdx.y], __float_as_uint(amax));
  }
  if (tid == 0) {
    float mnf[3], mxf[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const unsigned a = bbs[d], z = bbs[3 + d];
      mnf[d] = __uint_as_float((a & 0x80000000u) ? (a & 0x7fffffffu) : ~a);
      mxf[d] = __uint_as_float((z & 0x80000000u) ? (z & 0x7fffffffu) : ~z);
    }
    KeySpec k;
    key_spec_from_bbox(mnf, mxf, bbs[0] != 0xffffffffu, kbits, aniso, k);
    kspec = k;
  }
  __syncthreads();
  // (any monotone cell assignment gives a valid sort: the keys only order the points)
  const KeySpec ks = kspec;
  auto key_of = [&](const float4 p) -> unsigned { return key_of_point(p, ks, kbits); };
#if defined(SRRG2_MSORT_EXPERIMENT) && SRRG2_MSORT_EXPERIMENT == 3
  if (kbits > 0) return;  // (timing experiment: bounding box pass only)
#endif
  // ---- pass 2: histogram
  for (int i0 = tid; i0 < pd.nm; i0 += NPT * 1024) {
    float4 q[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) q[j] = (i0 + j * 1024 < pd.nm) ? load(i0 + j * 1024) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NPT; ++j)
      if (i0 + j * 1024 < pd.nm) atomicAdd(&hist[key_of(q[j])], 1);
  }
  __syncthreads();
  // ---- exclusive scan in place.  Wave w owns the cells [w * seg, (w + 1) * seg) and walks them 64 at a time (lane =
  // consecutive cell: no bank conflicts; a thread owning consecutive cells would put all 64 lanes on one bank), carrying
  // the running total; then every cell gets the total of the waves before its own.
  {
    const int seg = (ncell + 15) / 16;
    int carry     = 0;
    for (int c = wid * seg + lane; c - lane < min((wid + 1) * seg, ncell); c += 64) {
      const bool in = c < min((wid + 1) * seg, ncell);
      const int v   = in ? hist[c] : 0;
      int incl      = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      if (in) hist[c] = carry + incl - v;
      carry += __shfl(incl, 63);
    }
    if (lane == 0) wsum[wid] = carry;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wid; ++w) before += wsum[w];
    for (int c = wid * seg + lane; c < min((wid + 1) * seg, ncell); c += 64) hist[c] += before;
  }
  __syncthreads();
#if defined(SRRG2_MSORT_EXPERIMENT) && SRRG2_MSORT_EXPERIMENT == 2
  if (kbits > 0) return;  // (timing experiment: no scatter pass at all)
#endif
  // ---- pass 3: scatter (the caller's index travels in .w; the order inside a cell does not matter: see above)
  const float* nbase = nsrc ? nsrc + (size_t) pd.moff * nsf : nullptr;
  for (int i0 = tid; i0 < pd.nm; i0 += NPT * 1024) {
    float4 q[NPT], nq[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = i0 + j * 1024;
      q[j]  = i < pd.nm ? load(i) : make_float4(0.f, 0.f, 0.f, 0.f);
      nq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nbase && i < pd.nm) {
        const float* p = nbase + (size_t) i * nsf;
        nq[j]          = make_float4(p[0], p[1], dim == 3 ? p[2] : 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = i0 + j * 1024;
      if (i >= pd.nm) continue;
      const int pos = atomicAdd(&hist[key_of(q[j])], 1);
      q[j].w        = __int_as_float(seg0 + i);  // the caller's index within its problem
#if defined(SRRG2_MSORT_EXPERIMENT) && SRRG2_MSORT_EXPERIMENT == 1
      if (pos < 0) {  // (timing experiment: the scatter without its stores; wrong results)
#endif
      out_pts[pd.moff + pos] = q[j];
      if (nbase) out_nrm[pd.moff + pos] = nq[j];
#if defined(SRRG2_MSORT_EXPERIMENT) && SRRG2_MSORT_EXPERIMENT == 1
      }
#endif
    }
  }
}

// ============================================================================================
// launchers
// ============================================================================================
namespace srrg2amd {

void launch_ingest(const float* src, int stride_floats, int n, int dim, float4* dst, unsigned* maxabs_bits,
                   int finite_per_point, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_ingest, dim3((n + 255) / 256), dim3(256), 0, s, src, stride_floats, n, dim, dst, maxabs_bits,
                     finite_per_point);
}

void launch_ingest_batch(const float* src, int stride_floats, const ProblemDev* probs, int K, int max_nm, int dim,
                         float4* dst, unsigned* maxabs_bits, int finite_per_point, hipStream_t s) {
  if (K <= 0 || max_nm <= 0) return;
  int bx = (max_nm + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(k_ingest_batch, dim3(bx, K), dim3(256), 0, s, src, stride_floats, probs, dim, dst, maxabs_bits,
                     finite_per_point);
}

void launch_ingest_bbox(const float* src, int stride_floats, int n, int dim, float4* dst, unsigned* maxabs_bits, unsigned* mn,
                        unsigned* mx, int* nvalid, hipStream_t s, unsigned* ticket, unsigned* host_out, unsigned seq,
                        unsigned* block_out, unsigned* clear_after) {
  if (n <= 0) return;
  // (with the rows: few blocks -- every block ends in ONE atomic on the ticket word, and device-scope atomics on one address cost
  // ~80 ns each: on a 100 k-point cloud 48 .. 128 blocks of 256 threads take 9.6 - 10.5 us, 1 024-thread blocks 14 - 15 us; one block
  // per 8 Ki points between 64 and INGEST_BBOX_MAX_BLOCKS, so that a cloud of millions still has its loads in flight)
  if (ticket && host_out && block_out) {
    const int bx  = (n + 255) / 256;
    int cap       = n / 8192;
    cap           = cap < 64 ? 64 : (cap > INGEST_BBOX_MAX_BLOCKS ? INGEST_BBOX_MAX_BLOCKS : cap);
    hipLaunchKernelGGL(k_ingest_bbox, dim3(bx < cap ? bx : cap), dim3(256), 0, s, src, stride_floats, n, dim, dst, maxabs_bits, mn, mx, nvalid,
                       ticket, host_out, seq, block_out, clear_after);
    return;
  }
  const int bx = (n + 255) / 256;
  hipLaunchKernelGGL(k_ingest_bbox, dim3(bx < 256 ? bx : 256), dim3(256), 0, s, src, stride_floats, n, dim, dst, maxabs_bits, mn, mx, nvalid,
                     nullptr, nullptr, 0u, nullptr, nullptr);
}
void launch_bbox(const float4* pts, int n, unsigned* mn, unsigned* mx, int* nvalid, hipStream_t s) {
  if (n <= 0) return;
  const int bx = (n + 255) / 256;
  hipLaunchKernelGGL(k_bbox, dim3(bx < 64 ? bx : 64), dim3(256), 0, s, pts, n, mn, mx, nvalid);
}

void launch_grid_count(const GridDev& g, const float4* pts, int n, int* counts, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_grid_count, dim3((n + 255) / 256), dim3(256), 0, s, g, pts, n, counts);
}

void launch_count_nonzero(const int* counts, int n, int* out, hipStream_t s) {
  if (n <= 0) return;
  int nb = (n + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(k_count_nonzero, dim3(nb), dim3(256), 0, s, counts, n, out);
}

int scan_num_blocks(int n) {
  return (n + SCAN_TILE - 1) / SCAN_TILE;
}

void launch_exclusive_scan(int* data, int n, int* block_sums, int* grand_total, hipStream_t s, int* copy) {
  int nb = scan_num_blocks(n);
  hipLaunchKernelGGL(k_scan_tiles, dim3(nb), dim3(SCAN_THREADS), 0, s, data, n, block_sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, s, block_sums, nb, grand_total);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, s, data, n, block_sums, grand_total, copy);
}

void launch_grid_scatter(const GridDev& g, const float4* pts, const float4* nrm, int n, int* cursor, float4* out_pts,
                         float4* out_nrm, int* pos_of, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_grid_scatter, dim3((n + 255) / 256), dim3(256), 0, s, g, pts, nrm, n, cursor, out_pts, out_nrm, pos_of);
}

void launch_msort(const float4* pts, const float4* nrm, const ProblemDev* probs, int K, int max_nm, int kbits, int aniso,
                  unsigned* bb, int* counts, int* cursor, int* scan_sums, int* scan_total, float4* out_pts,
                  float4* out_nrm, hipStream_t s) {
  if (K <= 0 || max_nm <= 0) return;
  int bx = (max_nm + 255) / 256;
  if (bx > 1024) bx = 1024;
  dim3 grid(bx, K);
  if (((long long) K << kbits) > 0x3fffffffLL) return;  // (the caller lowers kbits first: upload_moving)
  const int ncell = K << kbits;
  // (few, grid-striding blocks per problem for the bounding box: its cost is the atomics, not the reads)
  hipLaunchKernelGGL(k_msort_bbox, dim3(bx < 32 ? bx : 32, K), dim3(256), 0, s, pts, probs, bb);
  hipLaunchKernelGGL(k_msort_count, grid, dim3(256), 0, s, pts, probs, bb, kbits, aniso, counts);
  launch_exclusive_scan(counts, ncell, scan_sums, scan_total, s, cursor);  // (the cursors beside the starts: no copy behind the scan)
  hipLaunchKernelGGL(k_msort_scatter, grid, dim3(256), 0, s, pts, nrm, probs, bb, kbits, aniso, cursor, out_pts, out_nrm);
}

// counts (ent == null: per-cell entry counts into list_start[0 .. ncell)) or fills the cell neighbour lists
void launch_cnl_build(const GridDev& g, const GridLists& L, const int4* offs, int noffs, int* list_start, uint2* box_at,
                      uint4* ent, hipStream_t s) {
  const int ncell = L.lnx * L.lny * L.lnz;
  if (ncell <= 0 || noffs <= 0) return;
  if (!ent) {
    hipLaunchKernelGGL(k_cnl_count, dim3((ncell + 255) / 256), dim3(256), 0, s, g, L, offs, noffs, ncell, list_start);
  } else {
    const int gcell = g.nx * g.ny * g.nz;
    hipLaunchKernelGGL(k_cnl_boxes, dim3((gcell + 255) / 256), dim3(256), 0, s, g, gcell, box_at);
    hipLaunchKernelGGL(k_cnl_fill, dim3((ncell + 255) / 256), dim3(256), 0, s, g, L, offs, noffs, ncell, list_start, box_at, ent);
  }
}

// false: the key space does not fit in LDS (the caller takes the ingest + global-histogram path)
bool launch_msort_local(const float* src, int sf, const float* nsrc, int nsf, const ProblemDev* probs, int K, int dim, int kbits,
                        int aniso, int segments, int max_nm, float4* out_pts, float4* out_nrm, unsigned* maxabs_bits,
                        hipStream_t s) {
  if (kbits > 15) return false;
  if (K <= 0) return true;
  // segments per problem: fill about half the chip's CUs (one 1024-thread workgroup each); segments get the coarser key
  // space of the big batches (16^3 cells for ~6-12 k points)
  const int seg_env = segments;
  // (measured on C4, profiles/archive/r2zk_ab_msort_segments.txt: 32 alignments 287 -> 305 k it/s with 4 segments, 8 alignments
  // 146 -> 161 k, 64 alignments 331 -> 345 k with 2; the passes lose 1.4 % of coherence, the sort goes from 118 to ~35 us;
  // 128 alignments 362 -> 372 k with 2 (profiles/archive/r2zs_env_tests.txt); 256: within noise, one workgroup per cloud)
  int G = seg_env > 0 ? seg_env : (K >= 256 ? 1 : (K >= 64 ? 2 : (128 / K < 8 ? 128 / K : 8)));
  // (round 6, late: a call of one to four clouds -- a tracker's set_moving, every frame -- gets a segment per ~3 Ki points, up to 64
  // workgroups in all: eight 1024-thread workgroups sorted a 100 k-point cloud in 67 us, a latency chain on eight CUs that the
  // frame's set_fixed then waits behind; 32 segments: ~20 us.  The passes lose a little coherence -- a tracker's compute() 0.194 ->
  // 0.198 ms -- the frame gains 30 us: 0.452-0.475 -> 0.423-0.426 ms.  C2 with the fixed cloud kept: unchanged, 0.168 ms)
  if (seg_env <= 0 && K <= 4) {
    const int by_size = max_nm / 3072, room = 64 / K;
    G = G > (by_size < room ? by_size : room) ? G : (by_size < room ? by_size : room);
  }
  if (G < 1) G = 1;
  if (G > 1) {
    if (!aniso) kbits = kbits < 12 ? kbits : 12;
    (void) hipMemsetAsync(maxabs_bits, 0, (size_t) K * sizeof(unsigned), s);
  }
  // (anisotropic keys: 2^15 cells when a workgroup sorts >= 8 Ki points -- clearing and scanning the histogram is then
  // no more than the points themselves -- else 2^12)
  if (aniso && kbits > 12 && max_nm / G < 8192) kbits = 12;
  const size_t lds = sizeof(int) << kbits;
  static bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_msort_local),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int) (sizeof(int) << 15)) == hipSuccess;
  if (!attr_ok) {
    (void) hipGetLastError();
    return false;
  }
  hipLaunchKernelGGL(k_msort_local, dim3(G, K), dim3(1024), lds, s, src, sf, nsrc, nsf, probs, dim, kbits, aniso, out_pts, out_nrm,
                     maxabs_bits);
  return true;
}

}  // namespace srrg2amd
