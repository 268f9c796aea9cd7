// kernels.hip -- hand-written gfx950 kernels of the aligner hot path.
//
//   ingest / grid build      : CorrespondenceFinder_ search-structure construction, once per setFixed
//                              (S/registration/correspondence_finder.h:80-91)
//   icp_step<DIM,PLANE>      : ONE kernel per slice per ICP iteration = finder->compute()
//                              (aligner_slice_processor_impl.cpp:39-48) fused with the factor's
//                              per-correspondence linearisation + robustifier + JtJ/Jtr reduction
//                              ([EXT] FactorCorrespondenceDriven_, SURVEY.md A1+A3+A4)
//   icp_control              : the rest of one _runSolver iteration on device: association check,
//                              prior factors, 6x6 solve, X <- X [+] dx, IterationStats, termination
//                              (multi_aligner_impl.cpp:104-127; aligner_termination_criteria_impl.cpp:24-65)
//   icp_init / icp_post / icp_finalize : compute() prologue/epilogue (multi_aligner_impl.cpp:52-95,163-181)
//
// Arithmetic follows DESIGN.md "arithmetic specification": float32 geometry in a fixed operation
// order without FMA contraction (-ffp-contract=off), products widened to double, sums carried as
// exact 64-bit fixed point so that the reduction order (lanes, waves, blocks, GPUs) cannot change
// a single bit of H, b or the statistics.
#include "kernels.h"

#include "det_math.h"

namespace {

__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// monotone float -> unsigned key (for atomicMin/atomicMax on floats of either sign)
__device__ __forceinline__ unsigned fkey(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ int cell_coord(float x, float o, float inv_h) {
  float u = (x - o) * inv_h;
  u       = fminf(fmaxf(u, -2048.f), 4096.f);
  return (int) floorf(u);
}

__device__ __forceinline__ float bound2_of(int r, float h) {
  float b = ((float) r - 0.01f) * h;
  return (b * b) * 0.9999f;
}

__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

}  // namespace

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
  if ((threadIdx.x & 63) == 0 && maxabs_bits && amax > 0.f) atomicMax(maxabs_bits, __float_as_uint(amax));
}

// ============================================================================================
// grid build
// ============================================================================================
__global__ void k_bbox(const float4* __restrict__ pts, int n, unsigned* __restrict__ mn, unsigned* __restrict__ mx,
                       int* __restrict__ nvalid) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  if (!finite3(p.x, p.y, p.z)) return;
  atomicMin(&mn[0], fkey(p.x)); atomicMax(&mx[0], fkey(p.x));
  atomicMin(&mn[1], fkey(p.y)); atomicMax(&mx[1], fkey(p.y));
  atomicMin(&mn[2], fkey(p.z)); atomicMax(&mx[2], fkey(p.z));
  atomicAdd(nvalid, 1);
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

__global__ void k_scan_add(int* __restrict__ data, int n, const int* __restrict__ block_sums,
                           const int* __restrict__ grand_total) {
  int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int add  = block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) data[base + k] += add;
  if (blockIdx.x == 0 && threadIdx.x == 0) data[n] = *grand_total;  // cell_start[ncell]
}

__global__ void k_grid_scatter(GridDev g, const float4* __restrict__ pts, const float4* __restrict__ nrm, int n,
                               int* __restrict__ cursor, float4* __restrict__ out_pts, float4* __restrict__ out_nrm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  if (!finite3(p.x, p.y, p.z)) return;
  int pos      = atomicAdd(&cursor[grid_cell_of(g, p)], 1);
  p.w          = __int_as_float(i);
  out_pts[pos] = p;
  if (nrm) out_nrm[pos] = nrm[i];
}

// ============================================================================================
// the fused ICP step kernel
// ============================================================================================
template <int DIM>
__device__ __forceinline__ void scan_cube(const GridDev& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                          int r, float& best, int& bidx, int& bpos) {
  int z0 = DIM == 3 ? max(cz - r, 0) : 0, z1 = DIM == 3 ? min(cz + r, g.nz - 1) : 0;
  int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
  int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
  if (x0 > x1) return;
  for (int z = z0; z <= z1; ++z) {
    for (int y = y0; y <= y1; ++y) {
      int row = (z * g.ny + y) * g.nx;
      int s = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
      for (int j = s; j < e; ++j) {
        float4 f = g.pts[j];
        float dx = f.x - qx, dy = f.y - qy;
        float d2 = dx * dx + dy * dy;
        if (DIM == 3) {
          float dz = f.z - qz;
          d2       = d2 + dz * dz;
        }
        int idx = __float_as_int(f.w);
        if (d2 < best || (d2 == best && idx < bidx)) {
          best = d2;
          bidx = idx;
          bpos = j;
        }
      }
    }
  }
}

template <int DIM, bool PLANE>
__global__ __launch_bounds__(256) void k_icp_step(SliceDev S, const ProblemDev* __restrict__ probs,
                                                  ProblemState* __restrict__ states, int slot) {
  constexpr int D    = DIM == 3 ? 6 : 3;
  constexpr int ROWS = PLANE ? 1 : DIM;
  const int prob     = blockIdx.y;
  ProblemState* st   = &states[prob];
  if (st->done || st->finished) return;
  const ProblemDev pd = probs[prob];

  // finder->setLocalMapInSensor(robot_in_sensor * X), aligner_slice_processor_impl.cpp:35
  float T[12];
  if (DIM == 3) {
    dm::se3_compose(S.Sinv, st->X, T);
  } else {
    float t9[9];
    dm::se2_compose(S.Sinv, st->X, t9);
    // spread the 3x3 into the 3x4 slots used below: rows [r0 r1 t]
    T[0] = t9[0]; T[1] = t9[1]; T[2] = 0.f; T[3] = t9[2];
    T[4] = t9[3]; T[5] = t9[4]; T[6] = 0.f; T[7] = t9[5];
    T[8] = 0.f; T[9] = 0.f; T[10] = 1.f; T[11] = 0.f;
  }
  const int kexp     = st->kexp[S.slice_idx];
  const double scale = dm::pow2(kexp);
  const int rk       = (st->phase == 1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  const float thr    = S.robust_thr;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const GridDev& g   = S.grid;
  const float b2_1   = bound2_of(1, g.h);

  long long acc[ACC_N];
#pragma unroll
  for (int a = 0; a < ACC_N; ++a) acc[a] = 0;

  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pd.nm; i += gridDim.x * blockDim.x) {
    const int gi   = pd.moff + i;
    const float4 p = S.mpts[gi];
    int match      = -1;
    float resp     = 0.f;
    uint8_t fstat  = SRRG2_FACTOR_SUPPRESSED;
    if (finite3(p.x, p.y, p.z)) {
      float qx, qy, qz = 0.f;
      if constexpr (DIM == 3) {
        qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
        qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
        qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
      } else {
        qx = (T[0] * p.x + T[1] * p.y) + T[3];
        qy = (T[4] * p.x + T[5] * p.y) + T[7];
      }
      const int cx = cell_coord(qx, g.ox, g.inv_h);
      const int cy = cell_coord(qy, g.oy, g.inv_h);
      const int cz = DIM == 3 ? cell_coord(qz, g.oz, g.inv_h) : 0;
      float best   = INFINITY;
      int bidx = NO_MATCH, bpos = 0;
      scan_cube<DIM>(g, qx, qy, qz, cx, cy, cz, 1, best, bidx, bpos);
      bool found = bidx != NO_MATCH && best <= g.gate2;
      if (!(found && best <= b2_1) && g.rmax > 1) {
        int r2 = g.rmax;
        if (found) {
          r2 = 1;
          while (r2 < g.rmax && bound2_of(r2, g.h) < best) ++r2;
        }
        if (r2 > 1) scan_cube<DIM>(g, qx, qy, qz, cx, cy, cz, r2, best, bidx, bpos);
      }
      found = bidx != NO_MATCH && best <= g.gate2;
      float4 nf = make_float4(0.f, 0.f, 0.f, 0.f);
      if (found && (PLANE || S.use_normal_gate)) nf = g.nrm[bpos];
      if (found && S.use_normal_gate) {
        const float4 nm = S.mnrm[gi];
        float dot;
        if (DIM == 3) {
          float rx = (T[0] * nm.x + T[1] * nm.y) + T[2] * nm.z;
          float ry = (T[4] * nm.x + T[5] * nm.y) + T[6] * nm.z;
          float rz = (T[8] * nm.x + T[9] * nm.y) + T[10] * nm.z;
          dot      = (nf.x * rx + nf.y * ry) + nf.z * rz;
        } else {
          float rx = T[0] * nm.x + T[1] * nm.y;
          float ry = T[4] * nm.x + T[5] * nm.y;
          dot      = nf.x * rx + nf.y * ry;
        }
        if (!(dot > S.normal_cos)) found = false;
      }
      if (found) {
        match          = bidx;
        resp           = best;
        const float4 f = g.pts[bpos];
        float J[ROWS][D];
        float e[ROWS];
        if constexpr (DIM == 3) {
          float m[ROWS][3];
          if (PLANE) {
            e[0]    = (nf.x * (qx - f.x) + nf.y * (qy - f.y)) + nf.z * (qz - f.z);
            m[0][0] = (T[0] * nf.x + T[4] * nf.y) + T[8] * nf.z;
            m[0][1] = (T[1] * nf.x + T[5] * nf.y) + T[9] * nf.z;
            m[0][2] = (T[2] * nf.x + T[6] * nf.y) + T[10] * nf.z;
          } else {
            const float q[3]  = {qx, qy, qz};
            const float ff[3] = {f.x, f.y, f.z};
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              e[r]    = q[r] - ff[r];
              m[r][0] = T[r * 4 + 0];
              m[r][1] = T[r * 4 + 1];
              m[r][2] = T[r * 4 + 2];
            }
          }
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            J[r][0]     = m[r][0];
            J[r][1]     = m[r][1];
            J[r][2]     = m[r][2];
            J[r][D - 3] = kk * (p.y * m[r][2] - p.z * m[r][1]);
            J[r][D - 2] = kk * (p.z * m[r][0] - p.x * m[r][2]);
            J[r][D - 1] = kk * (p.x * m[r][1] - p.y * m[r][0]);
          }
        } else {
          float m[ROWS][2];
          if (PLANE) {
            e[0]    = nf.x * (qx - f.x) + nf.y * (qy - f.y);
            m[0][0] = T[0] * nf.x + T[4] * nf.y;
            m[0][1] = T[1] * nf.x + T[5] * nf.y;
          } else {
            const float q[2]  = {qx, qy};
            const float ff[2] = {f.x, f.y};
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              e[r]    = q[r] - ff[r];
              m[r][0] = T[r * 4 + 0];
              m[r][1] = T[r * 4 + 1];
            }
          }
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            J[r][0] = m[r][0];
            J[r][1] = m[r][1];
            J[r][2] = m[r][1] * p.x - m[r][0] * p.y;
          }
        }
        float chi = e[0] * e[0];
#pragma unroll
        for (int r = 1; r < ROWS; ++r) chi = chi + e[r] * e[r];
        acc[ACC_N_CORR] += 1;
        if (isfinite(chi)) {
          float w = 1.f;
          bool kernelized = false;
          if (rk != SRRG2_ROBUST_NONE && !(chi < thr)) {
            kernelized = true;
            w = rk == SRRG2_ROBUST_CLAMP ? 0.f : (rk == SRRG2_ROBUST_SATURATED ? thr / chi : 1.0f / (1.0f + chi / thr));
          }
          const long long chi_fx = __double2ll_rn((double) chi * scale);
          if (kernelized) {
            fstat = SRRG2_FACTOR_KERNELIZED;
            acc[ACC_N_OUT] += 1;
            acc[ACC_CHI_OUT] += chi_fx;
          } else {
            fstat = SRRG2_FACTOR_INLIER;
            acc[ACC_N_IN] += 1;
            acc[ACC_CHI_IN] += chi_fx;
          }
          if (w != 0.f) {
#pragma unroll
            for (int a = 0; a < D; ++a) {
              double wj[ROWS];
#pragma unroll
              for (int r = 0; r < ROWS; ++r) wj[r] = (double) w * (double) J[r][a];
#pragma unroll
              for (int b = a; b < D; ++b) {
                double t = wj[0] * (double) J[0][b];
#pragma unroll
                for (int r = 1; r < ROWS; ++r) t = t + wj[r] * (double) J[r][b];
                acc[hidx(a, b)] += __double2ll_rn(t * scale);
              }
              double t = wj[0] * (double) e[0];
#pragma unroll
              for (int r = 1; r < ROWS; ++r) t = t + wj[r] * (double) e[r];
              acc[ACC_B + a] += __double2ll_rn(t * scale);
            }
          }
        }
      }
    }
    S.corr_fixed[gi] = match;
    S.corr_resp[gi]  = resp;
    S.corr_stat[gi]  = fstat;
  }

  // block reduction: wave shuffles, then 4 waves through LDS, then one 64-bit atomic per entry
  __shared__ long long red[4][ACC_N];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < ACC_N; ++a) {
    long long v = wave_sum(acc[a]);
    if (lane == 0) red[wid][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < ACC_N) {
    long long v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (v != 0) {
      unsigned long long* dst = S.acc + ((size_t) prob * S.slots + slot) * ACC_N + threadIdx.x;
      atomicAdd(dst, (unsigned long long) v);
    }
  }
}

// ============================================================================================
// control kernels (one thread per problem)
// ============================================================================================
namespace {

__device__ int slice_exponent(const CtlParams& C, const SliceCtl& s, int prob, int nm) {
  const bool plane = s.kind == SRRG2_SLICE_P2PLANE;
  const double kk  = C.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.0 : 1.0;
  const int dim    = C.variable_kind == SRRG2_SE2_RIGHT ? 2 : 3;
  const int rows   = plane ? 1 : dim;
  const float pinf = __uint_as_float(s.pinf_bits[prob]);
  const float ninf = __uint_as_float(s.ninf_bits[0]);
  double mb        = plane ? (1.7320508075688772 * (double) ninf) * 1.01 : 1.01;
  double pf        = (2.0 * kk) * (double) pinf;
  double jb        = mb * (pf > 1.0 ? pf : 1.0);
  double eb        = (mb * (double) s.gate) * 1.01;
  double mx        = jb > eb ? jb : eb;
  double B         = (double) rows * (mx * mx);
  return dm::fixed_point_exponent(nm, B);
}

__device__ float robust_weight(int kind, float thr, float chi, bool& kernelized) {
  if (kind == SRRG2_ROBUST_NONE || chi < thr) {
    kernelized = false;
    return 1.f;
  }
  kernelized = true;
  if (kind == SRRG2_ROBUST_CLAMP) return 0.f;
  if (kind == SRRG2_ROBUST_SATURATED) return thr / chi;
  return 1.0f / (1.0f + chi / thr);
}

// SE2PriorErrorFactor / SE3PriorErrorFactorAD ([EXT]) at the current estimate
template <int D>
__device__ void prior_linearize(int variable_kind, const SliceCtl& s, int rk, const float* X, double* H, double* b,
                                double& chi_out, int& status) {
  double e[D], J[D * D];
  for (int i = 0; i < D * D; ++i) J[i] = 0.0;
  float Zinv[12], E[12];
  if (D == 6) {
    dm::se3_inverse(s.prior_Z, Zinv);
    dm::se3_compose(Zinv, X, E);
    dm::se3_t2v_quat(E, e);
    double n2 = (e[3] * e[3] + e[4] * e[4]) + e[5] * e[5];
    double w  = n2 < 1.0 ? sqrt(1.0 - n2) : 0.0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) J[i * 6 + j] = (double) E[i * 4 + j];
    J[3 * 6 + 3] = w;      J[3 * 6 + 4] = -e[5];  J[3 * 6 + 5] = e[4];
    J[4 * 6 + 3] = e[5];   J[4 * 6 + 4] = w;      J[4 * 6 + 5] = -e[3];
    J[5 * 6 + 3] = -e[4];  J[5 * 6 + 4] = e[3];   J[5 * 6 + 5] = w;
  } else {
    dm::se2_inverse(s.prior_Z, Zinv);
    dm::se2_compose(Zinv, X, E);
    dm::se2_t2v(E, e);
    J[0] = (double) E[0]; J[1] = (double) E[1];
    J[3] = (double) E[3]; J[4] = (double) E[4];
    J[8] = 1.0;
  }
  double chi = 0.0;
  for (int i = 0; i < D; ++i) chi = chi + (e[i] * (double) s.prior_info[i]) * e[i];
  bool kernelized;
  float w = robust_weight(rk, s.robust_thr, (float) chi, kernelized);
  chi_out = chi;
  status  = !isfinite(chi) ? (int) SRRG2_FACTOR_SUPPRESSED
                           : (kernelized ? (int) SRRG2_FACTOR_KERNELIZED : (int) SRRG2_FACTOR_INLIER);
  for (int i = 0; i < D * D; ++i) H[i] = 0.0;
  for (int i = 0; i < D; ++i) b[i] = 0.0;
  if (status == SRRG2_FACTOR_SUPPRESSED) return;
  for (int a = 0; a < D; ++a) {
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
      for (int r = 0; r < D; ++r) t = t + (J[r * D + a] * (double) s.prior_info[r]) * J[r * D + c];
      H[a * D + c] = (double) w * t;
    }
    double t = 0.0;
    for (int r = 0; r < D; ++r) t = t + (J[r * D + a] * (double) s.prior_info[r]) * e[r];
    b[a] = (double) w * t;
  }
}

__device__ int num_correspondences(const CtlParams& C, const ProblemState* st) {
  int n = 0;
  for (int s = 0; s < C.nslices; ++s) {
    int c = C.slices[s].kind == SRRG2_SLICE_PRIOR ? 1 : st->ncorr[s];
    if (c >= 0) n += c;
  }
  return n;
}

__device__ double win_max(const double* b, int n) {
  double m = b[0];
  for (int i = 1; i < n; ++i)
    if (b[i] > m) m = b[i];
  return m;
}
__device__ double win_min(const double* b, int n) {
  double m = b[0];
  for (int i = 1; i < n; ++i)
    if (b[i] < m) m = b[i];
  return m;
}

// AlignerTerminationCriteriaStandard_::hasToStop, aligner_termination_criteria_impl.cpp:24-65
__device__ bool has_to_stop(const CtlParams& C, ProblemState* st, const srrg2_iteration_stats& cur) {
  int ncorr = num_correspondences(C, st);
  int ninl  = cur.num_inliers;
  int nout  = cur.num_outliers;
  float chi = cur.chi_inliers / (float) ninl;
  if (!ninl) return false;
  const int W = C.term.window_size;
  int slot    = st->w_count % W;
  st->w_corr[slot] = ncorr;
  st->w_inl[slot]  = ninl;
  st->w_out[slot]  = nout;
  st->w_chi[slot]  = (double) chi;
  st->w_count++;
  int n = st->w_count < W ? st->w_count : W;
  if (n < W) return false;
  if (win_max(st->w_out, n) - win_min(st->w_out, n) > (double) C.term.num_correspondences_range) return false;  // :46
  if (win_max(st->w_inl, n) - win_min(st->w_inl, n) > (double) C.term.num_inliers_range) return false;
  float chi_range = (float) (win_max(st->w_chi, n) - win_min(st->w_chi, n));
  if (chi_range > (float) C.term.num_outliers_range) return false;  // :53
  if (chi_range / (float) win_max(st->w_chi, n) > C.term.chi_epsilon) return false;
  return true;
}

template <int D>
__device__ void control_body(const CtlParams& C, ProblemState* st, srrg2_iteration_stats* stats, int prob, int slot) {
  // association check: association_good |= slice->correspondencesGood(), multi_aligner.h:126-138
  bool good = false;
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    if (sc.kind == SRRG2_SLICE_PRIOR) {
      good = true;  // aligner_slice_processor_prior.h:66-68
      continue;
    }
    const unsigned long long* acc = sc.acc + ((size_t) prob * sc.slots + slot) * ACC_N;
    int nc       = (int) (long long) acc[ACC_N_CORR];
    st->ncorr[s] = nc;
    good |= nc > sc.min_num_correspondences;  // aligner_slice_processor_impl.cpp:77-79
  }
  if (!good) {
    st->status = SRRG2_NOT_ENOUGH_CORRESPONDENCES;  // multi_aligner_impl.cpp:107-111
    st->done   = 1;
    return;
  }
  // solver->compute(): one Gauss-Newton iteration over all slices' factors
  double H[D * D], b[D], dx[D];
  for (int i = 0; i < D * D; ++i) H[i] = 0.0;
  for (int i = 0; i < D; ++i) b[i] = 0.0;
  srrg2_iteration_stats cur;
  cur.iteration = st->nstats;
  cur.num_inliers = cur.num_outliers = cur.num_suppressed = 0;
  double chi_in = 0.0, chi_out = 0.0;
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    const int rk = (st->phase == 1 && sc.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : sc.robust_kind;
    if (sc.kind == SRRG2_SLICE_PRIOR) {
      double pH[D * D], pb[D], pchi;
      int pstat;
      prior_linearize<D>(C.variable_kind, sc, rk, st->X, pH, pb, pchi, pstat);
      for (int i = 0; i < D * D; ++i) H[i] = H[i] + pH[i];
      for (int i = 0; i < D; ++i) b[i] = b[i] + pb[i];
      if (pstat == SRRG2_FACTOR_INLIER) {
        cur.num_inliers++;
        chi_in = chi_in + pchi;
      } else if (pstat == SRRG2_FACTOR_KERNELIZED) {
        cur.num_outliers++;
        chi_out = chi_out + pchi;
      } else {
        cur.num_suppressed++;
      }
      st->ninl[s] = pstat == SRRG2_FACTOR_INLIER ? 1 : 0;
      continue;
    }
    const unsigned long long* acc = sc.acc + ((size_t) prob * sc.slots + slot) * ACC_N;
    const double inv = dm::pow2(-st->kexp[s]);
    for (int a = 0; a < D; ++a) {
      for (int c = a; c < D; ++c) {
        double v     = (double) (long long) acc[hidx(a, c)] * inv;
        H[a * D + c] = H[a * D + c] + v;
        if (c != a) H[c * D + a] = H[c * D + a] + v;
      }
      b[a] = b[a] + (double) (long long) acc[ACC_B + a] * inv;
    }
    int n_in = (int) (long long) acc[ACC_N_IN], n_out = (int) (long long) acc[ACC_N_OUT];
    int n_c  = (int) (long long) acc[ACC_N_CORR];
    cur.num_inliers += n_in;
    cur.num_outliers += n_out;
    cur.num_suppressed += n_c - n_in - n_out;
    chi_in      = chi_in + (double) (long long) acc[ACC_CHI_IN] * inv;
    chi_out     = chi_out + (double) (long long) acc[ACC_CHI_OUT] * inv;
    st->ninl[s] = n_in;
  }
  cur.num_correspondences = num_correspondences(C, st);
  cur.chi_inliers         = (float) chi_in;
  cur.chi_outliers        = (float) chi_out;
  int bad                 = dm::solve<D>(H, b, dx);
  cur.solver_status       = bad ? 1 : 0;
  for (int i = 0; i < D * D; ++i) st->last_H[i] = H[i];
  for (int i = 0; i < D; ++i) {
    st->last_b[i]  = b[i];
    st->last_dx[i] = bad ? 0.0 : dx[i];
  }
  if (!bad) dm::box_plus(C.variable_kind, st->X, dx);  // solver Success: multi_aligner_impl.cpp:118-121
  if (st->nstats < C.max_stats) stats[(size_t) prob * C.max_stats + st->nstats] = cur;
  st->nstats++;
  if (C.has_term && has_to_stop(C, st, cur)) st->done = 1;  // :124-126
}

}  // namespace

// compute() prologue: term_crit->init, stats clear, _preCompute (prior init overrides the guess)
__global__ void k_icp_init(CtlParams C, const ProblemDev* __restrict__ probs, ProblemState* __restrict__ states,
                           const float* __restrict__ guesses, int tsize) {
  int prob = blockIdx.x * blockDim.x + threadIdx.x;
  if (prob >= C.K) return;
  ProblemState* st = &states[prob];
  for (int i = 0; i < 12; ++i) st->X[i] = i < tsize ? guesses[(size_t) prob * tsize + i] : 0.f;
  st->status   = SRRG2_FAIL;
  st->done     = 0;
  st->finished = 0;
  st->nstats   = 0;
  st->phase    = 0;
  st->w_count  = 0;
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    st->ncorr[s]       = 0;
    st->ninl[s]        = 0;
    if (sc.kind == SRRG2_SLICE_PRIOR) {
      st->kexp[s] = 0;
      if (sc.prior_sets_initial_guess) {  // aligner_slice_odometry_prior.cpp:19,34; aligner_slice_motion_model.hpp:69-70
        for (int i = 0; i < tsize; ++i) st->X[i] = sc.prior_Z[i];
      }
    } else {
      st->kexp[s] = slice_exponent(C, sc, prob, probs[(size_t) s * C.K + prob].nm);
    }
  }
}

__global__ void k_icp_control(CtlParams C, ProblemState* __restrict__ states, srrg2_iteration_stats* __restrict__ stats,
                              int slot) {
  int prob = blockIdx.x * blockDim.x + threadIdx.x;
  if (prob >= C.K) return;
  ProblemState* st = &states[prob];
  if (st->done || st->finished) return;
  if (C.variable_kind == SRRG2_SE2_RIGHT)
    control_body<3>(C, st, stats, prob, slot);
  else
    control_body<6>(C, st, stats, prob, slot);
}

// after the main _runSolver: multi_aligner_impl.cpp:75-85 and the start of _postCompute (:165-171)
__global__ void k_icp_post(CtlParams C, ProblemState* __restrict__ states, const srrg2_iteration_stats* __restrict__ stats) {
  int prob = blockIdx.x * blockDim.x + threadIdx.x;
  if (prob >= C.K) return;
  ProblemState* st = &states[prob];
  if (st->nstats == 0) {
    st->status   = SRRG2_FAIL;
    st->finished = 1;
    return;
  }
  int last = st->nstats < C.max_stats ? st->nstats - 1 : C.max_stats - 1;
  if (stats[(size_t) prob * C.max_stats + last].num_inliers < C.params.min_num_inliers) {
    st->status   = SRRG2_NOT_ENOUGH_INLIERS;
    st->finished = 1;
    return;
  }
  if (C.params.enable_inlier_only_runs) {
    st->phase = 1;  // _setClampRobustifiers
    st->done  = 0;
  }
}

// end of compute(): _pruneCorrespondences bookkeeping, fixTransform, Success (:88-94); fills ProblemOut
__global__ void k_icp_finalize(CtlParams C, ProblemState* __restrict__ states, ProblemOut* __restrict__ outs) {
  int prob = blockIdx.x * blockDim.x + threadIdx.x;
  if (prob >= C.K) return;
  ProblemState* st = &states[prob];
  if (!st->finished) {
    if (C.params.keep_only_inlier_correspondences) {
      for (int s = 0; s < C.nslices; ++s)
        if (C.slices[s].kind != SRRG2_SLICE_PRIOR) st->ncorr[s] = st->ninl[s];
    }
    if (C.variable_kind == SRRG2_SE2_RIGHT)
      dm::se2_fix_transform(st->X);
    else
      dm::se3_fix_transform(st->X);
    st->status = SRRG2_SUCCESS;
  }
  ProblemOut* o = &outs[prob];
  for (int i = 0; i < 12; ++i) o->X[i] = st->X[i];
  o->status = st->status;
  o->nstats = st->nstats;
  for (int s = 0; s < SRRG2_MAX_SLICES; ++s) o->ncorr[s] = st->ncorr[s];
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

void launch_bbox(const float4* pts, int n, unsigned* mn, unsigned* mx, int* nvalid, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_bbox, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, mn, mx, nvalid);
}

void launch_grid_count(const GridDev& g, const float4* pts, int n, int* counts, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_grid_count, dim3((n + 255) / 256), dim3(256), 0, s, g, pts, n, counts);
}

int scan_num_blocks(int n) {
  return (n + SCAN_TILE - 1) / SCAN_TILE;
}

void launch_exclusive_scan(int* data, int n, int* block_sums, int* grand_total, hipStream_t s) {
  int nb = scan_num_blocks(n);
  hipLaunchKernelGGL(k_scan_tiles, dim3(nb), dim3(SCAN_THREADS), 0, s, data, n, block_sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, s, block_sums, nb, grand_total);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, s, data, n, block_sums, grand_total);
}

void launch_grid_scatter(const GridDev& g, const float4* pts, const float4* nrm, int n, int* cursor, float4* out_pts,
                         float4* out_nrm, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_grid_scatter, dim3((n + 255) / 256), dim3(256), 0, s, g, pts, nrm, n, cursor, out_pts, out_nrm);
}

void launch_icp_step(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int slot,
                     int K, int max_nm, hipStream_t s) {
  if (K <= 0 || max_nm <= 0) return;
  int bx = (max_nm + 255) / 256;
  if (bx > 4096) bx = 4096;
  dim3 grid(bx, K);
  if (dim == 3) {
    if (plane)
      hipLaunchKernelGGL((k_icp_step<3, true>), grid, dim3(256), 0, s, S, probs, states, slot);
    else
      hipLaunchKernelGGL((k_icp_step<3, false>), grid, dim3(256), 0, s, S, probs, states, slot);
  } else {
    if (plane)
      hipLaunchKernelGGL((k_icp_step<2, true>), grid, dim3(256), 0, s, S, probs, states, slot);
    else
      hipLaunchKernelGGL((k_icp_step<2, false>), grid, dim3(256), 0, s, S, probs, states, slot);
  }
}

void launch_icp_init(const CtlParams& C, const ProblemDev* probs, ProblemState* states, const float* guesses, int tsize,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_icp_init, dim3((C.K + 63) / 64), dim3(64), 0, s, C, probs, states, guesses, tsize);
}
void launch_icp_control(const CtlParams& C, ProblemState* states, srrg2_iteration_stats* stats, int slot, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_control, dim3((C.K + 63) / 64), dim3(64), 0, s, C, states, stats, slot);
}
void launch_icp_post(const CtlParams& C, ProblemState* states, const srrg2_iteration_stats* stats, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_post, dim3((C.K + 63) / 64), dim3(64), 0, s, C, states, stats);
}
void launch_icp_finalize(const CtlParams& C, ProblemState* states, ProblemOut* outs, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_finalize, dim3((C.K + 63) / 64), dim3(64), 0, s, C, states, outs);
}

}  // namespace srrg2amd
