// posegraph.hip -- block-sparse Gauss-Newton + block-Jacobi PCG for the pose graph.
//
// Replaces global_solver->compute() as called from MultiGraphSLAM_::optimize()
// (S/system/multi_graph_slam_impl.cpp:300-317) on the graph built at :52-90 / :241-293: only pose variables
// (LocalMap2D/3D, S/mapping/local_map.h:64,75) and binary pose-pose factors (S/registration/loop_closure.h:110-111),
// so there is nothing to Schur-eliminate: block-sparse H build + preconditioned CG.
//
//   k_pg_edges<D>     one thread per factor: e, Ji, Jj, Omega products -> Ho[e] = Ji^T W Jj and the factor's
//                     contributions to H_ii, H_jj, b_i, b_j, chi (stored per factor: no atomics)
//   k_pg_vertices<D>  one thread per variable: gathers its factors' contributions in incidence order
//                     (deterministic), adds damping, handles Fixed variables, inverts the 6x6 block (preconditioner)
//   k_pg_pcg_init / k_pg_spmv / k_pg_update_xr / k_pg_update_p   the PCG loop; scalars (alpha, beta, convergence)
//                     are recomputed by every block from the per-block partial dot products, so the loop needs no
//                     host round trip and no atomics
//   k_pg_apply<D>     X_v <- X_v [+] dx_v
//
// float64 throughout (the float32 poses are the only float32 state).  Results agree with the CPU oracle to PCG
// tolerance, not bit for bit (dot-product order differs); tests bound the pose difference.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "det_math.h"
#include "host_util.h"

#define PG_THREADS 256
#define PG_ROWS 252  // scalar rows per block in the PCG kernels: a multiple of 6 and 3, so a variable never straddles blocks
#define PG_MAX_PARTIALS 4096
#define PG_TILES 4  // tiles of PG_ROWS rows per block in the element-wise PCG kernels: 4x fewer partial sums to re-add

namespace {

using srrg2amd::DevBuf;
using srrg2amd::fail;

struct PgScalars {      // device-resident scalars of one PCG solve
  double rz, bb, rr;
  int pcg_iters, done, bad, num_factors;
  double chi;
};

template <int D>
struct EdgeContrib {    // what one factor adds to the system
  double Cii[D * D], Cjj[D * D], bi[D], bj[D], chi;
};

template <int D>
__device__ void edge_linearize(const float* Xi, const float* Xj, const float* Z, double* err, double* Ji, double* Jj) {
  float Xi_inv[12], A[12], Zinv[12], Em[12];
#pragma unroll
  for (int k = 0; k < D * D; ++k) { Ji[k] = 0.0; Jj[k] = 0.0; }
  if (D == 6) {
    dm::se3_inverse(Xi, Xi_inv);
    dm::se3_compose(Xi_inv, Xj, A);
    dm::se3_inverse(Z, Zinv);
    dm::se3_compose(Zinv, A, Em);
    dm::se3_t2v_quat(Em, err);
    double n2 = (err[3] * err[3] + err[4] * err[4]) + err[5] * err[5];
    double w  = n2 < 1.0 ? sqrt(1.0 - n2) : 0.0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Jj[r * 6 + c] = (double) Em[r * 4 + c];
    Jj[3 * 6 + 3] = w;       Jj[3 * 6 + 4] = -err[5]; Jj[3 * 6 + 5] = err[4];
    Jj[4 * 6 + 3] = err[5];  Jj[4 * 6 + 4] = w;       Jj[4 * 6 + 5] = -err[3];
    Jj[5 * 6 + 3] = -err[4]; Jj[5 * 6 + 4] = err[3];  Jj[5 * 6 + 5] = w;
    double M[36];
    for (int k = 0; k < 36; ++k) M[k] = 0.0;
    double RAt[9], tA[3] = {(double) A[3], (double) A[7], (double) A[11]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) RAt[r * 3 + c] = (double) A[c * 4 + r];
    double tx[9] = {0, -tA[2], tA[1], tA[2], 0, -tA[0], -tA[1], tA[0], 0};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        M[r * 6 + c]           = RAt[r * 3 + c];
        M[(r + 3) * 6 + c + 3] = RAt[r * 3 + c];
        double s               = 0.0;
        for (int k = 0; k < 3; ++k) s = s + RAt[r * 3 + k] * tx[k * 3 + c];
        M[r * 6 + c + 3] = -2.0 * s;
      }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s = s + Jj[r * 6 + k] * M[k * 6 + c];
        Ji[r * 6 + c] = -s;
      }
  } else {
    dm::se2_inverse(Xi, Xi_inv);
    dm::se2_compose(Xi_inv, Xj, A);
    dm::se2_inverse(Z, Zinv);
    dm::se2_compose(Zinv, A, Em);
    dm::se2_t2v(Em, err);
    Jj[0] = (double) Em[0]; Jj[1] = (double) Em[1];
    Jj[3] = (double) Em[3]; Jj[4] = (double) Em[4];
    Jj[8] = 1.0;
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double RAt[4] = {(double) A[0], (double) A[3], (double) A[1], (double) A[4]};
    double StA[2] = {-(double) A[5], (double) A[2]};
    M[0] = RAt[0]; M[1] = RAt[1]; M[3] = RAt[2]; M[4] = RAt[3];
    M[2] = RAt[0] * StA[0] + RAt[1] * StA[1];
    M[5] = RAt[2] * StA[0] + RAt[3] * StA[1];
    M[8] = 1.0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s = s + Jj[r * 3 + k] * M[k * 3 + c];
        Ji[r * 3 + c] = -s;
      }
  }
}

template <int D>
__device__ void atwb(const double* A, const double* W, const double* B, double* C) {
  double WB[D * D];
  for (int r = 0; r < D; ++r)
    for (int c = 0; c < D; ++c) {
      double s = 0.0;
      for (int k = 0; k < D; ++k) s = s + W[r * D + k] * B[k * D + c];
      WB[r * D + c] = s;
    }
  for (int r = 0; r < D; ++r)
    for (int c = 0; c < D; ++c) {
      double s = 0.0;
      for (int k = 0; k < D; ++k) s = s + A[k * D + r] * WB[k * D + c];
      C[r * D + c] = s;
    }
}

template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_edges(int E, int T, const float* __restrict__ poses,
                                                         const int2* __restrict__ ij, const float* __restrict__ Z,
                                                         const double* __restrict__ omega,
                                                         const uint8_t* __restrict__ enabled, double* __restrict__ Ho,
                                                         EdgeContrib<D>* __restrict__ contrib) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  EdgeContrib<D>& C = contrib[e];
  if (!enabled[e]) {  // disabled factors are skipped (LoopClosure_ is created disabled, loop_closure.h:71)
    C.chi = 0.0;
    return;
  }
  const int2 v = ij[e];
  double err[D], Ji[D * D], Jj[D * D];
  edge_linearize<D>(poses + (size_t) v.x * T, poses + (size_t) v.y * T, Z + (size_t) e * T, err, Ji, Jj);
  const double* W = omega + (size_t) e * D * D;
  double We[D];
  for (int a = 0; a < D; ++a) {
    double s = 0.0;
    for (int k = 0; k < D; ++k) s = s + W[a * D + k] * err[k];
    We[a] = s;
  }
  double chi = 0.0;
  for (int a = 0; a < D; ++a) chi = chi + err[a] * We[a];
  C.chi = chi;
  atwb<D>(Ji, W, Ji, C.Cii);
  atwb<D>(Jj, W, Jj, C.Cjj);
  atwb<D>(Ji, W, Jj, Ho + (size_t) e * D * D);
  for (int r = 0; r < D; ++r) {
    double s = 0.0, t = 0.0;
    for (int k = 0; k < D; ++k) {
      s = s + Ji[k * D + r] * We[k];
      t = t + Jj[k * D + r] * We[k];
    }
    C.bi[r] = s;
    C.bj[r] = t;
  }
}

template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_vertices(int V, const uint8_t* __restrict__ fixed,
                                                            const int* __restrict__ inc_start,
                                                            const int* __restrict__ inc_edge,
                                                            const uint8_t* __restrict__ enabled,
                                                            const EdgeContrib<D>* __restrict__ contrib, double damping,
                                                            double* __restrict__ Hd, double* __restrict__ b,
                                                            double* __restrict__ Minv, PgScalars* __restrict__ sc) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  double H[D * D], bv[D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) H[k] = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) bv[k] = 0.0;
  if (fixed[v]) {  // VariableBase::Fixed (multi_graph_slam_impl.cpp:86): identity row, zero right-hand side
#pragma unroll
    for (int a = 0; a < D; ++a) H[a * D + a] = 1.0;
  } else {
    for (int q = inc_start[v]; q < inc_start[v + 1]; ++q) {
      const int code = inc_edge[q];
      const int e    = code >> 1;
      if (!enabled[e]) continue;
      const EdgeContrib<D>& C = contrib[e];
      const double* blk       = (code & 1) ? C.Cjj : C.Cii;
      const double* bb        = (code & 1) ? C.bj : C.bi;
#pragma unroll
      for (int k = 0; k < D * D; ++k) H[k] = H[k] + blk[k];
#pragma unroll
      for (int k = 0; k < D; ++k) bv[k] = bv[k] + bb[k];
    }
#pragma unroll
    for (int a = 0; a < D; ++a) H[a * D + a] = H[a * D + a] + damping;
  }
  // preconditioner block: inverse through D Cholesky solves (dm::solve solves A x = -rhs)
  double Mi[D * D];
  bool bad = false;
  for (int c = 0; c < D; ++c) {
    double rhs[D], x[D];
#pragma unroll
    for (int r = 0; r < D; ++r) rhs[r] = r == c ? -1.0 : 0.0;
    if (dm::solve<D>(H, rhs, x)) bad = true;
#pragma unroll
    for (int r = 0; r < D; ++r) Mi[r * D + c] = bad ? 0.0 : x[r];
  }
  if (bad) sc->bad = 1;
#pragma unroll
  for (int k = 0; k < D * D; ++k) {
    Hd[(size_t) v * D * D + k]   = H[k];
    Minv[(size_t) v * D * D + k] = Mi[k];
  }
#pragma unroll
  for (int k = 0; k < D; ++k) b[(size_t) v * D + k] = bv[k];
}

// Vertex-major copy of the off-diagonal blocks for the PCG loop: incidence q of a variable gets the D x D block that
// multiplies p[other] (H_ij, or its transpose for the `to` end), so k_pg_spmv streams contiguous memory instead of
// gathering 288-byte blocks by factor index and reading columns with a 48-byte stride.  Written once per
// linearisation, read once per PCG iteration.  other = -1: the term is skipped (disabled factor, fixed neighbour).
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_build_csr(int n_inc, const int2* __restrict__ ij,
                                                             const int* __restrict__ inc_edge,
                                                             const uint8_t* __restrict__ enabled,
                                                             const uint8_t* __restrict__ fixed,
                                                             const double* __restrict__ Ho, double* __restrict__ Hcsr,
                                                             int* __restrict__ inc_other) {
  const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t) n_inc * D * D) return;
  const int q = (int) (idx / (D * D)), k = (int) (idx - (size_t) q * D * D);
  const int code = inc_edge[q];
  const int e    = code >> 1;
  const int2 vv  = ij[e];
  const int other = (code & 1) ? vv.x : vv.y;
  const bool skip = !enabled[e] || fixed[other];
  const int r = k / D, c = k - r * D;
  Hcsr[idx] = skip ? 0.0 : ((code & 1) ? Ho[(size_t) e * D * D + c * D + r] : Ho[(size_t) e * D * D + k]);
  if (k == 0) inc_other[q] = skip ? -1 : other;
}

// ---- deterministic block reduction of one double per thread -> partial[blockIdx.x] ------------------------------
__device__ double block_sum(double v) {
  __shared__ double sh[PG_THREADS / 64];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < PG_THREADS / 64; ++w) t += sh[w];
  return t;
}

// every block sums the same partial array in the same order -> identical scalar in every block
__device__ double sum_partials(const double* __restrict__ partials, int n) {
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += PG_THREADS) v += partials[k];
  return block_sum(v);
}

__global__ __launch_bounds__(PG_THREADS) void k_pg_chi(int E, const uint8_t* __restrict__ enabled, const void* contrib,
                                                       int contrib_stride_doubles, int chi_offset_doubles,
                                                       double* __restrict__ partial_chi, int* __restrict__ partial_n) {
  double c = 0.0;
  int n    = 0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    if (enabled[e]) {
      c += ((const double*) contrib)[(size_t) e * contrib_stride_doubles + chi_offset_doubles];
      n += 1;
    }
  }
  double cs = block_sum(c);
  double ns = block_sum((double) n);
  if (threadIdx.x == 0) {
    partial_chi[blockIdx.x] = cs;
    partial_n[blockIdx.x]   = (int) ns;
  }
}

// n = V*D scalars; thread t handles entry t (vertex t/D, row t%D)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_pcg_init(int n, const double* __restrict__ b,
                                                            const double* __restrict__ Minv, double* __restrict__ x,
                                                            double* __restrict__ r, double* __restrict__ p,
                                                            double* __restrict__ part_rz, double* __restrict__ part_bb,
                                                            PgScalars* __restrict__ sc,
                                                            const double* __restrict__ partial_chi,
                                                            const int* __restrict__ partial_n, int n_chi_partials) {
  double rz = 0.0, bb = 0.0;
  for (int tile = 0; tile < PG_TILES; ++tile) {
    const int t = (blockIdx.x * PG_TILES + tile) * PG_ROWS + threadIdx.x;
    if (threadIdx.x < PG_ROWS && t < n) {
      const int v = t / D, row = t - v * D;
      double z = 0.0;
#pragma unroll
      for (int c = 0; c < D; ++c) z = z + Minv[((size_t) v * D + row) * D + c] * (-b[(size_t) v * D + c]);
      const double ri = -b[t];
      x[t] = 0.0;
      r[t] = ri;
      p[t] = z;
      rz   = rz + ri * z;
      bb   = bb + ri * ri;
    }
  }
  rz = block_sum(rz);
  bb = block_sum(bb);
  if (threadIdx.x == 0) {
    part_rz[blockIdx.x] = rz;
    part_bb[blockIdx.x] = bb;
  }
  if (blockIdx.x == 0) {  // statistics of the linearisation
    double c = 0.0, m = 0.0;
    for (int k = threadIdx.x; k < n_chi_partials; k += PG_THREADS) {
      c += partial_chi[k];
      m += (double) partial_n[k];
    }
    c = block_sum(c);
    m = block_sum(m);
    if (threadIdx.x == 0) {
      sc->chi         = c;
      sc->num_factors = (int) m;
      sc->pcg_iters   = 0;
      sc->done        = 0;
    }
  }
}

// Ap = A p; partial p.Ap.  First launch of an iteration also folds the scalars of the init / previous update.
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_spmv(int n, const uint8_t* __restrict__ fixed,
                                                        const int* __restrict__ inc_start,
                                                        const int* __restrict__ inc_other,
                                                        const double* __restrict__ Hd, const double* __restrict__ Hcsr,
                                                        const double* __restrict__ p, double* __restrict__ Ap,
                                                        double* __restrict__ part_pAp, const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const int t = blockIdx.x * PG_ROWS + threadIdx.x;
  double pap = 0.0;
  if (threadIdx.x < PG_ROWS && t < n) {
    const int v = t / D, row = t - v * D;
    double y = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) y = y + Hd[((size_t) v * D + row) * D + c] * p[(size_t) v * D + c];
    if (!fixed[v]) {
      // the D row-threads of a variable read consecutive rows of consecutive blocks: one contiguous stream per block
      for (int q = inc_start[v]; q < inc_start[v + 1]; ++q) {
        const int other = inc_other[q];
        if (other < 0) continue;
        const double* B = Hcsr + ((size_t) q * D + row) * D;
        double s        = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) s = s + B[c] * p[(size_t) other * D + c];
        y = y + s;
      }
    }
    Ap[t] = y;
    pap   = y * p[t];
  }
  pap = block_sum(pap);
  if (threadIdx.x == 0) part_pAp[blockIdx.x] = pap;
}

// x += alpha p ; r -= alpha Ap ; z = Minv r ; partial rr, rz_new.   alpha = rz / (p.Ap) from the partials.
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_update_xr(int n, int nblocks, const double* __restrict__ Minv,
                                                             const double* __restrict__ p, const double* __restrict__ Ap,
                                                             double* __restrict__ x, double* __restrict__ r,
                                                             double* __restrict__ z, const double* __restrict__ part_rz,
                                                             const double* __restrict__ part_pAp,
                                                             double* __restrict__ part_rr, double* __restrict__ part_rz_new,
                                                             const PgScalars* __restrict__ sc, int nblocks_spmv) {
  if (sc->done || sc->bad) return;
  const double rz  = sum_partials(part_rz, nblocks);
  const double pap = sum_partials(part_pAp, nblocks_spmv);
  const double alpha = pap > 0.0 ? rz / pap : 0.0;
  double rr = 0.0, rzn = 0.0;
  for (int tile = 0; tile < PG_TILES; ++tile) {
    const int t       = (blockIdx.x * PG_TILES + tile) * PG_ROWS + threadIdx.x;
    const bool active = threadIdx.x < PG_ROWS && t < n;
    if (active) {
      x[t] = x[t] + alpha * p[t];
      r[t] = r[t] - alpha * Ap[t];
    }
    __syncthreads();  // all D rows of a variable live in this tile (PG_ROWS is a multiple of D)
    if (active) {
      const int v = t / D, row = t - v * D;
      double zi = 0.0;
#pragma unroll
      for (int c = 0; c < D; ++c) zi = zi + Minv[((size_t) v * D + row) * D + c] * r[(size_t) v * D + c];
      z[t] = zi;
      rr   = rr + r[t] * r[t];
      rzn  = rzn + r[t] * zi;
    }
  }
  rr  = block_sum(rr);
  rzn = block_sum(rzn);
  if (threadIdx.x == 0) {
    part_rr[blockIdx.x]     = rr;
    part_rz_new[blockIdx.x] = rzn;
  }
}

// p = z + beta p ; beta = rz_new / rz ; convergence test |r| <= tol |b| ; block 0 publishes the scalars
__global__ __launch_bounds__(PG_THREADS) void k_pg_update_p(int n, int nblocks, double tol, const double* __restrict__ z,
                                                            double* __restrict__ p, const double* __restrict__ part_rz,
                                                            const double* __restrict__ part_rz_new,
                                                            const double* __restrict__ part_rr,
                                                            const double* __restrict__ part_bb, PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const double rz     = sum_partials(part_rz, nblocks);
  const double rz_new = sum_partials(part_rz_new, nblocks);
  const double rr     = sum_partials(part_rr, nblocks);
  const double bb     = sum_partials(part_bb, nblocks);
  const double beta   = rz != 0.0 ? rz_new / rz : 0.0;
  for (int tile = 0; tile < PG_TILES; ++tile) {
    const int t = (blockIdx.x * PG_TILES + tile) * PG_ROWS + threadIdx.x;
    if (threadIdx.x < PG_ROWS && t < n) p[t] = z[t] + beta * p[t];
  }
  // (the host swaps the rz / rz_new partial buffers for the next iteration: no in-kernel copy, no race)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc->rz = rz_new;
    sc->rr = rr;
    sc->bb = bb;
    sc->pcg_iters += 1;
    if (rr <= tol * tol * bb) sc->done = 1;
  }
}

template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_apply(int V, int T, int variable_kind, const uint8_t* __restrict__ fixed,
                                                         const double* __restrict__ x, float* __restrict__ poses,
                                                         const PgScalars* __restrict__ sc) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V || fixed[v] || sc->bad) return;
  double dx[D];
#pragma unroll
  for (int k = 0; k < D; ++k) dx[k] = x[(size_t) v * D + k];
  float X[12];
  for (int k = 0; k < T; ++k) X[k] = poses[(size_t) v * T + k];
  dm::box_plus(variable_kind, X, dx);
  for (int k = 0; k < T; ++k) poses[(size_t) v * T + k] = X[k];
}

}  // namespace

struct srrg2_posegraph_s {
  int kind = 2, D = 6, T = 12, device = 0;
  hipStream_t stream = nullptr;
  int V = 0, E = 0;
  DevBuf<float> poses, Z;
  DevBuf<uint8_t> fixed, enabled;
  DevBuf<int2> ij;
  DevBuf<double> omega, Hd, Ho, Hcsr, b, Minv, x, r, z, p, Ap, contrib;
  DevBuf<double> part_rz, part_rz_new, part_pAp, part_rr, part_bb, part_chi;
  DevBuf<int> part_n, inc_start, inc_edge, inc_other;
  DevBuf<PgScalars> sc;
  // host mirrors for the incremental interface (incidence lists are rebuilt lazily from these)
  std::vector<int> h_ij;
  std::vector<uint8_t> h_enabled, h_removed;
  bool inc_dirty = false;
};

namespace {

template <int D>
int pg_solve_t(srrg2_posegraph_s* g, const srrg2_posegraph_params* p, srrg2_posegraph_stats* stats, int* n_inout) {
  const int V = g->V, E = g->E, T = g->T;
  const int n = V * D;
  int rc;
  const int nb  = std::max((n + PG_ROWS - 1) / PG_ROWS, 1);          // SpMV blocks (one tile each)
  const int nbt = std::max((nb + PG_TILES - 1) / PG_TILES, 1);        // element-wise kernels: PG_TILES tiles per block
  const int nbv = std::max((V + PG_THREADS - 1) / PG_THREADS, 1);
  const int nbe = std::max((E + PG_THREADS - 1) / PG_THREADS, 1);
  const int nchi = std::min(nbe, 1024);
  if (nb > PG_MAX_PARTIALS * 64) return fail(SRRG2_E_UNSUPPORTED, "posegraph: too many variables");
  if ((rc = g->Hd.reserve((size_t) std::max(V, 1) * D * D))) return rc;
  if ((rc = g->Minv.reserve((size_t) std::max(V, 1) * D * D))) return rc;
  if ((rc = g->Ho.reserve((size_t) std::max(E, 1) * D * D))) return rc;
  if ((rc = g->Hcsr.reserve((size_t) std::max(2 * E, 1) * D * D))) return rc;
  if ((rc = g->inc_other.reserve((size_t) std::max(2 * E, 1)))) return rc;
  if ((rc = g->contrib.reserve((size_t) std::max(E, 1) * (sizeof(EdgeContrib<D>) / sizeof(double))))) return rc;
  for (DevBuf<double>* v : {&g->b, &g->x, &g->r, &g->z, &g->p, &g->Ap})
    if ((rc = v->reserve((size_t) std::max(n, 1)))) return rc;
  for (DevBuf<double>* v : {&g->part_rz, &g->part_rz_new, &g->part_pAp, &g->part_rr, &g->part_bb})
    if ((rc = v->reserve((size_t) nb))) return rc;
  if ((rc = g->part_chi.reserve((size_t) nchi))) return rc;
  if ((rc = g->part_n.reserve((size_t) nchi))) return rc;
  if ((rc = g->sc.reserve(1))) return rc;
  EdgeContrib<D>* contrib = reinterpret_cast<EdgeContrib<D>*>(g->contrib.p);
  const int cstride = (int) (sizeof(EdgeContrib<D>) / sizeof(double));
  const int chi_off = (int) (offsetof(EdgeContrib<D>, chi) / sizeof(double));
  int nstats = 0;
  for (int it = 0; it < p->max_iterations; ++it) {
    HIP_TRY(hipMemsetAsync(g->sc.p, 0, sizeof(PgScalars), g->stream));
    if (E > 0)
      hipLaunchKernelGGL(k_pg_edges<D>, dim3(nbe), dim3(PG_THREADS), 0, g->stream, E, T, g->poses.p, g->ij.p, g->Z.p,
                         g->omega.p, g->enabled.p, g->Ho.p, contrib);
    if (E > 0) {
      const size_t nel = (size_t) 2 * E * D * D;
      hipLaunchKernelGGL(k_pg_build_csr<D>, dim3((unsigned) ((nel + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS), 0,
                         g->stream, 2 * E, g->ij.p, g->inc_edge.p, g->enabled.p, g->fixed.p, g->Ho.p, g->Hcsr.p,
                         g->inc_other.p);
    }
    hipLaunchKernelGGL(k_pg_chi, dim3(nchi), dim3(PG_THREADS), 0, g->stream, E, g->enabled.p, (const void*) contrib,
                       cstride, chi_off, g->part_chi.p, g->part_n.p);
    hipLaunchKernelGGL(k_pg_vertices<D>, dim3(nbv), dim3(PG_THREADS), 0, g->stream, V, g->fixed.p, g->inc_start.p,
                       g->inc_edge.p, g->enabled.p, contrib, (double) p->damping, g->Hd.p, g->b.p, g->Minv.p, g->sc.p);
    hipLaunchKernelGGL(k_pg_pcg_init<D>, dim3(nbt), dim3(PG_THREADS), 0, g->stream, n, g->b.p, g->Minv.p, g->x.p, g->r.p,
                       g->p.p, g->part_rz.p, g->part_bb.p, g->sc.p, g->part_chi.p, g->part_n.p, nchi);
    PgScalars h{};
    int launched = 0;
    while (launched < p->pcg_max_iterations) {
      const int chunk = std::min(25, p->pcg_max_iterations - launched);
      for (int k = 0; k < chunk; ++k) {
        double* rz_cur = ((launched + k) & 1) ? g->part_rz_new.p : g->part_rz.p;
        double* rz_nxt = ((launched + k) & 1) ? g->part_rz.p : g->part_rz_new.p;
        hipLaunchKernelGGL(k_pg_spmv<D>, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, g->fixed.p, g->inc_start.p,
                           g->inc_other.p, g->Hd.p, g->Hcsr.p, g->p.p, g->Ap.p, g->part_pAp.p, g->sc.p);
        hipLaunchKernelGGL(k_pg_update_xr<D>, dim3(nbt), dim3(PG_THREADS), 0, g->stream, n, nbt, g->Minv.p, g->p.p, g->Ap.p,
                           g->x.p, g->r.p, g->z.p, rz_cur, g->part_pAp.p, g->part_rr.p, rz_nxt, g->sc.p, nb);
        hipLaunchKernelGGL(k_pg_update_p, dim3(nbt), dim3(PG_THREADS), 0, g->stream, n, nbt, (double) p->pcg_tolerance,
                           g->z.p, g->p.p, rz_cur, rz_nxt, g->part_rr.p, g->part_bb.p, g->sc.p);
      }
      launched += chunk;
      HIP_TRY(hipMemcpyAsync(&h, g->sc.p, sizeof(h), hipMemcpyDeviceToHost, g->stream));
      HIP_TRY(hipStreamSynchronize(g->stream));
      if (h.done || h.bad) break;
    }
    if (p->pcg_max_iterations == 0) {
      HIP_TRY(hipMemcpyAsync(&h, g->sc.p, sizeof(h), hipMemcpyDeviceToHost, g->stream));
      HIP_TRY(hipStreamSynchronize(g->stream));
    }
    hipLaunchKernelGGL(k_pg_apply<D>, dim3(nbv), dim3(PG_THREADS), 0, g->stream, V, T, g->kind, g->fixed.p, g->x.p,
                       g->poses.p, g->sc.p);
    HIP_TRY(hipGetLastError());
    srrg2_posegraph_stats st{};
    st.iteration      = it;
    st.num_factors    = h.num_factors;
    st.pcg_iterations = h.pcg_iters;
    st.solver_status  = h.bad ? 1 : 0;
    st.chi            = (float) h.chi;
    st.pcg_residual   = h.bb > 0.0 ? (float) std::sqrt(h.rr / h.bb) : 0.f;
    if (stats && n_inout && nstats < *n_inout) stats[nstats] = st;
    ++nstats;
    if (h.bad) break;
  }
  HIP_TRY(hipStreamSynchronize(g->stream));
  if (n_inout) *n_inout = nstats;
  return 0;
}

}  // namespace

extern "C" {

void srrg2_posegraph_default_params(srrg2_posegraph_params* p) {
  if (!p) return;
  p->max_iterations     = 10;
  p->pcg_max_iterations = 200;
  p->pcg_tolerance      = 1e-6f;
  p->damping            = 0.f;
}

}  // extern "C" (reopened below)
namespace {
// grow a device array keeping its first `keep` elements (DevBuf::reserve drops the content)
template <typename T>
int grow_keep(DevBuf<T>& b, size_t keep, size_t want) {
  if (want <= b.cap) return 0;
  DevBuf<T> nb;
  int rc;
  if ((rc = nb.reserve(want + want / 2 + 64))) return rc;
  if (keep > 0) HIP_TRY(hipMemcpy(nb.p, b.p, sizeof(T) * keep, hipMemcpyDeviceToDevice));
  b.release();
  b = nb;
  return 0;
}

// incidence lists in (vertex, edge id) order: fixed summation order of the diagonal blocks
int upload_incidence(srrg2_posegraph_s* g) {
  const int V = g->V, E = g->E;
  int rc;
  if ((rc = g->inc_start.reserve((size_t) V + 1))) return rc;
  if ((rc = g->inc_edge.reserve((size_t) std::max(2 * E, 1)))) return rc;
  std::vector<int> start((size_t) V + 1, 0), inc((size_t) std::max(2 * E, 1), 0);
  const int* ij = g->h_ij.data();
  for (int e = 0; e < E; ++e) {
    start[ij[2 * e] + 1]++;
    start[ij[2 * e + 1] + 1]++;
  }
  for (int v = 0; v < V; ++v) start[v + 1] += start[v];
  {
    std::vector<int> cur(start.begin(), start.end() - 1);
    for (int e = 0; e < E; ++e) {
      inc[cur[ij[2 * e]]++]     = 2 * e;
      inc[cur[ij[2 * e + 1]]++] = 2 * e + 1;
    }
  }
  HIP_TRY(hipMemcpy(g->inc_start.p, start.data(), sizeof(int) * start.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->inc_edge.p, inc.data(), sizeof(int) * inc.size(), hipMemcpyHostToDevice));
  g->inc_dirty = false;
  return 0;
}
}  // namespace
extern "C" {

int srrg2_posegraph_create(int variable_kind, int device, srrg2_posegraph_h* out) {
  if (!out || (variable_kind != SRRG2_SE2_RIGHT && variable_kind != SRRG2_SE3_QUAT_RIGHT))
    return fail(SRRG2_E_INVALID, "posegraph_create: variable kind must be SE2_RIGHT or SE3_QUAT_RIGHT");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(SRRG2_E_NO_DEVICE, "posegraph_create: no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(SRRG2_E_INVALID, "posegraph_create: bad device ordinal");
  srrg2_posegraph_s* g = new srrg2_posegraph_s();
  g->kind   = variable_kind;
  g->D      = variable_kind == SRRG2_SE2_RIGHT ? 3 : 6;
  g->T      = variable_kind == SRRG2_SE2_RIGHT ? 9 : 12;
  g->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
    delete g;
    return fail(SRRG2_E_HIP, "posegraph_create: cannot create stream");
  }
  *out = g;
  return 0;
}

int srrg2_posegraph_destroy(srrg2_posegraph_h g) {
  if (!g) return 0;
  (void) hipSetDevice(g->device);
  if (g->stream) (void) hipStreamSynchronize(g->stream);
  g->poses.release(); g->Z.release(); g->fixed.release(); g->enabled.release(); g->ij.release(); g->omega.release();
  g->Hd.release(); g->Ho.release(); g->Hcsr.release(); g->inc_other.release(); g->b.release(); g->Minv.release(); g->x.release(); g->r.release(); g->z.release();
  g->p.release(); g->Ap.release(); g->contrib.release(); g->part_rz.release(); g->part_rz_new.release();
  g->part_pAp.release(); g->part_rr.release(); g->part_bb.release(); g->part_chi.release(); g->part_n.release();
  g->inc_start.release(); g->inc_edge.release(); g->sc.release();
  if (g->stream) (void) hipStreamDestroy(g->stream);
  delete g;
  return 0;
}

int srrg2_posegraph_set(srrg2_posegraph_h g, int V, const float* poses, const uint8_t* fixed_mask, int E,
                        const int32_t* ij, const float* Z, const float* omega, const uint8_t* enabled) {
  if (!g || V < 0 || E < 0 || (V > 0 && !poses) || (E > 0 && (!ij || !Z)))
    return fail(SRRG2_E_INVALID, "posegraph_set: bad arguments");
  for (int e = 0; e < E; ++e)
    if (ij[2 * e] < 0 || ij[2 * e] >= V || ij[2 * e + 1] < 0 || ij[2 * e + 1] >= V || ij[2 * e] == ij[2 * e + 1])
      return fail(SRRG2_E_INVALID, "posegraph_set: bad edge endpoints");
  HIP_TRY(hipSetDevice(g->device));
  const int D = g->D, T = g->T;
  int rc;
  if ((rc = g->poses.reserve((size_t) std::max(V, 1) * T))) return rc;
  if ((rc = g->fixed.reserve((size_t) std::max(V, 1)))) return rc;
  if ((rc = g->ij.reserve((size_t) std::max(E, 1)))) return rc;
  if ((rc = g->Z.reserve((size_t) std::max(E, 1) * T))) return rc;
  if ((rc = g->omega.reserve((size_t) std::max(E, 1) * D * D))) return rc;
  if ((rc = g->enabled.reserve((size_t) std::max(E, 1)))) return rc;
  if ((rc = g->inc_start.reserve((size_t) V + 1))) return rc;
  if ((rc = g->inc_edge.reserve((size_t) std::max(2 * E, 1)))) return rc;
  std::vector<uint8_t> fx((size_t) std::max(V, 1), 0), en((size_t) std::max(E, 1), 1);
  if (fixed_mask) {
    for (int v = 0; v < V; ++v) fx[v] = fixed_mask[v] ? 1 : 0;
  } else if (V > 0) {
    fx[0] = 1;  // multi_graph_slam_impl.cpp:86
  }
  if (enabled)
    for (int e = 0; e < E; ++e) en[e] = enabled[e] ? 1 : 0;
  std::vector<double> om((size_t) std::max(E, 1) * D * D, 0.0);
  for (int e = 0; e < E; ++e)
    for (int a = 0; a < D; ++a)
      for (int b = 0; b < D; ++b)
        om[((size_t) e * D + a) * D + b] = omega ? (double) omega[((size_t) e * D + a) * D + b] : (a == b ? 1.0 : 0.0);
  HIP_TRY(hipMemcpy(g->poses.p, poses, sizeof(float) * (size_t) V * T, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->fixed.p, fx.data(), (size_t) std::max(V, 1), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->ij.p, ij, sizeof(int32_t) * 2 * (size_t) E, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->Z.p, Z, sizeof(float) * (size_t) E * T, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->omega.p, om.data(), sizeof(double) * om.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->enabled.p, en.data(), (size_t) std::max(E, 1), hipMemcpyHostToDevice));
  g->V = V;
  g->E = E;
  g->h_ij.assign(ij, ij + 2 * (size_t) E);
  g->h_enabled.assign(en.begin(), en.begin() + E);
  g->h_removed.assign((size_t) E, 0);
  return upload_incidence(g);
}

/* ---- incremental interface: the pose-graph lifecycle of MultiGraphSLAM_ (S/system/multi_graph_slam_impl.cpp:52-90
 * makeNewMap, :227-297 loopValidate, :300-317 optimize).  The arrays stay on the device between solves; appends grow
 * them in place, the incidence lists are rebuilt at the next solve. */
int srrg2_posegraph_add_variable(srrg2_posegraph_h g, const float* pose, int fixed, int* id_out) {
  if (!g || !pose) return fail(SRRG2_E_INVALID, "posegraph_add_variable: bad arguments");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  const int V = g->V, T = g->T;
  int rc;
  if ((rc = grow_keep(g->poses, (size_t) V * T, (size_t) (V + 1) * T))) return rc;
  if ((rc = grow_keep(g->fixed, (size_t) V, (size_t) V + 1))) return rc;
  const uint8_t fx = fixed ? 1 : 0;
  HIP_TRY(hipMemcpy(g->poses.p + (size_t) V * T, pose, sizeof(float) * (size_t) T, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->fixed.p + V, &fx, 1, hipMemcpyHostToDevice));
  if (id_out) *id_out = V;  // graph ids are indices
  g->V         = V + 1;
  g->inc_dirty = true;
  return 0;
}

int srrg2_posegraph_add_factor(srrg2_posegraph_h g, int i, int j, const float* Z, const float* information, int enabled,
                               int* id_out) {
  if (!g || !Z) return fail(SRRG2_E_INVALID, "posegraph_add_factor: bad arguments");
  if (i < 0 || i >= g->V || j < 0 || j >= g->V || i == j) return fail(SRRG2_E_INVALID, "posegraph_add_factor: bad endpoints");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  const int E = g->E, T = g->T, D = g->D;
  int rc;
  if ((rc = grow_keep(g->ij, (size_t) E, (size_t) E + 1))) return rc;
  if ((rc = grow_keep(g->Z, (size_t) E * T, (size_t) (E + 1) * T))) return rc;
  if ((rc = grow_keep(g->omega, (size_t) E * D * D, (size_t) (E + 1) * D * D))) return rc;
  if ((rc = grow_keep(g->enabled, (size_t) E, (size_t) E + 1))) return rc;
  double om[36];
  for (int a = 0; a < D; ++a)
    for (int b = 0; b < D; ++b) om[a * D + b] = information ? (double) information[a * D + b] : (a == b ? 1.0 : 0.0);
  const int2 e2    = make_int2(i, j);
  const uint8_t en = enabled ? 1 : 0;
  HIP_TRY(hipMemcpy(g->ij.p + E, &e2, sizeof(int2), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->Z.p + (size_t) E * T, Z, sizeof(float) * (size_t) T, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->omega.p + (size_t) E * D * D, om, sizeof(double) * (size_t) D * D, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->enabled.p + E, &en, 1, hipMemcpyHostToDevice));
  g->h_ij.push_back(i);
  g->h_ij.push_back(j);
  g->h_enabled.push_back(en);
  g->h_removed.push_back(0);
  if (id_out) *id_out = E;
  g->E         = E + 1;
  g->inc_dirty = true;
  return 0;
}

/* FactorBase::setEnabled (closures are added disabled and promoted by the validator, :238-241, :283-286) */
int srrg2_posegraph_set_factor_enabled(srrg2_posegraph_h g, int factor_id, int enabled) {
  if (!g || factor_id < 0 || factor_id >= g->E) return fail(SRRG2_E_INVALID, "posegraph_set_factor_enabled: bad factor id");
  if (g->h_removed[(size_t) factor_id]) return fail(SRRG2_E_STATE, "posegraph_set_factor_enabled: factor was removed");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  const uint8_t en = enabled ? 1 : 0;
  g->h_enabled[(size_t) factor_id] = en;
  HIP_TRY(hipMemcpy(g->enabled.p + factor_id, &en, 1, hipMemcpyHostToDevice));
  return 0;
}

/* FactorGraph::removeFactor (rejected closures, :279-281): the ids of the other factors do not change */
int srrg2_posegraph_remove_factor(srrg2_posegraph_h g, int factor_id) {
  if (!g || factor_id < 0 || factor_id >= g->E) return fail(SRRG2_E_INVALID, "posegraph_remove_factor: bad factor id");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  const uint8_t en = 0;
  g->h_enabled[(size_t) factor_id] = 0;
  g->h_removed[(size_t) factor_id] = 1;
  HIP_TRY(hipMemcpy(g->enabled.p + factor_id, &en, 1, hipMemcpyHostToDevice));
  return 0;
}

int srrg2_posegraph_size(srrg2_posegraph_h g, int* num_variables, int* num_factors, int* num_enabled_factors) {
  if (!g) return fail(SRRG2_E_INVALID, "posegraph_size: null handle");
  int nf = 0, ne = 0;
  for (int e = 0; e < g->E; ++e) {
    nf += g->h_removed[(size_t) e] ? 0 : 1;
    ne += g->h_enabled[(size_t) e] ? 1 : 0;
  }
  if (num_variables) *num_variables = g->V;
  if (num_factors) *num_factors = nf;
  if (num_enabled_factors) *num_enabled_factors = ne;
  return 0;
}

int srrg2_posegraph_set_enabled(srrg2_posegraph_h g, const uint8_t* enabled) {
  if (!g || !enabled) return fail(SRRG2_E_INVALID, "posegraph_set_enabled: bad arguments");
  HIP_TRY(hipSetDevice(g->device));
  std::vector<uint8_t> en((size_t) std::max(g->E, 1), 1);
  for (int e = 0; e < g->E; ++e) en[e] = (enabled[e] && !g->h_removed[(size_t) e]) ? 1 : 0;
  g->h_enabled.assign(en.begin(), en.begin() + g->E);
  HIP_TRY(hipMemcpy(g->enabled.p, en.data(), (size_t) std::max(g->E, 1), hipMemcpyHostToDevice));
  return 0;
}

int srrg2_posegraph_solve(srrg2_posegraph_h g, const srrg2_posegraph_params* p, srrg2_posegraph_stats* stats,
                          int* n_inout) {
  if (!g || !p || p->max_iterations < 0 || p->pcg_max_iterations < 0)
    return fail(SRRG2_E_INVALID, "posegraph_solve: bad arguments");
  HIP_TRY(hipSetDevice(g->device));
  if (g->V == 0) {
    if (n_inout) *n_inout = 0;
    return 0;
  }
  if (g->inc_dirty) {
    int rc = upload_incidence(g);
    if (rc) return rc;
  }
  return g->D == 6 ? pg_solve_t<6>(g, p, stats, n_inout) : pg_solve_t<3>(g, p, stats, n_inout);
}

int srrg2_posegraph_get_poses(srrg2_posegraph_h g, float* out) {
  if (!g || !out) return fail(SRRG2_E_INVALID, "posegraph_get_poses: bad arguments");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  HIP_TRY(hipMemcpy(out, g->poses.p, sizeof(float) * (size_t) g->V * g->T, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
