// posegraph.hip -- block-sparse Gauss-Newton + multigrid-preconditioned CG for the pose graph.
//
// Replaces global_solver->compute() as called from MultiGraphSLAM_::optimize()
// (S/system/multi_graph_slam_impl.cpp:300-317) on the graph built at :52-90 / :241-293: only pose variables
// (LocalMap2D/3D, S/mapping/local_map.h:64,75) and binary pose-pose factors (S/registration/loop_closure.h:110-111),
// so there is nothing to Schur-eliminate: block-sparse H build + preconditioned CG.
//
//   k_pg_edges<D>     one thread per factor: e, Ji, Jj, Omega products -> Ho[e] = Ji^T W Jj and the factor's
//                     contributions to H_ii, H_jj, b_i, b_j, chi (stored per factor: no atomics)
//   k_pg_vertices<D>  one thread per variable: gathers its factors' contributions in incidence order
//                     (deterministic), adds damping, handles Fixed variables, inverts the 6x6 block (level-0 smoother)
//   k_mg_*            the aggregation-multigrid preconditioner (see "Linear solver" below)
//   k_pg_pcg_init / k_pg_spmv / k_pg_update_xr / k_pg_converged / k_pg_dot_rz / k_pg_update_p   the PCG loop; scalars
//                     (alpha, beta, convergence) are recomputed by every block from the per-block partial dot
//                     products, so the loop needs no host round trip and no atomics
//   k_pg_apply<D>     X_v <- X_v [+] dx_v
//
// float64 throughout (the float32 poses are the only float32 state).  Results agree with the CPU oracle to PCG
// tolerance, not bit for bit (dot-product order differs); tests bound the pose difference.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <set>
#include <string>
#include <thread>
#if defined(__linux__)
#include <sched.h>
#endif
#include <vector>

#include <rocprim/rocprim.hpp>

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
  unsigned long long max_dx_bits;  // bit pattern of max |dx| over all variables (non-negative doubles order like their bits)
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

// =====================================================================================================================
// Linear solver: conjugate gradients preconditioned by a smoothed-aggregation multigrid V-cycle.
//
// Block-Jacobi PCG needs ~2700 iterations on BASELINE's C5 graph (50 000 poses on a 112 x 112 x 4 lattice of loop
// closures, one gauge vertex: the accumulated odometry drift is the slowest mode of a grounded lattice Laplacian) and
// was cut at 200 per Gauss-Newton iteration in round 1 without converging.  The hierarchy:
//   * aggregates by greedy pairwise matching (three passes per level: <= 8 poses per aggregate), a pose is matched with
//     its nearest unmatched graph neighbour; built on the host when the graph's structure changes;
//   * tentative interpolation T = the rigid-body motion of the aggregate: dx_i = T_i eta_I with T_i = Ad(X_i^-1 X_I) in
//     the (translation, quaternion-vector) coordinates of the right perturbation -- the near-null space of H (global
//     rigid motions) is represented exactly on every level;
//   * the interpolation is smoothed by one damped block-Jacobi sweep, Ps = (I - 0.66 Dinv H) T.  With T alone the
//     cycle's convergence degrades with the number of levels (C5, plain aggregation: 151-415 CG iterations per
//     Gauss-Newton iteration, profiles/archive/r2k); with Ps: 30-79.  The price is fill: a row of Ps holds the aggregates of a
//     pose and of its neighbours (C5: 4 blocks per row), the coarse operators have 40-120 blocks per row, so rows and
//     columns of the coarse levels are shared by 2-32 adjacent lanes (MgLevel::row_parts / col_parts / prow_parts);
//   * coarse operators by the Galerkin product Ps^T (H Ps) in two sparse products over patterns fixed on the host; every
//     output block is one lane group's fixed-order sum (no atomics: deterministic), float64; recomputed every
//     Gauss-Newton iteration on the device;
//   * V(1,1) cycle with damped block-Jacobi smoothing (omega = 0.7); the coarsest level (<= 32 nodes) is solved by its
//     dense inverse; small sparse levels run inside ONE single-workgroup launch (they are launch floors otherwise).
// The cycle is symmetric and positive definite, so PCG applies.  Off-diagonal blocks are stored once per factor
// (H_ij; the `to` end reads it transposed): (E + V) x 288 bytes per SpMV instead of (2E + V) x 288.  They stay float64:
// the smallest eigenvalue of a 50 000-pose graph with one gauge vertex is ~1e-9 of the largest, float32 blocks
// (relative error 6e-8) make the operator indefinite in exactly the drift modes the solve is about -- measured: with
// float32 blocks PCG needed 151 / 175 / 200 iterations on C5 and the coarsest Cholesky failed in the fourth
// Gauss-Newton iteration (profiles/archive/r2k_bench_c5_float32_blocks.json).
// =====================================================================================================================
#define MG_OMEGA 0.8     // (0.7 until round 6: swept again on the spanning-tree aggregates, profiles/r9/r9h_*)
#define MG_OMEGA_P 0.75  // damping of the interpolation smoother (0.66 = (4/3) / lambda_max until round 6; same sweep)
#define MG_MAX_LEVELS 16
#define MG_FUSE_NODES 170   // levels of at most this many nodes run inside the single-workgroup launch (one row per thread)
#define MG_COARSEST_NODES 32
#define MG_FUSE_BLOCKS 1200   // ... and at most this many off-diagonal blocks
#define MG_DOWN2_THREADS 256
#define MG_FUSE_LAST_NODES 16  // coarsest levels of at most this many nodes are solved inside the last down launch (k_mg_down2_coarsest)

struct MgLevel {  // device view of one level (level 0 = the pose graph without its Fixed variables' couplings)
  int n, ne, nc, nce, np, nq;
  double omega;                 // damping of the block-Jacobi smoother
  int row_parts, col_parts, prow_parts;  // lanes sharing one row of H / one column of Ps / one row of Ps: powers of two <= 32
  const int2* eij;              // [ne] endpoints of the off-diagonal blocks (this level's node ids)
  const int* inc_start;         // [n + 1]
  const int2* inc_adj;          // incidences: {other node, (edge << 1) | side (1: this node is the edge's second endpoint)}
  const int* agg;               // [n] aggregate of the next level, or -1 (Fixed variables)
  const int* rep0;              // [n] graph vertex whose pose represents this node
  // smoothed interpolation Ps (n x nc blocks): rows = this level's nodes (CSR, columns ascending), and by column (CSC)
  const int* prow_start;        // [n + 1]
  const int* pcol;              // [np]
  const int* prow_of;           // [np] row of every entry
  const int* pcsc_start;        // [nc + 1]
  const int* pcsc_ent;          // [np] entries of every column, rows ascending
  // Q = H Ps (n x nc blocks, CSR, columns ascending): the first half of the Galerkin product
  const int* qrow_start;        // [n + 1]
  const int* qcol;              // [nq]
  const int* qrow_of;           // [nq]
  double* Hd;                   // [n][D*D] diagonal blocks
  double* Ho;                   // [ne][D*D] off-diagonal blocks H_ij (i = eij.x, j = eij.y), stored once per edge
  float* P;                     // [n][D*D] tentative interpolation blocks (the aggregate's rigid motion)
  double* Ps;                   // [np][D*D] smoothed interpolation
  double* Q;                    // [nq][D*D]
  double* Dinv;                 // [n][D*D] inverse diagonal blocks (smoother)
  // float32 copies of Hd, Ho, Dinv, read by the V-cycle ONLY (k_mg_to_float, once per Gauss-Newton iteration).  The cycle
  // is a preconditioner: any fixed symmetric positive definite operator will do, and one built from the blocks rounded to
  // float32 -- used consistently in the down and the up sweep, accumulated in float64 -- is one (damped Jacobi stays
  // convergent under a 6e-8 relative perturbation, the coarsest inverse and every Galerkin product stay float64).  CG's
  // own operator (k_pg_spmv) reads the float64 blocks: THERE float32 loses the drift modes.
  // Halves the bytes of the six residual passes of a cycle (C5: 2 x 72 + 2 x 78 MB on levels 0 and 1).
  float* Hdf;
  float* Hof;
  float* Dinvf;
  float* Psf;                   // [np][D*D], [nq][D*D]: float32 copies of Ps and Q = H Ps (two-phase levels, k_mg_down2 / k_mg_up2)
  float* Qf;
  const int* qcsc_start;        // [nc + 1]  Q by column (entries of a column, rows ascending)
  const int* qcsc_ent;          // [nq]
  const int2* pcsc2;            // [np] / [nq]: the columns again as {entry, its row} (one load instead of a dependent pair)
  const int2* qcsc2;
  // product lists of the set-up (round 6): entry q of Q = H Ps is the sum over qp_list[qp_start[q] .. qp_start[q + 1]) of
  // H(y) * Ps[x] with y = -1: the row's diagonal block, else the incidence code (edge << 1) | side; coarse edge k of the Galerkin
  // product is the sum over gp_list[gp_start[k] ..) of Ps[x]^T * Q[y]; qdiag[e] = the entry of Q in the row and column of entry e
  // of Ps (the coarse diagonal blocks).  They are the sorted candidate lists of the pattern build with their origins as payload:
  // k_mg_hp / k_mg_galerkin found every product by a binary search per (entry, incidence), five in six of them in vain.
  // Column-ordered float32 copies of Ps and Q (round 6): block m of Psfc / Qfc is the block of entry pcsc2[m].x / qcsc2[m].x, so a
  // column is ONE contiguous run -- the restriction (level 0) and the down phases read their columns as a stream instead of 144-byte
  // blocks scattered over the row-ordered arrays (55.7 MB per restriction for 29 MB of blocks); pcsc_pos / qcsc_pos = where an entry's
  // block goes (the inverse of the column lists).
  float* Psfc;
  float* Qfc;
  const int* pcsc_pos;
  const int* qcsc_pos;
  const int* qp_start;
  const int2* qp_list;
  const int* gp_start;
  const int2* gp_list;
  const int* qdiag;
  double *x, *r, *res;          // [n][D] work vectors of the cycle
};

// A level and the next coarser one, BY VALUE in the kernel's arguments: pointers that come out of the argument segment are
// global pointers to the compiler (global_load / global_store); a pointer loaded from a record in device memory is a GENERIC
// one, every access through it a flat_load / flat_store -- issued to the LDS path as well and counted in both wait counters
// (round 4: 1 981 flat against 376 global accesses in this file) -- and the record itself a dependent load in front of the
// kernel's first useful one.
struct MgPair {
  MgLevel L, C;
};

// (k_mg_coarse_cycle walks several small levels inside one workgroup and reads their records from device memory)
template <typename T>
__device__ __forceinline__ T* mg_glob(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the builtins exist in the device pass only)
  __builtin_assume(!__builtin_amdgcn_is_shared((const void*) p) & !__builtin_amdgcn_is_private((const void*) p));
#endif
  return p;
}
__device__ __forceinline__ MgLevel mg_level(const MgLevel* __restrict__ levels, int l) {
  MgLevel L = levels[l];
  L.eij = mg_glob(L.eij); L.inc_start = mg_glob(L.inc_start); L.inc_adj = mg_glob(L.inc_adj);
  L.agg = mg_glob(L.agg); L.rep0 = mg_glob(L.rep0); L.prow_start = mg_glob(L.prow_start); L.pcol = mg_glob(L.pcol);
  L.prow_of = mg_glob(L.prow_of); L.pcsc_start = mg_glob(L.pcsc_start); L.pcsc_ent = mg_glob(L.pcsc_ent);
  L.qrow_start = mg_glob(L.qrow_start); L.qcol = mg_glob(L.qcol); L.qrow_of = mg_glob(L.qrow_of);
  L.Hd = mg_glob(L.Hd); L.Ho = mg_glob(L.Ho); L.P = mg_glob(L.P); L.Ps = mg_glob(L.Ps); L.Q = mg_glob(L.Q); L.Dinv = mg_glob(L.Dinv);
  L.Hdf = mg_glob(L.Hdf); L.Hof = mg_glob(L.Hof); L.Dinvf = mg_glob(L.Dinvf); L.Psf = mg_glob(L.Psf); L.Qf = mg_glob(L.Qf);
  L.qcsc_start = mg_glob(L.qcsc_start); L.qcsc_ent = mg_glob(L.qcsc_ent); L.pcsc2 = mg_glob(L.pcsc2); L.qcsc2 = mg_glob(L.qcsc2);
  L.qp_start = mg_glob(L.qp_start); L.qp_list = mg_glob(L.qp_list); L.gp_start = mg_glob(L.gp_start); L.gp_list = mg_glob(L.gp_list);
  L.qdiag = mg_glob(L.qdiag);
  L.Psfc = mg_glob(L.Psfc); L.Qfc = mg_glob(L.Qfc); L.pcsc_pos = mg_glob(L.pcsc_pos); L.qcsc_pos = mg_glob(L.qcsc_pos);
  L.x = mg_glob(L.x); L.r = mg_glob(L.r); L.res = mg_glob(L.res);
  return L;
}

// sum of w over the `parts` adjacent lanes that share an output (fixed butterfly: deterministic; every lane gets the sum)
template <int D>
__device__ __forceinline__ void mg_group_sum(double (&w)[D], int parts) {
  for (int off = parts >> 1; off >= 1; off >>= 1) {
#pragma unroll
    for (int r = 0; r < D; ++r) w[r] = w[r] + __shfl_xor(w[r], off);
  }
}

// position of `key` in the ascending list cols[lo, hi), or -1
__device__ __forceinline__ int mg_find(const int* __restrict__ cols, int lo, int hi, int key) {
  while (hi - lo > 4) {
    const int mid = (lo + hi) >> 1;
    if (cols[mid] <= key) lo = mid; else hi = mid;
  }
  for (int k = lo; k < hi; ++k)
    if (cols[k] == key) return k;
  return -1;
}

// ---- device pieces of the cycle; (tid, nth) = this thread / number of threads sharing the loop --------------------
template <int D>
__device__ __forceinline__ void mg_smooth0(const MgLevel& L, int tid, int nth) {
  for (int t = tid; t < L.n * D; t += nth) {
    const int v = t / D, row = t - v * D;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) s = s + (double) L.Dinvf[((size_t) v * D + row) * D + c] * L.r[(size_t) v * D + c];
    L.x[t] = L.omega * s;
  }
}

// y_t = (H x)_t for row t of this level.  The incidence records carry the neighbour, so the loads of four incidences
// (record, block row, neighbour's x) are independent and issued together: small levels are pure load latency.
// (BT: float64 blocks for CG's operator, the float32 copies inside the V-cycle)
template <int D, typename BT>
__device__ __forceinline__ double mg_row(const MgLevel& L, const BT* __restrict__ Hd, const BT* __restrict__ Ho,
                                         const double* __restrict__ x, int v, int row) {
  double y = 0.0;
#pragma unroll
  for (int c = 0; c < D; ++c) y = y + (double) Hd[((size_t) v * D + row) * D + c] * x[(size_t) v * D + c];
  const int q1 = L.inc_start[v + 1];
  for (int q0 = L.inc_start[v]; q0 < q1; q0 += 4) {
    int2 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = q0 + k < q1 ? L.inc_adj[q0 + k] : make_int2(-1, 0);
    double s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] = 0.0;
      if (a[k].x >= 0) {
        const BT* B      = Ho + (size_t) (a[k].y >> 1) * D * D;
        const double* xo = x + (size_t) a[k].x * D;
        if (a[k].y & 1) {  // this node is j: H_ji = H_ij^T
#pragma unroll
          for (int c = 0; c < D; ++c) s[k] = s[k] + (double) B[c * D + row] * xo[c];
        } else {
#pragma unroll
          for (int c = 0; c < D; ++c) s[k] = s[k] + (double) B[row * D + c] * xo[c];
        }
      }
    }
    y = y + ((s[0] + s[1]) + (s[2] + s[3]));
  }
  return y;
}

// the share of lane `part` (of `parts`) in (H x)_t: the diagonal block (part 0) and every parts-th group of four incidences
template <int D, typename BT>
__device__ __forceinline__ double mg_row_part(const MgLevel& L, const BT* __restrict__ Hd, const BT* __restrict__ Ho,
                                              const double* __restrict__ x, int v, int row, int part, int parts) {
  double y = 0.0;
  if (part == 0) {
#pragma unroll
    for (int c = 0; c < D; ++c) y = y + (double) Hd[((size_t) v * D + row) * D + c] * x[(size_t) v * D + c];
  }
  const int q1 = L.inc_start[v + 1];
  for (int q0 = L.inc_start[v] + 4 * part; q0 < q1; q0 += 4 * parts) {
    int2 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = q0 + k < q1 ? L.inc_adj[q0 + k] : make_int2(-1, 0);
    double s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] = 0.0;
      if (a[k].x >= 0) {
        const BT* B      = Ho + (size_t) (a[k].y >> 1) * D * D;
        const double* xo = x + (size_t) a[k].x * D;
        if (a[k].y & 1) {
#pragma unroll
          for (int c = 0; c < D; ++c) s[k] = s[k] + (double) B[c * D + row] * xo[c];
        } else {
#pragma unroll
          for (int c = 0; c < D; ++c) s[k] = s[k] + (double) B[row * D + c] * xo[c];
        }
      }
    }
    y = y + ((s[0] + s[1]) + (s[2] + s[3]));
  }
  return y;
}

// res = r - H x.  Coarse levels of a smoothed-aggregation hierarchy have long rows (C5: 40-120 blocks) and few of them:
// `row_parts` adjacent lanes share a row and add their shares with a fixed shuffle tree (deterministic).
template <int D>
__device__ __forceinline__ void mg_residual(const MgLevel& L, int tid, int nth) {
  const int parts = L.row_parts;
  if (parts == 1) {
    for (int t = tid; t < L.n * D; t += nth) {
      const int v = t / D, row = t - v * D;
      L.res[t]    = L.r[t] - mg_row<D, float>(L, L.Hdf, L.Hof, L.x, v, row);
    }
    return;
  }
  for (int t = tid; t < L.n * D * parts; t += nth) {  // (nth and the bound are multiples of `parts`: whole groups run)
    const int part = t & (parts - 1), u = t / parts;
    const int v = u / D, row = u - v * D;
    double y = mg_row_part<D, float>(L, L.Hdf, L.Hof, L.x, v, row, part, parts);
    for (int off = parts >> 1; off >= 1; off >>= 1) y = y + __shfl_xor(y, off);
    if (part == 0) L.res[u] = L.r[u] - y;
  }
}

// r_coarse = Ps^T res   (`col_parts` adjacent lanes share a column of Ps: a coarse node interpolates to 30-250 fine ones)
// (BT: the float32 copy of Ps wherever the level has one -- the same copy in the restriction and the prolongation, so the
// cycle stays symmetric; level 0's two passes over Ps were 2 x 58 MB of float64 blocks per cycle on C5)
template <int D, typename BT>
__device__ __forceinline__ void mg_restrict_t(const MgLevel& L, const BT* __restrict__ Ps, double* __restrict__ rc, int tid, int nth) {
  const int parts = L.col_parts;
  for (int t = tid; t < L.nc * D * parts; t += nth) {
    const int part = t & (parts - 1), u = t / parts;
    const int I = u / D, a = u - I * D;
    double s = 0.0;
    for (int m = L.pcsc_start[I] + part; m < L.pcsc_start[I + 1]; m += parts) {
      const int e = L.pcsc_ent[m], i = L.prow_of[e];
      const BT* B = Ps + (size_t) e * D * D;
#pragma unroll
      for (int b = 0; b < D; ++b) s = s + (double) B[b * D + a] * L.res[(size_t) i * D + b];
    }
    for (int off = parts >> 1; off >= 1; off >>= 1) s = s + __shfl_xor(s, off);
    if (part == 0) rc[u] = s;
  }
}
template <int D>
__device__ __forceinline__ void mg_restrict(const MgLevel& L, double* __restrict__ rc, int tid, int nth) {
  if (L.Psf) mg_restrict_t<D, float>(L, L.Psf, rc, tid, nth);
  else mg_restrict_t<D, double>(L, L.Ps, rc, tid, nth);
}

// x += Ps x_coarse   (`prow_parts` adjacent lanes share a row of Ps)
template <int D, typename BT>
__device__ __forceinline__ void mg_prolong_t(const MgLevel& L, const BT* __restrict__ Ps, const double* __restrict__ xc, int tid, int nth) {
  const int parts = L.prow_parts;
  for (int t = tid; t < L.n * D * parts; t += nth) {
    const int part = t & (parts - 1), u = t / parts;
    const int v = u / D, row = u - v * D;
    double s = 0.0;
    for (int e = L.prow_start[v] + part; e < L.prow_start[v + 1]; e += parts) {
      const BT* B      = Ps + (size_t) e * D * D;
      const double* xo = xc + (size_t) L.pcol[e] * D;
#pragma unroll
      for (int a = 0; a < D; ++a) s = s + (double) B[row * D + a] * xo[a];
    }
    for (int off = parts >> 1; off >= 1; off >>= 1) s = s + __shfl_xor(s, off);
    if (part == 0) L.x[u] = L.x[u] + s;
  }
}
template <int D>
__device__ __forceinline__ void mg_prolong(const MgLevel& L, const double* __restrict__ xc, int tid, int nth) {
  if (L.Psf) mg_prolong_t<D, float>(L, L.Psf, xc, tid, nth);
  else mg_prolong_t<D, double>(L, L.Ps, xc, tid, nth);
}

// x += omega Dinv res   (res = r - H x computed by mg_residual beforehand: Jacobi, no race)
template <int D>
__device__ __forceinline__ void mg_update(const MgLevel& L, int tid, int nth) {
  for (int t = tid; t < L.n * D; t += nth) {
    const int v = t / D, row = t - v * D;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) s = s + (double) L.Dinvf[((size_t) v * D + row) * D + c] * L.res[(size_t) v * D + c];
    L.x[t] = L.x[t] + L.omega * s;
  }
}

// x = Cinv r on the coarsest level (dense N x N inverse, N = n * D)
template <int D>
__device__ __forceinline__ void mg_coarsest(const MgLevel& L, const double* __restrict__ Cinv, int tid, int nth) {
  const int N = L.n * D;
  for (int t = tid; t < N; t += nth) {
    double s = 0.0;
    for (int c = 0; c < N; ++c) s = s + Cinv[(size_t) t * N + c] * L.r[c];
    L.x[t] = s;
  }
}

enum { MG_OP_SMOOTH0 = 0, MG_OP_RESIDUAL = 1, MG_OP_RESTRICT = 2, MG_OP_PROLONG = 3, MG_OP_UPDATE = 4 };

// one phase of the cycle on one (large) level
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_op(int op, MgPair LV,
                                                      const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  if (op == MG_OP_SMOOTH0) mg_smooth0<D>(L, tid, nth);
  else if (op == MG_OP_RESIDUAL) mg_residual<D>(L, tid, nth);
  else if (op == MG_OP_RESTRICT) mg_restrict<D>(L, LV.C.r, tid, nth);
  else if (op == MG_OP_PROLONG) mg_prolong<D>(L, LV.C.x, tid, nth);
  else mg_update<D>(L, tid, nth);
}

// ---- two-phase levels -------------------------------------------------------------------------------------------------
// A V(1,1) cycle visits a level with six dependent phases (smooth, residual, restrict | prolong, residual, update): six
// launches of which, below level 0, each is mostly launch floor (profiles/archive/r3k_pg_trace.txt: 4.3 us for a launch with
// nothing to do, 31 launches per CG iteration).  With Q = H Ps -- which the Galerkin product needs anyway -- the same cycle
// takes ONE phase down and ONE up per level:
//   down:  r_c = Ps^T (r - H x1) = Ps^T r - Q^T x1           x1 = omega D^-1 r is local to a node and written by whoever
//          x1_c = omega_c D_c^-1 r_c                          produces r: here for the next level
//   up:    x = x1 + Ps x_c + omega D^-1 (r - H x1 - Q x_c)    (r - H (x1 + Ps x_c) = r - H x1 - Q x_c), written to `res`:
//                                                             x1 of the neighbours is still being read
// Same operator up to rounding; float32 copies of the blocks as in the six-phase levels.  A node owns 8 * parts adjacent
// lanes (rows 0 .. D-1 of 8 slots, `parts` lanes per row): the D row results of a node meet by shuffles inside the wave.
// (one 256-thread workgroup per COARSE node: a column of Q holds hundreds of blocks on the coarse levels -- C5 level 1:
// 342 -- and there are few columns; 32 lanes share a row of the result, the D rows meet through LDS)
// (every lane takes WHOLE blocks of the column -- all D rows of the result from one 144-byte block load and one load of
// the vector's node, two blocks in flight -- instead of 32 lanes per row that each pick six strided words out of every
// block: a dense 788-row column was 25 chains of three dependent loads in a row, 34 us for 11 MB on C5's level 2;
// profiles/archive/r3p_*.  The columns are stored a second time as {entry, row} pairs: one load less on the chain.)
template <int D>
__device__ __forceinline__ void mg_block_tmulsub(const float* __restrict__ B, const double* __restrict__ xv, double sign,
                                                 double (&s)[D]) {
  // s += sign * B^T xv
  float b[D * D];
  if (D == 6) {
    const float4* B4 = reinterpret_cast<const float4*>(B);  // (blocks of 36 floats: 16-byte aligned)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 v = B4[k];
      b[4 * k] = v.x; b[4 * k + 1] = v.y; b[4 * k + 2] = v.z; b[4 * k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < D * D; ++k) b[k] = B[k];
  }
  double x[D];
#pragma unroll
  for (int c = 0; c < D; ++c) x[c] = xv[c];
#pragma unroll
  for (int r = 0; r < D; ++r) {
    const double xs = sign * x[r];
#pragma unroll
    for (int a = 0; a < D; ++a) s[a] = s[a] + (double) b[r * D + a] * xs;
  }
}

// the share of thread `tid` of `nth` in r_c(I) = Ps^T r - Q^T x1
template <int D>
__device__ __forceinline__ void mg_down2_column(const MgLevel& L, int I, int tid, int nth, double (&s)[D]) {
#pragma unroll
  for (int a = 0; a < D; ++a) s[a] = 0.0;
  for (int m = L.pcsc_start[I] + tid; m < L.pcsc_start[I + 1]; m += nth) {
    const int2 er = L.pcsc2[m];
    mg_block_tmulsub<D>(L.Psfc ? L.Psfc + (size_t) m * D * D : L.Psf + (size_t) er.x * D * D, L.r + (size_t) er.y * D, 1.0, s);
  }
  const int qe = L.qcsc_start[I + 1];
  for (int m0 = L.qcsc_start[I] + tid; m0 < qe; m0 += 2 * nth) {
    const int2 e0 = L.qcsc2[m0];
    const bool two = m0 + nth < qe;
    const int2 e1 = two ? L.qcsc2[m0 + nth] : e0;
    double t[D];
#pragma unroll
    for (int a = 0; a < D; ++a) t[a] = 0.0;
    mg_block_tmulsub<D>(L.Qfc ? L.Qfc + (size_t) m0 * D * D : L.Qf + (size_t) e0.x * D * D, L.x + (size_t) e0.y * D, -1.0, s);
    mg_block_tmulsub<D>(L.Qfc ? L.Qfc + (size_t) (two ? m0 + nth : m0) * D * D : L.Qf + (size_t) e1.x * D * D, L.x + (size_t) e1.y * D,
                        two ? -1.0 : 0.0, t);
#pragma unroll
    for (int a = 0; a < D; ++a) s[a] = s[a] + t[a];
  }
}

template <int D>
__global__ __launch_bounds__(MG_DOWN2_THREADS) void k_mg_down2(MgPair LV,
                                                               const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  const MgLevel C = LV.C;
  const int I     = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = MG_DOWN2_THREADS / 64;
  __shared__ double part[NW][D];
  __shared__ double rc[D];
  double s[D];
  mg_down2_column<D>(L, I, threadIdx.x, MG_DOWN2_THREADS, s);
#pragma unroll
  for (int a = 0; a < D; ++a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s[a] = s[a] + __shfl_xor(s[a], off);
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < D; ++a) part[wave][a] = s[a];
  }
  __syncthreads();
  if (threadIdx.x < D) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t = t + part[w][threadIdx.x];
    rc[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const int a = threadIdx.x;
    double x1 = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) x1 = x1 + (double) C.Dinvf[((size_t) I * D + a) * D + c] * rc[c];
    C.r[(size_t) I * D + a] = rc[a];
    C.x[(size_t) I * D + a] = C.omega * x1;
  }
}

// The down phase of the LAST two-phase level and the dense solve of the coarsest level behind it in one launch of one
// workgroup: with a dozen coarsest nodes the two launches (13 workgroups, then one) were 11 + 18 us of mostly launch floor
// and load latency per cycle (profiles/archive/r3p_*).  A wave per coarsest node, the waves meet in LDS, 8 lanes share a row of
// the dense inverse.
template <int D>
__global__ __launch_bounds__(1024) void k_mg_down2_coarsest(MgPair LV, const double* __restrict__ Cinv,
                                                            const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  const MgLevel C = LV.C;
  __shared__ double rc[MG_FUSE_LAST_NODES * D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = C.n * D;
  // (the rows of the inverse this thread will need: in flight while the columns are summed)
  const int row = threadIdx.x >> 3, part = threadIdx.x & 7;
  double ci[(MG_FUSE_LAST_NODES * D + 7) / 8];
#pragma unroll
  for (int k = 0; k < (MG_FUSE_LAST_NODES * D + 7) / 8; ++k) {
    const int c = part + 8 * k;
    ci[k] = (row < N && c < N) ? Cinv[(size_t) row * N + c] : 0.0;
  }
  for (int I = wave; I < C.n; I += 16) {
    double s[D];
    mg_down2_column<D>(L, I, lane, 64, s);
#pragma unroll
    for (int a = 0; a < D; ++a) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s[a] = s[a] + __shfl_xor(s[a], off);
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < D; ++a) rc[I * D + a] = s[a];
    }
  }
  __syncthreads();
  double y = 0.0;
#pragma unroll
  for (int k = 0; k < (MG_FUSE_LAST_NODES * D + 7) / 8; ++k) {
    const int c = part + 8 * k;
    if (c < N) y = y + ci[k] * rc[c];
  }
  y = y + __shfl_xor(y, 4);
  y = y + __shfl_xor(y, 2);
  y = y + __shfl_xor(y, 1);
  if (row < N && part == 0) {
    C.r[row] = rc[row];
    C.x[row] = y;
  }
}

// y += B xv (trans == false) or B^T xv (trans == true), whole block per lane
template <int D>
__device__ __forceinline__ void mg_block_mul(const float* __restrict__ B, const double* __restrict__ xv, bool trans, bool on,
                                             double (&y)[D]) {
  float b[D * D];
  if (D == 6) {
    const float4* B4 = reinterpret_cast<const float4*>(B);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 v = B4[k];
      b[4 * k] = v.x; b[4 * k + 1] = v.y; b[4 * k + 2] = v.z; b[4 * k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < D * D; ++k) b[k] = B[k];
  }
  double x[D];
#pragma unroll
  for (int c = 0; c < D; ++c) x[c] = on ? xv[c] : 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) t = t + (double) (trans ? b[c * D + a] : b[a * D + c]) * x[c];
    y[a] = y[a] + t;
  }
}

// xc: the coarse correction (levels[l + 1].res when that level is a two-phase level too, else its x)
// A node owns NL = 8 * parts adjacent lanes; every lane takes WHOLE blocks of the node's rows of H, Q and Ps (all D rows
// of the result from one 144-byte load, two blocks in flight), the lanes of a node meet by shuffles.
// (lane k of the NL lanes of node v; `on`: the node exists)
template <int D>
__device__ __forceinline__ void mg_up2_node(const MgLevel& L, const double* __restrict__ xc, int v, int k, int NL, bool on) {
  double y[D], p[D];
#pragma unroll
  for (int a = 0; a < D; ++a) y[a] = p[a] = 0.0;
  if (on) {
    if (k == 0) mg_block_mul<D>(L.Hdf + (size_t) v * D * D, L.x + (size_t) v * D, false, true, y);
    const int q1 = L.inc_start[v + 1];
    for (int q0 = L.inc_start[v] + k; q0 < q1; q0 += 2 * NL) {
      const int2 a0 = L.inc_adj[q0];
      const bool two = q0 + NL < q1;
      const int2 a1 = two ? L.inc_adj[q0 + NL] : a0;
      mg_block_mul<D>(L.Hof + (size_t) (a0.y >> 1) * D * D, L.x + (size_t) a0.x * D, (a0.y & 1) != 0, true, y);
      mg_block_mul<D>(L.Hof + (size_t) (a1.y >> 1) * D * D, L.x + (size_t) a1.x * D, (a1.y & 1) != 0, two, y);
    }
    const int qe = L.qrow_start[v + 1];
    for (int e0 = L.qrow_start[v] + k; e0 < qe; e0 += 2 * NL) {
      const bool two = e0 + NL < qe;
      const int e1 = two ? e0 + NL : e0;
      const int c0 = L.qcol[e0], c1 = L.qcol[e1];
      mg_block_mul<D>(L.Qf + (size_t) e0 * D * D, xc + (size_t) c0 * D, false, true, y);
      mg_block_mul<D>(L.Qf + (size_t) e1 * D * D, xc + (size_t) c1 * D, false, two, y);
    }
    for (int e = L.prow_start[v] + k; e < L.prow_start[v + 1]; e += NL)
      mg_block_mul<D>(L.Psf + (size_t) e * D * D, xc + (size_t) L.pcol[e] * D, false, true, p);
  }
  for (int off = NL >> 1; off >= 1; off >>= 1) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
      y[a] = y[a] + __shfl_xor(y[a], off);
      p[a] = p[a] + __shfl_xor(p[a], off);
    }
  }
  if (on && k < D) {
    double rr[D];
#pragma unroll
    for (int c = 0; c < D; ++c) rr[c] = L.r[(size_t) v * D + c] - y[c];
    double u = 0.0, pk = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      u  = u + (double) L.Dinvf[((size_t) v * D + k) * D + c] * rr[c];
      pk = k == c ? p[c] : pk;
    }
    L.res[(size_t) v * D + k] = L.x[(size_t) v * D + k] + pk + L.omega * u;
  }
}

template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_up2(MgPair LV, int parts, int xc_in_res,
                                                       const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  const double* __restrict__ xc = xc_in_res ? LV.C.res : LV.C.x;
  const int t  = blockIdx.x * blockDim.x + threadIdx.x;
  const int NL = 8 * parts;
  const int k = t & (NL - 1), v = t / NL;
  mg_up2_node<D>(L, xc, v, k, NL, v < L.n);
}

// The bottom of the cycle as ONE dense operator (round 6).  The last two-phase level L (C5: 100 nodes) and the dense coarsest
// level behind it (13 nodes) were two launches of dependent round trips -- k_mg_down2_coarsest 19.7 us (column starts -> {entry,
// row} pairs -> blocks and vectors -> LDS -> dense solve) and k_mg_up2 10.1 us -- for < 1 MB of operands.  What they compute is
// linear in the level's r:   x1 = S r (S = omega D^-1),  x_c = Cinv (Ps^T r - Q^T x1) = Cinv G^T r with G = Ps - S Q,
//   res = x1 + Ps x_c + S (r - H x1 - Q x_c) = (2 S - S H S) r + G Cinv G^T r = B r,
// an N x N matrix (N = 600) that the set-up forms once per hierarchy (k_bd_*: four small launches after the coarsest inverse) and the
// cycle applies in one launch with no index and no dependent load (k_mg_bottom_dense: ~5 us).  The same operator up to rounding
// (formed from the float64 blocks, stored as float32 like the cycle's other copies, upper triangle mirrored: exactly symmetric).
// A first attempt fused the two launches as they were, every workgroup of the up phase redoing the down phase for itself: 33.7 us
// against 29.8 (profiles/r9d).
template <int D, bool TRANSPOSE_A, typename TA, typename TB>
__device__ __forceinline__ void mg_block_mac(double (&w)[D * D], const TA* __restrict__ A, const TB* __restrict__ B);  // (below)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_bd_g(MgPair LV, double* __restrict__ G) {
  // G[i, A] = Ps[i, A] - S_i Q[i, A]   (dense N x M, zero where neither has an entry)
  const MgLevel L = LV.L;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L.n * L.nc) return;
  const int i = t / L.nc, A = t - i * L.nc;
  const int M = L.nc * D;
  double g[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) g[k] = 0.0;
  const int q = mg_find(L.qcol, L.qrow_start[i], L.qrow_start[i + 1], A);
  if (q >= 0) mg_block_mac<D, false>(g, L.Dinv + (size_t) i * D * D, L.Q + (size_t) q * D * D);
  const int e = mg_find(L.pcol, L.prow_start[i], L.prow_start[i + 1], A);
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c)
      G[(size_t) (i * D + r) * M + A * D + c] =
        (e >= 0 ? (L.Psf ? (double) L.Psf[(size_t) e * D * D + r * D + c] : L.Ps[(size_t) e * D * D + r * D + c]) : 0.0) - L.omega * g[r * D + c];
}
// W = G Cinv   (N x M)
__global__ __launch_bounds__(PG_THREADS) void k_bd_w(int N, int M, const double* __restrict__ G, const double* __restrict__ Cinv,
                                                      double* __restrict__ W) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * M) return;
  const int row = t / M, c = t - row * M;
  double s = 0.0;
  for (int k = 0; k < M; ++k) s = s + G[(size_t) row * M + k] * Cinv[(size_t) k * M + c];
  W[t] = s;
}
// A = 2 S - S H S into the dense accumulator (zeroed before): one thread per diagonal block / per off-diagonal block of H (written
// to both sides; the blocks of a level below level 0 are unique per node pair)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_bd_a(MgPair LV, double* __restrict__ Bacc) {
  const MgLevel L = LV.L;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L.n + L.ne) return;
  const int N = L.n * D;
  const double w = L.omega;
  int i, j;
  const double* H;
  if (t < L.n) {
    i = j = t;
    H = L.Hd + (size_t) t * D * D;
  } else {
    const int2 ij = L.eij[t - L.n];
    i = ij.x;
    j = ij.y;
    H = L.Ho + (size_t) (t - L.n) * D * D;
  }
  double T[D * D], U[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) T[k] = U[k] = 0.0;
  mg_block_mac<D, false>(T, H, L.Dinv + (size_t) j * D * D);  // H_ij Dinv_j
  mg_block_mac<D, false>(U, L.Dinv + (size_t) i * D * D, T);  // Dinv_i H_ij Dinv_j
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double v = -(w * w) * U[r * D + c];
      if (i == j) v = v + 2.0 * w * L.Dinv[(size_t) i * D * D + r * D + c];
      Bacc[(size_t) (i * D + r) * N + j * D + c] = v;
      if (i != j) Bacc[(size_t) (j * D + c) * N + i * D + r] = v;
    }
}
// B = A + W G^T, upper triangle computed and mirrored, stored as float32
__global__ __launch_bounds__(PG_THREADS) void k_bd_b(int N, int M, const double* __restrict__ W, const double* __restrict__ G,
                                                      const double* __restrict__ Bacc, float* __restrict__ Bf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * N) return;
  const int row = t / N, c = t - row * N;
  if (row > c) return;
  double s = Bacc[(size_t) row * N + c];
  for (int k = 0; k < M; ++k) s = s + W[(size_t) row * M + k] * G[(size_t) c * M + k];
  Bf[(size_t) row * N + c] = (float) s;
  Bf[(size_t) c * N + row] = (float) s;
}
// res = B r on the last two-phase level: 32 lanes per row
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_bottom_dense(MgPair LV, const float* __restrict__ Bf, const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  const int N = L.n * D;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t >> 5, part = t & 31;
  double s = 0.0;
  if (row < N)
    for (int c = part; c < N; c += 32) s = s + (double) Bf[(size_t) row * N + c] * L.r[c];
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) s = s + __shfl_xor(s, off);
  if (row < N && part == 0) L.res[row] = s;
}

// x += Ps x_coarse with the coarse correction taken from the coarse level's `res` (a two-phase level below a six-phase one)
// (xc_in_x: the coarse level ran its six phases -- its result is in its x)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_prolong_res(MgPair LV, int xc_in_x,
                                                               const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  mg_prolong<D>(L, xc_in_x ? LV.C.x : LV.C.res, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// levels lf .. nl-1 (small) + the coarsest level nl in ONE workgroup: down, coarsest solve, up
template <int D>
__global__ __launch_bounds__(1024) void k_mg_coarse_cycle(const MgLevel* __restrict__ levels, int lf, int nl,
                                                          const double* __restrict__ Cinv, int coarsest_dense,
                                                          const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const int tid = threadIdx.x, nth = blockDim.x;
  for (int l = lf; l < nl; ++l) {
    const MgLevel L = mg_level(levels, l);
    mg_smooth0<D>(L, tid, nth);
    __syncthreads();
    mg_residual<D>(L, tid, nth);
    __syncthreads();
    mg_restrict<D>(L, mg_glob(levels[l + 1].r), tid, nth);
    __syncthreads();
  }
  {
    const MgLevel L = mg_level(levels, nl);
    if (coarsest_dense) {
      mg_coarsest<D>(L, Cinv, tid, nth);
    } else {  // (coarsening stalled above the dense limit: smoothing only)
      mg_smooth0<D>(L, tid, nth);
      for (int k = 0; k < 3; ++k) {
        __syncthreads();
        mg_residual<D>(L, tid, nth);
        __syncthreads();
        mg_update<D>(L, tid, nth);
      }
    }
    __syncthreads();
  }
  for (int l = nl - 1; l >= lf; --l) {
    const MgLevel L = mg_level(levels, l);
    mg_prolong<D>(L, mg_glob(levels[l + 1].x), tid, nth);
    __syncthreads();
    mg_residual<D>(L, tid, nth);
    __syncthreads();
    mg_update<D>(L, tid, nth);
    __syncthreads();
  }
}

// ---- numeric set-up of the hierarchy (every Gauss-Newton iteration) -------------------------------------------------
// level 0: the diagonal blocks and the active factors' off-diagonal blocks in the level's compact edge order
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_pack0(int V, int ne, const int* __restrict__ act_edge,
                                                         const double* __restrict__ Hd, const double* __restrict__ Ho,
                                                         double* __restrict__ Hd0, double* __restrict__ Ho0) {
  const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t) V * D * D) Hd0[idx] = Hd[idx];
  if (idx < (size_t) ne * D * D) {
    const int k = (int) (idx / (D * D));
    Ho0[idx]    = Ho[(size_t) act_edge[k] * D * D + (idx - (size_t) k * D * D)];
  }
}

// P_i = Ad(X_i^-1 X_I): the aggregate's rigid motion seen from member i (right perturbations, rotation part = quaternion vector)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_interp(MgPair LV, int T,
                                                          const float* __restrict__ poses) {
  const MgLevel L = LV.L;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L.n) return;
  float* P = L.P + (size_t) i * D * D;
  const int I = L.agg[i];
#pragma unroll
  for (int k = 0; k < D * D; ++k) P[k] = 0.f;
  if (I < 0) return;
  const float* Xi = poses + (size_t) L.rep0[i] * T;
  const float* XI = poses + (size_t) LV.C.rep0[I] * T;  // (the pose of the aggregate's first member)
  float Xi_inv[12], A[12];
  if (D == 6) {
    dm::se3_inverse(Xi, Xi_inv);
    dm::se3_compose(Xi_inv, XI, A);
    const float t[3] = {A[3], A[7], A[11]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        P[r * 6 + c]           = A[r * 4 + c];
        P[(r + 3) * 6 + c + 3] = A[r * 4 + c];
      }
    // 2 [t]x R
    for (int c = 0; c < 3; ++c) {
      const float r0 = A[0 * 4 + c], r1 = A[1 * 4 + c], r2 = A[2 * 4 + c];
      P[0 * 6 + c + 3] = 2.f * (t[1] * r2 - t[2] * r1);
      P[1 * 6 + c + 3] = 2.f * (t[2] * r0 - t[0] * r2);
      P[2 * 6 + c + 3] = 2.f * (t[0] * r1 - t[1] * r0);
    }
  } else {
    dm::se2_inverse(Xi, Xi_inv);
    dm::se2_compose(Xi_inv, XI, A);
    P[0] = A[0]; P[1] = A[1]; P[2] = A[5];
    P[3] = A[3]; P[4] = A[4]; P[5] = -A[2];
    P[8] = 1.f;
  }
}

// w (D x D, row-major) += A B  or  A^T B, as D rank-1 updates (12 operand values live at a time)
template <int D, bool TRANSPOSE_A, typename TA, typename TB>
__device__ __forceinline__ void mg_block_mac(double (&w)[D * D], const TA* __restrict__ A, const TB* __restrict__ B) {
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double col[D], row[D];
#pragma unroll
    for (int r = 0; r < D; ++r) col[r] = (double) (TRANSPOSE_A ? A[a * D + r] : A[r * D + a]);
#pragma unroll
    for (int c = 0; c < D; ++c) row[c] = (double) B[a * D + c];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) w[r * D + c] = w[r * D + c] + col[r] * row[c];
  }
}

// Smoothed aggregation: Ps = (I - omega_p Dinv H) T with T the tentative (rigid-motion) interpolation.  Entry (i, A) of Ps:
//   [agg(i) = A] T_i  -  omega_p Dinv_i  sum_{j in N(i) + i, agg(j) = A} H_ij T_j.
// Piecewise-rigid interpolation alone leaves the V-cycle's convergence dependent on the number of levels (C5: 151-415 CG
// iterations per solve); one Jacobi sweep on the interpolation removes its high-energy part (C5: ~30 iterations).
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_psmooth(MgPair LV, double omega_p) {
  // one thread per (entry of Ps, part): every H block of the row is read once for the whole D x D block of the entry
  const MgLevel L = LV.L;
  const int parts = L.row_parts;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L.np * parts) return;  // (whole groups: the bound is a multiple of `parts`)
  const int part = t & (parts - 1), e = t / parts;
  const int i = L.prow_of[e], A = L.pcol[e];
  double w[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) w[k] = 0.0;
  const bool own = L.agg[i] == A;
  if (own && part == 0) mg_block_mac<D, false>(w, L.Hd + (size_t) i * D * D, L.P + (size_t) i * D * D);
  // (which incidences contribute is found first, as a bit mask: the block products then run once per CONTRIBUTING incidence of
  // the wave's busiest lane -- inside the search loop a lane sat through the product of every incidence any lane needed)
  const int q0 = L.inc_start[i] + part, q1 = L.inc_start[i + 1];
  unsigned long long todo = 0;
  for (int q = q0, b = 0; q < q1; q += parts, ++b) {
    const int2 adj = L.inc_adj[q];
    if (L.agg[adj.x] != A) continue;
    if (b < 64) {
      todo |= 1ull << b;
      continue;
    }
    const float* Tj = L.P + (size_t) adj.x * D * D;
    const double* B = L.Ho + (size_t) (adj.y >> 1) * D * D;
    if (adj.y & 1)
      mg_block_mac<D, true>(w, B, Tj);
    else
      mg_block_mac<D, false>(w, B, Tj);
  }
  while (todo) {
    const int b = __ffsll((long long) todo) - 1;
    todo &= todo - 1;
    const int2 adj  = L.inc_adj[q0 + b * parts];
    const float* Tj = L.P + (size_t) adj.x * D * D;
    const double* B = L.Ho + (size_t) (adj.y >> 1) * D * D;
    if (adj.y & 1)
      mg_block_mac<D, true>(w, B, Tj);
    else
      mg_block_mac<D, false>(w, B, Tj);
  }
  mg_group_sum<D * D>(w, parts);
  if (part != 0) return;
  double o[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) o[k] = 0.0;
  mg_block_mac<D, false>(o, L.Dinv + (size_t) i * D * D, w);
  double* out = L.Ps + (size_t) e * D * D;
#pragma unroll
  for (int k = 0; k < D * D; ++k) {
    o[k]   = (own ? (double) L.P[(size_t) i * D * D + k] : 0.0) - omega_p * o[k];
    out[k] = o[k];
  }
  // (the float32 copy the cycle restricts and prolongs with -- and, round 6, the set-up products multiply with: the coarse operators
  // are then the Galerkin products of exactly the interpolation the cycle uses, and the products fetch 144 instead of 288 bytes of it;
  // 16-byte stores: 36 scalar ones took this kernel from 132 to 223 us on level 0)
  if (L.Psf) {
    const size_t ec = L.Psfc ? (size_t) L.pcsc_pos[e] : 0;  // (and the column-ordered copy)
    if constexpr (D == 6) {
      float4* of = reinterpret_cast<float4*>(L.Psf + (size_t) e * D * D);
      float4* oc = reinterpret_cast<float4*>(L.Psfc + ec * D * D);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float4 v = make_float4((float) o[4 * k], (float) o[4 * k + 1], (float) o[4 * k + 2], (float) o[4 * k + 3]);
        of[k] = v;
        if (L.Psfc) oc[k] = v;
      }
    } else {
#pragma unroll
      for (int k = 0; k < D * D; ++k) {
        L.Psf[(size_t) e * D * D + k]  = (float) o[k];
        if (L.Psfc) L.Psfc[ec * D * D + k] = (float) o[k];
      }
    }
  }
}

// Q = H Ps.  One thread per (entry of Q, part): the look-ups of Ps[j, B] over the incidences j of row i are the
// expensive part (a binary search each), so a thread does them once for the whole D x D block; `row_parts` adjacent
// lanes share the incidences and add their blocks with the fixed butterfly.
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_hp(MgPair LV) {
  const MgLevel L = LV.L;
  const int parts = L.row_parts;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L.nq * parts) return;  // (whole groups: the bound is a multiple of `parts`)
  const int part = t & (parts - 1), q = t / parts;
  const int i = L.qrow_of[q], Bc = L.qcol[q];
  double w[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) w[k] = 0.0;
  if (part == 0) {
    const int e = mg_find(L.pcol, L.prow_start[i], L.prow_start[i + 1], Bc);
    if (e >= 0) mg_block_mac<D, false>(w, L.Hd + (size_t) i * D * D, L.Ps + (size_t) e * D * D);
  }
  for (int k = L.inc_start[i] + part; k < L.inc_start[i + 1]; k += parts) {
    const int2 adj = L.inc_adj[k];
    const int e    = mg_find(L.pcol, L.prow_start[adj.x], L.prow_start[adj.x + 1], Bc);
    if (e < 0) continue;
    const double* Pe = L.Ps + (size_t) e * D * D;
    const double* B  = L.Ho + (size_t) (adj.y >> 1) * D * D;
    if (adj.y & 1)
      mg_block_mac<D, true>(w, B, Pe);
    else
      mg_block_mac<D, false>(w, B, Pe);
  }
  mg_group_sum<D * D>(w, parts);
  if (part != 0) return;
  double* out = L.Q + (size_t) q * D * D;
#pragma unroll
  for (int k = 0; k < D * D; ++k) out[k] = w[k];
}

// Galerkin product, second half: Hc[A, B] = sum_i Ps[i, A]^T Q[i, B] over the rows of column A; the diagonal blocks
// and the blocks of the coarse edges (A < B).  One thread per (coarse block, part); fixed lists, fixed order: deterministic.
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_galerkin(MgPair LV) {
  const MgLevel L = LV.L;
  const MgLevel C = LV.C;
  const int parts = L.col_parts;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (L.nc + L.nce) * parts) return;
  const int part = t & (parts - 1), blk = t / parts;
  int A, B;
  double* out;
  if (blk < L.nc) {
    A = B = blk;
    out = C.Hd + (size_t) blk * D * D;
  } else {
    const int2 ab = C.eij[blk - L.nc];
    A = ab.x;
    B = ab.y;
    out = C.Ho + (size_t) (blk - L.nc) * D * D;
  }
  double acc[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) acc[k] = 0.0;
  for (int m = L.pcsc_start[A] + part; m < L.pcsc_start[A + 1]; m += parts) {
    const int e = L.pcsc_ent[m], i = L.prow_of[e];
    const int q = mg_find(L.qcol, L.qrow_start[i], L.qrow_start[i + 1], B);
    if (q < 0) continue;
    mg_block_mac<D, true>(acc, L.Ps + (size_t) e * D * D, L.Q + (size_t) q * D * D);
  }
  mg_group_sum<D * D>(acc, parts);
  if (part != 0) return;
#pragma unroll
  for (int k = 0; k < D * D; ++k) out[k] = acc[k];
}

// Q = H Ps and the Galerkin product over the product lists (MgLevel::qp_list / gp_list): `parts` adjacent lanes share an
// output block, lane `part` takes every parts-th product of its list (a fixed order) and the lanes add their blocks with the
// fixed butterfly -- deterministic, and the same list order on the host-built and the device-built structure.
template <int D, bool F32PS>
__global__ __launch_bounds__(PG_THREADS) void k_mg_hp_list(MgPair LV, int parts) {
  const MgLevel L = LV.L;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L.nq * parts) return;  // (whole groups: the bound is a multiple of `parts`)
  const int part = t & (parts - 1), q = t / parts;
  const int i = L.qrow_of[q];
  double w[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) w[k] = 0.0;
  const int k1 = L.qp_start[q + 1];
  for (int k = L.qp_start[q] + part; k < k1; k += parts) {
    const int2 pr    = L.qp_list[k];
    const double* Pd = L.Ps + (size_t) pr.x * D * D;
    const float* Pf  = L.Psf + (size_t) pr.x * D * D;
    const double* B  = pr.y < 0 ? L.Hd + (size_t) i * D * D : L.Ho + (size_t) (pr.y >> 1) * D * D;
    if (pr.y >= 0 && (pr.y & 1)) {
      if constexpr (F32PS) mg_block_mac<D, true>(w, B, Pf); else mg_block_mac<D, true>(w, B, Pd);
    } else {
      if constexpr (F32PS) mg_block_mac<D, false>(w, B, Pf); else mg_block_mac<D, false>(w, B, Pd);
    }
  }
  mg_group_sum<D * D>(w, parts);
  if (part != 0) return;
  double* out = L.Q + (size_t) q * D * D;
#pragma unroll
  for (int k = 0; k < D * D; ++k) out[k] = w[k];
}

template <int D, bool F32PS>
__global__ __launch_bounds__(PG_THREADS) void k_mg_galerkin_list(MgPair LV, int parts) {
  const MgLevel L = LV.L;
  const MgLevel C = LV.C;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (L.nc + L.nce) * parts) return;
  const int part = t & (parts - 1), blk = t / parts;
  double acc[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) acc[k] = 0.0;
  double* out;
  if (blk < L.nc) {  // diagonal block A: the entries of column A of Ps against the entries of Q in their rows, column A
    out = C.Hd + (size_t) blk * D * D;
    for (int m = L.pcsc_start[blk] + part; m < L.pcsc_start[blk + 1]; m += parts) {
      const int e = L.pcsc_ent[m];
      if constexpr (F32PS)
        mg_block_mac<D, true>(acc, L.Psf + (size_t) e * D * D, L.Q + (size_t) L.qdiag[e] * D * D);
      else
        mg_block_mac<D, true>(acc, L.Ps + (size_t) e * D * D, L.Q + (size_t) L.qdiag[e] * D * D);
    }
  } else {
    const int ke = blk - L.nc;
    out = C.Ho + (size_t) ke * D * D;
    const int k1 = L.gp_start[ke + 1];
    for (int k = L.gp_start[ke] + part; k < k1; k += parts) {
      const int2 pr = L.gp_list[k];
      if constexpr (F32PS)
        mg_block_mac<D, true>(acc, L.Psf + (size_t) pr.x * D * D, L.Q + (size_t) pr.y * D * D);
      else
        mg_block_mac<D, true>(acc, L.Ps + (size_t) pr.x * D * D, L.Q + (size_t) pr.y * D * D);
    }
  }
  mg_group_sum<D * D>(acc, parts);
  if (part != 0) return;
#pragma unroll
  for (int k = 0; k < D * D; ++k) out[k] = acc[k];
}

// inverse diagonal blocks of level l (the smoother)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_dinv(MgPair LV, PgScalars* __restrict__ sc) {
  const MgLevel L = LV.L;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= L.n) return;
  double H[D * D], Mi[D * D];
#pragma unroll
  for (int k = 0; k < D * D; ++k) H[k] = L.Hd[(size_t) v * D * D + k];
#pragma unroll
  for (int a = 0; a < D; ++a)  // (rows are accumulated independently: make the block exactly symmetric for the factorisation)
#pragma unroll
    for (int c = a + 1; c < D; ++c) H[c * D + a] = H[a * D + c];
  // A coarse block that is not positive definite belongs to an aggregate whose rigid motion is a null vector of H (a
  // component of the graph that no Fixed variable anchors): no smoothing there, the cycle leaves those coordinates alone.
  bool bad = false;
  for (int c = 0; c < D; ++c) {
    double rhs[D], x[D];
#pragma unroll
    for (int r = 0; r < D; ++r) rhs[r] = r == c ? -1.0 : 0.0;
    if (dm::solve<D>(H, rhs, x)) bad = true;
#pragma unroll
    for (int r = 0; r < D; ++r) Mi[r * D + c] = x[r];
  }
  (void) sc;
#pragma unroll
  for (int k = 0; k < D * D; ++k) L.Dinv[(size_t) v * D * D + k] = bad ? 0.0 : Mi[k];
}

// the V-cycle's float32 copies of one level's blocks (see MgLevel)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_to_float(MgPair LV, int with_p, int with_q) {
  const MgLevel L = LV.L;
  const size_t nd = (size_t) L.n * D * D, no = (size_t) L.ne * D * D;
  for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < nd || k < no; k += (size_t) gridDim.x * blockDim.x) {
    if (k < nd) {
      L.Hdf[k]   = (float) L.Hd[k];
      L.Dinvf[k] = (float) L.Dinv[k];
    }
    if (k < no) L.Hof[k] = (float) L.Ho[k];
  }
  // (Psf / Psfc: written by k_mg_psmooth; Qf / Qfc: k_mg_q_to_float below)
  (void) with_p;
  (void) with_q;
}
// the float32 copies of Q = H Ps of a two-phase level: row-ordered (the up phase) and column-ordered (the down phase); one entry per thread
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_q_to_float(MgPair LV) {
  const MgLevel L = LV.L;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= L.nq) return;
  const double* src = L.Q + (size_t) q * D * D;
  const size_t qc   = L.Qfc ? (size_t) L.qcsc_pos[q] : 0;
  if constexpr (D == 6) {
    float4* of = reinterpret_cast<float4*>(L.Qf + (size_t) q * D * D);
    float4* oc = reinterpret_cast<float4*>(L.Qfc + qc * D * D);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 v = make_float4((float) src[4 * k], (float) src[4 * k + 1], (float) src[4 * k + 2], (float) src[4 * k + 3]);
      of[k] = v;
      if (L.Qfc) oc[k] = v;
    }
  } else {
#pragma unroll
    for (int k = 0; k < D * D; ++k) {
      L.Qf[(size_t) q * D * D + k] = (float) src[k];
      if (L.Qfc) L.Qfc[qc * D * D + k] = (float) src[k];
    }
  }
}
__global__ __launch_bounds__(PG_THREADS) void k_st_invert(int m, const int* __restrict__ ent, int* __restrict__ pos) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) pos[ent[t]] = t;
}

// dense inverse of the coarsest operator: assemble, Cholesky in place, then one thread per column solves for the inverse
template <int D>
__global__ __launch_bounds__(1024) void k_mg_coarsest_inverse(MgPair LV, double* __restrict__ Aglobal,
                                                              double* __restrict__ Cinv, PgScalars* __restrict__ sc, int in_lds) {
  // (the factorisation is a chain of N column steps with three barriers each: with the matrix in LDS a step costs
  // ~2 us instead of ~12; the host asks for it when N x N doubles fit)
  extern __shared__ double A_lds[];
  double* __restrict__ A = in_lds ? A_lds : Aglobal;
  const MgLevel L = LV.L;
  const int N = L.n * D, tid = threadIdx.x, nth = blockDim.x;
  for (int k = tid; k < N * N; k += nth) A[k] = 0.0;
  __syncthreads();
  for (int k = tid; k < L.n * D * D; k += nth) {
    const int v = k / (D * D), rc = k - v * D * D, r = rc / D, c = rc - r * D;
    A[(size_t) (v * D + r) * N + v * D + c] = 0.5 * (L.Hd[(size_t) v * D * D + r * D + c] + L.Hd[(size_t) v * D * D + c * D + r]);
  }
  __syncthreads();
  for (int k = tid; k < L.ne * D * D; k += nth) {
    const int e = k / (D * D), rc = k - e * D * D, r = rc / D, c = rc - r * D;
    const int2 vv = L.eij[e];
    const double b = L.Ho[(size_t) e * D * D + rc];
    // (parallel factors between the same two poses share entries when level 0 itself is the coarsest level)
    atomicAdd(&A[(size_t) (vv.x * D + r) * N + vv.y * D + c], b);
    atomicAdd(&A[(size_t) (vv.y * D + c) * N + vv.x * D + r], b);
  }
  __syncthreads();
  // right-looking Cholesky, lower triangle, column by column.  A pivot that has (numerically) vanished belongs to a
  // component of the graph that no Fixed variable anchors (its rigid motion is a null vector of H; the right-hand side
  // has no component there and CG leaves it alone): the coarse correction of that coordinate is switched off by an
  // "infinitely stiff" pivot instead of failing the solve.
  __shared__ double s_thr;
  if (tid == 0) {
    double m = 0.0;
    for (int k = 0; k < N; ++k) m = fmax(m, A[(size_t) k * N + k]);
    s_thr = 1e-11 * m;
  }
  for (int k = 0; k < N; ++k) {
    __syncthreads();
    if (tid == 0) {
      const double d = A[(size_t) k * N + k];
      A[(size_t) k * N + k] = d > s_thr ? sqrt(d) : 1e150;
    }
    __syncthreads();
    const double piv = A[(size_t) k * N + k];
    for (int i = k + 1 + tid; i < N; i += nth) A[(size_t) i * N + k] = A[(size_t) i * N + k] / piv;
    __syncthreads();
    const int m = N - k - 1;
    for (int idx = tid; idx < m * m; idx += nth) {
      const int i = k + 1 + idx / m, j = k + 1 + idx % m;
      if (j <= i) A[(size_t) i * N + j] -= A[(size_t) i * N + k] * A[(size_t) j * N + k];
    }
  }
  __syncthreads();
  if (in_lds == 2) {
    // Y = L^-1 by a right-looking sweep over the rows (all columns at once: no thread walks a dependent chain of N^2
    // terms), in LDS behind the factor; then Cinv = Y^T Y, every entry on its own.  The pivot of a switched-off
    // coordinate is 1e150: its row and column of the inverse come out as ~0, like the column solves below.
    double* __restrict__ Y = A_lds + (size_t) N * N;
    for (int k = tid; k < N * N; k += nth) Y[k] = (k / N == k % N) ? 1.0 : 0.0;
    for (int i = 0; i < N; ++i) {
      __syncthreads();
      const double piv = A[(size_t) i * N + i];
      for (int c = tid; c <= i; c += nth) Y[(size_t) i * N + c] = Y[(size_t) i * N + c] / piv;
      __syncthreads();
      const int rows = N - i - 1, cols = i + 1;
      for (int idx = tid; idx < rows * cols; idx += nth) {
        const int j = i + 1 + idx / cols, c = idx - (idx / cols) * cols;
        Y[(size_t) j * N + c] -= A[(size_t) j * N + i] * Y[(size_t) i * N + c];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < N * N; idx += nth) {
      const int a = idx / N, b = idx - a * N;
      double sum = 0.0;
      for (int k = (a > b ? a : b); k < N; ++k) sum += Y[(size_t) k * N + a] * Y[(size_t) k * N + b];
      Cinv[idx] = sum;
    }
    return;
  }
  // column c of the inverse: L y = e_c, L^T x = y   (Cinv is symmetric: stored row = column)
  for (int c = tid; c < N; c += nth) {
    double* x = Cinv + (size_t) c * N;
    for (int i = 0; i < N; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = (c < i ? c : i); k < i; ++k) s -= A[(size_t) i * N + k] * x[k];
      x[i] = i < c ? 0.0 : s / A[(size_t) i * N + i];
    }
    for (int i = N - 1; i >= 0; --i) {
      double s = x[i];
      for (int k = i + 1; k < N; ++k) s -= A[(size_t) k * N + i] * x[k];
      x[i] = s / A[(size_t) i * N + i];
    }
  }
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

// ---- PCG ------------------------------------------------------------------------------------------------------------
// x = 0, r = -b -> level 0's r (the cycle's input); statistics of the linearisation
__global__ __launch_bounds__(PG_THREADS) void k_pg_pcg_init(int n, const double* __restrict__ b, double* __restrict__ x,
                                                            double* __restrict__ r, double* __restrict__ part_bb,
                                                            PgScalars* __restrict__ sc, const double* __restrict__ partial_chi,
                                                            const int* __restrict__ partial_n, int n_chi_partials) {
  double bb = 0.0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const double ri = -b[t];
    x[t] = 0.0;
    r[t] = ri;
    bb   = bb + ri * ri;
  }
  bb = block_sum(bb);
  if (threadIdx.x == 0) part_bb[blockIdx.x] = bb;
  if (blockIdx.x == 0) {
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

// after the cycle: z = level 0's x.  first: p = z, rz = r.z ; later: beta = rz_new / rz, p = z + beta p.
// (two kernels: the dot product needs a grid-wide sum before p can be updated)
__global__ __launch_bounds__(PG_THREADS) void k_pg_dot_rz(int n, const double* __restrict__ r, const double* __restrict__ z,
                                                          double* __restrict__ part_rz_new, const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  double s = 0.0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) s = s + r[t] * z[t];
  s = block_sum(s);
  if (threadIdx.x == 0) part_rz_new[blockIdx.x] = s;
}

__global__ __launch_bounds__(PG_THREADS) void k_pg_update_p(int n, int nblocks, int first, const double* __restrict__ z,
                                                            double* __restrict__ p, const double* __restrict__ part_rz,
                                                            const double* __restrict__ part_rz_new, PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const double rz_new = sum_partials(part_rz_new, nblocks);
  double beta         = 0.0;
  if (!first) {
    const double rz = sum_partials(part_rz, nblocks);
    beta            = rz != 0.0 ? rz_new / rz : 0.0;
  }
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) p[t] = first ? z[t] : z[t] + beta * p[t];
  if (blockIdx.x == 0 && threadIdx.x == 0) sc->rz = rz_new;
}

// Ap = H p on level 0 (float32 blocks, float64 accumulation); partial p.Ap
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_spmv(MgPair LV, const double* __restrict__ p,
                                                        double* __restrict__ Ap, double* __restrict__ part_pAp,
                                                        const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  double pap = 0.0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < L.n * D; t += gridDim.x * blockDim.x) {
    const int v = t / D, row = t - v * D;
    const double y = mg_row<D, double>(L, L.Hd, L.Ho, p, v, row);
    Ap[t] = y;
    pap   = pap + y * p[t];
  }
  pap = block_sum(pap);
  if (threadIdx.x == 0) part_pAp[blockIdx.x] = pap;
}

// x += alpha p ; r -= alpha Ap ; partial r.r ; convergence test |r| <= tol |b| (published by block 0)
__global__ __launch_bounds__(PG_THREADS) void k_pg_update_xr(int n, int nblocks, double tol, const double* __restrict__ p,
                                                             const double* __restrict__ Ap, double* __restrict__ x,
                                                             double* __restrict__ r, const double* __restrict__ part_rz,
                                                             const double* __restrict__ part_pAp, double* __restrict__ part_rr,
                                                             const double* __restrict__ part_bb, PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const double rz  = sum_partials(part_rz, nblocks);
  const double pap = sum_partials(part_pAp, nblocks);
  const double alpha = pap > 0.0 ? rz / pap : 0.0;
  double rr = 0.0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    x[t] = x[t] + alpha * p[t];
    const double ri = r[t] - alpha * Ap[t];
    r[t] = ri;
    rr   = rr + ri * ri;
  }
  rr = block_sum(rr);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = rr;
}

// block 0 of a tiny launch: sums r.r, publishes the iteration's scalars, raises `done`
__global__ __launch_bounds__(PG_THREADS) void k_pg_converged(int nblocks, double tol, const double* __restrict__ part_rr,
                                                             const double* __restrict__ part_bb, PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const double rr = sum_partials(part_rr, nblocks);
  const double bb = sum_partials(part_bb, nblocks);
  if (threadIdx.x == 0) {
    sc->rr = rr;
    sc->bb = bb;
    sc->pcg_iters += 1;
    if (rr <= tol * tol * bb) sc->done = 1;
  }
}

// ---- appended leaves eliminated exactly (round 6) -----------------------------------------------------------------------
// MultiGraphSLAM_::makeNewMap appends ONE variable and ONE factor per new local map (multi_graph_slam_impl.cpp:52-90).  Such a
// variable is a LEAF of the graph: eliminating it from H dx = -b (its Schur complement) leaves the system of the graph without
// it -- so the multigrid hierarchy built before the append, structure AND coarse space, stays exactly what that system needs
// (round 4 patched the new pose into / beside the aggregates: 1.5 - 3 x the CG iterations; round 5 rebuilt the structure:
// 12 - 15 ms per append).  The variables appended since the hierarchy was built (the `tail`: indices >= V0, each with exactly
// one factor to a variable of lower index, its parent) are eliminated children first,
//   K_v = S_v^-1 H_vp,  c_v = S_v^-1 b_v,  H_pp -= H_pv K_v,  b_p -= H_pv c_v     (S_v = H_vv with its own children's terms)
// CG runs on the first V0 variables, then dx_v = -(c_v + K_v dx_p) parents first.  A handful of sequential 6 x 6 operations:
// one thread.  Anything else that changed (a factor between two old variables, a flag, a long tail) rebuilds as before.
template <int D>
__global__ void k_pg_tail_down(int ntail, int V0, const int* __restrict__ parent, const int* __restrict__ ecode,
                               const uint8_t* __restrict__ fixed, double* __restrict__ Hd, double* __restrict__ b,
                               double* __restrict__ Minv, const double* __restrict__ Ho, double* __restrict__ K,
                               double* __restrict__ cv, PgScalars* __restrict__ sc) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int t = ntail - 1; t >= 0; --t) {
    const int v = V0 + t, p = parent[t], code = ecode[t];
    const double* Hb = Ho + (size_t) (code >> 1) * D * D;  // block (i, j) of the factor; the child is its j end when code & 1
    double S[D * D], Hvp[D * D];
    for (int k = 0; k < D * D; ++k) S[k] = Hd[(size_t) v * D * D + k];
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) Hvp[r * D + c] = (code & 1) ? Hb[c * D + r] : Hb[r * D + c];
    double Kv[D * D], cvv[D];
    bool bad = false;
    for (int col = 0; col <= D; ++col) {  // S X = [H_vp | b_v]  (dm::solve solves A x = -rhs)
      double rhs[D], x[D];
      for (int r = 0; r < D; ++r) rhs[r] = col < D ? -Hvp[r * D + col] : -b[(size_t) v * D + r];
      if (dm::solve<D>(S, rhs, x)) bad = true;
      for (int r = 0; r < D; ++r) {
        if (col < D) Kv[r * D + col] = bad ? 0.0 : x[r];
        else cvv[r] = bad ? 0.0 : x[r];
      }
    }
    if (bad) sc->bad = 1;
    const bool pfixed = fixed[p] != 0;
    for (int k = 0; k < D * D; ++k) K[(size_t) t * D * D + k] = pfixed ? 0.0 : Kv[k];
    for (int r = 0; r < D; ++r) cv[(size_t) t * D + r] = cvv[r];
    if (pfixed) continue;  // (an identity row: dx_p = 0, nothing to update)
    for (int r = 0; r < D; ++r) {  // H_pv = H_vp^T
      double sb = 0.0;
      for (int k = 0; k < D; ++k) sb = sb + Hvp[k * D + r] * cvv[k];
      b[(size_t) p * D + r] = b[(size_t) p * D + r] - sb;
      for (int c = 0; c < D; ++c) {
        double sh = 0.0;
        for (int k = 0; k < D; ++k) sh = sh + Hvp[k * D + r] * Kv[k * D + c];
        Hd[(size_t) p * D * D + r * D + c] = Hd[(size_t) p * D * D + r * D + c] - sh;
      }
    }
    if (p < V0) {  // the level-0 smoother's block of a parent inside the hierarchy (k_pg_vertices inverted the block before the update)
      double Hp[D * D];
      for (int k = 0; k < D * D; ++k) Hp[k] = Hd[(size_t) p * D * D + k];
      bool badp = false;
      for (int c = 0; c < D; ++c) {
        double rhs[D], x[D];
        for (int r = 0; r < D; ++r) rhs[r] = r == c ? -1.0 : 0.0;
        if (dm::solve<D>(Hp, rhs, x)) badp = true;
        for (int r = 0; r < D; ++r) Minv[(size_t) p * D * D + r * D + c] = badp ? 0.0 : x[r];
      }
      if (badp) sc->bad = 1;
    }
  }
}
template <int D>
__global__ void k_pg_tail_up(int ntail, int V0, const int* __restrict__ parent, const uint8_t* __restrict__ fixed,
                             const double* __restrict__ K, const double* __restrict__ cv, double* __restrict__ x,
                             const PgScalars* __restrict__ sc) {
  if (blockIdx.x != 0 || threadIdx.x != 0 || sc->bad) return;
  for (int t = 0; t < ntail; ++t) {
    const int v = V0 + t, p = parent[t];
    for (int r = 0; r < D; ++r) {
      double s = cv[(size_t) t * D + r];
      if (!fixed[p])
        for (int c = 0; c < D; ++c) s = s + K[(size_t) t * D * D + r * D + c] * x[(size_t) p * D + c];
      x[(size_t) v * D + r] = -s;
    }
  }
}

// ---- fused steps of a CG iteration (round 6) ------------------------------------------------------------------------------
// A CG iteration was 18 launches, of which four did node-local work between two grid-wide dependencies: level 0's first
// smoothing step x1 = omega Dinv r (now in the kernel that produces r), level 1's (in level 0's restriction), the cycle's last
// update x += omega Dinv res (in the r.z kernel), and the convergence test (block 0 of the cycle's first residual pass: the sum
// of update_xr's partials is complete behind the kernel boundary).  14 launches; same operations on the same operands in the same
// order per entry, so the same numbers as the unfused sequence (SRRG2_AMD_PG_FUSED_CG=0 keeps that one).
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_update_xr_smooth(int n, int nblocks, const double* __restrict__ p,
                                                                    const double* __restrict__ Ap, double* __restrict__ x,
                                                                    double* __restrict__ r, const double* __restrict__ part_rz,
                                                                    const double* __restrict__ part_pAp, double* __restrict__ part_rr,
                                                                    MgPair LV, const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  const double rz  = sum_partials(part_rz, nblocks);
  const double pap = sum_partials(part_pAp, nblocks);
  const double alpha = pap > 0.0 ? rz / pap : 0.0;
  __shared__ double sh_r[PG_ROWS];  // (PG_ROWS rows per tile, a multiple of D: a node never straddles tiles)
  double rr = 0.0;
  for (int base = blockIdx.x * PG_ROWS; base < n; base += gridDim.x * PG_ROWS) {
    const int t   = base + (int) threadIdx.x;
    const bool on = threadIdx.x < PG_ROWS && t < n;
    if (on) {
      x[t] = x[t] + alpha * p[t];
      const double ri = r[t] - alpha * Ap[t];
      r[t] = ri;
      rr   = rr + ri * ri;
      sh_r[threadIdx.x] = ri;
    }
    __syncthreads();
    if (on) {  // x1 = omega Dinv r of this row's node (mg_smooth0)
      const int v = t / D, row = t - v * D, l0 = (int) threadIdx.x - row;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < D; ++c) s = s + (double) L.Dinvf[((size_t) v * D + row) * D + c] * sh_r[l0 + c];
      L.x[t] = L.omega * s;
    }
    __syncthreads();
  }
  rr = block_sum(rr);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = rr;
}

// res = r - H x on level 0; `check`: block 0 first sums r.r of the update before it and publishes the iteration's scalars
// (k_pg_converged's work; the blocks that have already read `done` finish a pass nobody reads)
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_residual0(MgPair LV, int check, int nblocks, double tol,
                                                             const double* __restrict__ part_rr, const double* __restrict__ part_bb,
                                                             PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  if (check && blockIdx.x == 0) {
    const double rr = sum_partials(part_rr, nblocks);
    const double bb = sum_partials(part_bb, nblocks);
    if (threadIdx.x == 0) {
      sc->rr = rr;
      sc->bb = bb;
      sc->pcg_iters += 1;
      if (rr <= tol * tol * bb) sc->done = 1;
    }
  }
  mg_residual<D>(LV.L, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// r_c = Ps^T res on level 0 and x1_c = omega_c D_c^-1 r_c for level 1: a coarse node owns NL = 8 * parts adjacent lanes, every
// lane takes whole blocks of the column (as k_mg_down2), the lanes meet by shuffles
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_mg_restrict_smooth(MgPair LV, int parts, const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  const MgLevel C = LV.C;
  const int t  = blockIdx.x * blockDim.x + threadIdx.x;
  const int NL = 8 * parts;
  const int k = t & (NL - 1), I = t / NL;
  const bool on = I < L.nc;
  double s[D];
#pragma unroll
  for (int a = 0; a < D; ++a) s[a] = 0.0;
  if (on) {
    const int me = L.pcsc_start[I + 1];
    for (int m0 = L.pcsc_start[I] + k; m0 < me; m0 += 2 * NL) {
      const int2 e0  = L.pcsc2[m0];
      const bool two = m0 + NL < me;
      const int2 e1  = two ? L.pcsc2[m0 + NL] : e0;
      double u[D];
#pragma unroll
      for (int a = 0; a < D; ++a) u[a] = 0.0;
      mg_block_tmulsub<D>(L.Psfc ? L.Psfc + (size_t) m0 * D * D : L.Psf + (size_t) e0.x * D * D, L.res + (size_t) e0.y * D, 1.0, s);
      mg_block_tmulsub<D>(L.Psfc ? L.Psfc + (size_t) (two ? m0 + NL : m0) * D * D : L.Psf + (size_t) e1.x * D * D, L.res + (size_t) e1.y * D,
                          two ? 1.0 : 0.0, u);
#pragma unroll
      for (int a = 0; a < D; ++a) s[a] = s[a] + u[a];
    }
  }
  for (int off = NL >> 1; off >= 1; off >>= 1) {
#pragma unroll
    for (int a = 0; a < D; ++a) s[a] = s[a] + __shfl_xor(s[a], off);
  }
  if (on && k < D) {
    double x1 = 0.0, sk = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      x1 = x1 + (double) C.Dinvf[((size_t) I * D + k) * D + c] * s[c];
      sk = k == c ? s[c] : sk;
    }
    C.r[(size_t) I * D + k] = sk;
    C.x[(size_t) I * D + k] = C.omega * x1;
  }
}

// the cycle's last step z = x + omega Dinv res on level 0 and the partial r.z
template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_update_dot(int n, MgPair LV, const double* __restrict__ r,
                                                              double* __restrict__ part_rz_new, const PgScalars* __restrict__ sc) {
  if (sc->done || sc->bad) return;
  const MgLevel L = LV.L;
  double s = 0.0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int v = t / D, row = t - v * D;
    double u = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) u = u + (double) L.Dinvf[((size_t) v * D + row) * D + c] * L.res[(size_t) v * D + c];
    const double z = L.x[t] + L.omega * u;
    L.x[t] = z;
    s      = s + r[t] * z;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) part_rz_new[blockIdx.x] = s;
}

template <int D>
__global__ __launch_bounds__(PG_THREADS) void k_pg_apply(int V, int V0, int T, int variable_kind, const uint8_t* __restrict__ fixed,
                                                         const double* __restrict__ x, float* __restrict__ poses,
                                                         PgScalars* __restrict__ sc) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (sc->bad) return;
  double dx[D];
  double m = 0.0;
  const bool on = v < V && !fixed[v];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    dx[k] = on ? x[(size_t) v * D + k] : 0.0;
    if (v < V0) m = fmax(m, fabs(dx[k]));  // (an appended leaf's own step says nothing about the hierarchy's poses)
  }
  // the size of the step, for the host: a hierarchy built at nearly the same poses can be kept (pg_solve_t)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(&sc->max_dx_bits, (unsigned long long) __double_as_longlong(m));
  if (!on) return;
  float X[12];
  for (int k = 0; k < T; ++k) X[k] = poses[(size_t) v * T + k];
  dm::box_plus(variable_kind, X, dx);
  for (int k = 0; k < T; ++k) poses[(size_t) v * T + k] = X[k];
}

}  // namespace

// host side of one level: structure (built when the graph changes) + device buffers
struct MgLevelBufs {
  int n = 0, ne = 0, nc = 0, nce = 0, np = 0, nq = 0, row_parts = 1, col_parts = 1, prow_parts = 1, smoothed = 1;
  DevBuf<int2> eij;
  DevBuf<int> inc_start, agg, rep0, prow_start, pcol, prow_of, pcsc_start, pcsc_ent, qrow_start, qcol, qrow_of;
  DevBuf<int2> inc_adj;
  DevBuf<float> P, Hdf, Hof, Dinvf, Psf, Qf;
  DevBuf<int> qcsc_start, qcsc_ent;
  DevBuf<int2> pcsc2, qcsc2;
  DevBuf<double> Hd, Ho, Ps, Q, Dinv, x, r, res;
  DevBuf<float> Psfc, Qfc;                      // column-ordered copies (MgLevel::Psfc)
  DevBuf<int> pcsc_pos, qcsc_pos;
  DevBuf<unsigned long long> qp_list, gp_list;  // (int2 {x, y} = the low and the high word)
  DevBuf<int> qp_start, gp_start, qdiag;
  long long nqp = 0, ngp = 0;                   // products of Q = H Ps / of the coarse edges
  void release() {
    qp_list.release(); gp_list.release(); qp_start.release(); gp_start.release(); qdiag.release();
    Psfc.release(); Qfc.release(); pcsc_pos.release(); qcsc_pos.release();
    eij.release(); inc_start.release(); inc_adj.release(); agg.release(); rep0.release(); prow_start.release();
    pcol.release(); prow_of.release(); pcsc_start.release(); pcsc_ent.release(); qrow_start.release(); qcol.release();
    qrow_of.release(); Hd.release(); Ho.release(); P.release(); Ps.release(); Q.release(); Dinv.release(); x.release();
    r.release(); res.release(); Hdf.release(); Hof.release(); Dinvf.release(); Psf.release(); Qf.release();
    qcsc_start.release(); qcsc_ent.release(); pcsc2.release(); qcsc2.release();
  }
};

struct srrg2_posegraph_s {
  int kind = 2, D = 6, T = 12, device = 0;
  hipStream_t stream = nullptr;
  int V = 0, E = 0;
  DevBuf<float> poses, Z;
  DevBuf<uint8_t> fixed, enabled;
  DevBuf<int2> ij;
  DevBuf<double> omega, Hd, Ho, b, Minv, x, r, p, Ap, contrib;
  DevBuf<double> part_rz, part_rz_new, part_pAp, part_rr, part_bb, part_chi;
  DevBuf<int> part_n, inc_start, inc_edge, act_edge;
  DevBuf<PgScalars> sc;
  // multigrid hierarchy
  std::vector<MgLevelBufs*> levels;      // the current hierarchy: the first levels of the pool
  std::set<int> pg_force_tentative;      // levels whose smoothed interpolation exceeded the fill limit (this build)
  std::vector<MgLevelBufs*> level_pool;  // level objects with their device buffers, kept across rebuilds
  DevBuf<MgLevel> levels_dev;
  std::vector<MgLevel> level_views;      // host copies of the records in levels_dev (kernel arguments: MgPair)
  DevBuf<double> coarse_A, coarse_inv;
  DevBuf<double> bottom_acc, bottom_G, bottom_W;  // the dense bottom operator of the cycle (k_bd_*): A = 2S - SHS, G = Ps - SQ, W = G Cinv
  DevBuf<float> bottom_B;                         // ... and B = A + W G^T, what k_mg_bottom_dense applies
  int coarsest_dense = 1;
  // strategy knobs (srrg2_posegraph_tuning): defaults overridden by the SRRG2_AMD_PG_* environment ONCE, in
  // srrg2_posegraph_create; srrg2_posegraph_set_tuning replaces them
  struct Switches {
    int match_passes = 3;       // SRRG2_AMD_PG_PASSES
    double omega_p = 0.0;       // SRRG2_AMD_PG_OMEGA_P (set to MG_OMEGA_P at create)
    double omega = 0.0;         // SRRG2_AMD_PG_OMEGA   (set to MG_OMEGA at create)
    double lag_below = 0.05;    // SRRG2_AMD_PG_LAG
    bool two_phase = true;      // SRRG2_AMD_PG_TWO_PHASE
    bool use_graph = true;      // SRRG2_AMD_PG_GRAPH
    bool debug = false;         // SRRG2_AMD_PG_DEBUG
    bool keep_structure = true; // the hierarchy's structure survives a set() with the same topology
    bool device_structure = true;  // SRRG2_AMD_PG_DEVICE_STRUCTURE: the sparsity patterns of a level are built on the device
    bool product_lists = true;     // SRRG2_AMD_PG_PRODUCT_LISTS: the set-up products over the lists the pattern build leaves (round 6; 0: the searching kernels)
    int list_lane_products = 4;    // SRRG2_AMD_PG_LIST_LANES: products per lane the list kernels aim at
    bool column_copies = false;    // SRRG2_AMD_PG_COLUMN_COPIES (experiment, off): column-ordered float32 copies of Ps and Q for the down phases --
                                   // k_mg_down2 20.0 -> 17.4 / 10.7 -> 9.9 us, the restriction unchanged, the solve 85.3-87.5 -> 90.0 ms: the second
                                   // copy of Ps is 29 MB more per cycle and two more arrays to write per set-up (profiles/r9/r9s_*)
    bool setup_f32_ps = true;      // SRRG2_AMD_PG_SETUP_F32_PS: the set-up products read the float32 copy of the interpolation (round 6)
    bool l1_six = false;           // SRRG2_AMD_PG_L1_SIX (experiment): level 1 on six phases through H instead of two through Q
    bool tree_positions = true;    // SRRG2_AMD_PG_TREE_POSITIONS: the matching's geometry from a spanning tree of the measurements (round 6)
    bool fused_bottom = true;      // SRRG2_AMD_PG_FUSED_BOTTOM: the bottom of the cycle as one dense operator (k_mg_bottom_dense, round 6)
    bool fused_cg = true;          // SRRG2_AMD_PG_FUSED_CG: 14 launches per CG iteration instead of 18 (round 6; an A/B switch: same numbers)
  } sw;
  // scratch of the device-side pattern build (pg_device_patterns)
  DevBuf<unsigned long long> st_keys_a, st_keys_b;
  DevBuf<int> st_cnt, st_off, st_slot, st_ia, st_ib, st_counts;
  DevBuf<unsigned long long> st_vals;  // payload of the candidate keys (the product lists)
  DevBuf<unsigned> st_rle;             // run lengths of the sorted keys
  DevBuf<unsigned long long> st_total;
  unsigned long long st_offset_limit = 0x7fff0000ull;  // candidate lists beyond this many entries: the host build (32-bit offsets);
                                                       // SRRG2_AMD_PG_OFFSET_LIMIT lowers it (tests of that fallback)
  DevBuf<char> st_temp;
  double st_ms[5] = {0, 0, 0, 0, 0};  // (debug) P sorted / Q counted / Q sorted / columns + coarse edges counted / coarse edges sorted
  srrg2_posegraph_tuning tuning{};
  bool mg_dirty      = true;
  // the graph the hierarchy was built for, and what has been appended since (k_pg_tail_down / k_pg_tail_up)
  int hier_V = 0, hier_E = 0;
  int hier_builds = 0;                  // structure builds of this handle (srrg2_posegraph_structure_info)
  bool tail_pending = false;            // variables / factors appended since: classified at the next solve
  int ntail = 0;                        // > 0: the last hier_V .. V variables are eliminated leaves
  DevBuf<int> tail_parent, tail_ecode;
  DevBuf<double> tail_K, tail_c;
  // host mirrors for the incremental interface (incidence lists are rebuilt lazily from these)
  std::vector<int> h_ij;
  std::vector<float> h_Z;  // the measurements (the spanning-tree geometry of the matching, build_hierarchy)
  std::vector<uint8_t> h_enabled, h_removed, h_fixed;
  bool inc_dirty = false;
};

namespace {

template <typename T>
int upload(DevBuf<T>& b, const std::vector<T>& v) {
  int rc = b.reserve(std::max<size_t>(v.size(), 1));
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpy(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  return 0;
}

// A handful of host threads for the parallel regions of build_hierarchy (a dozen per build).  Started once per build;
// between regions they SPIN on the generation counter -- the regions follow each other within microseconds and the pool
// lives ~15 ms: creating 16 threads per region cost more than most regions' work on the 256-thread host, and waking them
// through a condition variable made a region's duration a lottery (2-8 ms for the same work).  Every worker acknowledges
// every generation, so two regions never overlap.
class HostPool {
 public:
  explicit HostPool(int n) : n_(std::max(n, 1)) {
    for (int t = 1; t < n_; ++t) threads_.emplace_back([this, t] { loop(t); });
  }
  ~HostPool() {
    stop_.store(true, std::memory_order_release);
    word_.store(((unsigned long long) ++gen_ << 32), std::memory_order_release);
    for (std::thread& th : threads_) th.join();
  }
  int size() const { return n_; }
  // CPUs this process may actually run on (affinity mask / cgroup cpuset), not the machine's: under a CPU quota or on an
  // oversubscribed host more spinning workers than usable cores make every region wait for the scheduler
  static int usable_cpus() {
    int n = (int) std::thread::hardware_concurrency();
#if defined(__linux__)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      const int c = CPU_COUNT(&set);
      if (c > 0) n = n > 0 ? std::min(n, c) : c;
    }
#endif
    return std::max(n, 1);
  }
  // fn(t) for t in [0, nt), nt <= size(); the caller runs t = 0 itself and returns when every PARTICIPATING t is done
  // (workers t >= nt only note the generation: a region never waits for a thread that has nothing to do in it)
  template <typename Fn>
  void run(int nt, Fn&& fn) {
    nt = std::min(nt, n_);
    if (nt <= 1) {
      fn(0);
      return;
    }
    job_ = [&fn](int t) { fn(t); };
    pending_.store(nt - 1, std::memory_order_relaxed);
    // generation and participant count travel in ONE word: a worker decides from the snapshot it woke up on, never from a
    // count that a later region has already overwritten
    word_.store(((unsigned long long) ++gen_ << 32) | (unsigned) nt, std::memory_order_release);
    fn(0);
    unsigned spins = 0;
    while (pending_.load(std::memory_order_acquire) != 0) relax(spins);
  }

 private:
  // spin briefly (the regions follow each other within microseconds), then give the core away
  static void relax(unsigned& spins) {
    if (++spins < 4096u) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#elif defined(__aarch64__)
      asm volatile("yield" ::: "memory");
#endif
    } else {
      std::this_thread::yield();
    }
  }
  void loop(int t) {
    unsigned seen = 0;
    for (;;) {
      unsigned long long w;
      unsigned spins = 0;
      while ((unsigned) ((w = word_.load(std::memory_order_acquire)) >> 32) == seen) relax(spins);
      if (stop_.load(std::memory_order_acquire)) return;
      // (a worker that was descheduled across several regions it had no part in catches up here: it can only have missed
      // generations whose count was <= t, or run() would still be waiting for it)
      seen = (unsigned) (w >> 32);
      if (t < (int) (unsigned) w) {
        job_(t);
        pending_.fetch_sub(1, std::memory_order_release);
      }
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::function<void(int)> job_;
  std::atomic<int> pending_{0};
  std::atomic<unsigned long long> word_{0};  // generation << 32 | participants of that generation
  unsigned gen_ = 0;                          // (written by the owning thread only)
  std::atomic<bool> stop_{false};
};

// Rows of a sparse pattern, each the sorted set of distinct columns that `row_fn(row, stamp, out)` pushes (it marks a column
// in `stamp` -- one int per column, initialised to -1 -- with the row's id to see it once).  The rows are independent: the
// pool's threads take contiguous chunks of them (as many threads as `work`, an estimate of the inner-loop steps, pays
// for), the chunks are copied to their places in parallel (the result does not depend on the number of threads).
template <typename RowFn>
void pattern_rows(HostPool& pool, int nrows, int ncols, long long work, std::vector<int>& start, std::vector<int>& cols,
                  std::vector<int>* row_of, RowFn row_fn) {
  const int nthreads = (int) std::max(1LL, std::min({(long long) pool.size(), work / 16384 + 1, (long long) nrows}));
  std::vector<std::vector<int>> chunk_cols((size_t) nthreads), chunk_len((size_t) nthreads);
  auto row_lo = [&](int t) { return (int) ((long long) nrows * t / nthreads); };
  pool.run(nthreads, [&](int t) {
    const int r0 = row_lo(t), r1 = row_lo(t + 1);
    std::vector<int> stamp((size_t) std::max(ncols, 1), -1), out;
    std::vector<int>& cc = chunk_cols[(size_t) t];
    std::vector<int>& ll = chunk_len[(size_t) t];
    ll.reserve((size_t) (r1 - r0));
    cc.reserve((size_t) (work / nthreads / 2 + 16));
    for (int r = r0; r < r1; ++r) {
      out.clear();
      row_fn(r, stamp, out);
      std::sort(out.begin(), out.end());
      cc.insert(cc.end(), out.begin(), out.end());
      ll.push_back((int) out.size());
    }
  });
  std::vector<size_t> offset((size_t) nthreads + 1, 0);
  for (int t = 0; t < nthreads; ++t) offset[(size_t) t + 1] = offset[(size_t) t] + chunk_cols[(size_t) t].size();
  const size_t total = offset[(size_t) nthreads];
  start.assign((size_t) nrows + 1, 0);
  cols.resize(total);
  if (row_of) row_of->resize(total);
  pool.run(nthreads, [&](int t) {
    const std::vector<int>& cc = chunk_cols[(size_t) t];
    if (!cc.empty()) std::memcpy(cols.data() + offset[(size_t) t], cc.data(), cc.size() * sizeof(int));
    size_t at = offset[(size_t) t];
    int r     = row_lo(t);
    for (int len : chunk_len[(size_t) t]) {
      if (row_of) std::fill(row_of->begin() + (long) at, row_of->begin() + (long) (at + (size_t) len), r);
      at += (size_t) len;
      start[(size_t) r + 1] = (int) at;  // (= the end of row r: the global offsets are exclusive prefix sums)
      ++r;
    }
  });
}

// The entries of a row-major pattern listed by column (rows ascending within a column) + the {entry, row} pairs the
// two-phase kernels read.  Thread t owns a contiguous range of columns and picks its entries out of one scan of all of
// them (the scan is cheap; a serial counting sort of C5's 546 000-entry Q was 1.5 ms).
void columns_of(HostPool& pool, int ncols, const std::vector<int>& col_of_entry, const std::vector<int>& row_of_entry,
                std::vector<int>& csc_start, std::vector<int>& csc_ent, std::vector<int2>& csc2) {
  const int ne = (int) col_of_entry.size();
  csc_start.assign((size_t) ncols + 1, 0);
  csc_ent.assign((size_t) std::max(ne, 1), 0);
  csc2.assign((size_t) std::max(ne, 1), make_int2(0, 0));
  for (int e = 0; e < ne; ++e) csc_start[(size_t) col_of_entry[(size_t) e] + 1]++;
  for (int c = 0; c < ncols; ++c) csc_start[(size_t) c + 1] += csc_start[(size_t) c];
  const int nthreads = (int) std::max(1LL, std::min({(long long) pool.size(), (long long) ne / 32768 + 1, (long long) std::max(ncols, 1)}));
  // column ranges of about equal numbers of entries
  std::vector<int> c_lo((size_t) nthreads + 1, ncols);
  c_lo[0] = 0;
  for (int t = 1; t < nthreads; ++t) {
    const int want = (int) ((long long) ne * t / nthreads);
    c_lo[(size_t) t] = (int) (std::lower_bound(csc_start.begin(), csc_start.end(), want) - csc_start.begin());
    c_lo[(size_t) t] = std::min(std::max(c_lo[(size_t) t], c_lo[(size_t) t - 1]), ncols);
  }
  pool.run(nthreads, [&](int t) {
    const int c0 = c_lo[(size_t) t], c1 = c_lo[(size_t) t + 1];
    if (c0 >= c1) return;
    std::vector<int> cur(csc_start.begin() + c0, csc_start.begin() + c1);
    for (int e = 0; e < ne; ++e) {
      const int c = col_of_entry[(size_t) e];
      if (c < c0 || c >= c1) continue;
      const int at = cur[(size_t) (c - c0)]++;
      csc_ent[(size_t) at] = e;  // (rows ascending: the entries are scanned in row order)
      csc2[(size_t) at]    = make_int2(e, row_of_entry[(size_t) e]);
    }
  });
}

// ---- the sparsity patterns of one level, built on the device ------------------------------------------------------------
// Every pattern of the hierarchy is a sorted set of (row, column) pairs: rows of Ps = the aggregates of a node and of its
// neighbours, rows of Q = H Ps = the union of the rows of Ps over a node and its neighbours, the coarse edges = the pairs
// (A, B > A) with B in the row of Q of some row of column A of Ps.  The host built them row by row with stamp arrays (13 ms of
// the 25 ms a structure build cost on C5, on 16 threads); here every pattern is: count the candidates, scan, write them as
// keys row * (nc + 1) + column, ONE radix sort, unique -- a canonical order, so the arrays are those of the host build entry for
// entry (SRRG2_AMD_PG_DEVICE_STRUCTURE=0 keeps the host build; tests/test_gpu_posegraph.py compares the two).  The column
// lists are stable sorts of the entries by column.  The host keeps what is sequential: the greedy matching.
typedef unsigned long long st_key;

__global__ __launch_bounds__(PG_THREADS) void k_st_p_candidates(int n, int nc, int smoothed, const int* __restrict__ agg,
                                                                const int* __restrict__ inc_start, const int2* __restrict__ inc_adj,
                                                                st_key* __restrict__ keys, int* __restrict__ slot_node) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const st_key none = (st_key) n * (st_key) (nc + 1);
  const int q0 = inc_start[v], q1 = inc_start[v + 1];
  const size_t base = (size_t) v + (size_t) q0;  // slot 0: the node itself, then its incidences
  const int a = agg[v];
  keys[base]      = a >= 0 ? (st_key) v * (st_key) (nc + 1) + (st_key) a : none;
  slot_node[base] = v;
  for (int q = q0; q < q1; ++q) {
    const int b = (a >= 0 && smoothed) ? agg[inc_adj[q].x] : -1;
    keys[base + 1 + (size_t) (q - q0)]      = b >= 0 ? (st_key) v * (st_key) (nc + 1) + (st_key) b : none;
    slot_node[base + 1 + (size_t) (q - q0)] = v;
  }
}

// counts[which] = number of keys below `none` among the nsel sorted unique keys (`none`, if present, is the last one)
__global__ void k_st_valid(const st_key* __restrict__ keys, const int* __restrict__ nsel, st_key none, int* __restrict__ counts,
                           int which) {
  const int m   = *nsel;
  counts[which] = (m > 0 && keys[m - 1] >= none) ? m - 1 : m;
}

// rows, columns and row starts of a sorted unique key list
__global__ __launch_bounds__(PG_THREADS) void k_st_decode(int nrows, int nc, const int* __restrict__ counts, int which,
                                                          const st_key* __restrict__ keys, int* __restrict__ start,
                                                          int* __restrict__ col, int* __restrict__ row_of) {
  const int m = counts[which];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) {
    const st_key k = keys[t];
    const int r    = (int) (k / (st_key) (nc + 1));
    if (col) col[t] = (int) (k - (st_key) r * (st_key) (nc + 1));
    if (row_of) row_of[t] = r;
  }
  if (t <= nrows) {  // start[t] = the first key of row t or beyond
    const st_key want = (st_key) t * (st_key) (nc + 1);
    int lo = 0, hi = m;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    start[t] = lo;
  }
}

// slot t = (node v, itself or one of its neighbours j): the row of Ps of j goes into the row of Q of v
__device__ __forceinline__ int st_slot_other(int t, int v, const int* __restrict__ inc_start, const int2* __restrict__ inc_adj) {
  const int s = t - (v + inc_start[v]);
  return s == 0 ? v : inc_adj[inc_start[v] + s - 1].x;
}
__global__ __launch_bounds__(PG_THREADS) void k_st_q_count(int slots, const int* __restrict__ slot_node, const int* __restrict__ agg,
                                                           const int* __restrict__ inc_start, const int2* __restrict__ inc_adj,
                                                           const int* __restrict__ prow_start, int* __restrict__ cnt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= slots) return;
  const int v = slot_node[t];
  const int j = st_slot_other(t, v, inc_start, inc_adj);
  cnt[t]      = agg[v] >= 0 ? prow_start[j + 1] - prow_start[j] : 0;
}
__global__ __launch_bounds__(PG_THREADS) void k_st_q_candidates(int slots, int nc, const int* __restrict__ slot_node,
                                                                const int* __restrict__ inc_start, const int2* __restrict__ inc_adj,
                                                                const int* __restrict__ prow_start, const int* __restrict__ pcol,
                                                                const int* __restrict__ cnt, const int* __restrict__ off,
                                                                st_key* __restrict__ keys, unsigned long long* __restrict__ vals) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= slots || cnt[t] == 0) return;
  const int v = slot_node[t];
  const int s = t - (v + inc_start[v]);
  const int2 adj = s == 0 ? make_int2(v, -1) : inc_adj[inc_start[v] + s - 1];  // (.y: the block of H this slot multiplies by; -1 = the diagonal one)
  const int j = adj.x;
  st_key* out = keys + off[t];
  unsigned long long* vo = vals + off[t];
  for (int e = prow_start[j], k = 0; e < prow_start[j + 1]; ++e, ++k) {
    out[k] = (st_key) v * (st_key) (nc + 1) + (st_key) pcol[e];
    vo[k]  = ((unsigned long long) (unsigned) adj.y << 32) | (unsigned) e;  // int2 {entry of Ps, block of H}
  }
}
// *total += sum of cnt[0, m) in 64 bits (the guards against fill and against 32-bit offsets look at this one)
__global__ __launch_bounds__(PG_THREADS) void k_st_sum64(int m, const int* __restrict__ cnt, unsigned long long* __restrict__ total) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c = t < m ? (unsigned long long) cnt[t] : 0ull;
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, c);
}
// counts[which] = off[m - 1] + cnt[m - 1] (the total behind an exclusive scan)
__global__ void k_st_total(int m, const int* __restrict__ cnt, const int* __restrict__ off, int* __restrict__ counts, int which) {
  counts[which] = m > 0 ? off[m - 1] + cnt[m - 1] : 0;
}
__global__ __launch_bounds__(PG_THREADS) void k_st_iota(int m, int* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) out[t] = t;
}
// column starts and the {entry, row} pairs of a pattern's entries sorted by column (rows ascending: the sort is stable)
__global__ __launch_bounds__(PG_THREADS) void k_st_columns(int m, int ncols, const int* __restrict__ sorted_col,
                                                           const int* __restrict__ ent, const int* __restrict__ row_of,
                                                           int* __restrict__ csc_start, int2* __restrict__ csc2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) csc2[t] = make_int2(ent[t], row_of[ent[t]]);
  if (t <= ncols) {
    int lo = 0, hi = m;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sorted_col[mid] < t) lo = mid + 1; else hi = mid;
    }
    csc_start[t] = lo;
  }
}
// entry e = Ps[i, A] puts the columns B > A of row i of Q into row A of the coarse pattern
__global__ __launch_bounds__(PG_THREADS) void k_st_ce_count(int np, const int* __restrict__ pcol, const int* __restrict__ prow_of,
                                                            const int* __restrict__ qrow_start, const int* __restrict__ qcol,
                                                            int* __restrict__ first, int* __restrict__ cnt, int* __restrict__ qdiag) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= np) return;
  const int A = pcol[e], i = prow_of[e];
  int lo = qrow_start[i], hi = qrow_start[i + 1];
  const int end = hi;
  while (lo < hi) {  // first column behind A
    const int mid = (lo + hi) >> 1;
    if (qcol[mid] <= A) lo = mid + 1; else hi = mid;
  }
  first[e] = lo;
  cnt[e]   = end - lo;
  qdiag[e] = lo - 1;  // (column A itself: the row of Q holds every column of the row of Ps)
}
__global__ __launch_bounds__(PG_THREADS) void k_st_ce_candidates(int np, int nc, const int* __restrict__ pcol,
                                                                 const int* __restrict__ qcol, const int* __restrict__ first,
                                                                 const int* __restrict__ cnt, const int* __restrict__ off,
                                                                 st_key* __restrict__ keys, unsigned long long* __restrict__ vals) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= np) return;
  const st_key A = (st_key) pcol[e];
  st_key* out    = keys + off[e];
  unsigned long long* vo = vals + off[e];
  for (int k = 0; k < cnt[e]; ++k) {
    out[k] = A * (st_key) (nc + 1) + (st_key) qcol[first[e] + k];
    vo[k]  = ((unsigned long long) (unsigned) (first[e] + k) << 32) | (unsigned) e;  // int2 {entry of Ps, entry of Q}
  }
}
__global__ __launch_bounds__(PG_THREADS) void k_st_ce_decode(int nc, const int* __restrict__ counts, int which,
                                                             const st_key* __restrict__ keys, int2* __restrict__ ceij) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= counts[which]) return;
  const st_key k = keys[t];
  const int A    = (int) (k / (st_key) (nc + 1));
  ceij[t]        = make_int2(A, (int) (k - (st_key) A * (st_key) (nc + 1)));
}

inline int st_bits(unsigned long long x) {  // number of bits needed for the values 0 .. x
  int b = 1;
  while (b < 64 && (x >> b) != 0) ++b;
  return b;
}
inline dim3 st_grid(size_t items) { return dim3((unsigned) std::max<size_t>((items + PG_THREADS - 1) / PG_THREADS, 1)); }

// sorted unique keys of keys_a[0, m) -> keys_a[0, counts[which]) (`none` keys dropped from the count); their run lengths stay in
// st_rle.  vals_out: the keys carry st_vals as payload -- a radix sort is stable, so the payloads of equal keys keep the order in
// which the candidates were written -- sorted into vals_out: with st_run_starts, the product lists of the pattern's entries.
// (rocPRIM directly: sort, run-length encode, scan; round 5 went through the hipCUB layer)
__global__ void k_st_set(int* __restrict__ p, int v) { *p = v; }
int st_sort_unique(srrg2_posegraph_s* g, int m, st_key none, int which, unsigned long long* vals_out = nullptr) {
  int rc;
  size_t t1 = 0, t2 = 0;
  const unsigned end_bit = (unsigned) std::min(64, st_bits(none));
  if ((rc = g->st_keys_b.reserve((size_t) std::max(m, 1))) || (rc = g->st_rle.reserve((size_t) std::max(m, 1)))) return rc;
  if (m == 0) {  // (nothing to sort: an empty pattern)
    HIP_TRY(hipMemsetAsync(g->st_counts.p + which, 0, sizeof(int), g->stream));
    return 0;
  }
  unsigned* nruns = reinterpret_cast<unsigned*>(g->st_counts.p + 7);
  if (vals_out)
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, g->st_keys_a.p, g->st_keys_b.p, g->st_vals.p, vals_out, (unsigned) m, 0u, end_bit, g->stream));
  else
    HIP_TRY(rocprim::radix_sort_keys(nullptr, t1, g->st_keys_a.p, g->st_keys_b.p, (unsigned) m, 0u, end_bit, g->stream));
  HIP_TRY(rocprim::run_length_encode(nullptr, t2, g->st_keys_b.p, (unsigned) m, g->st_keys_a.p, g->st_rle.p, nruns, g->stream));
  if ((rc = g->st_temp.reserve(std::max(t1, t2) + 256))) return rc;
  t1 = t2 = g->st_temp.cap;
  if (vals_out)
    HIP_TRY(rocprim::radix_sort_pairs(g->st_temp.p, t1, g->st_keys_a.p, g->st_keys_b.p, g->st_vals.p, vals_out, (unsigned) m, 0u, end_bit, g->stream));
  else
    HIP_TRY(rocprim::radix_sort_keys(g->st_temp.p, t1, g->st_keys_a.p, g->st_keys_b.p, (unsigned) m, 0u, end_bit, g->stream));
  HIP_TRY(rocprim::run_length_encode(g->st_temp.p, t2, g->st_keys_b.p, (unsigned) m, g->st_keys_a.p, g->st_rle.p, nruns, g->stream));
  hipLaunchKernelGGL(k_st_valid, dim3(1), dim3(1), 0, g->stream, g->st_keys_a.p, g->st_counts.p + 7, none, g->st_counts.p, which);
  return 0;  // (the unique keys are in st_keys_a again)
}
// starts[0 .. nruns] = the offsets of the runs of the last st_sort_unique (no `none` key among them), starts[nruns] = total
int st_run_starts(srrg2_posegraph_s* g, int nruns, int total, DevBuf<int>& starts) {
  int rc;
  if ((rc = starts.reserve((size_t) nruns + 1))) return rc;
  if (nruns > 0) {
    size_t t = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, t, g->st_rle.p, starts.p, 0, (size_t) nruns, rocprim::plus<int>(), g->stream));
    if ((rc = g->st_temp.reserve(t + 256))) return rc;
    t = g->st_temp.cap;
    HIP_TRY(rocprim::exclusive_scan(g->st_temp.p, t, g->st_rle.p, starts.p, 0, (size_t) nruns, rocprim::plus<int>(), g->stream));
  }
  hipLaunchKernelGGL(k_st_set, dim3(1), dim3(1), 0, g->stream, starts.p + nruns, total);
  return 0;
}
int st_exclusive_sum(srrg2_posegraph_s* g, int m, int which) {
  int rc;
  size_t t = 0;
  if ((rc = g->st_off.reserve((size_t) std::max(m, 1)))) return rc;
  if (m == 0) {
    HIP_TRY(hipMemsetAsync(g->st_counts.p + which, 0, sizeof(int), g->stream));
    return 0;
  }
  HIP_TRY(rocprim::exclusive_scan(nullptr, t, g->st_cnt.p, g->st_off.p, 0, (size_t) m, rocprim::plus<int>(), g->stream));
  if ((rc = g->st_temp.reserve(t + 256))) return rc;
  t = g->st_temp.cap;
  HIP_TRY(rocprim::exclusive_scan(g->st_temp.p, t, g->st_cnt.p, g->st_off.p, 0, (size_t) m, rocprim::plus<int>(), g->stream));
  hipLaunchKernelGGL(k_st_total, dim3(1), dim3(1), 0, g->stream, m, g->st_cnt.p, g->st_off.p, g->st_counts.p, which);
  return 0;
}
// the 64-bit sum of st_cnt[0, m), on the host
int st_total_of_counts(srrg2_posegraph_s* g, int m, unsigned long long* out) {
  int rc;
  if ((rc = g->st_total.reserve(1))) return rc;
  HIP_TRY(hipMemsetAsync(g->st_total.p, 0, sizeof(unsigned long long), g->stream));
  hipLaunchKernelGGL(k_st_sum64, st_grid((size_t) m), dim3(PG_THREADS), 0, g->stream, m, g->st_cnt.p, g->st_total.p);
  HIP_TRY(hipMemcpyAsync(out, g->st_total.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  return 0;
}
int st_read_count(srrg2_posegraph_s* g, int which, int* out) {
  HIP_TRY(hipMemcpyAsync(out, g->st_counts.p + which, sizeof(int), hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  return 0;
}
// entries of a pattern by column: csc_ent, csc_start, csc2
int st_columns(srrg2_posegraph_s* g, int m, int ncols, const int* col, const int* row_of, DevBuf<int>& csc_start,
               DevBuf<int>& csc_ent, DevBuf<int2>& csc2) {
  int rc;
  if ((rc = csc_start.reserve((size_t) ncols + 1)) || (rc = csc_ent.reserve((size_t) std::max(m, 1))) ||
      (rc = csc2.reserve((size_t) std::max(m, 1))) || (rc = g->st_ia.reserve((size_t) std::max(m, 1))) ||
      (rc = g->st_ib.reserve((size_t) std::max(m, 1))))
    return rc;
  if (m == 0) {  // (every column empty)
    HIP_TRY(hipMemsetAsync(csc_start.p, 0, sizeof(int) * ((size_t) ncols + 1), g->stream));
    return 0;
  }
  hipLaunchKernelGGL(k_st_iota, st_grid((size_t) m), dim3(PG_THREADS), 0, g->stream, m, g->st_ia.p);
  size_t t = 0;
  const unsigned end_bit = (unsigned) st_bits((unsigned long long) std::max(ncols, 1));
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, t, col, g->st_ib.p, g->st_ia.p, csc_ent.p, (unsigned) m, 0u, end_bit, g->stream));
  if ((rc = g->st_temp.reserve(t + 256))) return rc;
  t = g->st_temp.cap;
  HIP_TRY(rocprim::radix_sort_pairs(g->st_temp.p, t, col, g->st_ib.p, g->st_ia.p, csc_ent.p, (unsigned) m, 0u, end_bit, g->stream));
  hipLaunchKernelGGL(k_st_columns, st_grid((size_t) std::max(m, ncols + 1)), dim3(PG_THREADS), 0, g->stream, m, ncols,
                     g->st_ib.p, csc_ent.p, row_of, csc_start.p, csc2.p);
  return 0;
}

// The patterns of level L (n nodes, ne blocks, nc aggregates; L->agg, inc_start, inc_adj already on the device) -> L's pattern
// arrays, the counts, the coarse edges on the host.  `smoothed` in: wanted; out: what the fill guard allowed.  Returns 1 when the
// exact size of Q is over the limit of a smoothed level (the caller repeats the level with the tentative interpolation), 2 when
// a candidate list would need 64-bit offsets (the caller builds this hierarchy on the host).
int pg_device_patterns(srrg2_posegraph_s* g, MgLevelBufs* L, int n, int ne, int nc, long long q_limit, bool* smoothed, int* np_out,
                       int* nq_out, std::vector<int>* ceij) {
  int rc;
  if ((unsigned long long) n + 2ull * (unsigned long long) ne > g->st_offset_limit) return 2;
  const int slots  = n + 2 * ne;
  const st_key row = (st_key) (nc + 1);
  if ((rc = g->st_counts.reserve(8)) || (rc = g->st_keys_a.reserve((size_t) std::max(slots, 1))) ||
      (rc = g->st_slot.reserve((size_t) std::max(slots, 1))) || (rc = g->st_cnt.reserve((size_t) std::max(slots, 1))) ||
      (rc = L->prow_start.reserve((size_t) n + 1)) || (rc = L->qrow_start.reserve((size_t) n + 1)))
    return rc;
  int np = 0, nq = 0, bound = 0;
  auto t_stage = std::chrono::steady_clock::now();
  auto stage   = [&](int k) {
    const auto now = std::chrono::steady_clock::now();
    g->st_ms[k] += std::chrono::duration<double, std::milli>(now - t_stage).count();
    t_stage = now;
  };
  for (int attempt = 0; attempt < 2; ++attempt) {
    hipLaunchKernelGGL(k_st_p_candidates, st_grid((size_t) n), dim3(PG_THREADS), 0, g->stream, n, nc, *smoothed ? 1 : 0, L->agg.p,
                       L->inc_start.p, L->inc_adj.p, g->st_keys_a.p, g->st_slot.p);
    if ((rc = st_sort_unique(g, slots, (st_key) n * row, 0)) || (rc = st_read_count(g, 0, &np))) return rc;
    stage(0);
    if ((rc = L->pcol.reserve((size_t) std::max(np, 1))) || (rc = L->prow_of.reserve((size_t) std::max(np, 1)))) return rc;
    hipLaunchKernelGGL(k_st_decode, st_grid((size_t) std::max(np, n + 1)), dim3(PG_THREADS), 0, g->stream, n, nc, g->st_counts.p, 0,
                       g->st_keys_a.p, L->prow_start.p, L->pcol.p, L->prow_of.p);
    // candidates of Q = the fill guard's bound
    hipLaunchKernelGGL(k_st_q_count, st_grid((size_t) slots), dim3(PG_THREADS), 0, g->stream, slots, g->st_slot.p, L->agg.p,
                       L->inc_start.p, L->inc_adj.p, L->prow_start.p, g->st_cnt.p);
    unsigned long long bound64 = 0;
    if ((rc = st_total_of_counts(g, slots, &bound64))) return rc;
    stage(1);
    if (*smoothed && bound64 > (unsigned long long) (16 * q_limit)) {
      *smoothed = false;
      continue;
    }
    if (bound64 > g->st_offset_limit) return 2;
    if ((rc = st_exclusive_sum(g, slots, 1))) return rc;
    bound = (int) bound64;
    break;
  }
  if ((rc = g->st_keys_a.reserve((size_t) std::max(bound, 1))) || (rc = g->st_vals.reserve((size_t) std::max(bound, 1))) ||
      (rc = L->qp_list.reserve((size_t) std::max(bound, 1))))
    return rc;
  hipLaunchKernelGGL(k_st_q_candidates, st_grid((size_t) slots), dim3(PG_THREADS), 0, g->stream, slots, nc, g->st_slot.p,
                     L->inc_start.p, L->inc_adj.p, L->prow_start.p, L->pcol.p, g->st_cnt.p, g->st_off.p, g->st_keys_a.p, g->st_vals.p);
  if ((rc = st_sort_unique(g, bound, (st_key) n * row, 2, L->qp_list.p)) || (rc = st_read_count(g, 2, &nq))) return rc;
  stage(2);
  if (*smoothed && (long long) nq > q_limit) return 1;
  if ((rc = st_run_starts(g, nq, bound, L->qp_start))) return rc;  // (the candidates of an entry of Q = its products)
  L->nqp = bound;
  if ((rc = L->qcol.reserve((size_t) std::max(nq, 1))) || (rc = L->qrow_of.reserve((size_t) std::max(nq, 1)))) return rc;
  hipLaunchKernelGGL(k_st_decode, st_grid((size_t) std::max(nq, n + 1)), dim3(PG_THREADS), 0, g->stream, n, nc, g->st_counts.p, 2,
                     g->st_keys_a.p, L->qrow_start.p, L->qcol.p, L->qrow_of.p);
  if ((rc = st_columns(g, np, nc, L->pcol.p, L->prow_of.p, L->pcsc_start, L->pcsc_ent, L->pcsc2)) ||
      (rc = st_columns(g, nq, nc, L->qcol.p, L->qrow_of.p, L->qcsc_start, L->qcsc_ent, L->qcsc2)))
    return rc;
  // coarse edges
  int mce = 0, nce = 0;
  if ((rc = g->st_cnt.reserve((size_t) std::max(np, 1))) || (rc = g->st_ia.reserve((size_t) std::max(np, 1))) ||
      (rc = L->qdiag.reserve((size_t) std::max(np, 1))))
    return rc;
  hipLaunchKernelGGL(k_st_ce_count, st_grid((size_t) np), dim3(PG_THREADS), 0, g->stream, np, L->pcol.p, L->prow_of.p,
                     L->qrow_start.p, L->qcol.p, g->st_ia.p, g->st_cnt.p, L->qdiag.p);
  {
    unsigned long long mce64 = 0;
    if ((rc = st_total_of_counts(g, np, &mce64))) return rc;
    stage(3);
    if (mce64 > g->st_offset_limit) return 2;
    mce = (int) mce64;
  }
  if ((rc = st_exclusive_sum(g, np, 3))) return rc;
  if ((rc = g->st_keys_a.reserve((size_t) std::max(mce, 1))) || (rc = g->st_vals.reserve((size_t) std::max(mce, 1))) ||
      (rc = L->gp_list.reserve((size_t) std::max(mce, 1))))
    return rc;
  hipLaunchKernelGGL(k_st_ce_candidates, st_grid((size_t) np), dim3(PG_THREADS), 0, g->stream, np, nc, L->pcol.p, L->qcol.p,
                     g->st_ia.p, g->st_cnt.p, g->st_off.p, g->st_keys_a.p, g->st_vals.p);
  if ((rc = st_sort_unique(g, mce, (st_key) nc * row, 4, L->gp_list.p)) || (rc = st_read_count(g, 4, &nce))) return rc;
  if ((rc = st_run_starts(g, nce, mce, L->gp_start))) return rc;  // (the candidates of a coarse edge = its products)
  L->ngp = mce;
  ceij->assign(2 * (size_t) nce, 0);
  if (nce > 0) {
    // (decoded into the key scratch's other half, read back for the next level's matching)
    int2* out = reinterpret_cast<int2*>(g->st_keys_b.p);
    hipLaunchKernelGGL(k_st_ce_decode, st_grid((size_t) nce), dim3(PG_THREADS), 0, g->stream, nc, g->st_counts.p, 4, g->st_keys_a.p, out);
    HIP_TRY(hipMemcpyAsync(ceij->data(), out, sizeof(int2) * (size_t) nce, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
  }
  stage(4);
  *np_out = np;
  *nq_out = nq;
  return 0;
}

// Aggregation hierarchy from the graph's structure and the current poses (host; only when the structure changed).
int build_hierarchy(srrg2_posegraph_s* g) {
  const int V = g->V, E = g->E, D = g->D, T = g->T;
  const auto t_begin = std::chrono::steady_clock::now();
  double ms_match = 0.0, ms_pattern = 0.0, ms_inc = 0.0, ms_p = 0.0, ms_q = 0.0, ms_csc = 0.0, ms_ce = 0.0, ms_up = 0.0;
  auto ms_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  HostPool pool(std::min(16, HostPool::usable_cpus()));
  bool device_structure = g->sw.device_structure;
  for (double& m : g->st_ms) m = 0.0;
  g->pg_force_tentative.clear();
  g->levels.clear();  // (the levels' device buffers stay in g->level_pool: a rebuild reuses them, they only ever grow)
  std::vector<float> poses((size_t) std::max(V, 1) * T);
  HIP_TRY(hipMemcpy(poses.data(), g->poses.p, sizeof(float) * (size_t) V * T, hipMemcpyDeviceToHost));
  // The geometry the matching measures distances in.  The CURRENT poses are a poor one when the hierarchy is built from an initial
  // guess: C5's odometry integration has drifted by up to 62 m, two poses half a metre apart through a loop closure look metres
  // apart, the "nearest free neighbour" is then the odometry neighbour, the aggregates come out as pieces of the trajectory instead
  // of compact blobs -- and the cycle built on them needs 42 - 51 CG iterations per solve once the poses have converged, where
  // aggregates matched at the converged poses need 28 (the same graph after an append, round 5: [19, 26, 28, 29, ...]).  Round 6:
  // positions from a BREADTH-FIRST spanning tree of the measurements, rooted at the Fixed variables -- every pose is its tree
  // parent's composed with the factor's Z, a path of ~100 factors instead of up to 50 000, so neighbours agree to centimetres
  // whatever the initial guess is; it depends on the topology and the measurements only, like the structure it is used for.
  // (sw.tree_positions = false / SRRG2_AMD_PG_TREE_POSITIONS=0: the current poses, as before)
  auto position = [&](int v, float* out) {
    const float* X = poses.data() + (size_t) v * T;
    if (D == 6) { out[0] = X[3]; out[1] = X[7]; out[2] = X[11]; } else { out[0] = X[2]; out[1] = X[5]; out[2] = 0.f; }
  };
  // level 0: every variable is a node; edges = enabled factors between two free variables
  int n = V;
  std::vector<int> eij, rep0((size_t) n), act;
  std::vector<char> excluded((size_t) n, 0);
  for (int v = 0; v < n; ++v) {
    rep0[(size_t) v]     = v;
    excluded[(size_t) v] = g->h_fixed[(size_t) v] ? 1 : 0;
  }
  for (int e = 0; e < E; ++e) {
    const int i = g->h_ij[2 * (size_t) e], j = g->h_ij[2 * (size_t) e + 1];
    if (!g->h_enabled[(size_t) e] || g->h_fixed[(size_t) i] || g->h_fixed[(size_t) j]) continue;
    eij.push_back(i);
    eij.push_back(j);
    act.push_back(e);
  }
  int rc;
  if ((rc = upload(g->act_edge, act))) return rc;
  g->coarsest_dense = 1;
  const auto t_tree = std::chrono::steady_clock::now();
  if (g->sw.tree_positions && E > 0) {
    const std::vector<float>& hZ = g->h_Z;  // (the measurements' host mirror: set() / add_factor keep it)
    // adjacency of the free variables over level 0's edges (eij / act, above); the factors that touch a Fixed variable seed the tree
    const int ne0 = (int) (eij.size() / 2);
    // ({neighbour, factor code} side by side: the walk below is bound by its dependent look-ups -- 5.9 ms with the neighbour
    // fetched from the factor's endpoints, per visit)
    std::vector<int> a_start((size_t) V + 1, 0);
    std::vector<int2> a_adj((size_t) std::max(2 * ne0, 1));
    for (int k = 0; k < 2 * ne0; ++k) a_start[(size_t) eij[(size_t) k] + 1]++;
    for (int v = 0; v < V; ++v) a_start[(size_t) v + 1] += a_start[(size_t) v];
    {
      std::vector<int> cur(a_start.begin(), a_start.end() - 1);
      for (int k = 0; k < ne0; ++k) {
        const int i = eij[2 * (size_t) k], j = eij[2 * (size_t) k + 1];
        a_adj[(size_t) cur[(size_t) i]++] = make_int2(j, 2 * act[(size_t) k]);      // this end is the factor's first endpoint
        a_adj[(size_t) cur[(size_t) j]++] = make_int2(i, 2 * act[(size_t) k] + 1);  // ... its second
      }
    }
    if (g->sw.debug) std::fprintf(stderr, "  tree: adjacency %.2f ms\n", ms_since(t_tree));
    std::vector<char> seen((size_t) V, 0);
    std::vector<int> queue;
    queue.reserve((size_t) V);
    auto place = [&](int u, int w, int e, bool w_is_first) {  // X_w from X_u through factor e
      // (plain float products: this geometry ranks neighbours, centimetres matter and not the last bit -- dm::se3_compose's
      // float64 products with their conversions were 4 of the walk's 4.7 ms)
      float* Xw = poses.data() + (size_t) w * T;
      const float* A = poses.data() + (size_t) u * T;
      const float* Ze = hZ.data() + (size_t) e * T;
      const int R = D == 6 ? 3 : 2, W = R + 1;  // rotation size, row stride
      float B[12];
      if (w_is_first) {  // Z^-1 = [R^T, -R^T t]
        for (int r = 0; r < R; ++r) {
          float t = 0.f;
          for (int c = 0; c < R; ++c) {
            B[r * W + c] = Ze[c * W + r];
            t -= Ze[c * W + r] * Ze[c * W + R];
          }
          B[r * W + R] = t;
        }
      } else {
        for (int k = 0; k < R * W; ++k) B[k] = Ze[k];
      }
      float out[12];
      for (int r = 0; r < R; ++r) {
        for (int c = 0; c < W; ++c) {
          float v = c == R ? A[r * W + R] : 0.f;
          for (int k = 0; k < R; ++k) v += A[r * W + k] * B[k * W + c];
          out[r * W + c] = v;
        }
      }
      for (int k = 0; k < R * W; ++k) Xw[k] = out[k];
      if (D != 6) { Xw[6] = 0.f; Xw[7] = 0.f; Xw[8] = 1.f; }
    };
    // (the walk only records who is placed from whom through which factor; the poses follow in discovery order -- parents before
    // children -- with the factor's Z and the parent's pose prefetched a few steps ahead: every placement reads a random 48 bytes
    // of 9.6 MB of measurements, 50 000 dependent cache misses were 5 of the tree's 6 ms)
    struct Placed { int w, u, code; };
    std::vector<Placed> placed;
    placed.reserve((size_t) V);
    auto grow = [&]() {
      for (size_t head = 0; head < queue.size(); ++head) {
        const int u = queue[head];
        for (int k = a_start[(size_t) u]; k < a_start[(size_t) u + 1]; ++k) {
          const int w = a_adj[(size_t) k].x;
          if (seen[(size_t) w]) continue;
          seen[(size_t) w] = 1;
          placed.push_back({w, u, a_adj[(size_t) k].y});
          queue.push_back(w);
        }
      }
      queue.clear();
    };
    auto place_all = [&]() {
      const size_t np_ = placed.size();
      for (size_t k = 0; k < np_; ++k) {
        if (k + 12 < np_) {
          __builtin_prefetch(hZ.data() + (size_t) (placed[k + 12].code >> 1) * T);
          __builtin_prefetch(poses.data() + (size_t) placed[k + 12].u * T);
        }
        place(placed[k].u, placed[k].w, placed[k].code >> 1, (placed[k].code & 1) != 0);
      }
      placed.clear();
    };
    for (int v = 0; v < V; ++v) seen[(size_t) v] = g->h_fixed[(size_t) v] ? 1 : 0;  // (the Fixed variables keep their poses)
    for (int e = 0; e < E; ++e) {  // the free neighbours of the Fixed variables are the tree's first generation, in factor order
      if (!g->h_enabled[(size_t) e]) continue;
      const int i = g->h_ij[2 * (size_t) e], j = g->h_ij[2 * (size_t) e + 1];
      if (g->h_fixed[(size_t) i] && !seen[(size_t) j]) { seen[(size_t) j] = 1; place(i, j, e, false); queue.push_back(j); }
      else if (g->h_fixed[(size_t) j] && !seen[(size_t) i]) { seen[(size_t) i] = 1; place(j, i, e, true); queue.push_back(i); }
    }
    if (g->sw.debug) std::fprintf(stderr, "  tree: + seeds %.2f ms\n", ms_since(t_tree));
    grow();
    if (g->sw.debug) std::fprintf(stderr, "  tree: + walk %.2f ms\n", ms_since(t_tree));
    place_all();
    if (g->sw.debug) std::fprintf(stderr, "  tree: + poses %.2f ms\n", ms_since(t_tree));
    for (int v = 0; v < V; ++v)  // a component without a Fixed variable: its first variable keeps its pose
      if (!seen[(size_t) v]) { seen[(size_t) v] = 1; queue.push_back(v); grow(); place_all(); }
  }
  const double ms_tree = ms_since(t_tree);

  // matching passes per level: 2 -> aggregates of <= 4 poses, 3 -> <= 8 (fewer levels, slower convergence; measured in
  // DESIGN.md section 6)
  const int match_passes = g->sw.match_passes;
  for (int level = 0; level < MG_MAX_LEVELS; ++level) {
    const int ne = (int) (eij.size() / 2);
    if ((size_t) level >= g->level_pool.size()) g->level_pool.push_back(new MgLevelBufs());
    MgLevelBufs* L = g->level_pool[(size_t) level];
    g->levels.push_back(L);
    L->nc = L->nce = L->np = L->nq = 0;
    L->row_parts = L->col_parts = L->prow_parts = 1;
    L->smoothed = 1;
    L->n  = n;
    L->ne = ne;
    {  // lanes per row of H: ~8 incidences each
      const double avg = n > 0 ? 2.0 * ne / n : 0.0;
      int parts = 1;
      while (parts < 16 && 8.0 * parts < avg) parts *= 2;
      L->row_parts = parts;
    }
    // incidence lists in (node, edge) order
    const auto t_inc = std::chrono::steady_clock::now();
    std::vector<int> inc_start((size_t) n + 1, 0);
    std::vector<int2> inc_adj((size_t) std::max(2 * ne, 1), make_int2(-1, 0));
    for (int e = 0; e < ne; ++e) {
      inc_start[(size_t) eij[2 * (size_t) e] + 1]++;
      inc_start[(size_t) eij[2 * (size_t) e + 1] + 1]++;
    }
    for (int v = 0; v < n; ++v) inc_start[(size_t) v + 1] += inc_start[(size_t) v];
    {
      std::vector<int> cur(inc_start.begin(), inc_start.end() - 1);
      for (int e = 0; e < ne; ++e) {
        inc_adj[(size_t) cur[(size_t) eij[2 * (size_t) e]]++]     = make_int2(eij[2 * (size_t) e + 1], 2 * e);
        inc_adj[(size_t) cur[(size_t) eij[2 * (size_t) e + 1]]++] = make_int2(eij[2 * (size_t) e], 2 * e + 1);
      }
    }
    std::vector<int2> e2((size_t) std::max(ne, 1));
    for (int e = 0; e < ne; ++e) e2[(size_t) e] = make_int2(eij[2 * (size_t) e], eij[2 * (size_t) e + 1]);
    if ((rc = L->eij.reserve(e2.size()))) return rc;
    HIP_TRY(hipMemcpy(L->eij.p, e2.data(), sizeof(int2) * e2.size(), hipMemcpyHostToDevice));
    if ((rc = upload(L->inc_start, inc_start)) || (rc = upload(L->rep0, rep0))) return rc;
    if ((rc = L->inc_adj.reserve(inc_adj.size()))) return rc;
    HIP_TRY(hipMemcpy(L->inc_adj.p, inc_adj.data(), sizeof(int2) * inc_adj.size(), hipMemcpyHostToDevice));
    if ((rc = L->Hd.reserve((size_t) std::max(n, 1) * D * D)) || (rc = L->Ho.reserve((size_t) std::max(ne, 1) * D * D)) ||
        (rc = L->P.reserve((size_t) std::max(n, 1) * D * D)) || (rc = L->Dinv.reserve((size_t) std::max(n, 1) * D * D)) ||
        (rc = L->x.reserve((size_t) std::max(n, 1) * D)) || (rc = L->r.reserve((size_t) std::max(n, 1) * D)) ||
        (rc = L->res.reserve((size_t) std::max(n, 1) * D)) || (rc = L->Hdf.reserve((size_t) std::max(n, 1) * D * D)) ||
        (rc = L->Hof.reserve((size_t) std::max(ne, 1) * D * D)) || (rc = L->Dinvf.reserve((size_t) std::max(n, 1) * D * D)))
      return rc;
    ms_inc += ms_since(t_inc);
    int free_nodes = 0;
    for (int v = 0; v < n; ++v) free_nodes += excluded[(size_t) v] ? 0 : 1;
    if (free_nodes <= MG_COARSEST_NODES && level > 0) break;  // this is the coarsest level
    if (level == 0 && free_nodes <= MG_COARSEST_NODES && n <= MG_COARSEST_NODES) break;
    // two passes of greedy pairwise matching with the nearest unmatched neighbour: aggregates of <= 4 nodes
    const auto t_match = std::chrono::steady_clock::now();
    std::vector<int> agg((size_t) n);
    for (int v = 0; v < n; ++v) agg[(size_t) v] = excluded[(size_t) v] ? -1 : v;  // singletons first
    int nagg = n;
    {
      // pass structure: cur[v] = aggregate of v after the previous pass, named by its first member (an aggregate a that takes a
      // partner b keeps the name a, and a < b: every node below a has been matched by the time a is visited), so `cur[a] == a`
      // says "a names an aggregate" and the representative position of aggregate a is node a's.  The candidates of an
      // aggregate are the aggregates of its members' graph neighbours, enumerated through the members' incidence lists: the
      // same candidate set as a neighbour list between aggregates (round 5 built one per pass from the edges: count, scan,
      // fill -- three passes over 2 ne entries where this needs one over n), and the minimum with its tie-break does not care
      // how often or in which order a candidate shows up: the same matching, 6.7 -> ~4 ms on C5.
      std::vector<float> pos(3 * (size_t) n);
      for (int v = 0; v < n; ++v) position(rep0[(size_t) v], pos.data() + 3 * (size_t) v);
      std::vector<int> cur(agg), mstart, mlist, match((size_t) n);
      for (int pass = 0; pass < match_passes; ++pass) {
        const bool direct = pass == 0;  // (every aggregate is one node)
        if (!direct) {  // members of every aggregate (CSR by aggregate name, members ascending)
          mstart.assign((size_t) n + 1, 0);
          mlist.resize((size_t) n);
          for (int v = 0; v < n; ++v)
            if (cur[(size_t) v] >= 0) mstart[(size_t) cur[(size_t) v] + 1]++;
          for (int a = 0; a < n; ++a) mstart[(size_t) a + 1] += mstart[(size_t) a];
          std::vector<int> fill(mstart.begin(), mstart.end() - 1);
          for (int v = 0; v < n; ++v)
            if (cur[(size_t) v] >= 0) mlist[(size_t) fill[(size_t) cur[(size_t) v]]++] = v;
        }
        std::fill(match.begin(), match.end(), -1);
        for (int a = 0; a < n; ++a) {
          if (cur[(size_t) a] != a || match[(size_t) a] >= 0) continue;
          const float* pa = pos.data() + 3 * (size_t) a;
          int best = -1;
          float bd = 3.0e38f;
          const int m0 = direct ? 0 : mstart[(size_t) a], m1 = direct ? 1 : mstart[(size_t) a + 1];
          for (int m = m0; m < m1; ++m) {
            const int v = direct ? a : mlist[(size_t) m];
            for (int k = inc_start[(size_t) v]; k < inc_start[(size_t) v + 1]; ++k) {
              const int b = cur[(size_t) inc_adj[(size_t) k].x];
              if (b < 0 || b == a || match[(size_t) b] >= 0) continue;
              const float* pb = pos.data() + 3 * (size_t) b;
              const float d = (pa[0] - pb[0]) * (pa[0] - pb[0]) + (pa[1] - pb[1]) * (pa[1] - pb[1]) + (pa[2] - pb[2]) * (pa[2] - pb[2]);
              if (d < bd || (d == bd && b < best)) { bd = d; best = b; }
            }
          }
          match[(size_t) a] = a;
          if (best >= 0) match[(size_t) best] = a;
        }
        for (int v = 0; v < n; ++v)
          if (cur[(size_t) v] >= 0) cur[(size_t) v] = match[(size_t) cur[(size_t) v]];
      }
      // renumber the aggregates in order of their first member
      std::vector<int> renum((size_t) n, -1);
      nagg = 0;
      for (int v = 0; v < n; ++v) {
        if (cur[(size_t) v] < 0) continue;
        if (renum[(size_t) cur[(size_t) v]] < 0) renum[(size_t) cur[(size_t) v]] = nagg++;
        agg[(size_t) v] = renum[(size_t) cur[(size_t) v]];
      }
    }
    if (nagg == 0 || nagg > (int) (0.8 * free_nodes)) {  // coarsening stalled: this level is the coarsest
      if (free_nodes > 256) g->coarsest_dense = 0;
      break;
    }
    ms_match += ms_since(t_match);
    const auto t_pattern = std::chrono::steady_clock::now();
    const int nc = nagg;
    std::vector<int> crep0((size_t) nc, -1);
    for (int v = 0; v < n; ++v)  // representative of an aggregate = its first member
      if (agg[(size_t) v] >= 0 && crep0[(size_t) agg[(size_t) v]] < 0) crep0[(size_t) agg[(size_t) v]] = rep0[(size_t) v];
    // pattern of the smoothed interpolation: row i = the aggregates of i and of its neighbours.  Fill guard: a hub (a
    // pose with thousands of factors) puts its whole row into the rows of Q = H Ps of all its neighbours -- quadratic in
    // its degree.  When the pattern of Q would exceed 64 blocks per node (C5: 11 / 43 / 77 on its three levels) this
    // level falls back to the tentative interpolation (row = the node's own aggregate, no smoothing).
    bool smoothed = g->sw.omega_p != 0.0 &&
                    !g->pg_force_tentative.count(level);
    const long long q_limit = 64LL * std::max(n, 4096);
    int np = 0, nq = 0;
    std::vector<int> ceij;
    if (device_structure) {
      // the patterns on the device (pg_device_patterns): the arrays land in L's buffers, the coarse edges come back
      if ((rc = upload(L->agg, agg))) return rc;
      const int r = pg_device_patterns(g, L, n, ne, nc, q_limit, &smoothed, &np, &nq, &ceij);
      if (r < 0) return r;
      if (r == 1) g->pg_force_tentative.insert(level);
      if (r == 2) device_structure = false;
      if (r != 0) {  // this level again: without smoothing / on the host
        --level;
        g->levels.pop_back();
        continue;
      }
      ms_pattern += ms_since(t_pattern);
    } else {
    std::vector<int> prow_start, pcol, prow_of;
    for (int attempt = 0; attempt < 2; ++attempt) {
      pattern_rows(pool, n, nc, (long long) n + 2LL * ne, prow_start, pcol, &prow_of, [&](int v, std::vector<int>& stamp, std::vector<int>& out) {
        if (agg[(size_t) v] < 0) return;
        stamp[(size_t) agg[(size_t) v]] = v;
        out.push_back(agg[(size_t) v]);
        if (!smoothed) return;
        for (int q = inc_start[(size_t) v]; q < inc_start[(size_t) v + 1]; ++q) {
          const int a = agg[(size_t) inc_adj[(size_t) q].x];
          if (a >= 0 && stamp[(size_t) a] != v) {
            stamp[(size_t) a] = v;
            out.push_back(a);
          }
        }
      });
      if (!smoothed) break;
      // upper bound of the size of Q's pattern (before duplicates are merged): cheap, and enough to stop a blow-up
      // before it is computed; the exact size is checked again below
      long long bound = 0;
      {
        const int nt = (int) std::max(1LL, std::min((long long) pool.size(), ((long long) n + 2LL * ne) / 32768 + 1));
        std::vector<long long> part((size_t) nt, 0);
        pool.run(nt, [&](int t) {
          long long b = 0;
          for (int v = (int) ((long long) n * t / nt); v < (int) ((long long) n * (t + 1) / nt); ++v) {
            b += prow_start[(size_t) v + 1] - prow_start[(size_t) v];
            for (int q = inc_start[(size_t) v]; q < inc_start[(size_t) v + 1]; ++q) {
              const int j = inc_adj[(size_t) q].x;
              b += prow_start[(size_t) j + 1] - prow_start[(size_t) j];
            }
          }
          part[(size_t) t] = b;
        });
        for (long long b : part) bound += b;
      }
      if (bound <= 16 * q_limit) break;
      smoothed = false;
    }
    ms_p += ms_since(t_pattern);
    const auto t_q = std::chrono::steady_clock::now();
    np = (int) pcol.size();
    std::vector<int> pcsc_start, pcsc_ent;
    std::vector<int2> pcsc2;
    columns_of(pool, nc, pcol, prow_of, pcsc_start, pcsc_ent, pcsc2);
    // pattern of Q = H Ps: row i = union of the rows of Ps over i and its neighbours
    std::vector<int> qrow_start, qcol, qrow_of;
    const long long q_work = (long long) np * (1 + (n > 0 ? 2LL * ne / n : 0));
    pattern_rows(pool, n, nc, q_work, qrow_start, qcol, &qrow_of, [&](int v, std::vector<int>& stamp, std::vector<int>& out) {
      if (agg[(size_t) v] < 0) return;
      auto add_row = [&](int j) {
        for (int e = prow_start[(size_t) j]; e < prow_start[(size_t) j + 1]; ++e) {
          const int a = pcol[(size_t) e];
          if (stamp[(size_t) a] != v) {
            stamp[(size_t) a] = v;
            out.push_back(a);
          }
        }
      };
      add_row(v);
      for (int q = inc_start[(size_t) v]; q < inc_start[(size_t) v + 1]; ++q) add_row(inc_adj[(size_t) q].x);
    });
    if (smoothed && (long long) qcol.size() > q_limit) {
      // (exact size over the limit: this level again, without smoothing)
      g->pg_force_tentative.insert(level);
      --level;
      g->levels.pop_back();
      continue;
    }
    nq = (int) qcol.size();
    ms_q += ms_since(t_q);
    const auto t_csc = std::chrono::steady_clock::now();
    // Q by column (two-phase levels: r_c = Ps^T r - Q^T x1)
    std::vector<int> qcsc_start, qcsc_ent;
    std::vector<int2> qcsc2;
    columns_of(pool, nc, qcol, qrow_of, qcsc_start, qcsc_ent, qcsc2);
    ms_csc += ms_since(t_csc);
    const auto t_ce = std::chrono::steady_clock::now();
    // coarse edges (A < B): B in the row of Q of some row of column A of Ps
    std::vector<int> ce_start, ce_col;
    pattern_rows(pool, nc, nc, (long long) nq * 4, ce_start, ce_col, nullptr, [&](int A, std::vector<int>& stamp, std::vector<int>& out) {
      for (int m = pcsc_start[(size_t) A]; m < pcsc_start[(size_t) A + 1]; ++m) {
        const int i = prow_of[(size_t) pcsc_ent[(size_t) m]];
        // (the row is ascending: skip to the first column behind A)
        const int* end = qcol.data() + qrow_start[(size_t) i + 1];
        const int* first = qcol.data() + qrow_start[(size_t) i];
        for (const int* q = std::upper_bound(first, end, A); q < end; ++q)
          if (stamp[(size_t) *q] != A) {
            stamp[(size_t) *q] = A;
            out.push_back(*q);
          }
      }
    });
    ceij.resize(2 * ce_col.size());
    for (int A = 0; A < nc; ++A)
      for (int k = ce_start[(size_t) A]; k < ce_start[(size_t) A + 1]; ++k) {
        ceij[2 * (size_t) k]     = A;
        ceij[2 * (size_t) k + 1] = ce_col[(size_t) k];
      }
    // The product lists (MgLevel::qp_list / gp_list / qdiag), in the order the device build's stable sort leaves them: an entry
    // of Q takes the slots of its row in order -- the node itself, then its incidences -- and every slot's row of Ps ascending;
    // a coarse edge (A, B) takes the entries of column A of Ps in row order.  Two passes each (count, fill), rows / columns
    // dealt to the pool's threads: every output list belongs to one row resp. one column.
    std::vector<int> qp_start((size_t) nq + 1, 0), gp_start(ce_col.size() + 1, 0), qdiag((size_t) std::max(np, 1), 0);
    std::vector<unsigned long long> qp_list, gp_list;
    {
      const int nt = (int) std::max(1LL, std::min((long long) pool.size(), ((long long) nq + np) / 16384 + 1));
      auto rows_of = [&](int t, int total) { return std::make_pair((int) ((long long) total * t / nt), (int) ((long long) total * (t + 1) / nt)); };
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
          for (size_t q = 0; q < (size_t) nq; ++q) qp_start[q + 1] += qp_start[q];
          qp_list.resize((size_t) std::max(qp_start[(size_t) nq], 1));
        }
        pool.run(nt, [&](int t) {
          std::vector<int> colpos((size_t) std::max(nc, 1), -1), cur;
          const auto rr = rows_of(t, n);
          for (int v = rr.first; v < rr.second; ++v) {
            if (agg[(size_t) v] < 0) continue;
            const int q0 = qrow_start[(size_t) v], q1 = qrow_start[(size_t) v + 1];
            for (int q = q0; q < q1; ++q) colpos[(size_t) qcol[(size_t) q]] = q;
            if (pass == 1) cur.assign(qp_start.begin() + q0, qp_start.begin() + q1);
            auto slot = [&](int j, int hcode) {
              for (int e = prow_start[(size_t) j]; e < prow_start[(size_t) j + 1]; ++e) {
                const int q = colpos[(size_t) pcol[(size_t) e]];
                if (pass == 0)
                  qp_start[(size_t) q + 1]++;
                else
                  qp_list[(size_t) cur[(size_t) (q - q0)]++] = ((unsigned long long) (unsigned) hcode << 32) | (unsigned) e;
              }
            };
            slot(v, -1);
            for (int k = inc_start[(size_t) v]; k < inc_start[(size_t) v + 1]; ++k) slot(inc_adj[(size_t) k].x, inc_adj[(size_t) k].y);
          }
        });
      }
      const int nce_h = (int) ce_col.size();
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
          for (size_t k = 0; k < (size_t) nce_h; ++k) gp_start[k + 1] += gp_start[k];
          gp_list.resize((size_t) std::max(gp_start[(size_t) nce_h], 1));
        }
        pool.run(nt, [&](int t) {
          std::vector<int> cur;
          const auto cr = rows_of(t, nc);
          for (int A = cr.first; A < cr.second; ++A) {
            const int k0 = ce_start[(size_t) A], k1 = ce_start[(size_t) A + 1];
            if (pass == 1) cur.assign(gp_start.begin() + k0, gp_start.begin() + k1);
            for (int m = pcsc_start[(size_t) A]; m < pcsc_start[(size_t) A + 1]; ++m) {
              const int e = pcsc_ent[(size_t) m], i = prow_of[(size_t) e];
              const int* first = qcol.data() + qrow_start[(size_t) i];
              const int* end   = qcol.data() + qrow_start[(size_t) i + 1];
              const int* q     = std::upper_bound(first, end, A);
              if (pass == 0) qdiag[(size_t) e] = (int) (q - qcol.data()) - 1;
              for (; q < end; ++q) {
                const int k = (int) (std::lower_bound(ce_col.begin() + k0, ce_col.begin() + k1, *q) - ce_col.begin());
                if (pass == 0)
                  gp_start[(size_t) k + 1]++;
                else
                  gp_list[(size_t) cur[(size_t) (k - k0)]++] = ((unsigned long long) (unsigned) (q - qcol.data()) << 32) | (unsigned) e;
              }
            }
          }
        });
      }
    }
    ms_ce += ms_since(t_ce);
    ms_pattern += ms_since(t_pattern);
    const auto t_up0 = std::chrono::steady_clock::now();
    if ((rc = upload(L->agg, agg)) || (rc = upload(L->prow_start, prow_start)) || (rc = upload(L->pcol, pcol)) ||
        (rc = upload(L->prow_of, prow_of)) || (rc = upload(L->pcsc_start, pcsc_start)) || (rc = upload(L->pcsc_ent, pcsc_ent)) ||
        (rc = upload(L->qrow_start, qrow_start)) || (rc = upload(L->qcol, qcol)) || (rc = upload(L->qrow_of, qrow_of)) ||
        (rc = upload(L->qcsc_start, qcsc_start)) || (rc = upload(L->qcsc_ent, qcsc_ent)) ||
        (rc = upload(L->pcsc2, pcsc2)) || (rc = upload(L->qcsc2, qcsc2)) || (rc = upload(L->qp_start, qp_start)) ||
        (rc = upload(L->qp_list, qp_list)) || (rc = upload(L->gp_start, gp_start)) || (rc = upload(L->gp_list, gp_list)) ||
        (rc = upload(L->qdiag, qdiag)))
      return rc;
    L->nqp = qp_start[(size_t) nq];
    L->ngp = gp_start[ce_col.size()];
    ms_up += ms_since(t_up0);
    }  // (host patterns)
    const int nce = (int) (ceij.size() / 2);
    const auto t_up = std::chrono::steady_clock::now();
    L->nc  = nc;
    L->nce = nce;
    L->np  = np;
    L->nq  = nq;
    L->smoothed = smoothed ? 1 : 0;
    {  // lanes per column of Ps: ~8 entries each
      const double avg = nc > 0 ? (double) np / nc : 0.0;
      int parts = 1;
      while (parts < 32 && 8.0 * parts < avg) parts *= 2;
      L->col_parts = parts;
      const double avg_row = n > 0 ? (double) np / n : 0.0;  // lanes per row of Ps: ~4 entries each
      parts = 1;
      while (parts < 8 && 4.0 * parts < avg_row) parts *= 2;
      L->prow_parts = parts;
    }
    if ((rc = L->Ps.reserve((size_t) std::max(np, 1) * D * D)) || (rc = L->Q.reserve((size_t) std::max(nq, 1) * D * D)) ||
        (rc = L->Psf.reserve((size_t) std::max(np, 1) * D * D)) || (rc = L->Qf.reserve((size_t) std::max(nq, 1) * D * D)) ||
        (rc = L->Psfc.reserve((size_t) std::max(np, 1) * D * D)) || (rc = L->Qfc.reserve((size_t) std::max(nq, 1) * D * D)) ||
        (rc = L->pcsc_pos.reserve((size_t) std::max(np, 1))) || (rc = L->qcsc_pos.reserve((size_t) std::max(nq, 1))))
      return rc;
    // (where an entry's block goes in the column-ordered copies: the inverse of the column lists, which are on the device either way)
    if (np > 0) hipLaunchKernelGGL(k_st_invert, st_grid((size_t) np), dim3(PG_THREADS), 0, g->stream, np, L->pcsc_ent.p, L->pcsc_pos.p);
    if (nq > 0) hipLaunchKernelGGL(k_st_invert, st_grid((size_t) nq), dim3(PG_THREADS), 0, g->stream, nq, L->qcsc_ent.p, L->qcsc_pos.p);
    ms_up += ms_since(t_up);
    // next level
    n = nc;
    eij.swap(ceij);
    rep0.swap(crep0);
    excluded.assign((size_t) n, 0);
  }
  if (g->sw.debug) {
    std::fprintf(stderr, "posegraph hierarchy:");
    for (MgLevelBufs* L : g->levels)
      std::fprintf(stderr, " %d nodes / %d blocks (P %d, Q %d%s; products %lld + %lld) ->", L->n, L->ne, L->np, L->nq, L->smoothed ? "" : ", tentative",
                   L->nc > 0 ? L->nqp : 0LL, L->nc > 0 ? L->ngp : 0LL);
    std::fprintf(stderr, " coarsest %s; built in %.1f ms (spanning-tree positions %.1f, matching on the host %.1f, patterns on the %s %.1f)\n",
                 g->coarsest_dense ? "dense" : "smoothed", ms_since(t_begin), ms_tree, ms_match, device_structure ? "device" : "host", ms_pattern);
    if (device_structure)
      std::fprintf(stderr, "  on the device: P sorted %.1f, Q counted %.1f, Q sorted %.1f, columns + coarse edges counted %.1f, coarse edges sorted %.1f ms\n",
                   g->st_ms[0], g->st_ms[1], g->st_ms[2], g->st_ms[3], g->st_ms[4]);
    std::fprintf(stderr, "  incidences + uploads %.1f, P pattern %.1f, Q pattern %.1f, column lists %.1f, coarse edges %.1f, uploads %.1f ms\n",
                 ms_inc, ms_p, ms_q, ms_csc, ms_ce, ms_up);
  }
  // device views
  const int nl = (int) g->levels.size();
  std::vector<MgLevel> views((size_t) nl);
  for (int l = 0; l < nl; ++l) {
    MgLevelBufs* L = g->levels[(size_t) l];
    MgLevel& v     = views[(size_t) l];
    v.n = L->n; v.ne = L->ne; v.nc = L->nc; v.nce = L->nce; v.np = L->np; v.nq = L->nq;
    v.omega = g->sw.omega;
    v.row_parts = L->row_parts; v.col_parts = L->col_parts; v.prow_parts = L->prow_parts;
    v.eij = L->eij.p; v.inc_start = L->inc_start.p; v.inc_adj = L->inc_adj.p; v.agg = L->agg.p; v.rep0 = L->rep0.p;
    v.prow_start = L->prow_start.p; v.pcol = L->pcol.p; v.prow_of = L->prow_of.p; v.pcsc_start = L->pcsc_start.p;
    v.pcsc_ent = L->pcsc_ent.p; v.qrow_start = L->qrow_start.p; v.qcol = L->qcol.p; v.qrow_of = L->qrow_of.p;
    v.Hdf = L->Hdf.p; v.Hof = L->Hof.p; v.Dinvf = L->Dinvf.p;
    v.Psf = l + 1 < nl ? L->Psf.p : nullptr; v.Qf = l + 1 < nl ? L->Qf.p : nullptr;
    v.qcsc_start = L->qcsc_start.p; v.qcsc_ent = L->qcsc_ent.p; v.pcsc2 = L->pcsc2.p; v.qcsc2 = L->qcsc2.p;
    v.Psfc = (l + 1 < nl && g->sw.column_copies) ? L->Psfc.p : nullptr;
    v.Qfc  = (l + 1 < nl && g->sw.column_copies) ? L->Qfc.p : nullptr;
    v.pcsc_pos = L->pcsc_pos.p; v.qcsc_pos = L->qcsc_pos.p;
    v.qp_start = L->qp_start.p; v.qp_list = reinterpret_cast<const int2*>(L->qp_list.p); v.gp_start = L->gp_start.p;
    v.gp_list = reinterpret_cast<const int2*>(L->gp_list.p); v.qdiag = L->qdiag.p;
    v.Hd = L->Hd.p; v.Ho = L->Ho.p; v.P = L->P.p; v.Ps = L->Ps.p; v.Q = L->Q.p; v.Dinv = L->Dinv.p; v.x = L->x.p;
    v.r = L->r.p; v.res = L->res.p;
  }
  if ((rc = g->levels_dev.reserve((size_t) nl))) return rc;
  HIP_TRY(hipMemcpy(g->levels_dev.p, views.data(), sizeof(MgLevel) * (size_t) nl, hipMemcpyHostToDevice));
  g->level_views = views;
  const size_t N = (size_t) g->levels.back()->n * D;
  if (g->coarsest_dense) {
    if ((rc = g->coarse_A.reserve(std::max<size_t>(N * N, 1))) || (rc = g->coarse_inv.reserve(std::max<size_t>(N * N, 1)))) return rc;
  }
  g->mg_dirty = false;
  g->hier_V = V;
  g->hier_E = E;
  g->hier_builds++;
  g->ntail = 0;
  g->tail_pending = false;
  return 0;
}

// The variables / factors appended since the hierarchy was built: a forest of leaves (every new variable free, with exactly one
// factor to a variable of lower index, every new factor such a factor, enabled), at most 32 of them -> their parents and factor
// codes on the device, g->ntail set.  Anything else -> false: the caller rebuilds the hierarchy.
bool pg_classify_tail(srrg2_posegraph_s* g) {
  const int V0 = g->hier_V, E0 = g->hier_E, nt = g->V - V0;
  g->ntail = 0;
  if (!g->sw.keep_structure) return false;  // (the knob's meaning: every change rebuilds)
  if (nt <= 0 || nt > 32 || g->E - E0 != nt) return false;
  std::vector<int> parent((size_t) nt, -1), code((size_t) nt, 0);
  for (int e = E0; e < g->E; ++e) {
    if (!g->h_enabled[(size_t) e] || g->h_removed[(size_t) e]) return false;
    const int i = g->h_ij[2 * (size_t) e], j = g->h_ij[2 * (size_t) e + 1];
    const int c = std::max(i, j), p = std::min(i, j);
    if (c < V0 || parent[(size_t) (c - V0)] >= 0) return false;
    parent[(size_t) (c - V0)] = p;
    code[(size_t) (c - V0)]   = (e << 1) | (c == j ? 1 : 0);
  }
  for (int t = 0; t < nt; ++t)
    if (parent[(size_t) t] < 0 || g->h_fixed[(size_t) (V0 + t)]) return false;
  const int D = g->D;
  if (upload(g->tail_parent, parent) || upload(g->tail_ecode, code) || g->tail_K.reserve((size_t) nt * D * D) ||
      g->tail_c.reserve((size_t) nt * D))
    return false;
  g->ntail = nt;
  return true;
}

template <int D>
int pg_solve_t(srrg2_posegraph_s* g, const srrg2_posegraph_params* p, srrg2_posegraph_stats* stats, int* n_inout) {
  const int V = g->V, E = g->E, T = g->T;
  int rc;
  if (!g->mg_dirty && g->tail_pending) {  // appended since the hierarchy was built: leaves to eliminate, or a rebuild
    g->tail_pending = false;
    if (!pg_classify_tail(g)) g->mg_dirty = true;
  }
  if (g->mg_dirty && (rc = build_hierarchy(g))) return rc;
  // CG and the hierarchy cover the first Vc variables; the tail behind them is eliminated (k_pg_tail_down / k_pg_tail_up)
  const int ntail = g->ntail, Vc = V - ntail;
  const int n = Vc * D;  // CG's unknowns
  const int nl  = (int) g->levels.size() - 1;  // index of the coarsest level
  const int nb  = std::max(std::min((n + PG_ROWS - 1) / PG_ROWS, 1024), 1);  // grid-stride element-wise kernels (tiles of PG_ROWS rows)
  const int nbv = std::max((V + PG_THREADS - 1) / PG_THREADS, 1);
  const int nbe = std::max((E + PG_THREADS - 1) / PG_THREADS, 1);
  const int nchi = std::min(nbe, 1024);
  if ((rc = g->Hd.reserve((size_t) std::max(V, 1) * D * D))) return rc;
  if ((rc = g->Minv.reserve((size_t) std::max(V, 1) * D * D))) return rc;
  if ((rc = g->Ho.reserve((size_t) std::max(E, 1) * D * D))) return rc;
  if ((rc = g->contrib.reserve((size_t) std::max(E, 1) * (sizeof(EdgeContrib<D>) / sizeof(double))))) return rc;
  for (DevBuf<double>* v : {&g->b, &g->x, &g->r, &g->p, &g->Ap})
    if ((rc = v->reserve((size_t) std::max(V * D, 1)))) return rc;
  for (DevBuf<double>* v : {&g->part_rz, &g->part_rz_new, &g->part_pAp, &g->part_rr, &g->part_bb})
    if ((rc = v->reserve((size_t) nb))) return rc;
  if ((rc = g->part_chi.reserve((size_t) nchi))) return rc;
  if ((rc = g->part_n.reserve((size_t) nchi))) return rc;
  if ((rc = g->sc.reserve(1))) return rc;
  EdgeContrib<D>* contrib = reinterpret_cast<EdgeContrib<D>*>(g->contrib.p);
  const int cstride = (int) (sizeof(EdgeContrib<D>) / sizeof(double));
  const int chi_off = (int) (offsetof(EdgeContrib<D>, chi) / sizeof(double));
  MgLevelBufs* L0 = g->levels[0];
  auto pair = [&](int l) {  // level l and the next coarser one (the coarsest level's partner is itself: never read)
    MgPair P;
    P.L = g->level_views[(size_t) l];
    P.C = g->level_views[(size_t) std::min(l + 1, (int) g->level_views.size() - 1)];
    return P;
  };
  // first level that runs inside the single-workgroup launch
  int lf = nl;
  for (int l = 0; l < nl; ++l)  // (a small but dense level -- C5: 100 nodes, 4950 blocks -- is 1.4 MB of blocks: not for one workgroup)
    if (g->levels[(size_t) l]->n <= MG_FUSE_NODES && g->levels[(size_t) l]->ne <= MG_FUSE_BLOCKS) { lf = l; break; }
  auto blocks_for = [](int items) { return std::max(std::min((items + PG_THREADS - 1) / PG_THREADS, 2048), 1); };
  // damping of the Jacobi sweep that smooths the interpolation (0: plain aggregation)
  const double omega_p = g->sw.omega_p;
  // z = V-cycle(r): input levels[0].r (= g->r aliased below), output levels[0].x
  // levels 1 .. lf-1 take ONE launch down and ONE up (k_mg_down2 / k_mg_up2) instead of three each; level 0, where the
  // passes are long enough to be bound by their bytes (Q is 2.2 x the size of H there), keeps its six phases
  const bool two_phase = g->sw.two_phase;
  auto parts2 = [&](const MgLevelBufs* Lb, bool down) {
    int p2 = down ? Lb->col_parts : std::max(Lb->row_parts, Lb->prow_parts);
    return std::min(std::max(p2, 1), 8);
  };
  // (fused CG steps, k_pg_update_xr_smooth ...: `head` = level 0's x1 is already there, `check` = the first residual pass carries the
  // convergence test, `tail` = the caller's k_pg_update_dot applies the last update; sw.fused_cg)
  const bool fused_cg = g->sw.fused_cg && two_phase;
  const double tol_d  = (double) p->pcg_tolerance;
  // the bottom of the cycle as one dense operator (k_bd_*, k_mg_bottom_dense): the conditions of k_mg_down2_coarsest and a level small
  // enough for an N x N matrix
  const bool dense_bottom = g->sw.fused_bottom && two_phase && g->coarsest_dense && lf == nl && lf >= 2 && g->levels[(size_t) nl]->n <= MG_FUSE_LAST_NODES &&
                            g->levels[(size_t) nl]->n == g->levels[(size_t) nl - 1]->nc && g->levels[(size_t) nl - 1]->n * D <= 1024;
  if (dense_bottom) {
    const size_t Nb = (size_t) g->levels[(size_t) nl - 1]->n * D, Mb = (size_t) g->levels[(size_t) nl]->n * D;
    if ((rc = g->bottom_B.reserve(Nb * Nb)) || (rc = g->bottom_acc.reserve(Nb * Nb)) || (rc = g->bottom_G.reserve(Nb * Mb)) ||
        (rc = g->bottom_W.reserve(Nb * Mb)))
      return rc;
  }
  // level 1 on its six phases (through H: C5 18 MB per residual pass) instead of the two phases through Q (31 MB each way), with the
  // fused restriction: five launches for two, fewer bytes (experiment switch SRRG2_AMD_PG_L1_SIX; needs a two-phase level 2 below)
  const bool l1_six = g->sw.l1_six && fused_cg && lf >= 3;
  auto vcycle2 = [&](bool head, bool check, bool tail) {
    const MgLevelBufs* L0b = g->levels[0];
    const int bl0 = blocks_for(L0b->n * D), bc0 = blocks_for(L0b->nc * D * L0b->col_parts), br0 = blocks_for(L0b->n * D * L0b->row_parts);
    if (!head) hipLaunchKernelGGL(k_mg_op<D>, dim3(bl0), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_SMOOTH0, pair(0), g->sc.p);
    if (fused_cg)
      hipLaunchKernelGGL(k_mg_residual0<D>, dim3(br0), dim3(PG_THREADS), 0, g->stream, pair(0), check ? 1 : 0, nb, tol_d, g->part_rr.p,
                         g->part_bb.p, g->sc.p);
    else
      hipLaunchKernelGGL(k_mg_op<D>, dim3(br0), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESIDUAL, pair(0), g->sc.p);
    if (fused_cg && L0b->nc > 0) {  // restriction + x1 of level 1 in one launch
      const int pp = std::min(std::max(L0b->col_parts / 2, 1), 8);
      hipLaunchKernelGGL(k_mg_restrict_smooth<D>, dim3((unsigned) (((size_t) L0b->nc * 8 * pp + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS),
                         0, g->stream, pair(0), pp, g->sc.p);
    } else {
      hipLaunchKernelGGL(k_mg_op<D>, dim3(bc0), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESTRICT, pair(0), g->sc.p);
      if (lf > 1)  // x1 of level 1 (the levels below get theirs from k_mg_down2)
        hipLaunchKernelGGL(k_mg_op<D>, dim3(blocks_for(g->levels[1]->n * D)), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_SMOOTH0,
                           pair(1), g->sc.p);
    }
    // (the last two-phase level's down phase and the dense coarsest solve share a launch when the coarsest level is tiny)
    const bool fuse_last = g->coarsest_dense && lf == nl && lf >= 2 && g->levels[(size_t) nl]->n <= MG_FUSE_LAST_NODES &&
                           g->levels[(size_t) nl]->n == g->levels[(size_t) nl - 1]->nc;
    const bool bottom = dense_bottom;  // (k_mg_bottom_dense: that launch and the up phase of level lf - 1 as one dense operator)
    for (int l = 1; l < lf; ++l) {
      const MgLevelBufs* Lb = g->levels[(size_t) l];
      if (l == 1 && l1_six) {  // level 1 through H instead of Q: residual, then restriction + x1 of level 2
        hipLaunchKernelGGL(k_mg_op<D>, dim3(blocks_for(Lb->n * D * Lb->row_parts)), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESIDUAL, pair(1), g->sc.p);
        const int pp = std::min(std::max(Lb->col_parts / 2, 1), 8);
        hipLaunchKernelGGL(k_mg_restrict_smooth<D>, dim3((unsigned) (((size_t) Lb->nc * 8 * pp + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS),
                           0, g->stream, pair(1), pp, g->sc.p);
        continue;
      }
      if (l == lf - 1 && bottom)
        hipLaunchKernelGGL(k_mg_bottom_dense<D>, dim3((unsigned) (((size_t) Lb->n * D * 32 + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS), 0,
                           g->stream, pair(l), g->bottom_B.p, g->sc.p);
      else if (l == lf - 1 && fuse_last)
        hipLaunchKernelGGL(k_mg_down2_coarsest<D>, dim3(1), dim3(1024), 0, g->stream, pair(l), g->coarse_inv.p, g->sc.p);
      else if (Lb->nc > 0)
        hipLaunchKernelGGL(k_mg_down2<D>, dim3((unsigned) Lb->nc), dim3(MG_DOWN2_THREADS), 0, g->stream, pair(l), g->sc.p);
    }
    if (!fuse_last)
      hipLaunchKernelGGL(k_mg_coarse_cycle<D>, dim3(1), dim3(1024), 0, g->stream, g->levels_dev.p, lf, nl, g->coarse_inv.p,
                         g->coarsest_dense, g->sc.p);
    for (int l = lf - 1 - (bottom ? 1 : 0); l >= 1; --l) {
      const MgLevelBufs* Lb = g->levels[(size_t) l];
      if (l == 1 && l1_six) {  // x = x1 + Ps x_c (level 2's result is in its res), residual, update: the result in level 1's x
        hipLaunchKernelGGL(k_mg_prolong_res<D>, dim3(blocks_for(Lb->n * D * Lb->prow_parts)), dim3(PG_THREADS), 0, g->stream, pair(1), 0, g->sc.p);
        hipLaunchKernelGGL(k_mg_op<D>, dim3(blocks_for(Lb->n * D * Lb->row_parts)), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESIDUAL, pair(1), g->sc.p);
        hipLaunchKernelGGL(k_mg_op<D>, dim3(blocks_for(Lb->n * D)), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_UPDATE, pair(1), g->sc.p);
        continue;
      }
      const int pp = parts2(Lb, false);
      hipLaunchKernelGGL(k_mg_up2<D>, dim3((unsigned) (((size_t) Lb->n * 8 * pp + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS), 0, g->stream, pair(l), pp,
                         l + 1 < lf ? 1 : 0, g->sc.p);
    }
    const int bp0 = blocks_for(L0b->n * D * L0b->prow_parts);
    if (lf > 1)
      hipLaunchKernelGGL(k_mg_prolong_res<D>, dim3(bp0), dim3(PG_THREADS), 0, g->stream, pair(0), l1_six ? 1 : 0, g->sc.p);
    else
      hipLaunchKernelGGL(k_mg_op<D>, dim3(bp0), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_PROLONG, pair(0), g->sc.p);
    hipLaunchKernelGGL(k_mg_op<D>, dim3(br0), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESIDUAL, pair(0), g->sc.p);
    if (!tail) hipLaunchKernelGGL(k_mg_op<D>, dim3(bl0), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_UPDATE, pair(0), g->sc.p);
  };
  const bool fused_path = fused_cg && lf >= 1 && nl >= 1;  // (the two-phase cycle runs, with the fused CG steps around it)
  auto vcycle = [&]() {
    if (two_phase && lf >= 1 && nl >= 1) {
      vcycle2(false, false, false);
      return;
    }
    for (int l = 0; l < lf; ++l) {
      const MgLevelBufs* Lb = g->levels[(size_t) l];
      const int bl = blocks_for(Lb->n * D), bc = blocks_for(Lb->nc * D * Lb->col_parts), br = blocks_for(Lb->n * D * Lb->row_parts);
      hipLaunchKernelGGL(k_mg_op<D>, dim3(bl), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_SMOOTH0, pair(l), g->sc.p);
      hipLaunchKernelGGL(k_mg_op<D>, dim3(br), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESIDUAL, pair(l), g->sc.p);
      hipLaunchKernelGGL(k_mg_op<D>, dim3(bc), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESTRICT, pair(l), g->sc.p);
    }
    hipLaunchKernelGGL(k_mg_coarse_cycle<D>, dim3(1), dim3(1024), 0, g->stream, g->levels_dev.p, lf, nl, g->coarse_inv.p,
                       g->coarsest_dense, g->sc.p);
    for (int l = lf - 1; l >= 0; --l) {
      const int bl = blocks_for(g->levels[(size_t) l]->n * D), br = blocks_for(g->levels[(size_t) l]->n * D * g->levels[(size_t) l]->row_parts);
      const int bp = blocks_for(g->levels[(size_t) l]->n * D * g->levels[(size_t) l]->prow_parts);
      hipLaunchKernelGGL(k_mg_op<D>, dim3(bp), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_PROLONG, pair(l), g->sc.p);
      hipLaunchKernelGGL(k_mg_op<D>, dim3(br), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_RESIDUAL, pair(l), g->sc.p);
      hipLaunchKernelGGL(k_mg_op<D>, dim3(bl), dim3(PG_THREADS), 0, g->stream, (int) MG_OP_UPDATE, pair(l), g->sc.p);
    }
  };
  // SRRG2_AMD_PG_LAG: largest step (max |dx| over all variables) below which the next iteration keeps the hierarchy; 0 = never
  const double lag_below = g->sw.lag_below;
  bool hierarchy_fresh   = false;
  double prev_max_dx     = 1e300;
  int nstats = 0;
  constexpr int PCG_CHUNK = 10;  // CG iterations between two looks at the convergence flag
  const bool use_graph = g->sw.use_graph;
  hipGraphExec_t chunk_exec = nullptr;
  bool graph_failed = false;
  struct ExecGuard {  // (every return path below releases the instantiated graph)
    hipGraphExec_t& e;
    ~ExecGuard() { if (e) (void) hipGraphExecDestroy(e); }
  } exec_guard{chunk_exec};
  for (int it = 0; it < p->max_iterations; ++it) {
    HIP_TRY(hipMemsetAsync(g->sc.p, 0, sizeof(PgScalars), g->stream));
    if (E > 0)
      hipLaunchKernelGGL(k_pg_edges<D>, dim3(nbe), dim3(PG_THREADS), 0, g->stream, E, T, g->poses.p, g->ij.p, g->Z.p,
                         g->omega.p, g->enabled.p, g->Ho.p, contrib);
    hipLaunchKernelGGL(k_pg_chi, dim3(nchi), dim3(PG_THREADS), 0, g->stream, E, g->enabled.p, (const void*) contrib,
                       cstride, chi_off, g->part_chi.p, g->part_n.p);
    hipLaunchKernelGGL(k_pg_vertices<D>, dim3(nbv), dim3(PG_THREADS), 0, g->stream, V, g->fixed.p, g->inc_start.p,
                       g->inc_edge.p, g->enabled.p, contrib, (double) p->damping, g->Hd.p, g->b.p, g->Minv.p, g->sc.p);
    // hierarchy numerics: level 0 = float32 copies; then interpolation, Galerkin product, smoother of every level
    {
      if (ntail > 0)  // the appended leaves folded into their parents' blocks and right-hand sides
        hipLaunchKernelGGL(k_pg_tail_down<D>, dim3(1), dim3(64), 0, g->stream, ntail, Vc, g->tail_parent.p, g->tail_ecode.p, g->fixed.p,
                           g->Hd.p, g->b.p, g->Minv.p, g->Ho.p, g->tail_K.p, g->tail_c.p, g->sc.p);
      const size_t nel = std::max((size_t) Vc, (size_t) L0->ne) * D * D;
      hipLaunchKernelGGL(k_mg_pack0<D>, dim3((unsigned) ((nel + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS), 0, g->stream,
                         Vc, L0->ne, g->act_edge.p, g->Hd.p, g->Ho.p, L0->Hd.p, L0->Ho.p);
      HIP_TRY(hipMemcpyAsync(L0->Dinv.p, g->Minv.p, sizeof(double) * (size_t) Vc * D * D, hipMemcpyDeviceToDevice, g->stream));
      // The interpolation is the aggregates' rigid motion at the CURRENT poses and the coarse operators are Galerkin
      // products of the CURRENT H: 3.7 ms per Gauss-Newton iteration on C5.  Once the last step moved no variable by more
      // than `lag_below` the poses -- hence P, Ps and, to first order, H -- are what they were: the hierarchy of the
      // previous iteration is kept (still a fixed symmetric positive definite preconditioner; level 0's own blocks are
      // always the fresh ones).  Lagging it while the poses still move does not precondition at all.
      const bool reuse = lag_below > 0.0 && it > 0 && hierarchy_fresh && prev_max_dx < lag_below;
      if (reuse)
        hipLaunchKernelGGL(k_mg_to_float<D>, dim3(4096), dim3(PG_THREADS), 0, g->stream, pair(0), 0, 0);
      for (int l = 0; l < nl && !reuse; ++l) {
        MgLevelBufs* L = g->levels[(size_t) l];
        auto grid_of = [](size_t items) { return dim3((unsigned) std::max<size_t>((items + PG_THREADS - 1) / PG_THREADS, 1)); };
        hipLaunchKernelGGL(k_mg_interp<D>, dim3(blocks_for(L->n)), dim3(PG_THREADS), 0, g->stream, pair(l), T, g->poses.p);
        // Lanes per row of H in the two set-up products that walk rows (k_mg_psmooth, k_mg_hp): a quarter of the cycle's
        // `row_parts` where the launch stays above ~64 k lanes.  The cycle's kernels do a block-vector product per incidence and
        // want ~8 incidences per lane; these do a look-up and a block-BLOCK product per incidence into 36 accumulators, and the
        // butterfly that adds the lanes' blocks costs as much as several of them (C5's levels 1 / 2: Q 651 -> 557, 431 -> 381 us,
        // Ps 121 -> 87, 68 -> 47 us; the 100-node level needs all its lanes; the Galerkin product by COLUMNS of Ps is 1.3-1.6 x
        // slower with half the lanes: profiles/r8k_ab_setup_lanes.txt).
        auto setup_pair = [&](size_t items) {
          MgPair sp = pair(l);
          for (int k = 0; k < 2; ++k)
            if (sp.L.row_parts >= 2 && items * (size_t) (sp.L.row_parts / 2) >= 65536) sp.L.row_parts /= 2;
          return sp;
        };
        const MgPair sp_p = setup_pair((size_t) L->np), sp_q = setup_pair((size_t) L->nq);
        hipLaunchKernelGGL(k_mg_psmooth<D>, grid_of((size_t) L->np * sp_p.L.row_parts), dim3(PG_THREADS), 0, g->stream,
                           sp_p, L->smoothed ? omega_p : 0.0);
        if (g->sw.product_lists) {
          // lanes per output block: ~list_lane_products products each (C5: 3.3 / 14 / 60 products per entry of Q on levels 0 - 2)
          auto lanes_for = [&](double per_block, int cap) {
            int lanes = 1;
            while (lanes < cap && per_block > (double) g->sw.list_lane_products * lanes) lanes *= 2;
            return lanes;
          };
          const int lq = lanes_for(L->nq > 0 ? (double) L->nqp / L->nq : 0.0, 16);
          const int lg = lanes_for(L->nc + L->nce > 0 ? (double) (L->ngp + L->np) / (L->nc + L->nce) : 0.0, 32);
          if (g->sw.setup_f32_ps && pair(l).L.Psf) {  // (the interpolation as float32 operand: MgLevel::Psf, written by k_mg_psmooth)
            hipLaunchKernelGGL((k_mg_hp_list<D, true>), grid_of((size_t) L->nq * lq), dim3(PG_THREADS), 0, g->stream, pair(l), lq);
            hipLaunchKernelGGL((k_mg_galerkin_list<D, true>), grid_of((size_t) (L->nc + L->nce) * lg), dim3(PG_THREADS), 0, g->stream, pair(l), lg);
          } else {
            hipLaunchKernelGGL((k_mg_hp_list<D, false>), grid_of((size_t) L->nq * lq), dim3(PG_THREADS), 0, g->stream, pair(l), lq);
            hipLaunchKernelGGL((k_mg_galerkin_list<D, false>), grid_of((size_t) (L->nc + L->nce) * lg), dim3(PG_THREADS), 0, g->stream, pair(l), lg);
          }
        } else {
        hipLaunchKernelGGL(k_mg_hp<D>, grid_of((size_t) L->nq * sp_q.L.row_parts), dim3(PG_THREADS), 0, g->stream, sp_q);
        hipLaunchKernelGGL(k_mg_galerkin<D>, grid_of((size_t) (L->nc + L->nce) * L->col_parts), dim3(PG_THREADS), 0, g->stream,
                           pair(l));
        }
        hipLaunchKernelGGL(k_mg_dinv<D>, dim3((unsigned) ((L->nc + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS), 0, g->stream,
                           pair(l + 1), g->sc.p);
      }
      if (g->coarsest_dense && !reuse) {
        const size_t Nc       = (size_t) g->levels[(size_t) nl]->n * D;
        const size_t lds_need = Nc * Nc * sizeof(double);
        static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mg_coarsest_inverse<D>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
        // 2: factor and L^-1 in LDS (row-sweep inverse); 1: factor in LDS; 0: everything in global memory
        const int in_lds = !lds_ok ? 0 : (2 * lds_need <= 150 * 1024 ? 2 : (lds_need <= 150 * 1024 ? 1 : 0));
        if (!lds_ok) (void) hipGetLastError();
        hipLaunchKernelGGL(k_mg_coarsest_inverse<D>, dim3(1), dim3(1024), (size_t) in_lds * lds_need, g->stream, pair(nl),
                           g->coarse_A.p, g->coarse_inv.p, g->sc.p, in_lds);
      }
      if (dense_bottom && !reuse) {  // B of the last two-phase level (behind its D^-1, Q, Ps and the coarsest inverse)
        const MgLevelBufs* Lb = g->levels[(size_t) nl - 1];
        const int Nb = Lb->n * D, Mb = g->levels[(size_t) nl]->n * D;
        auto grid_of = [](size_t items) { return dim3((unsigned) std::max<size_t>((items + PG_THREADS - 1) / PG_THREADS, 1)); };
        HIP_TRY(hipMemsetAsync(g->bottom_acc.p, 0, sizeof(double) * (size_t) Nb * Nb, g->stream));
        hipLaunchKernelGGL(k_bd_g<D>, grid_of((size_t) Lb->n * Lb->nc), dim3(PG_THREADS), 0, g->stream, pair(nl - 1), g->bottom_G.p);
        hipLaunchKernelGGL(k_bd_w, grid_of((size_t) Nb * Mb), dim3(PG_THREADS), 0, g->stream, Nb, Mb, g->bottom_G.p, g->coarse_inv.p, g->bottom_W.p);
        hipLaunchKernelGGL(k_bd_a<D>, grid_of((size_t) Lb->n + Lb->ne), dim3(PG_THREADS), 0, g->stream, pair(nl - 1), g->bottom_acc.p);
        hipLaunchKernelGGL(k_bd_b, grid_of((size_t) Nb * Nb), dim3(PG_THREADS), 0, g->stream, Nb, Mb, g->bottom_W.p, g->bottom_G.p, g->bottom_acc.p,
                           g->bottom_B.p);
      }
      hierarchy_fresh = true;
      // the V-cycle's float32 copies of every level's blocks (the coarsest level is inverted, not cycled through)
      for (int l = 0; l <= nl && !reuse; ++l) {
        const MgLevelBufs* L = g->levels[(size_t) l];
        const size_t items   = std::max((size_t) L->n, (size_t) L->ne) * D * D;
        hipLaunchKernelGGL(k_mg_to_float<D>, dim3((unsigned) std::min<size_t>(std::max<size_t>((items + PG_THREADS - 1) / PG_THREADS, 1), 4096)),
                           dim3(PG_THREADS), 0, g->stream, pair(l), 1, 0);
        // (Q in float32, row- and column-ordered: read by the two-phase levels only -- level 0's Q is 2.7 x its Ps)
        const bool with_q = two_phase && l >= 1 && l < lf && !(l == 1 && g->sw.l1_six && g->sw.fused_cg && lf >= 3);
        if (with_q && L->nq > 0 && pair(l).L.Qf)
          hipLaunchKernelGGL(k_mg_q_to_float<D>, dim3((unsigned) ((L->nq + PG_THREADS - 1) / PG_THREADS)), dim3(PG_THREADS), 0, g->stream, pair(l));
      }
    }
    // PCG: r lives in level 0's r (the cycle's input), z = level 0's x (its output)
    double* r = L0->r.p;
    double* z = L0->x.p;
    hipLaunchKernelGGL(k_pg_pcg_init, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, g->b.p, g->x.p, r, g->part_bb.p, g->sc.p,
                       g->part_chi.p, g->part_n.p, nchi);
    if (fused_path) {
      vcycle2(false, false, true);
      hipLaunchKernelGGL(k_pg_update_dot<D>, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, pair(0), r, g->part_rz.p, g->sc.p);
    } else {
      vcycle();
      hipLaunchKernelGGL(k_pg_dot_rz, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, r, z, g->part_rz.p, g->sc.p);
    }
    hipLaunchKernelGGL(k_pg_update_p, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, nb, 1, z, g->p.p, g->part_rz.p, g->part_rz.p, g->sc.p);
    PgScalars h{};
    int launched = 0;
    auto launch_iterations = [&](int first, int count) {
      for (int k = 0; k < count; ++k) {
        double* rz_cur = ((first + k) & 1) ? g->part_rz_new.p : g->part_rz.p;
        double* rz_nxt = ((first + k) & 1) ? g->part_rz.p : g->part_rz_new.p;
        hipLaunchKernelGGL(k_pg_spmv<D>, dim3(nb), dim3(PG_THREADS), 0, g->stream, pair(0), g->p.p, g->Ap.p, g->part_pAp.p, g->sc.p);
        if (fused_path) {
          hipLaunchKernelGGL(k_pg_update_xr_smooth<D>, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, nb, g->p.p, g->Ap.p, g->x.p, r, rz_cur,
                             g->part_pAp.p, g->part_rr.p, pair(0), g->sc.p);
          vcycle2(true, true, true);
          hipLaunchKernelGGL(k_pg_update_dot<D>, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, pair(0), r, rz_nxt, g->sc.p);
        } else {
        hipLaunchKernelGGL(k_pg_update_xr, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, nb, (double) p->pcg_tolerance, g->p.p,
                           g->Ap.p, g->x.p, r, rz_cur, g->part_pAp.p, g->part_rr.p, g->part_bb.p, g->sc.p);
        hipLaunchKernelGGL(k_pg_converged, dim3(1), dim3(PG_THREADS), 0, g->stream, nb, (double) p->pcg_tolerance, g->part_rr.p,
                           g->part_bb.p, g->sc.p);
        vcycle();
        hipLaunchKernelGGL(k_pg_dot_rz, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, r, z, rz_nxt, g->sc.p);
        }
        hipLaunchKernelGGL(k_pg_update_p, dim3(nb), dim3(PG_THREADS), 0, g->stream, n, nb, 0, z, g->p.p, rz_cur, rz_nxt, g->sc.p);
      }
    };
    while (launched < p->pcg_max_iterations) {
      const int chunk = std::min(PCG_CHUNK, p->pcg_max_iterations - launched);
      // A CG iteration is ~30 small launches: issued one by one the host's launch rate (not the kernels) sets the pace.
      // A chunk of PCG_CHUNK iterations (an even number: the ping-pong of the rz partials repeats) is captured once per
      // solve() into a HIP graph and replayed; every pointer and scalar it carries is fixed for the whole call.
      if (chunk == PCG_CHUNK && use_graph && !chunk_exec && !graph_failed) {
        hipGraph_t graph = nullptr;
        const auto t_cap = std::chrono::steady_clock::now();
        if (hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          launch_iterations(0, PCG_CHUNK);
          if (hipStreamEndCapture(g->stream, &graph) != hipSuccess || !graph ||
              hipGraphInstantiate(&chunk_exec, graph, nullptr, nullptr, 0) != hipSuccess)
            chunk_exec = nullptr;
          if (graph) (void) hipGraphDestroy(graph);
        }
        if (!chunk_exec) {
          graph_failed = true;
          (void) hipGetLastError();
        }
        if (g->sw.debug)
          std::fprintf(stderr, "posegraph: CG chunk captured and instantiated in %.2f ms\n",
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cap).count());
      }
      if (chunk == PCG_CHUNK && chunk_exec) {
        HIP_TRY(hipGraphLaunch(chunk_exec, g->stream));
      } else {
        launch_iterations(launched, chunk);
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
    if (ntail > 0)
      hipLaunchKernelGGL(k_pg_tail_up<D>, dim3(1), dim3(64), 0, g->stream, ntail, Vc, g->tail_parent.p, g->fixed.p, g->tail_K.p, g->tail_c.p,
                         g->x.p, g->sc.p);
    hipLaunchKernelGGL(k_pg_apply<D>, dim3(nbv), dim3(PG_THREADS), 0, g->stream, V, Vc, T, g->kind, g->fixed.p, g->x.p,
                       g->poses.p, g->sc.p);
    HIP_TRY(hipGetLastError());
    if (lag_below > 0.0 && it + 1 < p->max_iterations) {  // the size of this step decides about the next iteration's hierarchy
      PgScalars h2{};
      HIP_TRY(hipMemcpyAsync(&h2, g->sc.p, sizeof(h2), hipMemcpyDeviceToHost, g->stream));
      HIP_TRY(hipStreamSynchronize(g->stream));
      std::memcpy(&prev_max_dx, &h2.max_dx_bits, sizeof(double));
      if (g->sw.debug) std::fprintf(stderr, "posegraph: iteration %d, max |dx| %.3e\n", it, prev_max_dx);
    }
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
  p->pcg_max_iterations = 600;
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

void srrg2_posegraph_default_tuning(srrg2_posegraph_tuning* t) {
  if (!t) return;
  std::memset(t, 0, sizeof(*t));
  t->match_passes   = 3;
  t->two_phase      = 1;
  t->use_graph      = 1;
  t->debug          = 0;
  t->keep_structure = 1;
  t->omega_p        = (float) MG_OMEGA_P;
  t->omega          = (float) MG_OMEGA;
  t->lag_below      = 0.05f;
  t->device_structure = 1;
}

static void apply_tuning(srrg2_posegraph_s* g, const srrg2_posegraph_tuning& t) {
  // (the built-in dampings are doubles: a knob left at its float default keeps the exact built-in value)
  srrg2_posegraph_tuning d;
  srrg2_posegraph_default_tuning(&d);
  const bool structure_changes = g->sw.match_passes != t.match_passes || (g->sw.omega_p != 0.0) != (t.omega_p != 0.f) ||
                                 g->sw.device_structure != (t.device_structure != 0);  // (the same arrays either way: an A/B switch)
  g->sw.match_passes   = t.match_passes;
  g->sw.omega_p        = t.omega_p == d.omega_p ? (double) MG_OMEGA_P : (double) t.omega_p;
  g->sw.omega          = t.omega == d.omega ? (double) MG_OMEGA : (double) t.omega;
  g->sw.lag_below      = (double) t.lag_below;
  g->sw.two_phase      = t.two_phase != 0;
  g->sw.use_graph      = t.use_graph != 0;
  g->sw.debug          = t.debug != 0;
  g->sw.keep_structure = t.keep_structure != 0;
  g->sw.device_structure = t.device_structure != 0;
  if (structure_changes) g->mg_dirty = true;  // (aggregate sizes / smoothed patterns are part of the structure)
  g->tuning = t;
}

int srrg2_posegraph_get_tuning(srrg2_posegraph_h g, srrg2_posegraph_tuning* t) {
  if (!g || !t) return fail(SRRG2_E_INVALID, "posegraph_get_tuning: null argument");
  *t = g->tuning;
  return 0;
}

int srrg2_posegraph_set_tuning(srrg2_posegraph_h g, const srrg2_posegraph_tuning* t) {
  if (!g || !t) return fail(SRRG2_E_INVALID, "posegraph_set_tuning: null argument");
  if (t->match_passes < 1 || t->match_passes > 8 || !(t->omega > 0.f) || !(t->omega < 2.f) || !(t->omega_p >= 0.f) ||
      !(t->omega_p < 2.f) || !(t->lag_below >= 0.f))
    return fail(SRRG2_E_INVALID, "posegraph_set_tuning: value out of range");
  apply_tuning(g, *t);
  return 0;
}

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
  {  // the strategy knobs: defaults, overridden by the environment ONCE, here
    srrg2_posegraph_tuning t;
    srrg2_posegraph_default_tuning(&t);
    auto geti = [](const char* name, int32_t& v) { if (const char* e = std::getenv(name)) v = (int32_t) std::atoi(e); };
    auto getf = [](const char* name, float& v) { if (const char* e = std::getenv(name)) v = (float) std::atof(e); };
    geti("SRRG2_AMD_PG_PASSES", t.match_passes);
    geti("SRRG2_AMD_PG_TWO_PHASE", t.two_phase);
    geti("SRRG2_AMD_PG_GRAPH", t.use_graph);
    geti("SRRG2_AMD_PG_KEEP_STRUCTURE", t.keep_structure);
    geti("SRRG2_AMD_PG_DEVICE_STRUCTURE", t.device_structure);
    getf("SRRG2_AMD_PG_OMEGA_P", t.omega_p);
    getf("SRRG2_AMD_PG_OMEGA", t.omega);
    getf("SRRG2_AMD_PG_LAG", t.lag_below);
    if (std::getenv("SRRG2_AMD_PG_DEBUG")) t.debug = 1;
    if (const char* e = std::getenv("SRRG2_AMD_PG_FUSED_CG")) g->sw.fused_cg = std::atoi(e) != 0;
    if (const char* e = std::getenv("SRRG2_AMD_PG_COLUMN_COPIES")) g->sw.column_copies = std::atoi(e) != 0;
    if (const char* e = std::getenv("SRRG2_AMD_PG_SETUP_F32_PS")) g->sw.setup_f32_ps = std::atoi(e) != 0;
    if (const char* e = std::getenv("SRRG2_AMD_PG_L1_SIX")) g->sw.l1_six = std::atoi(e) != 0;
    if (const char* e = std::getenv("SRRG2_AMD_PG_TREE_POSITIONS")) g->sw.tree_positions = std::atoi(e) != 0;
    if (const char* e = std::getenv("SRRG2_AMD_PG_FUSED_BOTTOM")) g->sw.fused_bottom = std::atoi(e) != 0;
    if (const char* e = std::getenv("SRRG2_AMD_PG_PRODUCT_LISTS")) g->sw.product_lists = std::atoi(e) != 0;
    if (const char* e = std::getenv("SRRG2_AMD_PG_LIST_LANES")) g->sw.list_lane_products = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("SRRG2_AMD_PG_OFFSET_LIMIT")) g->st_offset_limit = std::min<unsigned long long>(std::strtoull(e, nullptr, 10), 0x7fff0000ull);
    if (t.match_passes < 1) t.match_passes = 1;
    apply_tuning(g, t);
  }
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
  g->Hd.release(); g->Ho.release(); g->b.release(); g->Minv.release(); g->x.release(); g->r.release();
  g->p.release(); g->Ap.release(); g->contrib.release(); g->part_rz.release(); g->part_rz_new.release();
  g->part_pAp.release(); g->part_rr.release(); g->part_bb.release(); g->part_chi.release(); g->part_n.release();
  g->inc_start.release(); g->inc_edge.release(); g->sc.release(); g->act_edge.release(); g->levels_dev.release();
  g->coarse_A.release(); g->coarse_inv.release();
  g->bottom_acc.release(); g->bottom_G.release(); g->bottom_W.release(); g->bottom_B.release();
  g->tail_parent.release(); g->tail_ecode.release(); g->tail_K.release(); g->tail_c.release();
  g->st_keys_a.release(); g->st_keys_b.release(); g->st_cnt.release(); g->st_off.release(); g->st_slot.release();
  g->st_ia.release(); g->st_ib.release(); g->st_counts.release(); g->st_total.release(); g->st_temp.release();
  g->st_vals.release(); g->st_rle.release();
  for (MgLevelBufs* L : g->level_pool) {
    L->release();
    delete L;
  }
  g->level_pool.clear();
  g->levels.clear();
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
  // The hierarchy's STRUCTURE depends on the topology (edges, fixed and enabled masks) and -- through the matching's
  // nearest-neighbour rule, a quality heuristic -- on the poses at the time it was built; its NUMERICS are recomputed by
  // every Gauss-Newton iteration.  A set() that only brings new poses / measurements keeps the structure (the case of an
  // optimize() repeated on an unchanged graph, or of a caller that re-uploads the graph every time).
  const bool same_topology = g->sw.keep_structure && !g->mg_dirty && g->V == V && g->E == E && (int) g->h_fixed.size() == V &&
                             (int) g->h_enabled.size() == E && std::memcmp(g->h_ij.data(), ij, sizeof(int32_t) * 2 * (size_t) E) == 0 &&
                             std::memcmp(g->h_fixed.data(), fx.data(), (size_t) V) == 0 &&
                             std::memcmp(g->h_enabled.data(), en.data(), (size_t) E) == 0 &&
                             std::find(g->h_removed.begin(), g->h_removed.end(), (uint8_t) 1) == g->h_removed.end();
  g->V = V;
  g->E = E;
  g->h_Z.assign(Z, Z + (size_t) E * T);
  if (same_topology) return 0;  // (incidence lists and hierarchy structure are still those of this topology)
  g->h_ij.assign(ij, ij + 2 * (size_t) E);
  g->h_enabled.assign(en.begin(), en.begin() + E);
  g->h_removed.assign((size_t) E, 0);
  g->h_fixed.assign(fx.begin(), fx.begin() + V);
  g->mg_dirty = true;
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
  g->h_fixed.push_back(fx);
  g->inc_dirty = true;
  g->tail_pending = true;  // (the next solve eliminates it as a leaf or rebuilds the hierarchy: pg_classify_tail)
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
  g->h_Z.insert(g->h_Z.end(), Z, Z + T);
  g->h_enabled.push_back(en);
  g->h_removed.push_back(0);
  if (id_out) *id_out = E;
  g->E         = E + 1;
  g->inc_dirty = true;
  g->tail_pending = true;
  return 0;
}

/* FactorBase::setEnabled (closures are added disabled and promoted by the validator, :238-241, :283-286) */
int srrg2_posegraph_set_factor_enabled(srrg2_posegraph_h g, int factor_id, int enabled) {
  if (!g || factor_id < 0 || factor_id >= g->E) return fail(SRRG2_E_INVALID, "posegraph_set_factor_enabled: bad factor id");
  if (g->h_removed[(size_t) factor_id]) return fail(SRRG2_E_STATE, "posegraph_set_factor_enabled: factor was removed");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  const uint8_t en = enabled ? 1 : 0;
  if (g->h_enabled[(size_t) factor_id] != en) g->mg_dirty = true;  // (the hierarchy covers the enabled factors)
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
  if (g->h_enabled[(size_t) factor_id]) g->mg_dirty = true;
  g->h_enabled[(size_t) factor_id] = 0;
  g->h_removed[(size_t) factor_id] = 1;
  HIP_TRY(hipMemcpy(g->enabled.p + factor_id, &en, 1, hipMemcpyHostToDevice));
  return 0;
}

int srrg2_posegraph_structure_info(srrg2_posegraph_h g, int* hierarchy_builds, int* eliminated_leaves) {
  if (!g) return fail(SRRG2_E_INVALID, "posegraph_structure_info: null handle");
  if (hierarchy_builds) *hierarchy_builds = g->hier_builds;
  if (eliminated_leaves) *eliminated_leaves = g->ntail;
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
  g->mg_dirty = true;
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
