/*
 * o_posegraph.c -- CPU oracle of the pose-graph Gauss-Newton solve.  TEST INFRASTRUCTURE ONLY (oracle.h).
 *
 * Caller in the reference: MultiGraphSLAM_::optimize(), S/system/multi_graph_slam_impl.cpp:300-317
 * (graph->bindFactors(); global_solver->setGraph(graph); global_solver->compute()).  Graph construction:
 * makeNewMap :52-90 (variable + odometry factor, first variable Fixed :86), loopValidate :227-297 (closures
 * enabled/removed).  Variable = VariableSE2RightAD / VariableSE3QuaternionRightAD (S/mapping/local_map.h:64,75),
 * factor = SE2/SE3PosePoseGeodesicErrorFactor (S/registration/loop_closure.h:110-111).
 *
 * PARITY UNPINNED: the solver and the factor live in srrg2_solver, which is not under /root/reference and is not
 * version-pinned.  First-principles definition: e = t2v(Z^-1 Xi^-1 Xj) with
 * the variable's own chart, right perturbations, H dx = -b solved by block-Jacobi PCG (or densely, for the tests).
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct o_posegraph {
  int kind, D, tsize;
  int V, E;
  float* poses;
  uint8_t* fixed;
  int32_t* ij;
  float* Z;
  double* omega; /* E x D x D */
  uint8_t* enabled;
  int direct; /* 1 = dense Cholesky instead of PCG (small graphs, pins the PCG) */
};

static int pg_fail(int code, const char* msg) {
  /* reuse the aligner's error slot through a local copy: oracle_last_error() is defined in o_aligner.c */
  fprintf(stderr, "oracle posegraph: %s\n", msg);
  return code;
}

int oracle_posegraph_create(int variable_kind, o_posegraph** out) {
  if (!out || (variable_kind != SRRG2_SE2_RIGHT && variable_kind != SRRG2_SE3_QUAT_RIGHT))
    return pg_fail(SRRG2_E_INVALID, "create: variable kind must be SE2_RIGHT or SE3_QUAT_RIGHT");
  o_posegraph* g = (o_posegraph*) calloc(1, sizeof(o_posegraph));
  g->kind        = variable_kind;
  g->D           = variable_kind == SRRG2_SE2_RIGHT ? 3 : 6;
  g->tsize       = variable_kind == SRRG2_SE2_RIGHT ? 9 : 12;
  *out           = g;
  return 0;
}

static void pg_release(o_posegraph* g) {
  free(g->poses);
  free(g->fixed);
  free(g->ij);
  free(g->Z);
  free(g->omega);
  free(g->enabled);
  g->poses = NULL; g->fixed = NULL; g->ij = NULL; g->Z = NULL; g->omega = NULL; g->enabled = NULL;
}

int oracle_posegraph_destroy(o_posegraph* g) {
  if (!g) return 0;
  pg_release(g);
  free(g);
  return 0;
}

int oracle_posegraph_set_direct(o_posegraph* g, int enable) {
  if (!g) return SRRG2_E_INVALID;
  g->direct = enable;
  return 0;
}

int oracle_posegraph_set(o_posegraph* g, int V, const float* poses, const uint8_t* fixed_mask, int E, const int32_t* ij,
                         const float* Z, const float* omega, const uint8_t* enabled) {
  if (!g || V < 0 || E < 0 || (V > 0 && !poses) || (E > 0 && (!ij || !Z))) return pg_fail(SRRG2_E_INVALID, "set");
  for (int e = 0; e < E; ++e) {
    if (ij[2 * e] < 0 || ij[2 * e] >= V || ij[2 * e + 1] < 0 || ij[2 * e + 1] >= V || ij[2 * e] == ij[2 * e + 1])
      return pg_fail(SRRG2_E_INVALID, "set: bad edge endpoints");
  }
  pg_release(g);
  const int D = g->D, T = g->tsize;
  g->V = V;
  g->E = E;
  g->poses   = (float*) malloc(sizeof(float) * (size_t) (V > 0 ? V : 1) * T);
  g->fixed   = (uint8_t*) calloc((size_t) (V > 0 ? V : 1), 1);
  g->ij      = (int32_t*) malloc(sizeof(int32_t) * (size_t) (E > 0 ? E : 1) * 2);
  g->Z       = (float*) malloc(sizeof(float) * (size_t) (E > 0 ? E : 1) * T);
  g->omega   = (double*) malloc(sizeof(double) * (size_t) (E > 0 ? E : 1) * D * D);
  g->enabled = (uint8_t*) malloc((size_t) (E > 0 ? E : 1));
  memcpy(g->poses, poses, sizeof(float) * (size_t) V * T);
  if (fixed_mask) {
    for (int v = 0; v < V; ++v) g->fixed[v] = fixed_mask[v] ? 1 : 0;
  } else if (V > 0) {
    g->fixed[0] = 1; /* multi_graph_slam_impl.cpp:86 */
  }
  memcpy(g->ij, ij, sizeof(int32_t) * (size_t) E * 2);
  memcpy(g->Z, Z, sizeof(float) * (size_t) E * T);
  for (int e = 0; e < E; ++e) {
    for (int a = 0; a < D; ++a)
      for (int b = 0; b < D; ++b)
        g->omega[((size_t) e * D + a) * D + b] = omega ? (double) omega[((size_t) e * D + a) * D + b] : (a == b ? 1.0 : 0.0);
    g->enabled[e] = enabled ? (enabled[e] ? 1 : 0) : 1;
  }
  return 0;
}

int oracle_posegraph_set_enabled(o_posegraph* g, const uint8_t* enabled) {
  if (!g || !enabled) return SRRG2_E_INVALID;
  for (int e = 0; e < g->E; ++e) g->enabled[e] = enabled[e] ? 1 : 0;
  return 0;
}

int oracle_posegraph_get_poses(o_posegraph* g, float* out) {
  if (!g || !out) return SRRG2_E_INVALID;
  memcpy(out, g->poses, sizeof(float) * (size_t) g->V * g->tsize);
  return 0;
}

/* ---- factor: e, Ji, Jj of one edge ------------------------------------------------------------------------- */
static void edge_linearize(const o_posegraph* g, int e, double* err, double* Ji, double* Jj) {
  const int D = g->D, T = g->tsize;
  const float* Xi = g->poses + (size_t) g->ij[2 * e] * T;
  const float* Xj = g->poses + (size_t) g->ij[2 * e + 1] * T;
  const float* Z  = g->Z + (size_t) e * T;
  float Xi_inv[12], A[12], Zinv[12], Em[12];
  memset(Ji, 0, sizeof(double) * D * D);
  memset(Jj, 0, sizeof(double) * D * D);
  if (D == 6) {
    o_se3_inverse(Xi, Xi_inv);
    o_se3_compose(Xi_inv, Xj, A);
    o_se3_inverse(Z, Zinv);
    o_se3_compose(Zinv, A, Em);
    o_se3_t2v_quat(Em, err);
    double n2 = (err[3] * err[3] + err[4] * err[4]) + err[5] * err[5];
    double w  = n2 < 1.0 ? sqrt(1.0 - n2) : 0.0;
    /* Jj = d t2v(E v2t(d)) / d d = diag(E_R, w I + [v]x) */
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Jj[r * 6 + c] = (double) Em[r * 4 + c];
    Jj[3 * 6 + 3] = w;       Jj[3 * 6 + 4] = -err[5]; Jj[3 * 6 + 5] = err[4];
    Jj[4 * 6 + 3] = err[5];  Jj[4 * 6 + 4] = w;       Jj[4 * 6 + 5] = -err[3];
    Jj[5 * 6 + 3] = -err[4]; Jj[5 * 6 + 4] = err[3];  Jj[5 * 6 + 5] = w;
    /* perturbing Xi on the right acts on E's right as the conjugate by A = Xi^-1 Xj, negated:
     * M = [[R_A^T, -2 R_A^T [t_A]x], [0, R_A^T]],  Ji = -Jj M */
    double M[36];
    memset(M, 0, sizeof(M));
    double RAt[9], tA[3] = {(double) A[3], (double) A[7], (double) A[11]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) RAt[r * 3 + c] = (double) A[c * 4 + r];
    double tx[9] = {0, -tA[2], tA[1], tA[2], 0, -tA[0], -tA[1], tA[0], 0};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        M[r * 6 + c]           = RAt[r * 3 + c];
        M[(r + 3) * 6 + c + 3] = RAt[r * 3 + c];
        double s               = 0.0;
        for (int k = 0; k < 3; ++k) s = s + RAt[r * 3 + k] * tx[k * 3 + c];
        M[r * 6 + c + 3] = -2.0 * s;
      }
    }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s = s + Jj[r * 6 + k] * M[k * 6 + c];
        Ji[r * 6 + c] = -s;
      }
  } else {
    o_se2_inverse(Xi, Xi_inv);
    o_se2_compose(Xi_inv, Xj, A);
    o_se2_inverse(Z, Zinv);
    o_se2_compose(Zinv, A, Em);
    o_se2_t2v(Em, err);
    Jj[0] = (double) Em[0]; Jj[1] = (double) Em[1];
    Jj[3] = (double) Em[3]; Jj[4] = (double) Em[4];
    Jj[8] = 1.0;
    /* M = [[R_A^T, R_A^T S t_A], [0, 1]], S = [[0,-1],[1,0]] */
    double M[9];
    memset(M, 0, sizeof(M));
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

/* C = A^T W B (DxD) */
static void atwb(int D, const double* A, const double* W, const double* B, double* C) {
  double WB[36];
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

static void atwv(int D, const double* A, const double* W, const double* v, double* out) {
  double Wv[6];
  for (int r = 0; r < D; ++r) {
    double s = 0.0;
    for (int k = 0; k < D; ++k) s = s + W[r * D + k] * v[k];
    Wv[r] = s;
  }
  for (int r = 0; r < D; ++r) {
    double s = 0.0;
    for (int k = 0; k < D; ++k) s = s + A[k * D + r] * Wv[k];
    out[r] = s;
  }
}

/* inverse of an SPD DxD block through Cholesky; returns 1 if not PD */
static int spd_inverse(int D, const double* A, double* Ainv) {
  for (int c = 0; c < D; ++c) {
    double rhs[6], x[6];
    for (int r = 0; r < D; ++r) rhs[r] = r == c ? -1.0 : 0.0; /* o_solve solves A x = -rhs */
    if (o_solve(D, A, rhs, x)) return 1;
    for (int r = 0; r < D; ++r) Ainv[r * D + c] = x[r];
  }
  return 0;
}

typedef struct {
  int D, V, E;
  double* Hd; /* V x D x D */
  double* Ho; /* E x D x D = Ji^T W Jj */
  double* b;  /* V x D */
} pg_system;

static void spmv(const o_posegraph* g, const pg_system* S, const double* x, double* y) {
  const int D = S->D;
  for (int v = 0; v < S->V; ++v)
    for (int r = 0; r < D; ++r) {
      double s = 0.0;
      for (int c = 0; c < D; ++c) s = s + S->Hd[((size_t) v * D + r) * D + c] * x[(size_t) v * D + c];
      y[(size_t) v * D + r] = s;
    }
  for (int e = 0; e < S->E; ++e) {
    if (!g->enabled[e]) continue;
    int i = g->ij[2 * e], j = g->ij[2 * e + 1];
    if (g->fixed[i] || g->fixed[j]) continue;
    const double* B = S->Ho + (size_t) e * D * D;
    for (int r = 0; r < D; ++r) {
      double s = 0.0, t = 0.0;
      for (int c = 0; c < D; ++c) {
        s = s + B[r * D + c] * x[(size_t) j * D + c]; /* y_i += Hij x_j */
        t = t + B[c * D + r] * x[(size_t) i * D + c]; /* y_j += Hij^T x_i */
      }
      y[(size_t) i * D + r] = y[(size_t) i * D + r] + s;
      y[(size_t) j * D + r] = y[(size_t) j * D + r] + t;
    }
  }
}

static double dot(size_t n, const double* a, const double* b) {
  double s = 0.0;
  for (size_t k = 0; k < n; ++k) s = s + a[k] * b[k];
  return s;
}

static int dense_solve(const o_posegraph* g, const pg_system* S, double* x) {
  const int D = S->D;
  const size_t n = (size_t) S->V * D;
  double* A = (double*) calloc(n * n, sizeof(double));
  for (int v = 0; v < S->V; ++v)
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) A[((size_t) v * D + r) * n + (size_t) v * D + c] = S->Hd[((size_t) v * D + r) * D + c];
  for (int e = 0; e < S->E; ++e) {
    if (!g->enabled[e]) continue;
    int i = g->ij[2 * e], j = g->ij[2 * e + 1];
    if (g->fixed[i] || g->fixed[j]) continue;
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) {
        double v = S->Ho[((size_t) e * D + r) * D + c];
        A[((size_t) i * D + r) * n + (size_t) j * D + c] += v;
        A[((size_t) j * D + c) * n + (size_t) i * D + r] += v;
      }
  }
  /* in-place Cholesky A = L L^T, then solve L L^T x = -b */
  int bad = 0;
  for (size_t j = 0; j < n && !bad; ++j) {
    double s = A[j * n + j];
    for (size_t k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    if (!(s > 0.0)) { bad = 1; break; }
    double d     = sqrt(s);
    A[j * n + j] = d;
    for (size_t i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (size_t k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / d;
    }
  }
  if (!bad) {
    for (size_t i = 0; i < n; ++i) {
      double s = -S->b[i];
      for (size_t k = 0; k < i; ++k) s -= A[i * n + k] * x[k];
      x[i] = s / A[i * n + i];
    }
    for (size_t ii = n; ii-- > 0;) {
      double s = x[ii];
      for (size_t k = ii + 1; k < n; ++k) s -= A[k * n + ii] * x[k];
      x[ii] = s / A[ii * n + ii];
    }
  }
  free(A);
  return bad;
}

int oracle_posegraph_solve(o_posegraph* g, const srrg2_posegraph_params* p, srrg2_posegraph_stats* stats, int* n_inout) {
  if (!g || !p) return pg_fail(SRRG2_E_INVALID, "solve");
  const int D = g->D, V = g->V, E = g->E, T = g->tsize;
  const size_t n = (size_t) V * D;
  pg_system S;
  S.D  = D; S.V = V; S.E = E;
  S.Hd = (double*) malloc(sizeof(double) * (size_t) (V > 0 ? V : 1) * D * D);
  S.Ho = (double*) malloc(sizeof(double) * (size_t) (E > 0 ? E : 1) * D * D);
  S.b  = (double*) malloc(sizeof(double) * (n > 0 ? n : 1));
  double* Minv = (double*) malloc(sizeof(double) * (size_t) (V > 0 ? V : 1) * D * D);
  double* x  = (double*) malloc(sizeof(double) * (n > 0 ? n : 1));
  double* r  = (double*) malloc(sizeof(double) * (n > 0 ? n : 1));
  double* z  = (double*) malloc(sizeof(double) * (n > 0 ? n : 1));
  double* pp = (double*) malloc(sizeof(double) * (n > 0 ? n : 1));
  double* Ap = (double*) malloc(sizeof(double) * (n > 0 ? n : 1));
  int nstats = 0;
  for (int it = 0; it < p->max_iterations; ++it) {
    srrg2_posegraph_stats st;
    memset(&st, 0, sizeof(st));
    st.iteration = it;
    memset(S.Hd, 0, sizeof(double) * (size_t) V * D * D);
    memset(S.Ho, 0, sizeof(double) * (size_t) E * D * D);
    memset(S.b, 0, sizeof(double) * n);
    double chi = 0.0;
    for (int e = 0; e < E; ++e) {
      if (!g->enabled[e]) continue; /* disabled factors are skipped, loop_closure.h:71 */
      st.num_factors++;
      double err[6], Ji[36], Jj[36], blk[36], v6[6];
      edge_linearize(g, e, err, Ji, Jj);
      const double* W = g->omega + (size_t) e * D * D;
      int i = g->ij[2 * e], j = g->ij[2 * e + 1];
      double We[6];
      for (int a = 0; a < D; ++a) {
        double s = 0.0;
        for (int k = 0; k < D; ++k) s = s + W[a * D + k] * err[k];
        We[a] = s;
      }
      for (int a = 0; a < D; ++a) chi = chi + err[a] * We[a];
      atwb(D, Ji, W, Ji, blk);
      for (int k = 0; k < D * D; ++k) S.Hd[(size_t) i * D * D + k] += blk[k];
      atwb(D, Jj, W, Jj, blk);
      for (int k = 0; k < D * D; ++k) S.Hd[(size_t) j * D * D + k] += blk[k];
      atwb(D, Ji, W, Jj, S.Ho + (size_t) e * D * D);
      atwv(D, Ji, W, err, v6);
      for (int k = 0; k < D; ++k) S.b[(size_t) i * D + k] += v6[k];
      atwv(D, Jj, W, err, v6);
      for (int k = 0; k < D; ++k) S.b[(size_t) j * D + k] += v6[k];
    }
    st.chi = (float) chi;
    /* fixed variables are removed from the system: identity row, zero rhs (multi_graph_slam_impl.cpp:86) */
    int bad = 0;
    for (int v = 0; v < V; ++v) {
      double* Hv = S.Hd + (size_t) v * D * D;
      if (g->fixed[v]) {
        for (int a = 0; a < D; ++a) {
          for (int c = 0; c < D; ++c) Hv[a * D + c] = a == c ? 1.0 : 0.0;
          S.b[(size_t) v * D + a] = 0.0;
        }
      } else {
        for (int a = 0; a < D; ++a) Hv[a * D + a] += (double) p->damping;
      }
      if (spd_inverse(D, Hv, Minv + (size_t) v * D * D)) bad = 1;
    }
    memset(x, 0, sizeof(double) * n);
    if (!bad) {
      if (g->direct) {
        bad = dense_solve(g, &S, x);
      } else {
        /* block-Jacobi preconditioned CG on H x = -b */
        const double bnorm = sqrt(dot(n, S.b, S.b));
        for (size_t k = 0; k < n; ++k) r[k] = -S.b[k];
        double rz = 0.0;
        for (int v = 0; v < V; ++v)
          for (int a = 0; a < D; ++a) {
            double s = 0.0;
            for (int c = 0; c < D; ++c) s = s + Minv[((size_t) v * D + a) * D + c] * r[(size_t) v * D + c];
            z[(size_t) v * D + a] = s;
          }
        memcpy(pp, z, sizeof(double) * n);
        rz = dot(n, r, z);
        double rnorm = bnorm;
        int k = 0;
        if (bnorm > 0.0) {
          for (k = 0; k < p->pcg_max_iterations; ++k) {
            spmv(g, &S, pp, Ap);
            double pAp = dot(n, pp, Ap);
            if (!(pAp > 0.0)) break;
            double alpha = rz / pAp;
            for (size_t q = 0; q < n; ++q) {
              x[q] = x[q] + alpha * pp[q];
              r[q] = r[q] - alpha * Ap[q];
            }
            rnorm = sqrt(dot(n, r, r));
            if (rnorm <= (double) p->pcg_tolerance * bnorm) {
              ++k;
              break;
            }
            for (int v = 0; v < V; ++v)
              for (int a = 0; a < D; ++a) {
                double s = 0.0;
                for (int c = 0; c < D; ++c) s = s + Minv[((size_t) v * D + a) * D + c] * r[(size_t) v * D + c];
                z[(size_t) v * D + a] = s;
              }
            double rz_new = dot(n, r, z);
            double beta   = rz_new / rz;
            rz            = rz_new;
            for (size_t q = 0; q < n; ++q) pp[q] = z[q] + beta * pp[q];
          }
        }
        st.pcg_iterations = k;
        st.pcg_residual   = bnorm > 0.0 ? (float) (rnorm / bnorm) : 0.f;
      }
    }
    st.solver_status = bad ? 1 : 0;
    if (!bad) {
      for (int v = 0; v < V; ++v) {
        if (g->fixed[v]) continue;
        o_box_plus(g->kind, g->poses + (size_t) v * T, x + (size_t) v * D);
      }
    }
    if (stats && n_inout && nstats < *n_inout) stats[nstats] = st;
    ++nstats;
    if (bad) break;
  }
  if (n_inout) *n_inout = nstats;
  free(S.Hd); free(S.Ho); free(S.b); free(Minv); free(x); free(r); free(z); free(pp); free(Ap);
  return 0;
}

/* total chi2 of the enabled factors at the current poses (for tests) */
double oracle_posegraph_chi(o_posegraph* g) {
  double chi = 0.0;
  const int D = g->D;
  for (int e = 0; e < g->E; ++e) {
    if (!g->enabled[e]) continue;
    double err[6], Ji[36], Jj[36];
    edge_linearize(g, e, err, Ji, Jj);
    const double* W = g->omega + (size_t) e * D * D;
    for (int a = 0; a < D; ++a)
      for (int b = 0; b < D; ++b) chi += err[a] * W[a * D + b] * err[b];
  }
  return chi;
}

/* e, Ji, Jj of one edge (for the finite-difference tests) */
int oracle_posegraph_edge(o_posegraph* g, int e, double* err, double* Ji, double* Jj) {
  if (!g || e < 0 || e >= g->E) return SRRG2_E_INVALID;
  edge_linearize(g, e, err, Ji, Jj);
  return 0;
}
