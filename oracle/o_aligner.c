/*
 * o_aligner.c -- CPU oracle of the multi-cue aligner.  TEST INFRASTRUCTURE ONLY (oracle.h).
 *
 * Control flow restated from the reference (S/ = srrg2_slam_interfaces/src/srrg2_slam_interfaces/):
 *   compute()                       S/registration/aligners/multi_aligner_impl.cpp:47-95
 *   _runSolver()                    multi_aligner_impl.cpp:98-128
 *   _computeCorrespondencesPerSlices S/registration/aligners/multi_aligner.h:126-138
 *   _preCompute()/_postCompute()    multi_aligner_impl.cpp:131-141,163-181
 *   _setClampRobustifiers/_restore  multi_aligner_impl.cpp:184-211
 *   _pruneCorrespondences()         multi_aligner_impl.cpp:214-263
 *   numCorrespondences()            multi_aligner_impl.cpp:275-285
 *   slice setMovingInFixed/computeCorrespondences/correspondencesGood
 *                                   S/registration/aligners/aligner_slice_processor_impl.cpp:20-48,77-79
 *   prior slices                    S/registration/aligners/aligner_slice_processor_prior.h:41-97,
 *                                   aligner_slice_odometry_prior.cpp:6-37, aligner_slice_motion_model.hpp:44-79
 *   termination criterion           S/registration/aligners/aligner_termination_criteria_impl.cpp:10-65
 *
 * PARITY UNPINNED for the arithmetic below the slice interface (finder, factors, robustifiers,
 * 6x6 solve): those live in srrg2_solver / srrg2_core / downstream packages that are not under
 * /root/reference.  They are defined from first principles in DESIGN.md
 * ("arithmetic specification") and implemented here in the stated operation order.
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ACC_N 32
#define ACC_CHI_IN 27
#define ACC_CHI_OUT 28
#define ACC_N_IN 29
#define ACC_N_OUT 30
#define ACC_N_CORR 31
#define MAX_TERM_WINDOW 64

static char g_err[256] = "";
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
const char* oracle_last_error(void) {
  return g_err;
}
void oracle_set_error(const char* msg) { /* shared error slot (o_scene.c) */
  snprintf(g_err, sizeof(g_err), "%s", msg);
}
int oracle_abi_version(void) {
  return SRRG2_AMD_ABI_VERSION;
}

/* index of H(a,b), a<=b, in the 21-entry upper-triangular layout of a 6x6 */
static inline int hidx(int a, int b) {
  return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

/* ---- search grid ------------------------------------------------------------ */
typedef struct {
  int dim;
  float o[3];
  float h, inv_h;
  int n[3];
  int* cell_start; /* ncell + 1 */
  int* order;      /* sorted position -> original fixed index */
  float* pts;      /* sorted coordinates, dim floats per point */
  int npts;
  int rmax;
  float gate2;
} o_grid;

static void grid_free(o_grid* g) {
  free(g->cell_start);
  free(g->order);
  free(g->pts);
  memset(g, 0, sizeof(*g));
}

static inline float bound2_of(int r, float h) {
  float b = ((float) r - 0.01f) * h;
  return (b * b) * 0.9999f;
}

static inline int cell_coord(float x, float o, float inv_h) {
  float u = (x - o) * inv_h;
  u       = fminf(fmaxf(u, -2048.f), 4096.f);
  return (int) floorf(u);
}

static int point_finite(const float* p, int dim) {
  for (int d = 0; d < dim; ++d) {
    if (!isfinite(p[d])) {
      return 0;
    }
  }
  return 1;
}

static void grid_build(o_grid* g, int dim, const float* pts, int n, float gate, float cell_size) {
  grid_free(g);
  g->dim   = dim;
  g->gate2 = gate * gate;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  int first = 1;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + (size_t) i * dim;
    if (!point_finite(p, dim)) {
      continue;
    }
    for (int d = 0; d < dim; ++d) {
      if (first || p[d] < mn[d]) mn[d] = p[d];
      if (first || p[d] > mx[d]) mx[d] = p[d];
    }
    first = 0;
  }
  float h = cell_size > 0.f ? cell_size : gate * 0.25f;
  if (!(h > 0.f)) {
    h = 1.f;
  }
  for (;;) {
    double cells = 1.0;
    int ok       = 1;
    for (int d = 0; d < dim; ++d) {
      double nd = floor(((double) mx[d] - (double) mn[d]) / (double) h) + 1.0;
      if (nd > 1024.0) ok = 0;
      cells *= nd;
    }
    if (ok && cells <= 16777216.0) {
      break;
    }
    h *= 2.f;
  }
  g->h     = h;
  g->inv_h = 1.0f / h;
  for (int d = 0; d < 3; ++d) {
    g->o[d] = d < dim ? mn[d] : 0.f;
    g->n[d] = 1;
  }
  for (int d = 0; d < dim; ++d) {
    int c   = first ? 0 : cell_coord(mx[d], g->o[d], g->inv_h);
    g->n[d] = c + 1;
  }
  int ncell     = g->n[0] * g->n[1] * g->n[2];
  g->cell_start = (int*) calloc((size_t) ncell + 1, sizeof(int));
  int* cell_of  = (int*) malloc(sizeof(int) * (size_t) (n > 0 ? n : 1));
  int valid     = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + (size_t) i * dim;
    if (!point_finite(p, dim)) {
      cell_of[i] = -1;
      continue;
    }
    int c[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
      c[d] = cell_coord(p[d], g->o[d], g->inv_h);
      if (c[d] < 0) c[d] = 0;
      if (c[d] >= g->n[d]) c[d] = g->n[d] - 1;
    }
    int ci     = (c[2] * g->n[1] + c[1]) * g->n[0] + c[0];
    cell_of[i] = ci;
    g->cell_start[ci + 1]++;
    ++valid;
  }
  for (int c = 0; c < ncell; ++c) {
    g->cell_start[c + 1] += g->cell_start[c];
  }
  g->npts     = valid;
  g->order    = (int*) malloc(sizeof(int) * (size_t) (valid > 0 ? valid : 1));
  g->pts      = (float*) malloc(sizeof(float) * (size_t) (valid > 0 ? valid : 1) * dim);
  int* cursor = (int*) malloc(sizeof(int) * (size_t) (ncell > 0 ? ncell : 1));
  memcpy(cursor, g->cell_start, sizeof(int) * (size_t) ncell);
  for (int i = 0; i < n; ++i) {
    if (cell_of[i] < 0) continue;
    int pos       = cursor[cell_of[i]]++;
    g->order[pos] = i;
    memcpy(g->pts + (size_t) pos * dim, pts + (size_t) i * dim, sizeof(float) * dim);
  }
  free(cursor);
  free(cell_of);
  int r = 1;
  while (bound2_of(r, h) < g->gate2 && r < 4096) {
    ++r;
  }
  g->rmax = r;
}

/* squared distance in the specified operation order */
static inline float dist2(const float* f, const float* q, int dim) {
  float dx = f[0] - q[0];
  float dy = f[1] - q[1];
  float d  = dx * dx + dy * dy;
  if (dim == 3) {
    float dz = f[2] - q[2];
    d        = d + dz * dz;
  }
  return d;
}

static void grid_scan_cube(const o_grid* g, const float* q, const int* c, int r, float* best_d2, int* best_idx) {
  int z0 = g->dim == 3 ? c[2] - r : 0, z1 = g->dim == 3 ? c[2] + r : 0;
  if (z0 < 0) z0 = 0;
  if (z1 >= g->n[2]) z1 = g->n[2] - 1;
  int y0 = c[1] - r, y1 = c[1] + r;
  if (y0 < 0) y0 = 0;
  if (y1 >= g->n[1]) y1 = g->n[1] - 1;
  int x0 = c[0] - r, x1 = c[0] + r;
  if (x0 < 0) x0 = 0;
  if (x1 >= g->n[0]) x1 = g->n[0] - 1;
  if (x0 > x1) return;
  for (int z = z0; z <= z1; ++z) {
    for (int y = y0; y <= y1; ++y) {
      int row = (z * g->n[1] + y) * g->n[0];
      int s = g->cell_start[row + x0], e = g->cell_start[row + x1 + 1];
      for (int j = s; j < e; ++j) {
        float d2 = dist2(g->pts + (size_t) j * g->dim, q, g->dim);
        int idx  = g->order[j];
        if (d2 < *best_d2 || (d2 == *best_d2 && idx < *best_idx)) {
          *best_d2  = d2;
          *best_idx = idx;
        }
      }
    }
  }
}

/* exact gated nearest neighbour: returns the fixed index or -1 */
static int grid_query(const o_grid* g, const float* q, float* d2_out) {
  int c[3] = {0, 0, 0};
  for (int d = 0; d < g->dim; ++d) {
    c[d] = cell_coord(q[d], g->o[d], g->inv_h);
  }
  float best_d2 = INFINITY;
  int best_idx  = 0x7fffffff;
  grid_scan_cube(g, q, c, 1, &best_d2, &best_idx);
  int found = best_idx != 0x7fffffff && best_d2 <= g->gate2;
  if (!(found && best_d2 <= bound2_of(1, g->h)) && g->rmax > 1) {
    int r2 = g->rmax;
    if (found) {
      r2 = 1;
      while (r2 < g->rmax && bound2_of(r2, g->h) < best_d2) {
        ++r2;
      }
    }
    if (r2 > 1) {
      grid_scan_cube(g, q, c, r2, &best_d2, &best_idx);
    }
  }
  if (best_idx == 0x7fffffff || !(best_d2 <= g->gate2)) {
    return -1;
  }
  *d2_out = best_d2;
  return best_idx;
}

static int brute_query(const float* fixed, int nf, int dim, const float* q, float gate2, float* d2_out) {
  float best_d2 = INFINITY;
  int best_idx  = -1;
  for (int j = 0; j < nf; ++j) {
    const float* f = fixed + (size_t) j * dim;
    if (!point_finite(f, dim)) continue;
    float d2 = dist2(f, q, dim);
    if (d2 < best_d2) { /* ascending j: strict '<' keeps the smallest index on ties */
      best_d2  = d2;
      best_idx = j;
    }
  }
  if (best_idx < 0 || !(best_d2 <= gate2)) {
    return -1;
  }
  *d2_out = best_d2;
  return best_idx;
}

/* ---- slices ------------------------------------------------------------------- */
typedef struct {
  srrg2_slice_config cfg;
  int robust_kind; /* currently bound robustifier (bindRobustifier) */
  float robust_thr;
  float* fixed;
  float* fixed_n;
  int nf;
  float* moving;
  float* moving_n;
  int nm;
  float pinf; /* max |coordinate| of the finite moving points */
  float finf; /* max |coordinate| of the finite fixed points (given-correspondences slices) */
  srrg2_correspondence* given; /* SRRG2_FINDER_CORRESPONDENCES: the locked correspondences */
  int ngiven;
  float ninf; /* max |component| of the fixed normals */
  o_grid grid;
  int grid_valid;
  srrg2_correspondence* corr; /* owned by the slice, aligner_slice_processor.h:156 */
  uint8_t* fstat;
  int ncorr;
  int corr_cap;
  float prior_Z[12];
  int has_prior;
  int64_t acc[ACC_N];
  int k;
  double prior_H[36], prior_b[6], prior_chi;
  int prior_status;
  int shares; /* 1 + index of the slice whose clouds this one reads (srrg2_aligner_share_clouds); 0: its own */
} o_slice;

typedef struct {
  int count, window;
  double buf[MAX_TERM_WINDOW];
} o_window;

struct o_aligner {
  int kind, dim, dof, tsize;
  srrg2_aligner_params params;
  int has_term;
  srrg2_termination_params term;
  o_window w_corr, w_inl, w_out, w_chi;
  o_slice slices[SRRG2_MAX_SLICES];
  int nslices;
  float X[12];
  int status;
  srrg2_iteration_stats* stats;
  int nstats, stats_cap;
  int bruteforce;
  double last_H[36], last_b[6], last_dx[6];
};

static void identity(int kind, float* T) {
  if (kind == SRRG2_SE2_RIGHT) {
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(T, I, sizeof(I));
  } else {
    const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(T, I, sizeof(I));
  }
}

void srrg2_oracle_default_slice(srrg2_slice_config* c, int kind) {
  memset(c, 0, sizeof(*c));
  c->kind                      = SRRG2_SLICE_P2P;
  c->finder                    = SRRG2_FINDER_NN_GATED;
  c->robustifier               = SRRG2_ROBUST_NONE;
  c->robustifier_chi_threshold = 1.f;
  c->min_num_correspondences   = 0;
  c->finder_max_distance       = 1.f;
  c->finder_normal_cos         = -2.f;
  identity(kind, c->sensor_in_robot);
  for (int i = 0; i < 6; ++i) c->prior_information_diag[i] = kind == SRRG2_SE2_RIGHT ? 100.f : 1.f;
  c->prior_sets_initial_guess = 1;
}

int oracle_aligner_create(int variable_kind, o_aligner** out) {
  if (!out || variable_kind < 0 || variable_kind > 2) return fail(SRRG2_E_INVALID, "create: bad variable kind");
  o_aligner* a = (o_aligner*) calloc(1, sizeof(o_aligner));
  a->kind      = variable_kind;
  a->dim       = variable_kind == SRRG2_SE2_RIGHT ? 2 : 3;
  a->dof       = variable_kind == SRRG2_SE2_RIGHT ? 3 : 6;
  a->tsize     = variable_kind == SRRG2_SE2_RIGHT ? 9 : 12;
  a->params.max_iterations  = 10; /* aligner.h:30 */
  a->params.min_num_inliers = 10; /* multi_aligner.h:45 */
  identity(variable_kind, a->X);
  a->status = SRRG2_FAIL; /* aligner.h:56 */
  *out      = a;
  return 0;
}

static void slice_free(o_slice* s) {
  free(s->fixed);
  free(s->fixed_n);
  free(s->moving);
  free(s->moving_n);
  free(s->corr);
  free(s->fstat);
  free(s->given);
  grid_free(&s->grid);
  memset(s, 0, sizeof(*s));
}

int oracle_aligner_destroy(o_aligner* h) {
  if (!h) return 0;
  for (int i = 0; i < h->nslices; ++i) slice_free(&h->slices[i]);
  free(h->stats);
  free(h);
  return 0;
}

int oracle_aligner_set_params(o_aligner* h, const srrg2_aligner_params* p) {
  if (!h || !p || p->max_iterations < 0) return fail(SRRG2_E_INVALID, "set_params");
  h->params = *p;
  return 0;
}

int oracle_aligner_set_termination(o_aligner* h, const srrg2_termination_params* p) {
  if (!h) return fail(SRRG2_E_INVALID, "set_termination");
  if (!p) {
    h->has_term = 0;
    return 0;
  }
  if (p->window_size < 1 || p->window_size > MAX_TERM_WINDOW) return fail(SRRG2_E_INVALID, "window_size");
  h->has_term = 1;
  h->term     = *p;
  return 0;
}

int oracle_aligner_add_slice(o_aligner* h, const srrg2_slice_config* c, int* idx) {
  if (!h || !c) return fail(SRRG2_E_INVALID, "add_slice");
  if (h->nslices >= SRRG2_MAX_SLICES) return fail(SRRG2_E_INVALID, "too many slices");
  if (c->kind != SRRG2_SLICE_PRIOR && c->finder != SRRG2_FINDER_NN_GATED && c->finder != SRRG2_FINDER_PROJECTIVE &&
      c->finder != SRRG2_FINDER_CORRESPONDENCES)
    return fail(SRRG2_E_INVALID, "cue slice needs a finder"); /* aligner_slice_processor_impl.cpp:13-16 */
  if (c->finder == SRRG2_FINDER_CORRESPONDENCES && c->kind != SRRG2_SLICE_P2P && c->kind != SRRG2_SLICE_P2PLANE)
    return fail(SRRG2_E_INVALID, "given correspondences drive point-to-point / point-to-plane factors");
  if (c->finder == SRRG2_FINDER_PROJECTIVE || c->kind == SRRG2_SLICE_REPROJECTION) {
    if (h->dim != 3) return fail(SRRG2_E_UNSUPPORTED, "projective finder / reprojection factor are SE(3) only");
    if (c->finder != SRRG2_FINDER_PROJECTIVE) return fail(SRRG2_E_INVALID, "reprojection slice needs the projective finder");
    if (c->image_rows <= 0 || c->image_cols <= 0 || !(c->depth_min > 0.f) || !(c->depth_max >= c->depth_min) ||
        !(c->camera_matrix[0] > 0.f) || !(c->camera_matrix[4] > 0.f))
      return fail(SRRG2_E_INVALID, "projective slice: bad camera / image / depth range");
  }
  o_slice* s = &h->slices[h->nslices];
  memset(s, 0, sizeof(*s));
  s->cfg         = *c;
  s->robust_kind = c->robustifier;
  s->robust_thr  = c->robustifier_chi_threshold;
  if (idx) *idx = h->nslices;
  h->nslices++;
  return 0;
}

int oracle_aligner_clear_slices(o_aligner* h) {
  if (!h) return fail(SRRG2_E_INVALID, "clear_slices");
  for (int i = 0; i < h->nslices; ++i) slice_free(&h->slices[i]);
  h->nslices = 0;
  return 0;
}

int oracle_aligner_set_robustifier(o_aligner* h, int si, int kind, float thr) {
  if (!h || si < 0 || si >= h->nslices) return fail(SRRG2_E_INVALID, "set_robustifier");
  h->slices[si].cfg.robustifier               = kind;
  h->slices[si].cfg.robustifier_chi_threshold = thr;
  return 0;
}

static float* gather(const float* src, int stride_bytes, int n, int dim) {
  float* dst = (float*) malloc(sizeof(float) * (size_t) (n > 0 ? n : 1) * dim);
  for (int i = 0; i < n; ++i) {
    const float* p = (const float*) ((const char*) src + (size_t) i * stride_bytes);
    for (int d = 0; d < dim; ++d) dst[(size_t) i * dim + d] = p[d];
  }
  return dst;
}

static float max_abs_finite(const float* v, int n, int dim, int per_point) {
  float m = 0.f;
  for (int i = 0; i < n; ++i) {
    const float* p = v + (size_t) i * dim;
    if (per_point && !point_finite(p, dim)) continue;
    for (int d = 0; d < dim; ++d) {
      float a = fabsf(p[d]);
      if (isfinite(a) && a > m) m = a;
    }
  }
  return m;
}

static int set_fixed_one(o_aligner* h, int si, const float* coords, int cs, const float* normals, int ns, int n);
static int set_moving_one(o_aligner* h, int si, const float* coords, int cs, const float* normals, int ns, int n);

/* srrg2_aligner_share_clouds: in the reference two slices with the same fixed_slice_name / moving_slice_name bind to the same
 * cloud objects of the scene (aligner_slice_processor_base_impl.cpp:27-50).  The oracle stays literal: every slice keeps
 * its OWN finder and its own copy of the clouds (the reference's slices each run their finder); sharing only means that
 * what is set on the source is set on the sharing slices too. */
int oracle_aligner_share_clouds(o_aligner* h, int si, int source) {
  if (!h || si < 0 || si >= h->nslices) return fail(SRRG2_E_INVALID, "share_clouds");
  o_slice* s = &h->slices[si];
  if (source == -1) {
    s->shares = 0;
    return 0;
  }
  if (source < 0 || source >= h->nslices || source == si) return fail(SRRG2_E_INVALID, "share_clouds: two different cue slices");
  o_slice* o = &h->slices[source];
  if (s->cfg.kind == SRRG2_SLICE_PRIOR || o->cfg.kind == SRRG2_SLICE_PRIOR || o->shares)
    return fail(SRRG2_E_INVALID, "share_clouds: bad source");
  if (s->cfg.finder != SRRG2_FINDER_PROJECTIVE || o->cfg.finder != SRRG2_FINDER_PROJECTIVE)
    return fail(SRRG2_E_UNSUPPORTED, "share_clouds: both slices must use the projective finder");
  s->shares = source + 1;
  if (o->fixed) set_fixed_one(h, si, o->fixed, h->dim * 4, o->fixed_n, h->dim * 4, o->nf);
  if (o->moving) set_moving_one(h, si, o->moving, h->dim * 4, o->moving_n, h->dim * 4, o->nm);
  return 0;
}

int oracle_aligner_set_fixed(o_aligner* h, int si, const float* coords, int cs, const float* normals, int ns, int n,
                             int mem) {
  if (!h || si < 0 || si >= h->nslices || n < 0 || (n > 0 && !coords)) return fail(SRRG2_E_INVALID, "set_fixed");
  if (mem != SRRG2_MEM_HOST) return fail(SRRG2_E_UNSUPPORTED, "oracle takes host memory only");
  if (h->slices[si].cfg.kind == SRRG2_SLICE_PRIOR) return fail(SRRG2_E_INVALID, "set_fixed on a prior slice");
  if (h->slices[si].shares) return fail(SRRG2_E_STATE, "set_fixed: this slice shares the clouds of another slice");
  int rc = set_fixed_one(h, si, coords, cs, normals, ns, n);
  for (int t = 0; t < h->nslices && !rc; ++t)
    if (h->slices[t].shares == si + 1) rc = set_fixed_one(h, t, coords, cs, normals, ns, n);
  return rc;
}

static int set_fixed_one(o_aligner* h, int si, const float* coords, int cs, const float* normals, int ns, int n) {
  o_slice* s = &h->slices[si];
  free(s->fixed);
  free(s->fixed_n);
  s->fixed   = gather(coords, cs, n, h->dim);
  s->fixed_n = normals ? gather(normals, ns, n, h->dim) : NULL;
  s->nf      = n;
  s->ninf    = s->fixed_n ? max_abs_finite(s->fixed_n, n, h->dim, 0) : 0.f;
  s->finf    = max_abs_finite(s->fixed, n, h->dim, 1);
  s->grid_valid = 0; /* _fixed_changed_flag, correspondence_finder.h:80-83 */
  return 0;
}

int oracle_aligner_set_moving(o_aligner* h, int si, const float* coords, int cs, const float* normals, int ns, int n,
                              int mem) {
  if (!h || si < 0 || si >= h->nslices || n < 0 || (n > 0 && !coords)) return fail(SRRG2_E_INVALID, "set_moving");
  if (mem != SRRG2_MEM_HOST) return fail(SRRG2_E_UNSUPPORTED, "oracle takes host memory only");
  if (h->slices[si].cfg.kind == SRRG2_SLICE_PRIOR) return fail(SRRG2_E_INVALID, "set_moving on a prior slice");
  if (h->slices[si].shares) return fail(SRRG2_E_STATE, "set_moving: this slice shares the clouds of another slice");
  int rc = set_moving_one(h, si, coords, cs, normals, ns, n);
  for (int t = 0; t < h->nslices && !rc; ++t)
    if (h->slices[t].shares == si + 1) rc = set_moving_one(h, t, coords, cs, normals, ns, n);
  return rc;
}

static int set_moving_one(o_aligner* h, int si, const float* coords, int cs, const float* normals, int ns, int n) {
  o_slice* s = &h->slices[si];
  free(s->moving);
  free(s->moving_n);
  s->moving   = gather(coords, cs, n, h->dim);
  s->moving_n = normals ? gather(normals, ns, n, h->dim) : NULL;
  s->nm       = n;
  s->pinf     = max_abs_finite(s->moving, n, h->dim, 1);
  if (n > s->corr_cap) {
    free(s->corr);
    free(s->fstat);
    s->corr     = (srrg2_correspondence*) malloc(sizeof(srrg2_correspondence) * (size_t) n);
    s->fstat    = (uint8_t*) malloc((size_t) n);
    s->corr_cap = n;
  }
  s->ncorr = 0;
  return 0;
}

/* slice->setSensorInRobot with the transform looked up on every setMovingInFixed, aligner_slice_processor_impl.cpp:20-36 */
int oracle_aligner_set_sensor_in_robot(o_aligner* h, int si, const float* T) {
  if (!h || !T || si < 0 || si >= h->nslices) return fail(SRRG2_E_INVALID, "set_sensor_in_robot");
  for (int i = 0; i < h->tsize; ++i)
    if (!isfinite(T[i])) return fail(SRRG2_E_INVALID, "set_sensor_in_robot: non-finite transform");
  memcpy(h->slices[si].cfg.sensor_in_robot, T, sizeof(float) * h->tsize);
  return 0;
}

int oracle_aligner_set_prior_measurement(o_aligner* h, int si, const float* T) {
  if (!h || !T || si < 0 || si >= h->nslices) return fail(SRRG2_E_INVALID, "set_prior_measurement");
  o_slice* s = &h->slices[si];
  if (s->cfg.kind != SRRG2_SLICE_PRIOR) return fail(SRRG2_E_INVALID, "not a prior slice");
  memcpy(s->prior_Z, T, sizeof(float) * h->tsize);
  s->has_prior = 1;
  return 0;
}

int oracle_aligner_set_moving_in_fixed(o_aligner* h, const float* T) {
  if (!h || !T) return fail(SRRG2_E_INVALID, "set_moving_in_fixed");
  memcpy(h->X, T, sizeof(float) * h->tsize);
  return 0;
}

int oracle_aligner_get_moving_in_fixed(o_aligner* h, float* T) {
  if (!h || !T) return fail(SRRG2_E_INVALID, "get_moving_in_fixed");
  memcpy(T, h->X, sizeof(float) * h->tsize);
  return 0;
}

int oracle_aligner_set_bruteforce(o_aligner* h, int e) {
  if (!h) return fail(SRRG2_E_INVALID, "set_bruteforce");
  h->bruteforce = e;
  return 0;
}

/* ---- finder: AlignerSliceProcessor_::setMovingInFixed + computeCorrespondences -------- */
static void finder_transform(const o_aligner* a, const o_slice* s, float* T) {
  /* finder->setLocalMapInSensor(robot_in_sensor * X), aligner_slice_processor_impl.cpp:35 */
  float Sinv[12];
  if (a->dim == 3) {
    o_se3_inverse(s->cfg.sensor_in_robot, Sinv);
    o_se3_compose(Sinv, a->X, T);
  } else {
    o_se2_inverse(s->cfg.sensor_in_robot, Sinv);
    o_se2_compose(Sinv, a->X, T);
  }
}

static inline void xform3(const float* T, const float* p, float* q) {
  q[0] = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3];
  q[1] = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7];
  q[2] = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11];
}
static inline void rot3(const float* T, const float* p, float* q) {
  q[0] = (T[0] * p[0] + T[1] * p[1]) + T[2] * p[2];
  q[1] = (T[4] * p[0] + T[5] * p[1]) + T[6] * p[2];
  q[2] = (T[8] * p[0] + T[9] * p[1]) + T[10] * p[2];
}
static inline void xform2(const float* T, const float* p, float* q) {
  q[0] = (T[0] * p[0] + T[1] * p[1]) + T[2];
  q[1] = (T[3] * p[0] + T[4] * p[1]) + T[5];
}
static inline void rot2(const float* T, const float* p, float* q) {
  q[0] = T[0] * p[0] + T[1] * p[1];
  q[1] = T[3] * p[0] + T[4] * p[1];
}

static int slice_compute_correspondences_projective(o_aligner* a, o_slice* s);

static int slice_compute_correspondences(o_aligner* a, o_slice* s) {
  const int dim = a->dim;
  if (s->cfg.finder == SRRG2_FINDER_PROJECTIVE) return slice_compute_correspondences_projective(a, s);
  if (!s->fixed || !s->moving) return fail(SRRG2_E_STATE, "cue slice without fixed/moving cloud");
  if (s->cfg.finder == SRRG2_FINDER_CORRESPONDENCES) {
    /* "we keep the correspondences locked during optimization", multi_loop_detector_hbst_impl.cpp:343 */
    if (!s->given) return fail(SRRG2_E_STATE, "given-correspondences slice without correspondences");
    if (s->ngiven > s->corr_cap) {
      free(s->corr);
      free(s->fstat);
      s->corr     = (srrg2_correspondence*) malloc(sizeof(srrg2_correspondence) * (size_t) s->ngiven);
      s->fstat    = (uint8_t*) malloc((size_t) s->ngiven);
      s->corr_cap = s->ngiven;
    }
    for (int c = 0; c < s->ngiven; ++c) {
      if (s->given[c].fixed_idx < 0 || s->given[c].fixed_idx >= s->nf || s->given[c].moving_idx < 0 ||
          s->given[c].moving_idx >= s->nm)
        return fail(SRRG2_E_INVALID, "correspondence index out of range");
      s->corr[c] = s->given[c];
    }
    s->ncorr = s->ngiven;
    return 0;
  }
  float gate  = s->cfg.finder_max_distance;
  float gate2 = gate * gate;
  if (!a->bruteforce && !s->grid_valid) {
    grid_build(&s->grid, dim, s->fixed, s->nf, gate, s->cfg.finder_cell_size);
    s->grid_valid = 1;
  }
  float T[12];
  finder_transform(a, s, T);
  int use_ncos = s->cfg.finder_normal_cos > -1.f && s->fixed_n && s->moving_n;
  int nc       = 0;
  for (int i = 0; i < s->nm; ++i) {
    const float* p = s->moving + (size_t) i * dim;
    if (!point_finite(p, dim)) continue;
    float q[3] = {0, 0, 0};
    if (dim == 3)
      xform3(T, p, q);
    else
      xform2(T, p, q);
    float d2 = 0.f;
    int j    = a->bruteforce ? brute_query(s->fixed, s->nf, dim, q, gate2, &d2) : grid_query(&s->grid, q, &d2);
    if (j < 0) continue;
    if (use_ncos) {
      const float* nm = s->moving_n + (size_t) i * dim;
      const float* nf = s->fixed_n + (size_t) j * dim;
      float rn[3]     = {0, 0, 0};
      float dot;
      if (dim == 3) {
        rot3(T, nm, rn);
        dot = (nf[0] * rn[0] + nf[1] * rn[1]) + nf[2] * rn[2];
      } else {
        rot2(T, nm, rn);
        dot = nf[0] * rn[0] + nf[1] * rn[1];
      }
      if (!(dot > s->cfg.finder_normal_cos)) continue;
    }
    s->corr[nc].fixed_idx  = j;
    s->corr[nc].moving_idx = i;
    s->corr[nc].response   = d2;
    ++nc;
  }
  s->ncorr = nc;
  return 0;
}


/* ---- projective finder (first principles): pinhole projection of the transformed moving
 * points into the organised fixed cloud; z-buffer keeps, per pixel, the moving point of minimum depth, ties to the
 * smaller moving index. ---------------------------------------------------------------------------------------- */
#define PIX_BOUND 8.0f
static int project_point(const srrg2_slice_config* c, const float* q, float* u_out, float* v_out) {
  if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) return -1;
  if (!(q[2] >= c->depth_min) || !(q[2] <= c->depth_max)) return -1;
  const float* K = c->camera_matrix;
  float u  = (K[0] * q[0]) / q[2] + K[2];
  float v  = (K[4] * q[1]) / q[2] + K[5];
  float uf = u + 0.5f, vf = v + 0.5f;
  if (!(uf >= 0.f) || !(uf < (float) c->image_cols) || !(vf >= 0.f) || !(vf < (float) c->image_rows)) return -1;
  *u_out = u;
  *v_out = v;
  return (int) floorf(vf) * c->image_cols + (int) floorf(uf);
}

static int slice_compute_correspondences_projective(o_aligner* a, o_slice* s) {
  if (!s->fixed || !s->moving) return fail(SRRG2_E_STATE, "cue slice without fixed/moving cloud");
  const int npix = s->cfg.image_rows * s->cfg.image_cols;
  if (s->nf != npix) return fail(SRRG2_E_STATE, "projective finder: fixed cloud must be organised rows x cols");
  float T[12];
  finder_transform(a, s, T);
  float* zb_depth = (float*) malloc(sizeof(float) * (size_t) npix);
  int* zb_idx     = (int*) malloc(sizeof(int) * (size_t) npix);
  for (int k = 0; k < npix; ++k) {
    zb_depth[k] = INFINITY;
    zb_idx[k]   = 0x7fffffff;
  }
  for (int i = 0; i < s->nm; ++i) {
    const float* p = s->moving + (size_t) i * 3;
    if (!point_finite(p, 3)) continue;
    float q[3], u, v;
    xform3(T, p, q);
    int pix = project_point(&s->cfg, q, &u, &v);
    if (pix < 0) continue;
    if (q[2] < zb_depth[pix] || (q[2] == zb_depth[pix] && i < zb_idx[pix])) {
      zb_depth[pix] = q[2];
      zb_idx[pix]   = i;
    }
  }
  const float gate = s->cfg.finder_max_distance;
  const float lim2 = (2.f * gate) * (2.f * gate);
  int use_ncos     = s->cfg.finder_normal_cos > -1.f && s->fixed_n && s->moving_n;
  int nc           = 0;
  for (int i = 0; i < s->nm; ++i) {
    const float* p = s->moving + (size_t) i * 3;
    if (!point_finite(p, 3)) continue;
    float q[3], u, v;
    xform3(T, p, q);
    int pix = project_point(&s->cfg, q, &u, &v);
    if (pix < 0 || zb_idx[pix] != i) continue;
    const float* f = s->fixed + (size_t) pix * 3;
    if (!point_finite(f, 3)) continue;
    float dd = fabsf(f[2] - q[2]);
    if (!(dd <= gate)) continue;
    if (!(dist2(f, q, 3) <= lim2)) continue;
    if (use_ncos) {
      const float* nm = s->moving_n + (size_t) i * 3;
      const float* nf = s->fixed_n + (size_t) pix * 3;
      float rn[3];
      rot3(T, nm, rn);
      float dot = (nf[0] * rn[0] + nf[1] * rn[1]) + nf[2] * rn[2];
      if (!(dot > s->cfg.finder_normal_cos)) continue;
    }
    s->corr[nc].fixed_idx  = pix;
    s->corr[nc].moving_idx = i;
    s->corr[nc].response   = dd;
    ++nc;
  }
  s->ncorr = nc;
  free(zb_depth);
  free(zb_idx);
  return 0;
}

/* ---- factor: per-correspondence linearisation + fixed-point accumulation ------------- */
static inline float robust_weight(int kind, float thr, float chi, int* kernelized) {
  if (kind == SRRG2_ROBUST_NONE || chi < thr) {
    *kernelized = 0;
    return 1.f;
  }
  *kernelized = 1;
  switch (kind) {
    case SRRG2_ROBUST_CLAMP: return 0.f;
    case SRRG2_ROBUST_SATURATED: return thr / chi;
    default: return 1.0f / (1.0f + chi / thr); /* Cauchy */
  }
}

static int slice_exponent(const o_aligner* a, const o_slice* s) {
  const int plane = s->cfg.kind == SRRG2_SLICE_P2PLANE;
  const int repro = s->cfg.kind == SRRG2_SLICE_REPROJECTION;
  const int proj  = s->cfg.finder == SRRG2_FINDER_PROJECTIVE;
  const double kk = a->kind == SRRG2_SE3_QUAT_RIGHT ? 2.0 : 1.0;
  const int rows  = plane ? 1 : (repro ? 2 : a->dim);
  double mb       = plane ? (1.7320508075688772 * (double) s->ninf) * 1.01 : 1.01;
  if (repro) {
    const double K0 = (double) s->cfg.camera_matrix[0], K4 = (double) s->cfg.camera_matrix[4];
    const double tx = (double) s->cfg.image_cols / K0, ty = (double) s->cfg.image_rows / K4;
    const double gb = (((K0 > K4 ? K0 : K4) / (double) s->cfg.depth_min) * (1.0 + (tx > ty ? tx : ty))) * 1.01;
    mb              = (1.7320508075688772 * gb) * 1.01;
  }
  double pf       = (2.0 * kk) * (double) s->pinf;
  double jb       = mb * (pf > 1.0 ? pf : 1.0);
  double eb       = (mb * (double) s->cfg.finder_max_distance) * 1.01;
  if (proj) eb = (mb * (2.0 * (double) s->cfg.finder_max_distance)) * 1.01;
  if (repro) eb = (double) PIX_BOUND * 1.01;
  int n_terms = s->nm;
  if (s->cfg.finder == SRRG2_FINDER_CORRESPONDENCES) {
    /* no gate bounds the residual: |e_r| <= sqrt3 (sqrt3 |p|inf + |t|inf + |f|inf) with the CURRENT estimate */
    const float* X = a->X;
    double tmax    = 0.0;
    if (a->dim == 3) {
      for (int r = 0; r < 3; ++r) tmax = fabs((double) X[r * 4 + 3]) > tmax ? fabs((double) X[r * 4 + 3]) : tmax;
    } else {
      for (int r = 0; r < 2; ++r) tmax = fabs((double) X[r * 3 + 2]) > tmax ? fabs((double) X[r * 3 + 2]) : tmax;
    }
    eb      = ((mb * 1.7320508075688772) * ((1.7320508075688772 * (double) s->pinf + tmax) + (double) s->finf)) * 1.01;
    n_terms = s->ngiven > 1 ? s->ngiven : 1;
  }
  double mx       = jb > eb ? jb : eb;
  double B        = (double) rows * (mx * mx);
  return o_fixed_point_exponent(n_terms, B);
}

/* Fixed-point terms: every product (w 2^k J_ra) * J_rb is rounded ONCE, to nearest-even, onto the
 * integer grid by a fused multiply-add onto FX_MAGIC = 1.5 * 2^52 (a double in [2^52, 2^53) has ulp 1, so the fma result
 * is FX_MAGIC + integer; the rows of one correspondence are chained r = 0, 1, ... on the same accumulator).  The integer is
 * read back from the bit pattern.  |integer| < 2^51 is guaranteed by slice_exponent.  C99 fma() is exact (one rounding)
 * by definition; -march=x86-64-v3 compiles it to vfmadd. */
#define FX_MAGIC 6755399441055744.0
static inline int64_t fx_bits(double biased) {
  int64_t a, m;
  const double magic = FX_MAGIC;
  memcpy(&a, &biased, 8);
  memcpy(&m, &magic, 8);
  return a - m;
}

static int slice_linearize(o_aligner* a, o_slice* s) {
  const int dim   = a->dim;
  const int D     = a->dof;
  const int plane = s->cfg.kind == SRRG2_SLICE_P2PLANE;
  const int repro = s->cfg.kind == SRRG2_SLICE_REPROJECTION;
  if (plane && !s->fixed_n) return fail(SRRG2_E_STATE, "point-to-plane slice without fixed normals");
  const float kk = a->kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const int k    = slice_exponent(a, s);
  s->k           = k;
  memset(s->acc, 0, sizeof(s->acc));
  float T[12];
  finder_transform(a, s, T);
  for (int c = 0; c < s->ncorr; ++c) {
    const int i    = s->corr[c].moving_idx;
    const int j    = s->corr[c].fixed_idx;
    const float* p = s->moving + (size_t) i * dim;
    const float* f = s->fixed + (size_t) j * dim;
    float J[3][6];
    float e[3];
    int rows;
    int invalid = 0;
    if (dim == 3) {
      float q[3];
      xform3(T, p, q);
      float m[3][3];
      if (repro) {
        /* e = pi(q) - pi(f); rows g_r = d pi / d q, m_r = T_R^T g_r */
        const float* K = s->cfg.camera_matrix;
        rows           = 2;
        if (!(f[2] > 0.f)) {
          invalid = 1;
          e[0] = e[1] = 0.f;
          memset(m, 0, sizeof(m));
        } else {
          float uq = (K[0] * q[0]) / q[2] + K[2], vq = (K[4] * q[1]) / q[2] + K[5];
          float uf = (K[0] * f[0]) / f[2] + K[2], vf = (K[4] * f[1]) / f[2] + K[5];
          e[0]     = uq - uf;
          e[1]     = vq - vf;
          if (!(fabsf(e[0]) <= PIX_BOUND) || !(fabsf(e[1]) <= PIX_BOUND)) invalid = 1;
          float iz   = 1.0f / q[2];
          float g[2][3];
          g[0][0] = K[0] * iz; g[0][1] = 0.f;       g[0][2] = -(((K[0] * q[0]) * iz) * iz);
          g[1][0] = 0.f;       g[1][1] = K[4] * iz; g[1][2] = -(((K[4] * q[1]) * iz) * iz);
          for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 3; ++k) m[r][k] = (T[0 * 4 + k] * g[r][0] + T[1 * 4 + k] * g[r][1]) + T[2 * 4 + k] * g[r][2];
        }
      } else if (plane) {
        const float* n = s->fixed_n + (size_t) j * 3;
        rows           = 1;
        e[0]           = (n[0] * (q[0] - f[0]) + n[1] * (q[1] - f[1])) + n[2] * (q[2] - f[2]);
        m[0][0]        = (T[0] * n[0] + T[4] * n[1]) + T[8] * n[2];
        m[0][1]        = (T[1] * n[0] + T[5] * n[1]) + T[9] * n[2];
        m[0][2]        = (T[2] * n[0] + T[6] * n[1]) + T[10] * n[2];
      } else {
        rows = 3;
        for (int r = 0; r < 3; ++r) {
          e[r]    = q[r] - f[r];
          m[r][0] = T[r * 4 + 0];
          m[r][1] = T[r * 4 + 1];
          m[r][2] = T[r * 4 + 2];
        }
      }
      for (int r = 0; r < rows; ++r) {
        J[r][0] = m[r][0];
        J[r][1] = m[r][1];
        J[r][2] = m[r][2];
        J[r][3] = kk * (p[1] * m[r][2] - p[2] * m[r][1]);
        J[r][4] = kk * (p[2] * m[r][0] - p[0] * m[r][2]);
        J[r][5] = kk * (p[0] * m[r][1] - p[1] * m[r][0]);
      }
    } else {
      float q[2];
      xform2(T, p, q);
      float m[2][2];
      if (plane) {
        const float* n = s->fixed_n + (size_t) j * 2;
        rows           = 1;
        e[0]           = n[0] * (q[0] - f[0]) + n[1] * (q[1] - f[1]);
        m[0][0]        = T[0] * n[0] + T[3] * n[1];
        m[0][1]        = T[1] * n[0] + T[4] * n[1];
      } else {
        rows = 2;
        for (int r = 0; r < 2; ++r) {
          e[r]    = q[r] - f[r];
          m[r][0] = T[r * 3 + 0];
          m[r][1] = T[r * 3 + 1];
        }
      }
      for (int r = 0; r < rows; ++r) {
        J[r][0] = m[r][0];
        J[r][1] = m[r][1];
        J[r][2] = m[r][1] * p[0] - m[r][0] * p[1];
      }
    }
    float chi = e[0] * e[0];
    for (int r = 1; r < rows; ++r) chi = chi + e[r] * e[r];
    s->acc[ACC_N_CORR] += 1;
    if (!isfinite(chi) || invalid) {
      s->fstat[c] = SRRG2_FACTOR_SUPPRESSED;
      continue;
    }
    int kernelized;
    float w = robust_weight(s->robust_kind, s->robust_thr, chi, &kernelized);
    const double scale = ldexp(1.0, k);
    const int64_t chi_fx = fx_bits(fma((double) chi, scale, FX_MAGIC)); /* chi 2^k is exact: rne to the grid */
    if (kernelized) {
      s->fstat[c] = SRRG2_FACTOR_KERNELIZED;
      s->acc[ACC_N_OUT] += 1;
      s->acc[ACC_CHI_OUT] += chi_fx;
    } else {
      s->fstat[c] = SRRG2_FACTOR_INLIER;
      s->acc[ACC_N_IN] += 1;
      s->acc[ACC_CHI_IN] += chi_fx;
    }
    if (w == 0.f) continue;
    const double ws = (double) w * scale; /* exact */
    for (int aa = 0; aa < D; ++aa) {
      double wj[3];
      for (int r = 0; r < rows; ++r) wj[r] = ws * (double) J[r][aa]; /* exact: 24 x 24 bits */
      for (int bb = aa; bb < D; ++bb) {
        double t = FX_MAGIC;
        for (int r = 0; r < rows; ++r) t = fma(wj[r], (double) J[r][bb], t);
        s->acc[hidx(aa, bb)] += fx_bits(t);
      }
      double t = FX_MAGIC;
      for (int r = 0; r < rows; ++r) t = fma(wj[r], (double) e[r], t);
      s->acc[21 + aa] += fx_bits(t);
    }
  }
  return 0;
}

/* ---- prior factor (SE2PriorErrorFactor / SE3PriorErrorFactorAD, [EXT]) ------------------ */
static void prior_linearize(o_aligner* a, o_slice* s) {
  const int D = a->dof;
  double e[6], J[36];
  memset(J, 0, sizeof(J));
  float Zinv[12], E[12];
  if (a->dim == 3) {
    o_se3_inverse(s->prior_Z, Zinv);
    o_se3_compose(Zinv, a->X, E);
    o_se3_t2v_quat(E, e);
    /* w from the normalised quaternion: w = sqrt(1 - |v|^2) of the t2v output */
    double n2 = (e[3] * e[3] + e[4] * e[4]) + e[5] * e[5];
    double w  = n2 < 1.0 ? sqrt(1.0 - n2) : 0.0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) J[i * 6 + j] = (double) E[i * 4 + j];
    J[3 * 6 + 3] = w;
    J[3 * 6 + 4] = -e[5];
    J[3 * 6 + 5] = e[4];
    J[4 * 6 + 3] = e[5];
    J[4 * 6 + 4] = w;
    J[4 * 6 + 5] = -e[3];
    J[5 * 6 + 3] = -e[4];
    J[5 * 6 + 4] = e[3];
    J[5 * 6 + 5] = w;
  } else {
    o_se2_inverse(s->prior_Z, Zinv);
    o_se2_compose(Zinv, a->X, E);
    o_se2_t2v(E, e);
    J[0 * 3 + 0] = (double) E[0];
    J[0 * 3 + 1] = (double) E[1];
    J[1 * 3 + 0] = (double) E[3];
    J[1 * 3 + 1] = (double) E[4];
    J[2 * 3 + 2] = 1.0;
  }
  double chi = 0.0;
  for (int i = 0; i < D; ++i) chi = chi + (e[i] * (double) s->cfg.prior_information_diag[i]) * e[i];
  int kernelized;
  float w = robust_weight(s->robust_kind, s->robust_thr, (float) chi, &kernelized);
  s->prior_chi    = chi;
  s->prior_status = !isfinite(chi) ? SRRG2_FACTOR_SUPPRESSED : (kernelized ? SRRG2_FACTOR_KERNELIZED : SRRG2_FACTOR_INLIER);
  memset(s->prior_H, 0, sizeof(s->prior_H));
  memset(s->prior_b, 0, sizeof(s->prior_b));
  if (s->prior_status == SRRG2_FACTOR_SUPPRESSED) return;
  for (int aa = 0; aa < D; ++aa) {
    for (int bb = 0; bb < D; ++bb) {
      double t = 0.0;
      for (int r = 0; r < D; ++r) t = t + (J[r * D + aa] * (double) s->cfg.prior_information_diag[r]) * J[r * D + bb];
      s->prior_H[aa * D + bb] = (double) w * t;
    }
    double t = 0.0;
    for (int r = 0; r < D; ++r) t = t + (J[r * D + aa] * (double) s->cfg.prior_information_diag[r]) * e[r];
    s->prior_b[aa] = (double) w * t;
  }
}

/* ---- termination criterion (aligner_termination_criteria_impl.cpp) ----------------------- */
static void win_reset(o_window* w, int window) {
  w->count  = 0;
  w->window = window;
}
static void win_add(o_window* w, double v) {
  w->buf[w->count % w->window] = v;
  w->count++;
}
static int win_n(const o_window* w) {
  return w->count < w->window ? w->count : w->window;
}
static double win_max(const o_window* w) {
  double m = w->buf[0];
  for (int i = 1; i < win_n(w); ++i)
    if (w->buf[i] > m) m = w->buf[i];
  return m;
}
static double win_min(const o_window* w) {
  double m = w->buf[0];
  for (int i = 1; i < win_n(w); ++i)
    if (w->buf[i] < m) m = w->buf[i];
  return m;
}
static double win_range(const o_window* w) {
  return win_max(w) - win_min(w);
}

static int num_correspondences(const o_aligner* a) {
  int n = 0;
  for (int i = 0; i < a->nslices; ++i) {
    int c = a->slices[i].cfg.kind == SRRG2_SLICE_PRIOR ? 1 : a->slices[i].ncorr; /* prior.h:75-77 */
    if (c >= 0) n += c;
  }
  return n;
}

static int has_to_stop(o_aligner* a) {
  /* aligner_termination_criteria_impl.cpp:24-65, quirks preserved */
  const srrg2_iteration_stats* cur = &a->stats[a->nstats - 1];
  int ncorr                        = num_correspondences(a);
  int ninl                         = cur->num_inliers;
  int nout                         = cur->num_outliers;
  float chi                        = cur->chi_inliers / (float) ninl;
  if (!ninl) return 0;
  win_add(&a->w_corr, ncorr);
  win_add(&a->w_inl, ninl);
  win_add(&a->w_out, nout);
  win_add(&a->w_chi, (double) chi);
  if (win_n(&a->w_corr) < a->term.window_size) return 0;
  if (win_range(&a->w_out) > (double) a->term.num_correspondences_range) return 0; /* :46 */
  if (win_range(&a->w_inl) > (double) a->term.num_inliers_range) return 0;
  if ((float) win_range(&a->w_chi) > (float) a->term.num_outliers_range) return 0; /* :53 */
  if ((float) win_range(&a->w_chi) / (float) win_max(&a->w_chi) > a->term.chi_epsilon) return 0;
  return 1;
}

/* ---- one solver->compute(): ONE Gauss-Newton iteration (multi_aligner.h:61-62) ----------- */
static void push_stats(o_aligner* a, const srrg2_iteration_stats* st) {
  if (a->nstats == a->stats_cap) {
    a->stats_cap = a->stats_cap ? a->stats_cap * 2 : 32;
    a->stats     = (srrg2_iteration_stats*) realloc(a->stats, sizeof(srrg2_iteration_stats) * (size_t) a->stats_cap);
  }
  a->stats[a->nstats++] = *st;
}

static int solver_compute(o_aligner* a) {
  const int D = a->dof;
  double H[36], b[6], dx[6];
  memset(H, 0, sizeof(H));
  memset(b, 0, sizeof(b));
  srrg2_iteration_stats st;
  memset(&st, 0, sizeof(st));
  double chi_in = 0.0, chi_out = 0.0;
  for (int si = 0; si < a->nslices; ++si) {
    o_slice* s = &a->slices[si];
    if (s->cfg.kind == SRRG2_SLICE_PRIOR) {
      prior_linearize(a, s);
      for (int i = 0; i < D * D; ++i) H[i] = H[i] + s->prior_H[i];
      for (int i = 0; i < D; ++i) b[i] = b[i] + s->prior_b[i];
      if (s->prior_status == SRRG2_FACTOR_INLIER) {
        st.num_inliers++;
        chi_in = chi_in + s->prior_chi;
      } else if (s->prior_status == SRRG2_FACTOR_KERNELIZED) {
        st.num_outliers++;
        chi_out = chi_out + s->prior_chi;
      } else {
        st.num_suppressed++;
      }
      continue;
    }
    int rc = slice_linearize(a, s);
    if (rc) return rc;
    for (int aa = 0; aa < D; ++aa) {
      for (int bb = aa; bb < D; ++bb) {
        double v       = ldexp((double) s->acc[hidx(aa, bb)], -s->k);
        H[aa * D + bb] = H[aa * D + bb] + v;
        if (bb != aa) H[bb * D + aa] = H[bb * D + aa] + v;
      }
      b[aa] = b[aa] + ldexp((double) s->acc[21 + aa], -s->k);
    }
    st.num_inliers += (int) s->acc[ACC_N_IN];
    st.num_outliers += (int) s->acc[ACC_N_OUT];
    st.num_suppressed += (int) (s->acc[ACC_N_CORR] - s->acc[ACC_N_IN] - s->acc[ACC_N_OUT]);
    chi_in  = chi_in + ldexp((double) s->acc[ACC_CHI_IN], -s->k);
    chi_out = chi_out + ldexp((double) s->acc[ACC_CHI_OUT], -s->k);
  }
  st.iteration           = a->nstats;
  st.num_correspondences = num_correspondences(a);
  st.chi_inliers         = (float) chi_in;
  st.chi_outliers        = (float) chi_out;
  int bad                = o_solve(D, H, b, dx);
  st.solver_status       = bad ? 1 : 0;
  memcpy(a->last_H, H, sizeof(H));
  memcpy(a->last_b, b, sizeof(b));
  if (!bad) {
    memcpy(a->last_dx, dx, sizeof(dx));
    o_box_plus(a->kind, a->X, dx);
  } else {
    memset(a->last_dx, 0, sizeof(a->last_dx));
  }
  push_stats(a, &st);
  return 0;
}

/* ---- _runSolver (multi_aligner_impl.cpp:98-128) --------------------------------------------- */
static int run_solver(o_aligner* a, int iterations, int use_term) {
  float backup[12];
  memcpy(backup, a->X, sizeof(backup));
  for (int it = 0; it < iterations; ++it) {
    /* _computeCorrespondencesPerSlices(backup), multi_aligner.h:126-138 */
    memcpy(a->X, backup, sizeof(backup));
    int good = 0;
    for (int si = 0; si < a->nslices; ++si) {
      o_slice* s = &a->slices[si];
      if (s->cfg.kind == SRRG2_SLICE_PRIOR) {
        good |= 1; /* aligner_slice_processor_prior.h:66-68 */
        continue;
      }
      int rc = slice_compute_correspondences(a, s);
      if (rc) return rc;
      good |= s->ncorr > s->cfg.min_num_correspondences; /* aligner_slice_processor_impl.cpp:77-79 */
    }
    if (!good) {
      a->status = SRRG2_NOT_ENOUGH_CORRESPONDENCES; /* :108 */
      memcpy(a->X, backup, sizeof(backup));
      break;
    }
    int rc = solver_compute(a);
    if (rc) return rc;
    if (a->stats[a->nstats - 1].solver_status == 0) {
      memcpy(backup, a->X, sizeof(backup)); /* :118-121 */
    } else {
      memcpy(a->X, backup, sizeof(backup));
    }
    if (use_term && has_to_stop(a)) break; /* :124-126 */
  }
  return 0;
}

static void prune_correspondences(o_aligner* a) {
  /* multi_aligner_impl.cpp:214-263: keep FactorStats::Status::Inlier of the last iteration */
  for (int si = 0; si < a->nslices; ++si) {
    o_slice* s = &a->slices[si];
    if (s->cfg.kind == SRRG2_SLICE_PRIOR || !s->ncorr) continue;
    int keep = 0;
    for (int c = 0; c < s->ncorr; ++c) {
      if (s->fstat[c] == SRRG2_FACTOR_INLIER) {
        s->corr[keep]  = s->corr[c];
        s->fstat[keep] = s->fstat[c];
        ++keep;
      }
    }
    s->ncorr = keep;
  }
}

int oracle_aligner_compute(o_aligner* h, int* status_out) {
  if (!h) return fail(SRRG2_E_INVALID, "compute");
  o_aligner* a = h;
  if (a->has_term) { /* term_crit->init(this), :55-57 */
    win_reset(&a->w_corr, a->term.window_size);
    win_reset(&a->w_inl, a->term.window_size);
    win_reset(&a->w_out, a->term.window_size);
    win_reset(&a->w_chi, a->term.window_size);
  }
  a->nstats = 0; /* :58-59 */
  /* _preCompute(): bindRobustifier + init per slice, :131-141 */
  for (int si = 0; si < a->nslices; ++si) {
    o_slice* s     = &a->slices[si];
    s->robust_kind = s->cfg.robustifier;
    s->robust_thr  = s->cfg.robustifier_chi_threshold;
    if (s->cfg.kind == SRRG2_SLICE_PRIOR) {
      if (!s->has_prior) return fail(SRRG2_E_STATE, "prior slice without measurement"); /* prior_impl.cpp:16,20 */
      if (s->cfg.prior_sets_initial_guess) {
        memcpy(a->X, s->prior_Z, sizeof(float) * a->tsize); /* aligner_slice_odometry_prior.cpp:19,34 */
      }
    }
  }
  int rc = run_solver(a, a->params.max_iterations, a->has_term); /* :72 */
  if (rc) return rc;
  if (a->nstats == 0) { /* :75-78 */
    a->status = SRRG2_FAIL;
    if (status_out) *status_out = a->status;
    return 0;
  }
  if (a->stats[a->nstats - 1].num_inliers < a->params.min_num_inliers) { /* :81-85 */
    a->status = SRRG2_NOT_ENOUGH_INLIERS;
    if (status_out) *status_out = a->status;
    return 0;
  }
  /* _postCompute(), :163-181 */
  if (a->params.enable_inlier_only_runs) {
    for (int si = 0; si < a->nslices; ++si) { /* _setClampRobustifiers, :184-201 */
      o_slice* s = &a->slices[si];
      if (s->cfg.robustifier != SRRG2_ROBUST_NONE) {
        s->robust_kind = SRRG2_ROBUST_CLAMP;
        s->robust_thr  = s->cfg.robustifier_chi_threshold;
      }
    }
    rc = run_solver(a, a->params.max_iterations, a->has_term);
    for (int si = 0; si < a->nslices; ++si) { /* _restoreRobustifiers, :204-211 */
      a->slices[si].robust_kind = a->slices[si].cfg.robustifier;
      a->slices[si].robust_thr  = a->slices[si].cfg.robustifier_chi_threshold;
    }
    if (rc) return rc;
  }
  if (a->params.keep_only_inlier_correspondences) {
    prune_correspondences(a);
  }
  if (a->dim == 3) /* fixTransform, :91-93 */
    o_se3_fix_transform(a->X);
  else
    o_se2_fix_transform(a->X);
  a->status = SRRG2_SUCCESS;
  if (status_out) *status_out = a->status;
  return 0;
}

int oracle_aligner_status(o_aligner* h, int* s) {
  if (!h || !s) return fail(SRRG2_E_INVALID, "status");
  *s = h->status;
  return 0;
}

int oracle_aligner_get_iteration_stats(o_aligner* h, srrg2_iteration_stats* buf, int* n) {
  if (!h || !n) return fail(SRRG2_E_INVALID, "get_iteration_stats");
  if (buf) {
    int m = *n < h->nstats ? *n : h->nstats;
    memcpy(buf, h->stats, sizeof(srrg2_iteration_stats) * (size_t) m);
  }
  *n = h->nstats;
  return 0;
}

int oracle_aligner_num_correspondences(o_aligner* h, int* n) {
  if (!h || !n) return fail(SRRG2_E_INVALID, "num_correspondences");
  *n = num_correspondences(h);
  return 0;
}

int oracle_aligner_get_correspondences(o_aligner* h, int si, srrg2_correspondence* buf, int* n) {
  if (!h || !n || si < 0 || si >= h->nslices) return fail(SRRG2_E_INVALID, "get_correspondences");
  o_slice* s = &h->slices[si];
  if (buf) {
    int m = *n < s->ncorr ? *n : s->ncorr;
    memcpy(buf, s->corr, sizeof(srrg2_correspondence) * (size_t) m);
  }
  *n = s->ncorr;
  return 0;
}

int oracle_aligner_get_factor_status(o_aligner* h, int si, uint8_t* buf, int* n) {
  if (!h || !n || si < 0 || si >= h->nslices) return fail(SRRG2_E_INVALID, "get_factor_status");
  o_slice* s = &h->slices[si];
  if (buf) {
    int m = *n < s->ncorr ? *n : s->ncorr;
    memcpy(buf, s->fstat, (size_t) m);
  }
  *n = s->ncorr;
  return 0;
}

int oracle_aligner_linearize_once(o_aligner* h, int si, int64_t* acc32, int* k_out) {
  if (!h || si < 0 || si >= h->nslices) return fail(SRRG2_E_INVALID, "linearize_once");
  o_slice* s = &h->slices[si];
  if (s->cfg.kind == SRRG2_SLICE_PRIOR) return fail(SRRG2_E_INVALID, "linearize_once on prior");
  s->robust_kind = s->cfg.robustifier;
  s->robust_thr  = s->cfg.robustifier_chi_threshold;
  int rc         = slice_compute_correspondences(h, s);
  if (rc) return rc;
  rc = slice_linearize(h, s);
  if (rc) return rc;
  if (acc32) memcpy(acc32, s->acc, sizeof(s->acc));
  if (k_out) *k_out = s->k;
  return 0;
}

int oracle_aligner_get_last_system(o_aligner* h, double* H, double* b, double* dx) {
  if (!h) return fail(SRRG2_E_INVALID, "get_last_system");
  if (H) memcpy(H, h->last_H, sizeof(double) * h->dof * h->dof);
  if (b) memcpy(b, h->last_b, sizeof(double) * h->dof);
  if (dx) memcpy(dx, h->last_dx, sizeof(double) * h->dof);
  return 0;
}

/* srrg2_aligner_get_information: H of the last Gauss-Newton iteration as the product hands it out (float32, D x D) */
int oracle_aligner_get_information(o_aligner* h, float* H) {
  if (!h || !H) return fail(SRRG2_E_INVALID, "get_information");
  for (int i = 0; i < h->dof * h->dof; ++i) H[i] = (float) h->last_H[i];
  return 0;
}

int oracle_aligner_set_correspondences(o_aligner* h, int si, const srrg2_correspondence* corr, int n) {
  if (!h || si < 0 || si >= h->nslices || n < 0 || (n > 0 && !corr)) return fail(SRRG2_E_INVALID, "set_correspondences");
  o_slice* s = &h->slices[si];
  if (s->cfg.finder != SRRG2_FINDER_CORRESPONDENCES) return fail(SRRG2_E_INVALID, "set_correspondences: wrong finder kind");
  free(s->given);
  s->given = (srrg2_correspondence*) malloc(sizeof(srrg2_correspondence) * (size_t) (n > 0 ? n : 1));
  if (n > 0) memcpy(s->given, corr, sizeof(srrg2_correspondence) * (size_t) n);
  s->ngiven = n;
  return 0;
}

int oracle_aligner_compute_batch_correspondences(o_aligner* h, int K, const float* coords, int cs, const float* normals,
                                                 int ns, const int32_t* offsets, int mem,
                                                 const srrg2_correspondence* corr, const int32_t* corr_offsets,
                                                 const float* guesses, srrg2_batch_result* results) {
  /* loop body of multi_loop_detector_hbst_impl.cpp:296-374, one candidate after the other */
  if (!h || K < 0 || !offsets || !corr_offsets || !guesses || !results) return fail(SRRG2_E_INVALID, "compute_batch_correspondences");
  for (int k = 0; k < K; ++k) {
    const float* c = (const float*) ((const char*) coords + (size_t) offsets[k] * cs);
    const float* n = normals ? (const float*) ((const char*) normals + (size_t) offsets[k] * ns) : NULL;
    int rc         = oracle_aligner_set_moving(h, 0, c, cs, n, ns, offsets[k + 1] - offsets[k], mem);
    if (rc) return rc;
    rc = oracle_aligner_set_correspondences(h, 0, corr + corr_offsets[k], corr_offsets[k + 1] - corr_offsets[k]);
    if (rc) return rc;
    rc = oracle_aligner_set_moving_in_fixed(h, guesses + (size_t) k * h->tsize);
    if (rc) return rc;
    int st;
    rc = oracle_aligner_compute(h, &st);
    if (rc) return rc;
    memset(&results[k], 0, sizeof(results[k]));
    memcpy(results[k].moving_in_fixed, h->X, sizeof(float) * h->tsize);
    results[k].status         = st;
    results[k].num_iterations = h->nstats;
    if (h->nstats) results[k].last = h->stats[h->nstats - 1];
    int nc = 0;
    oracle_aligner_num_correspondences(h, &nc); /* after _pruneCorrespondences */
    results[k].num_correspondences = nc;
    if (h->nstats)
      for (int i = 0; i < h->dof * h->dof; ++i) results[k].information[i] = (float) h->last_H[i];
  }
  return 0;
}

int oracle_aligner_compute_batch(o_aligner* h, int K, const float* coords, int cs, const float* normals, int ns,
                                 const int32_t* offsets, int mem, const float* guesses, srrg2_batch_result* results) {
  /* loop body of multi_loop_detector_brute_force_impl.cpp:64-91 */
  if (!h || K < 0 || !offsets || !guesses || !results) return fail(SRRG2_E_INVALID, "compute_batch");
  for (int k = 0; k < K; ++k) {
    const float* c = (const float*) ((const char*) coords + (size_t) offsets[k] * cs);
    const float* n = normals ? (const float*) ((const char*) normals + (size_t) offsets[k] * ns) : NULL;
    int rc         = oracle_aligner_set_moving(h, 0, c, cs, n, ns, offsets[k + 1] - offsets[k], mem);
    if (rc) return rc;
    rc = oracle_aligner_set_moving_in_fixed(h, guesses + (size_t) k * h->tsize);
    if (rc) return rc;
    int st;
    rc = oracle_aligner_compute(h, &st);
    if (rc) return rc;
    memset(&results[k], 0, sizeof(results[k]));
    memcpy(results[k].moving_in_fixed, h->X, sizeof(float) * h->tsize);
    results[k].status         = st;
    results[k].num_iterations = h->nstats;
    if (h->nstats) results[k].last = h->stats[h->nstats - 1];
    int nc = 0;
    oracle_aligner_num_correspondences(h, &nc); /* after _pruneCorrespondences */
    results[k].num_correspondences = nc;
    if (h->nstats)
      for (int i = 0; i < h->dof * h->dof; ++i) results[k].information[i] = (float) h->last_H[i];
  }
  return 0;
}
