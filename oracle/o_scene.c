/*
 * o_scene.c -- CPU oracle of the tracker-side scene steps.
 * TEST INFRASTRUCTURE ONLY (see oracle.h): restates, single threaded and in the reference's order,
 *   - MergerCorrespondenceHomo_::compute()   S/mapping/merger_correspondence_homo_impl.cpp:11-125
 *   - the SceneClipper_ contract             S/mapping/scene_clipper.h:17-122 (interface only in the
 *     reference; the ball policy is this build's concrete clipper, documented in include/srrg2_slam_amd.h)
 * PARITY UNPINNED for the float32 arithmetic (Eigen's Isometry * Vector and squaredNorm are restated in a
 * fixed operation order); the control flow -- what is merged, skipped, appended, in which order -- follows
 * the cited lines.  Points are {x, y, z, 0}; Valid <=> finite coordinates.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

struct o_scene {
  int dim;
  int n, cap;
  int has_normals;
  float* pts; /* 4 floats per point */
  float* nrm; /* 4 floats per point */
  int* gidx;  /* local -> global indices of the last clip into this scene */
  int ng;
};

extern void oracle_set_error(const char* msg);
static int s_fail(const char* msg) {
  oracle_set_error(msg);
  return SRRG2_E_INVALID;
}

static int reserve(o_scene* s, int n) {
  if (n <= s->cap) return 0;
  int cap   = n + n / 2 + 16;
  float* p  = (float*) realloc(s->pts, (size_t) cap * 4 * sizeof(float));
  float* q  = (float*) realloc(s->nrm, (size_t) cap * 4 * sizeof(float));
  if (!p || !q) return s_fail("scene: out of memory");
  s->pts = p;
  s->nrm = q;
  s->cap = cap;
  return 0;
}

int oracle_scene_create(int dim, o_scene** out) {
  if ((dim != 2 && dim != 3) || !out) return s_fail("scene_create: bad arguments");
  o_scene* s = (o_scene*) calloc(1, sizeof(o_scene));
  if (!s) return s_fail("scene_create: out of memory");
  s->dim = dim;
  *out   = s;
  return 0;
}

int oracle_scene_destroy(o_scene* s) {
  if (!s) return 0;
  free(s->pts);
  free(s->nrm);
  free(s->gidx);
  free(s);
  return 0;
}

int oracle_scene_set(o_scene* s, const float* coords, int cs, const float* normals, int ns, int n) {
  if (!s || n < 0 || (n > 0 && !coords)) return s_fail("scene_set: bad arguments");
  int rc = reserve(s, n);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    const float* c = (const float*) ((const char*) coords + (size_t) i * cs);
    float* p       = s->pts + 4 * (size_t) i;
    p[0] = c[0]; p[1] = c[1]; p[2] = s->dim == 3 ? c[2] : 0.f; p[3] = 0.f;
    float* q = s->nrm + 4 * (size_t) i;
    q[0] = q[1] = q[2] = q[3] = 0.f;
    if (normals) {
      const float* m = (const float*) ((const char*) normals + (size_t) i * ns);
      q[0] = m[0]; q[1] = m[1]; q[2] = s->dim == 3 ? m[2] : 0.f;
    }
  }
  s->n           = n;
  s->has_normals = normals != NULL;
  s->ng          = 0;
  return 0;
}

int oracle_scene_size(o_scene* s, int* n) {
  if (!s || !n) return s_fail("scene_size: bad arguments");
  *n = s->n;
  return 0;
}

int oracle_scene_get(o_scene* s, float* coords_out, float* normals_out, int capacity, int* n) {
  if (!s || !n) return s_fail("scene_get: bad arguments");
  int m = s->n < capacity ? s->n : capacity;
  for (int i = 0; i < m && coords_out; ++i)
    for (int d = 0; d < s->dim; ++d) coords_out[(size_t) i * s->dim + d] = s->pts[4 * (size_t) i + d];
  for (int i = 0; i < m && normals_out; ++i)
    for (int d = 0; d < s->dim; ++d) normals_out[(size_t) i * s->dim + d] = s->nrm[4 * (size_t) i + d];
  *n = s->n;
  return 0;
}

int oracle_scene_global_indices(o_scene* s, int32_t* buf, int* n_inout) {
  if (!s || !n_inout) return s_fail("scene_global_indices: bad arguments");
  int m = s->ng < *n_inout ? s->ng : *n_inout;
  for (int i = 0; i < m && buf; ++i) buf[i] = s->gidx[i];
  *n_inout = s->ng;
  return 0;
}

/* T: SE(3) row-major 3x4, SE(2) row-major 3x3 -> rows of [R|t] as 3x4 with the unused entries zero */
static void load_transform(int dim, const float* T, float* M) {
  if (dim == 3) {
    memcpy(M, T, 12 * sizeof(float));
  } else {
    M[0] = T[0]; M[1] = T[1]; M[2] = 0.f; M[3] = T[2];
    M[4] = T[3]; M[5] = T[4]; M[6] = 0.f; M[7] = T[5];
    M[8] = 0.f; M[9] = 0.f; M[10] = 1.f; M[11] = 0.f;
  }
}

static void xform_point(int dim, const float* M, const float* p, float* q) {
  if (dim == 3) {
    q[0] = ((M[0] * p[0] + M[1] * p[1]) + M[2] * p[2]) + M[3];
    q[1] = ((M[4] * p[0] + M[5] * p[1]) + M[6] * p[2]) + M[7];
    q[2] = ((M[8] * p[0] + M[9] * p[1]) + M[10] * p[2]) + M[11];
  } else {
    q[0] = (M[0] * p[0] + M[1] * p[1]) + M[3];
    q[1] = (M[4] * p[0] + M[5] * p[1]) + M[7];
    q[2] = 0.f;
  }
  q[3] = 0.f;
}

static void rotate_normal(int dim, const float* M, const float* n, float* r) {
  if (dim == 3) {
    r[0] = (M[0] * n[0] + M[1] * n[1]) + M[2] * n[2];
    r[1] = (M[4] * n[0] + M[5] * n[1]) + M[6] * n[2];
    r[2] = (M[8] * n[0] + M[9] * n[1]) + M[10] * n[2];
  } else {
    r[0] = M[0] * n[0] + M[1] * n[1];
    r[1] = M[4] * n[0] + M[5] * n[1];
    r[2] = 0.f;
  }
  r[3] = 0.f;
}

static int valid_point(int dim, const float* p) {
  return isfinite(p[0]) && isfinite(p[1]) && (dim == 2 || isfinite(p[2]));
}

/* SceneClipper_::compute(), ball policy; scene_clipper.h:64-68 (local_map_in_robot = robot_in_local_map^-1),
 * :98-101 (globalIndices), :24-28 (status) */
int oracle_scene_clip_ball(o_scene* full, const float* robot_in_local_map, float range, o_scene* clipped, int* status) {
  if (!full || !clipped || !robot_in_local_map || full->dim != clipped->dim || full == clipped)
    return s_fail("scene_clip_ball: bad arguments");
  const int dim = full->dim;
  float Linv[12], M[12];
  if (dim == 3) {
    o_se3_inverse(robot_in_local_map, Linv);
  } else {
    o_se2_inverse(robot_in_local_map, Linv);
  }
  load_transform(dim, Linv, M);
  int rc = reserve(clipped, full->n);
  if (rc) return rc;
  int* g = (int*) realloc(clipped->gidx, (size_t) (full->n + 1) * sizeof(int));
  if (!g) return s_fail("scene_clip_ball: out of memory");
  clipped->gidx        = g;
  clipped->has_normals = full->has_normals;
  const float range2   = range * range;
  int k = 0;
  for (int i = 0; i < full->n; ++i) {
    const float* p = full->pts + 4 * (size_t) i;
    if (!valid_point(dim, p)) continue;
    float q[4];
    xform_point(dim, M, p, q);
    const float d2 = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    if (!(d2 <= range2)) continue;
    memcpy(clipped->pts + 4 * (size_t) k, q, 4 * sizeof(float));
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (full->has_normals) rotate_normal(dim, M, full->nrm + 4 * (size_t) i, r);
    memcpy(clipped->nrm + 4 * (size_t) k, r, 4 * sizeof(float));
    g[k] = i;
    ++k;
  }
  clipped->n  = k;
  clipped->ng = k;
  if (status) *status = full->n == 0 ? SRRG2_CLIPPER_READY : SRRG2_CLIPPER_SUCCESSFUL;
  return 0;
}

static int append_point(o_scene* scene, o_scene* meas, const float* M, int index) {
  int rc = reserve(scene, scene->n + 1);
  if (rc) return rc;
  /* point_meas.transformInPlace(measurement_in_scene): coordinates and normal (:37, :110) */
  xform_point(scene->dim, M, meas->pts + 4 * (size_t) index, scene->pts + 4 * (size_t) scene->n);
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  if (meas->has_normals) rotate_normal(scene->dim, M, meas->nrm + 4 * (size_t) index, r);
  memcpy(scene->nrm + 4 * (size_t) scene->n, r, 4 * sizeof(float));
  scene->n++;
  return 0;
}

/* MergerCorrespondenceHomo_::compute(), merger_correspondence_homo_impl.cpp:11-125 */
int oracle_scene_merge(o_scene* scene, o_scene* meas, const float* measurement_in_scene, const srrg2_correspondence* corr,
                       int ncorr, const srrg2_merger_params* p, srrg2_merge_result* out) {
  if (!scene || !meas || !measurement_in_scene || !p || !out || scene->dim != meas->dim || scene == meas)
    return s_fail("scene_merge: bad arguments");
  if (ncorr > 0 && !corr) return s_fail("scene_merge: null correspondences");
  const int dim = scene->dim;
  float M[12];
  load_transform(dim, measurement_in_scene, M);
  memset(out, 0, sizeof(*out));
  out->status = SRRG2_MERGER_INITIALIZING; /* :15 */
  const int n_scene = scene->n, n_meas = meas->n;
  int rc;
  if (scene->n == 0 && meas->has_normals) scene->has_normals = 1; /* a fresh scene takes the measurement's fields */
  if (ncorr < 0) {
    /* :30-41 no correspondences set: every Valid measurement point is transformed and added */
    for (int i = 0; i < n_meas; ++i) {
      if (!valid_point(dim, meas->pts + 4 * (size_t) i)) continue;
      if ((rc = append_point(scene, meas, M, i))) return rc;
      out->num_added++;
    }
  } else {
    for (int c = 0; c < ncorr; ++c) /* the reference asserts these (:53-54); here they are errors */
      if (corr[c].fixed_idx < 0 || corr[c].fixed_idx >= n_scene || corr[c].moving_idx < 0 || corr[c].moving_idx >= n_meas)
        return s_fail("scene_merge: correspondence index out of range");
    unsigned char* merged = (unsigned char*) calloc((size_t) n_meas + 1, 1);
    if (!merged) return s_fail("scene_merge: out of memory");
    out->num_correspondences = ncorr;
    /* :51-79 for all correspondences, in order; the scene size does not change */
    for (int c = 0; c < ncorr; ++c) {
      float* ps       = scene->pts + 4 * (size_t) corr[c].fixed_idx;
      const float* pm = meas->pts + 4 * (size_t) corr[c].moving_idx;
      if (!(corr[c].response < p->maximum_response)) continue; /* :60 */
      float q[4];
      xform_point(dim, M, pm, q); /* :62-63 */
      const float dx = q[0] - ps[0], dy = q[1] - ps[1], dz = q[2] - ps[2];
      const float d2 = (dx * dx + dy * dy) + dz * dz; /* :66-67 */
      if (!(d2 < p->maximum_distance_geometry_squared)) continue; /* :69 */
      /* :71 point_scene = point_meas (all fields; the normal stays in the measurement frame), :74 mean coordinates */
      float* ns = scene->nrm + 4 * (size_t) corr[c].fixed_idx;
      if (meas->has_normals) {
        memcpy(ns, meas->nrm + 4 * (size_t) corr[c].moving_idx, 4 * sizeof(float));
      } else {
        ns[0] = ns[1] = ns[2] = ns[3] = 0.f;
      }
      ps[0] = (q[0] + ps[0]) * 0.5f;
      ps[1] = (q[1] + ps[1]) * 0.5f;
      ps[2] = dim == 3 ? (q[2] + ps[2]) * 0.5f : 0.f;
      merged[corr[c].moving_idx] = 1; /* :75 */
    }
    int num_merged = 0;
    for (int i = 0; i < n_meas; ++i) num_merged += merged[i];
    out->num_merged = num_merged;
    /* :92-115 merge target not reached: add the unmerged Valid measurement points, in index order */
    if ((unsigned) num_merged < (unsigned) p->target_number_of_merges) {
      for (int i = 0; i < n_meas; ++i) {
        if (merged[i]) continue;
        if (!valid_point(dim, meas->pts + 4 * (size_t) i)) continue;
        if ((rc = append_point(scene, meas, M, i))) {
          free(merged);
          return rc;
        }
        out->num_added++;
      }
    }
    free(merged);
  }
  out->scene_size = scene->n;
  out->status     = SRRG2_MERGER_SUCCESS; /* :122 */
  return 0;
}
