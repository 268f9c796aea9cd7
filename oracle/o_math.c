/*
 * o_math.c -- deterministic scalar / SE(2) / SE(3) math of the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: srrg2_core's geometry2d/3d
 * (t2v, v2t, fixTransform; used at S/registration/aligners/multi_aligner_impl.cpp:92 and
 * T/test_motion_model_slice.cpp:81) is not vendored under /root/reference, so these are
 * first-principles definitions (SURVEY.md section 8c, table row "factor arithmetic ... t2v").
 *
 * Everything here uses only + - * / sqrt on IEEE doubles in a fixed operation order and is
 * compiled with -ffp-contract=off, so a device implementation following the same order is
 * bit-identical.  sin/cos/atan are evaluated with fixed polynomials (classic fdlibm-style
 * argument reduction + minimax kernels) instead of libm, for the same reason.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

/* ---- sin / cos ------------------------------------------------------------ */
static const double INV_PIO2 = 6.36619772367581382433e-01;
static const double PIO2_HI  = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
static const double PIO2_LO  = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI */

static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

void o_sincos(double x, double* s, double* c) {
  /* valid for |x| < ~1e5 (the callers pass Gauss-Newton increments) */
  double kx = x * INV_PIO2;
  long long k = (long long) (kx + (kx >= 0.0 ? 0.5 : -0.5));
  double kd = (double) k;
  double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
  double z = r * r;
  double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
  double sn = r + (r * z) * ps;
  double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
  double cs = (1.0 - 0.5 * z) + (z * z) * pc;
  switch ((int) (k & 3)) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

/* ---- atan / atan2 --------------------------------------------------------- */
static const double ATANHI[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                 9.82793723247329054082e-01, 1.57079632679489655800e+00};
static const double ATANLO[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                 1.39033110312309984516e-17, 6.12323399573676603587e-17};
static const double AT[11]    = {3.33333333333329318027e-01,  -1.99999999998764832476e-01,
                                 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                                 9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                                 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                                 4.97687799461593236017e-02,  -3.65315727442169155270e-02,
                                 1.62858201153657823623e-02};
static const double PI_HI = 3.14159265358979311600e+00;
static const double PI_LO = 1.22464679914735317720e-16;

static double o_atan_pos(double x) { /* x >= 0 */
  int id;
  if (x < 0.4375) {
    id = -1;
  } else if (x < 1.1875) {
    if (x < 0.6875) {
      id = 0;
      x  = (2.0 * x - 1.0) / (2.0 + x);
    } else {
      id = 1;
      x  = (x - 1.0) / (x + 1.0);
    }
  } else if (x < 2.4375) {
    id = 2;
    x  = (x - 1.5) / (1.0 + 1.5 * x);
  } else {
    id = 3;
    x  = -1.0 / x;
  }
  double z  = x * x;
  double w  = z * z;
  double s1 = z * (AT[0] + w * (AT[2] + w * (AT[4] + w * (AT[6] + w * (AT[8] + w * AT[10])))));
  double s2 = w * (AT[1] + w * (AT[3] + w * (AT[5] + w * (AT[7] + w * AT[9]))));
  if (id < 0) {
    return x - x * (s1 + s2);
  }
  return ATANHI[id] - ((x * (s1 + s2) - ATANLO[id]) - x);
}

double o_atan2(double y, double x) {
  if (x == 0.0 && y == 0.0) {
    return 0.0;
  }
  double ay = y < 0.0 ? -y : y;
  double ax = x < 0.0 ? -x : x;
  double z;
  if (ax == 0.0) {
    z = 0.5 * PI_HI;
  } else {
    z = o_atan_pos(ay / ax);
    if (x < 0.0) {
      z = PI_HI - (z - PI_LO);
    }
  }
  return y < 0.0 ? -z : z;
}

/* ---- SE(3): row-major 3x4 [R|t] ------------------------------------------- */
#define R_(T, i, j) ((T)[(i) *4 + (j)])
#define T_(T, i) ((T)[(i) *4 + 3])

void o_se3_compose(const float* A, const float* B, float* C) {
  float out[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double v = ((double) R_(A, i, 0) * (double) R_(B, 0, j) + (double) R_(A, i, 1) * (double) R_(B, 1, j)) +
                 (double) R_(A, i, 2) * (double) R_(B, 2, j);
      out[i * 4 + j] = (float) v;
    }
    double t = (((double) R_(A, i, 0) * (double) T_(B, 0) + (double) R_(A, i, 1) * (double) T_(B, 1)) +
                (double) R_(A, i, 2) * (double) T_(B, 2)) +
               (double) T_(A, i);
    out[i * 4 + 3] = (float) t;
  }
  memcpy(C, out, sizeof(out));
}

void o_se3_inverse(const float* A, float* Ainv) {
  float out[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      out[i * 4 + j] = R_(A, j, i);
    }
    double t = ((double) R_(A, 0, i) * (double) T_(A, 0) + (double) R_(A, 1, i) * (double) T_(A, 1)) +
               (double) R_(A, 2, i) * (double) T_(A, 2);
    out[i * 4 + 3] = (float) (-t);
  }
  memcpy(Ainv, out, sizeof(out));
}

/* ---- SE(2): row-major 3x3 homogeneous ------------------------------------- */
void o_se2_compose(const float* A, const float* B, float* C) {
  float out[9];
  for (int i = 0; i < 2; ++i) {
    for (int j = 0; j < 2; ++j) {
      double v = (double) A[i * 3 + 0] * (double) B[0 * 3 + j] + (double) A[i * 3 + 1] * (double) B[1 * 3 + j];
      out[i * 3 + j] = (float) v;
    }
    double t = ((double) A[i * 3 + 0] * (double) B[0 * 3 + 2] + (double) A[i * 3 + 1] * (double) B[1 * 3 + 2]) +
               (double) A[i * 3 + 2];
    out[i * 3 + 2] = (float) t;
  }
  out[6] = 0.f;
  out[7] = 0.f;
  out[8] = 1.f;
  memcpy(C, out, sizeof(out));
}

void o_se2_inverse(const float* A, float* Ainv) {
  float out[9];
  out[0] = A[0];
  out[1] = A[3];
  out[3] = A[1];
  out[4] = A[4];
  double tx = (double) A[0] * (double) A[2] + (double) A[3] * (double) A[5];
  double ty = (double) A[1] * (double) A[2] + (double) A[4] * (double) A[5];
  out[2]    = (float) (-tx);
  out[5]    = (float) (-ty);
  out[6]    = 0.f;
  out[7]    = 0.f;
  out[8]    = 1.f;
  memcpy(Ainv, out, sizeof(out));
}

/* ---- exponential-like maps -------------------------------------------------- */
static void quat_to_R(double w, double x, double y, double z, double* R) {
  double xx = x * x, yy = y * y, zz = z * z;
  double xy = x * y, xz = x * z, yz = y * z;
  double wx = w * x, wy = w * y, wz = w * z;
  R[0] = 1.0 - 2.0 * (yy + zz);
  R[1] = 2.0 * (xy - wz);
  R[2] = 2.0 * (xz + wy);
  R[3] = 2.0 * (xy + wz);
  R[4] = 1.0 - 2.0 * (xx + zz);
  R[5] = 2.0 * (yz - wx);
  R[6] = 2.0 * (xz - wy);
  R[7] = 2.0 * (yz + wx);
  R[8] = 1.0 - 2.0 * (xx + yy);
}

void o_se3_v2t(int variable_kind, const double* v, double* R, double* t) {
  t[0] = v[0];
  t[1] = v[1];
  t[2] = v[2];
  if (variable_kind == SRRG2_SE3_QUAT_RIGHT) {
    /* v[3..5] = imaginary part of a unit quaternion with w >= 0 */
    double n2 = (v[3] * v[3] + v[4] * v[4]) + v[5] * v[5];
    if (n2 < 1.0) {
      double w = sqrt(1.0 - n2);
      quat_to_R(w, v[3], v[4], v[5], R);
    } else {
      R[0] = R[4] = R[8] = 1.0;
      R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0.0;
    }
  } else {
    /* Euler: R = Rx(a) Ry(b) Rz(c) */
    double sa, ca, sb, cb, sc, cc;
    o_sincos(v[3], &sa, &ca);
    o_sincos(v[4], &sb, &cb);
    o_sincos(v[5], &sc, &cc);
    R[0] = cb * cc;
    R[1] = -(cb * sc);
    R[2] = sb;
    R[3] = ca * sc + (sa * sb) * cc;
    R[4] = ca * cc - (sa * sb) * sc;
    R[5] = -(sa * cb);
    R[6] = sa * sc - (ca * sb) * cc;
    R[7] = sa * cc + (ca * sb) * sc;
    R[8] = ca * cb;
  }
}

/* rotation matrix (double, row-major 3x3) -> unit quaternion (w,x,y,z), w >= 0 */
static void R_to_quat(const double* R, double* q) {
  double tr = (R[0] + R[4]) + R[8];
  double w, x, y, z;
  if (tr > 0.0) {
    double s = sqrt(tr + 1.0) * 2.0;
    w        = 0.25 * s;
    x        = (R[7] - R[5]) / s;
    y        = (R[2] - R[6]) / s;
    z        = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    double s = sqrt(((1.0 + R[0]) - R[4]) - R[8]) * 2.0;
    w        = (R[7] - R[5]) / s;
    x        = 0.25 * s;
    y        = (R[1] + R[3]) / s;
    z        = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    double s = sqrt(((1.0 + R[4]) - R[0]) - R[8]) * 2.0;
    w        = (R[2] - R[6]) / s;
    x        = (R[1] + R[3]) / s;
    y        = 0.25 * s;
    z        = (R[5] + R[7]) / s;
  } else {
    double s = sqrt(((1.0 + R[8]) - R[0]) - R[4]) * 2.0;
    w        = (R[3] - R[1]) / s;
    x        = (R[2] + R[6]) / s;
    y        = (R[5] + R[7]) / s;
    z        = 0.25 * s;
  }
  double n = sqrt(((w * w + x * x) + y * y) + z * z);
  w        = w / n;
  x        = x / n;
  y        = y / n;
  z        = z / n;
  if (w < 0.0) {
    w = -w;
    x = -x;
    y = -y;
    z = -z;
  }
  q[0] = w;
  q[1] = x;
  q[2] = y;
  q[3] = z;
}

void o_se3_t2v_quat(const float* T, double* v) {
  double R[9], q[4];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      R[i * 3 + j] = (double) R_(T, i, j);
    }
  }
  R_to_quat(R, q);
  v[0] = (double) T_(T, 0);
  v[1] = (double) T_(T, 1);
  v[2] = (double) T_(T, 2);
  v[3] = q[1];
  v[4] = q[2];
  v[5] = q[3];
}

void o_se2_t2v(const float* T, double* v) {
  v[0] = (double) T[2];
  v[1] = (double) T[5];
  v[2] = o_atan2((double) T[3], (double) T[0]);
}

void o_se3_fix_transform(float* T) {
  double R[9], q[4], Rn[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      R[i * 3 + j] = (double) R_(T, i, j);
    }
  }
  R_to_quat(R, q);
  quat_to_R(q[0], q[1], q[2], q[3], Rn);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      R_(T, i, j) = (float) Rn[i * 3 + j];
    }
  }
}

void o_se2_fix_transform(float* T) {
  double c = (double) T[0], s = (double) T[3];
  double n = sqrt(c * c + s * s);
  if (n > 0.0) {
    c = c / n;
    s = s / n;
  } else {
    c = 1.0;
    s = 0.0;
  }
  T[0] = (float) c;
  T[1] = (float) (-s);
  T[3] = (float) s;
  T[4] = (float) c;
  T[6] = 0.f;
  T[7] = 0.f;
  T[8] = 1.f;
}

void o_box_plus(int variable_kind, float* X, const double* dx) {
  if (variable_kind == SRRG2_SE2_RIGHT) {
    double s, c;
    o_sincos(dx[2], &s, &c);
    double D[6] = {c, -s, dx[0], s, c, dx[1]};
    float out[9];
    for (int i = 0; i < 2; ++i) {
      double a0 = (double) X[i * 3 + 0], a1 = (double) X[i * 3 + 1], a2 = (double) X[i * 3 + 2];
      out[i * 3 + 0] = (float) (a0 * D[0] + a1 * D[3]);
      out[i * 3 + 1] = (float) (a0 * D[1] + a1 * D[4]);
      out[i * 3 + 2] = (float) ((a0 * D[2] + a1 * D[5]) + a2);
    }
    out[6] = 0.f;
    out[7] = 0.f;
    out[8] = 1.f;
    memcpy(X, out, sizeof(out));
    return;
  }
  double R[9], t[3];
  o_se3_v2t(variable_kind, dx, R, t);
  float out[12];
  for (int i = 0; i < 3; ++i) {
    double a0 = (double) R_(X, i, 0), a1 = (double) R_(X, i, 1), a2 = (double) R_(X, i, 2);
    for (int j = 0; j < 3; ++j) {
      out[i * 4 + j] = (float) ((a0 * R[0 * 3 + j] + a1 * R[1 * 3 + j]) + a2 * R[2 * 3 + j]);
    }
    out[i * 4 + 3] = (float) (((a0 * t[0] + a1 * t[1]) + a2 * t[2]) + (double) T_(X, i));
  }
  memcpy(X, out, sizeof(out));
}

/* ---- dense solve H dx = -b (H symmetric positive definite) -------------------
 * Square-root-free L D L^T, one division per column and none in the substitutions: the operation order is the
 * specification shared with the GPU's control step (srrg2_slam_interfaces_amd/csrc/det_math.h, dm::solve: statement
 * for statement).  Returns 1 for a pivot <= 0 or a non-finite solution.
 * ([EXT] the reference's Solver is srrg2_solver's, not under /root/reference: parity unpinned) */
int o_solve(int D, const double* H, const double* b, double* dx) {
  double L[36];
  double d[6], inv[6], y[6];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < D; ++j) {
    double s = H[j * D + j];
    for (int k = 0; k < j; ++k) {
      s = s - (L[j * D + k] * L[j * D + k]) * d[k];
    }
    if (!(s > 0.0)) {
      return 1;
    }
    d[j]   = s;
    inv[j] = 1.0 / s;
    for (int i = j + 1; i < D; ++i) {
      double v = H[i * D + j];
      for (int k = 0; k < j; ++k) {
        v = v - (L[i * D + k] * L[j * D + k]) * d[k];
      }
      L[i * D + j] = v * inv[j];
    }
  }
  for (int i = 0; i < D; ++i) {
    double s = -b[i];
    for (int k = 0; k < i; ++k) {
      s = s - L[i * D + k] * y[k];
    }
    y[i] = s;
  }
  for (int i = D - 1; i >= 0; --i) {
    double s = y[i] * inv[i];
    for (int k = i + 1; k < D; ++k) {
      s = s - L[k * D + i] * dx[k];
    }
    dx[i] = s;
  }
  for (int i = 0; i < D; ++i) {
    if (!(dx[i] == dx[i]) || dx[i] > 1e300 || dx[i] < -1e300) {
      return 1;
    }
  }
  return 0;
}

/* ---- fixed-point exponent ---------------------------------------------------- */
static int ceil_log2_double(double v) { /* smallest e with 2^e >= v, v > 0 finite */
  uint64_t bits;
  memcpy(&bits, &v, sizeof(bits));
  int e          = (int) ((bits >> 52) & 0x7ff) - 1023;
  uint64_t mant  = bits & 0xfffffffffffffULL;
  if (mant != 0) {
    e += 1;
  }
  return e;
}

int o_fixed_point_exponent(int n_terms, double term_bound) {
  if (n_terms < 1) {
    n_terms = 1;
  }
  if (!(term_bound > 1e-30)) {
    term_bound = 1e-30;
  }
  int lb = ceil_log2_double(term_bound);
  int k  = 62 - ceil_log2_double((double) n_terms) - lb;
  if (k > 50 - lb) { /* every scaled term stays below 2^50: exact magic-number double->int conversion */
    k = 50 - lb;
  }
  if (k > 50) {
    k = 50;
  }
  if (k < -64) {
    k = -64;
  }
  return k;
}
