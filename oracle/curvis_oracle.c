/* curvis_oracle.c -- CPU oracle for the CurVis per-pixel geodesic hot path.
 * TEST INFRASTRUCTURE ONLY -- see curvis_oracle.h for scope, flavours and parity status.
 *
 * Build: gcc -O2 -std=c11 -mfma -ffp-contract=off (oracle/Makefile).  -ffp-contract=off is
 * mandatory: Rust never contracts a*b+c, and the only fused operations allowed are the
 * explicit fma() calls inside cv_math.h (CVO_CV flavour).
 */
#define _GNU_SOURCE
#include "curvis_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../curvis_amd/csrc/cv_math.h" /* elementary functions only (CVO_CV flavour) */

#define RUST_PI 3.14159265358979311600e+00 /* std::f64::consts::PI */

/* ------------------------------------------------------------------ Rust scalar semantics */
static uint32_t rust_as_u32(double v) { /* `as u32`: NaN -> 0, saturating, truncating */
  if (!(v == v)) return 0u;
  if (v <= 0.0) return 0u;
  if (v >= 4294967295.0) return 4294967295u;
  return (uint32_t)v;
}
static double rust_rem_euclid(double a, double b) { /* f64::rem_euclid */
  double r = fmod(a, b);
  return (r < 0.0) ? r + fabs(b) : r;
}
static double rust_signum(double v) { /* f64::signum */
  if (v != v) return v;
  return signbit(v) ? -1.0 : 1.0;
}

/* ------------------------------------------------------------------ nalgebra 0.33.0 restatements
 * (crate not under /root/reference; arithmetic order from the published source):
 *   Vector3 dot: a0*b0 + a1*b1 + a2*b2, left to right (base/blas.rs, U3 special case)
 *   norm = sqrt(dot(v,v));  normalize = component / norm
 *   cross: (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx)
 *   M*v and M*N: column-wise gemv, y_i = ((m_i0*x0) + m_i1*x1) + m_i2*x2
 */
static double vec3_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double vec3_norm(const double a[3]) { return sqrt(vec3_dot(a, a)); }
static void vec3_cross(const double a[3], const double b[3], double c[3]) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  c[0] = x;
  c[1] = y;
  c[2] = z;
}
static void vec3_normalize(const double a[3], double o[3]) {
  double n = vec3_norm(a);
  o[0] = a[0] / n;
  o[1] = a[1] / n;
  o[2] = a[2] / n;
}
static void mat3_identity(double m[9]) {
  for (int i = 0; i < 9; ++i) m[i] = 0.0;
  m[0] = m[4] = m[8] = 1.0;
}
static void mat3_vec(const double m[9], const double v[3], double out[3]) {
  double o0 = (m[0] * v[0] + m[1] * v[1]) + m[2] * v[2];
  double o1 = (m[3] * v[0] + m[4] * v[1]) + m[5] * v[2];
  double o2 = (m[6] * v[0] + m[7] * v[1]) + m[8] * v[2];
  out[0] = o0;
  out[1] = o1;
  out[2] = o2;
}
static void mat3_mul(const double a[9], const double b[9], double c[9]) {
  double t[9];
  for (int j = 0; j < 3; ++j) {
    double col[3] = {b[0 + j], b[3 + j], b[6 + j]}, o[3];
    mat3_vec(a, col, o);
    t[0 + j] = o[0];
    t[3 + j] = o[1];
    t[6 + j] = o[2];
  }
  memcpy(c, t, sizeof t);
}
static void mat3_transpose(const double a[9], double t[9]) {
  double r[9] = {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]};
  memcpy(t, r, sizeof r);
}

/* Rotation3::face_towards(dir, up): columns [x y z], z = dir.normalize(),
 * x = up.cross(z).normalize(), y = z.cross(x).normalize() */
void cvo_face_towards(const double dir[3], const double up[3], double m[9]) {
  double z[3], x[3], y[3], t[3];
  vec3_normalize(dir, z);
  vec3_cross(up, z, t);
  vec3_normalize(t, x);
  vec3_cross(z, x, t);
  vec3_normalize(t, y);
  m[0] = x[0];
  m[1] = y[0];
  m[2] = z[0];
  m[3] = x[1];
  m[4] = y[1];
  m[5] = z[1];
  m[6] = x[2];
  m[7] = y[2];
  m[8] = z[2];
}

/* src/algebra.rs:16-38 Orientation::new + :64-74 rotation_matrix_from_forward_up_pairs */
int cvo_orientation_new(const double fwd[3], const double up[3], double rot[9], double inv_rot[9], double up_out[3]) {
  double c[3];
  vec3_cross(fwd, up, c);
  if (vec3_norm(c) == 0.0) return -1; /* panic!("Forward and up vectors must not be parallel") */
  const double ex[3] = {1.0, 0.0, 0.0}, ez[3] = {0.0, 0.0, 1.0};
  double r1[9], r2[9], r1i[9];
  cvo_face_towards(ex, ez, r1);
  cvo_face_towards(fwd, up, r2);
  mat3_transpose(r1, r1i); /* Rotation::inverse = transpose */
  mat3_mul(r2, r1i, rot);
  mat3_transpose(rot, inv_rot);
  if (up_out) mat3_vec(rot, ez, up_out);
  return 0;
}

/* src/algebra.rs:106-116 */
static void normalize_theta_phi(double theta, double phi, double *to, double *po) {
  if (theta < 0.0) {
    theta = fabs(theta);
    phi = phi + RUST_PI;
  }
  *to = theta;
  *po = rust_rem_euclid(phi, 2.0 * RUST_PI);
}
void cvo_normalize_theta_phi(double theta, double phi, double *to, double *po) {
  normalize_theta_phi(theta, phi, to, po);
}
void cvo_mat3_vec(const double m[9], const double v[3], double out[3]) { mat3_vec(m, v, out); }

/* ------------------------------------------------------------------ src/cameras.rs */
int cvo_camera_new(cvo_camera *c, const double pos[4], const double fwd[3], const double up[3], double focal,
                   double diag, uint32_t rx, uint32_t ry) {
  if (focal <= 0.0) return -2;       /* :92 */
  if (diag <= 0.0) return -3;        /* :95 */
  if (rx == 0 || ry == 0) return -4; /* :98 */
  if (cvo_orientation_new(fwd, up, c->rot, c->inv_rot, c->up) != 0) return -1;
  for (int i = 0; i < 4; ++i) c->pos[i] = pos[i];
  for (int i = 0; i < 3; ++i) c->fwd[i] = fwd[i];
  double aspect = (double)rx / (double)ry; /* :107-110 */
  double aspect2 = aspect * aspect;
  c->sensor_h = sqrt(diag * diag / (aspect2 + 1.0));
  c->sensor_w = aspect * c->sensor_h;
  c->focal = focal;
  c->res_x = rx;
  c->res_y = ry;
  return 0;
}
void cvo_camera_outward_camera_space(const cvo_camera *c, uint32_t px, uint32_t py, double out[3]) { /* :150-164 */
  double res_x = (double)c->res_x, res_y = (double)c->res_y;
  double h = 0.5 - ((double)py / res_y);
  double w = ((double)px / res_x) - 0.5;
  double v[3] = {c->focal * 1.0, -c->sensor_w * w, c->sensor_h * h};
  vec3_normalize(v, out);
}
void cvo_camera_outward_world(const cvo_camera *c, uint32_t px, uint32_t py, double out[3]) { /* :169-172 */
  double v[3];
  cvo_camera_outward_camera_space(c, px, py, v);
  mat3_vec(c->rot, v, out);
}

/* ------------------------------------------------------------------ interp 1.0.3: interp_slice
 * (crate not under /root/reference).  m_i = dy_i/dx_i, c_i = y_i - x_i*m_i,
 * i = min(prev_index(x, xp), n-2), prev_index = index of last leading element < xp (0 if none),
 * result m_i*xp + c_i (linear extrapolation outside the table). */
void cvo_interp_slice(const double *x, const double *y, size_t n, const double *xp, size_t np, double *out) {
  if (n == 0) {
    for (size_t k = 0; k < np; ++k) out[k] = 0.0;
    return;
  }
  if (n == 1) {
    for (size_t k = 0; k < np; ++k) out[k] = y[0];
    return;
  }
  double *m = (double *)malloc((n - 1) * sizeof(double));
  double *c = (double *)malloc((n - 1) * sizeof(double));
  for (size_t i = 0; i + 1 < n; ++i) {
    double dx = x[i + 1] - x[i], dy = y[i + 1] - y[i];
    m[i] = dy / dx;
    c[i] = y[i] - x[i] * m[i];
  }
  for (size_t k = 0; k < np; ++k) {
    double q = xp[k];
    size_t cnt = 0; /* take_while(x < xp) */
    while (cnt < n && x[cnt] < q) ++cnt;
    size_t i = cnt ? cnt - 1 : 0;
    if (i > n - 2) i = n - 2;
    out[k] = m[i] * q + c[i];
  }
  free(m);
  free(c);
}

void cvo_samples_free(cvo_samples *s) {
  if (!s) return;
  free(s->a);
  free(s->e);
  free(s->s);
  s->a = s->e = s->s = NULL;
  s->n = 0;
}

/* ------------------------------------------------------------------ flavour instantiations */
static void sincos_two_calls(double x, double *s, double *c) { /* llvm.sin.f64 and llvm.cos.f64 left as two libcalls */
  *s = sin(x);
  *c = cos(x);
}
static void sincos_glibc(double x, double *s, double *c) { sincos(x, s, c); } /* ... merged into one sincos() libcall */

/* 1 (default): the render loops step with update_memo (every metric function once per step: identical bits, 2-3 x fewer libm
 * calls for the Interstellar metric); 0: with the literal update, call for call what the reference executes.  See
 * curvis_oracle_impl.inc.  Not for use while renders are running on other threads. */
static int g_metric_memo = 1;
void cvo_set_metric_memo(int on) { g_metric_memo = on != 0; }
int cvo_get_metric_memo(void) { return g_metric_memo; }

#define M_FRAME(name) F(name)
#define F(name) name##_libm
#define M_SIN sin
#define M_COS cos
#define M_ACOS acos
#define M_ATAN atan
#define M_ATAN2 atan2
#define M_LOG log
#define M_SINCOS sincos_two_calls
#define M_SIN_INL(x, s) sin(x)
#include "curvis_oracle_impl.inc"
#undef F
#undef M_SINCOS
#undef M_SIN_INL

/* CVO_LIBM_SINCOS: sin and cos of one value inside one reference function come from ONE glibc sincos() call (what
 * LLVM's FSIN+FCOS -> FSINCOS combine produces on *-linux-gnu when both land in one block); every lone sin stays sin */
#define F(name) name##_sc
#define M_SINCOS sincos_glibc
#define M_SIN_INL(x, s) sin(x)
#include "curvis_oracle_impl.inc"
#undef F
#undef M_SIN_INL

/* CVO_LIBM_SINCOS_INL: as above, and g33's theta.sin() (src/metrics.rs:68) is the sincos() sine too -- the case in
 * which object_position_diff_contr and object_momentum_diff_cov are both inlined into update_relativistic_object
 * and the three llvm.sin.f64(theta) are CSE'd before the combine */
#define F(name) name##_sci
#define M_SIN_INL(x, s) (s)
#include "curvis_oracle_impl.inc"
#undef F
#undef M_SIN
#undef M_COS
#undef M_ACOS
#undef M_ATAN
#undef M_ATAN2
#undef M_LOG
#undef M_SINCOS
#undef M_SIN_INL

#undef M_FRAME
#define M_FRAME(name) name##_sc /* the product's cv_frame_host.h: glibc, one sincos() per pair */
#define F(name) name##_cv
#define M_SIN cv_sin
#define M_COS cv_cos
#define M_ACOS cv_acos
#define M_ATAN cv_atan
#define M_ATAN2 cv_atan2
#define M_LOG cv_log
#define M_SINCOS cv_sincos /* cv_sin(x) and cv_cos(x) ARE the two results of cv_sincos(x) */
#define M_SIN_INL(x, s) cv_sin(x)
#include "curvis_oracle_impl.inc"
#undef F
#undef M_SIN
#undef M_COS
#undef M_ACOS
#undef M_ATAN
#undef M_ATAN2
#undef M_LOG
#undef M_SINCOS
#undef M_SIN_INL
#undef M_FRAME

#define DISPATCH(fl, name, ...)                                   \
  ((fl) == CVO_CV                 ? name##_cv(__VA_ARGS__)        \
   : (fl) == CVO_LIBM_SINCOS      ? name##_sc(__VA_ARGS__)        \
   : (fl) == CVO_LIBM_SINCOS_INL  ? name##_sci(__VA_ARGS__)       \
                                  : name##_libm(__VA_ARGS__))

/* the six elementary functions of a flavour over arrays (op: 0 sin, 1 cos, 2 atan, 3 acos, 4 log, 5 atan2(a, b)):
 * lets the tests compare cv_math.h with glibc -- an independent libm -- on the very arguments the Euler loop
 * produces, not only on random ones */
void cvo_math_array(int fl, int op, const double *a, const double *b, double *out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const double x = a[i], y = b ? b[i] : 0.0;
    double r;
    if (fl == CVO_CV) {
      r = op == 0 ? cv_sin(x) : op == 1 ? cv_cos(x) : op == 2 ? cv_atan(x) : op == 3 ? cv_acos(x) : op == 4 ? cv_log(x) : cv_atan2(x, y);
    } else {
      r = op == 0 ? sin(x) : op == 1 ? cos(x) : op == 2 ? atan(x) : op == 3 ? acos(x) : op == 4 ? log(x) : atan2(x, y);
    }
    out[i] = r;
  }
}

/* compute_photon_trajectory (src/systems.rs:77-92): out[k][0..7] = (x, p_cov) BEFORE the k-th Euler step */
void cvo_photon_trajectory(int fl, const cvo_metric *m, const double x0[4], const double p0[4], uint32_t iterations,
                           double delta, double *out) {
  double x[4], p[4];
  memcpy(x, x0, sizeof x);
  memcpy(p, p0, sizeof p);
  for (uint32_t k = 0; k < iterations; ++k) {
    memcpy(out + 8 * (size_t)k, x, sizeof x);
    memcpy(out + 8 * (size_t)k + 4, p, sizeof p);
    DISPATCH(fl, update, m, x, p, delta);
  }
}

int cvo_rotation_between(int fl, const double a[3], const double b[3], double m[9]) {
  return DISPATCH(fl, rotation_between, a, b, m);
}
int cvo_rotation_from_two_vectors(int fl, const double a[3], const double b[3], double m[9]) {
  return DISPATCH(fl, rotation_from_two_vectors, a, b, m);
}
void cvo_from_axis_angle(int fl, const double u[3], double angle, double m[9]) {
  DISPATCH(fl, from_axis_angle, u, angle, m);
}
void cvo_vector3_from_theta_phi(int fl, double theta, double phi, double v[3]) {
  DISPATCH(fl, vector3_from_theta_phi, theta, phi, v);
}
void cvo_theta_phi_from_vector3(int fl, const double v[3], double *theta, double *phi) {
  DISPATCH(fl, theta_phi_from_vector3, v, theta, phi);
}
double cvo_metric_r(int fl, const cvo_metric *m, double l) { return DISPATCH(fl, r, m, l); }
double cvo_metric_r_squared(int fl, const cvo_metric *m, double l) { return DISPATCH(fl, r_squared, m, l); }
double cvo_metric_r_derivative(int fl, const cvo_metric *m, double l) { return DISPATCH(fl, r_derivative, m, l); }
void cvo_new_photon(int fl, const cvo_metric *m, const double pos[4], const double dir[3], double x[4], double p[4]) {
  DISPATCH(fl, new_photon, m, pos, dir, x, p);
}
void cvo_update(int fl, const cvo_metric *m, double x[4], double p[4], double delta) { /* always the literal step */
  DISPATCH(fl, update, m, x, p, delta);
}
void cvo_update_memo(int fl, const cvo_metric *m, double x[4], double p[4], double delta) { /* every metric function once */
  DISPATCH(fl, update_memo, m, x, p, delta);
}
int cvo_escape_photon(int fl, const cvo_metric *m, double x[4], double p[4], double delta, uint32_t max_iter,
                      double max_radius, uint32_t *steps) {
  return DISPATCH(fl, escape_photon, m, x, p, delta, max_iter, max_radius, steps);
}
void cvo_vector_to_direction(int fl, const cvo_metric *m, const double p_cov[4], const double x[4], double dir[3]) {
  DISPATCH(fl, vector_to_direction, m, p_cov, x, dir);
}
double cvo_squared_norm_cov(int fl, const cvo_metric *m, const double p_cov[4], const double x[4]) {
  return DISPATCH(fl, squared_norm_cov, m, p_cov, x);
}
void cvo_sky_indices(int fl, const cvo_sky *s, const double v[3], uint32_t *x, uint32_t *y) {
  DISPATCH(fl, sky_indices, s, v, x, y);
}
int cvo_sky_pixel(int fl, const cvo_sky *s, const double v[3], uint8_t rgba[4]) {
  return DISPATCH(fl, sky_pixel, s, v, rgba, NULL, NULL);
}
int cvo_render_image(int fl, const cvo_metric *m, const cvo_camera *c, const cvo_sky *pos, const cvo_sky *neg,
                     uint32_t max_iter, double max_radius, double delta, uint32_t row_begin, uint32_t row_step,
                     uint8_t *rgb, cvo_ray_debug *dbg, cvo_stats *stats) {
  return DISPATCH(fl, render_image, m, c, pos, neg, max_iter, max_radius, delta, row_begin, row_step, rgb, dbg,
                  stats);
}
int cvo_compute_escape_angle(int fl, const cvo_metric *m, double l, double alpha, double delta, uint32_t max_iter,
                             double max_radius, double *angle, uint32_t *steps) {
  return DISPATCH(fl, compute_escape_angle, m, l, alpha, delta, max_iter, max_radius, angle, steps);
}
int cvo_doubly_sample(int fl, const cvo_metric *m, double l, double delta, uint32_t max_iter, double max_radius,
                      double a_min, double a_max, size_t n0, size_t max_iterations, double thr1, double thr2,
                      cvo_samples *out) {
  return DISPATCH(fl, doubly_sample, m, l, delta, max_iter, max_radius, a_min, a_max, n0, max_iterations, thr1, thr2,
                  out);
}
int cvo_render_image_efficient(int fl, const cvo_metric *m, const cvo_camera *c, const cvo_sky *pos,
                               const cvo_sky *neg, uint32_t max_iter, double max_radius, double delta,
                               uint32_t alpha_nums, uint32_t max_iterations_sampling, double thr1, double thr2,
                               uint8_t *rgb, cvo_samples *samples_out, cvo_stats *stats) {
  return DISPATCH(fl, render_image_efficient, m, c, pos, neg, max_iter, max_radius, delta, alpha_nums,
                  max_iterations_sampling, thr1, thr2, rgb, samples_out, stats);
}

int cvo_render_image_direct(int fl, const cvo_metric *m, const cvo_camera *c, const cvo_sky *pos, const cvo_sky *neg,
                            uint32_t max_iter, double max_radius, double delta, uint8_t *rgb, cvo_stats *stats) {
  return DISPATCH(fl, render_image_direct, m, c, pos, neg, max_iter, max_radius, delta, rgb, stats);
}

/* ------------------------------------------------------------------ src/csv.rs:24-62 load_path */
static int parse_rust_f64(const char *s, size_t len, double *out) {
  /* str::parse::<f64>: no surrounding whitespace allowed, whole token must be consumed */
  if (len == 0 || len > 255) return -1;
  char buf[256];
  memcpy(buf, s, len);
  buf[len] = 0;
  if (buf[0] == ' ' || buf[0] == '\t' || buf[len - 1] == ' ' || buf[len - 1] == '\t' || buf[len - 1] == '\r') return -1;
  char *end = NULL;
  double v = strtod(buf, &end);
  if (end != buf + len) return -1;
  *out = v;
  return 0;
}

int cvo_load_path(const char *csv, cvo_path *out) {
  FILE *f = fopen(csv, "rb");
  if (!f) return -1;
  size_t cap = 1024, n = 0;
  out->pos = (double *)malloc(cap * 4 * sizeof(double));
  out->fwd = (double *)malloc(cap * 3 * sizeof(double));
  out->up = (double *)malloc(cap * 3 * sizeof(double));
  char *line = NULL;
  size_t lcap = 0;
  ssize_t got;
  size_t index = 0;
  int rc = 0;
  while ((got = getline(&line, &lcap, f)) >= 0) {
    if (index++ == 0) continue; /* header skipped */
    size_t len = (size_t)got;
    /* BufRead::lines strips "\n" and "\r\n" */
    if (len && line[len - 1] == '\n') --len;
    if (len && line[len - 1] == '\r') --len;
    double v[10];
    size_t start = 0;
    int k = 0;
    for (size_t i = 0; i <= len && k < 10; ++i) {
      if (i == len || line[i] == ',') {
        if (parse_rust_f64(line + start, i - start, &v[k]) != 0) {
          rc = -2;
          goto done;
        }
        ++k;
        start = i + 1;
      }
    }
    if (k < 10) {
      rc = -3;
      goto done;
    }
    if (n == cap) {
      cap *= 2;
      out->pos = (double *)realloc(out->pos, cap * 4 * sizeof(double));
      out->fwd = (double *)realloc(out->fwd, cap * 3 * sizeof(double));
      out->up = (double *)realloc(out->up, cap * 3 * sizeof(double));
    }
    memcpy(out->pos + 4 * n, v, 4 * sizeof(double));
    memcpy(out->fwd + 3 * n, v + 4, 3 * sizeof(double));
    memcpy(out->up + 3 * n, v + 7, 3 * sizeof(double));
    ++n;
  }
done:
  free(line);
  fclose(f);
  out->n = n;
  if (rc != 0) cvo_path_free(out);
  return rc;
}
void cvo_path_free(cvo_path *p) {
  free(p->pos);
  free(p->fwd);
  free(p->up);
  p->pos = p->fwd = p->up = NULL;
  p->n = 0;
}

/* src/interpolation.rs:63-112, including the (k, k+1)-with-frac-of-(k-1, k) off-by-one */
int cvo_path_camera(const cvo_path *p, double t, double pos[4], double fwd[3], double up[3]) {
  const size_t n = p->n;
  if (n == 0) return -2;
  const double min_time = p->pos[0], max_time = p->pos[4 * (n - 1)];
  if (t < min_time) return -1;
  if (t > max_time) return -1;
  double t1 = min_time, t2 = max_time;
  size_t i = 0;
  while (t > p->pos[4 * i]) {
    if (i + 1 >= n) return -2; /* positions[i + 1] out of bounds */
    t1 = p->pos[4 * i];
    t2 = p->pos[4 * (i + 1)];
    i += 1;
    if (i >= n) return -2;
  }
  double frac = (t - t1) / (t2 - t1);
  size_t i1 = i, i2 = i + 1;
  if (i2 >= n) return -2; /* index out of bounds panic */
  if (!(frac >= 0.0 && frac <= 1.0)) return -1; /* :35-37 */
  for (int k = 0; k < 4; ++k) { /* v1 + frac * (v2 - v1) */
    double a = p->pos[4 * i1 + k], b = p->pos[4 * i2 + k];
    pos[k] = a + frac * (b - a);
  }
  for (int k = 0; k < 3; ++k) {
    double a = p->fwd[3 * i1 + k], b = p->fwd[3 * i2 + k];
    fwd[k] = a + frac * (b - a);
    a = p->up[3 * i1 + k];
    b = p->up[3 * i2 + k];
    up[k] = a + frac * (b - a);
  }
  return 0;
}

/* src/rendering.rs:224-238 */
size_t cvo_times_of_frames(double min_time, double max_time, double frame_rate, double *out, size_t cap) {
  double delta_time = 1.0 / frame_rate;
  size_t n = 0;
  double t = min_time;
  while (t < max_time) {
    if (out && n < cap) out[n] = t;
    ++n;
    t += delta_time;
  }
  return n;
}
