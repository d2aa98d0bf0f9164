/* cv_efficient.h -- per-ray / per-pixel functions of the reference's "efficient" renderer
 * (RelativisticSystem::render_image_efficient, src/systems.rs:333-527): the variant the
 * `curvis image` / `curvis video` commands call (src/rendering.rs:97, :299).
 *
 *   E2  compute_escape_angle            src/systems.rs:203-261 (+ :144-187)
 *   E1  per-pixel steps 2, 4 and 5      src/systems.rs:405-433, :491-523
 *
 * nalgebra 0.33.0 / interp 1.0.3 arithmetic is restated in the evaluation order of those crates
 * (see DESIGN.md section 3); elementary functions are cv_math.h.  __host__ __device__ like
 * cv_device.h: the x86 build is a test vehicle and the host side of the sampler, never a
 * replacement for the kernels.
 */
#ifndef CURVIS_CV_EFFICIENT_H
#define CURVIS_CV_EFFICIENT_H

#include "cv_device.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace cvk {

CV_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
CV_HD double norm3(const double *a) { return CV_SQRT(dot3(a, a)); }
CV_HD void cross3(const double *a, const double *b, double *c) {
  const double x = a[1] * b[2] - a[2] * b[1];
  const double y = a[2] * b[0] - a[0] * b[2];
  const double z = a[0] * b[1] - a[1] * b[0];
  c[0] = x;
  c[1] = y;
  c[2] = z;
}
CV_HD void mat3_identity(double *m) {
  m[0] = 1.0; m[1] = 0.0; m[2] = 0.0;
  m[3] = 0.0; m[4] = 1.0; m[5] = 0.0;
  m[6] = 0.0; m[7] = 0.0; m[8] = 1.0;
}

/* Elementary functions of the rotations below.  CvMath (cv_math.h) is what every per-ray / per-pixel evaluation uses, on
 * the device and in its x86 twin.  The two once-per-frame host values of render_image_efficient (cam_bg, rot_bg:
 * src/systems.rs:393-397, :411) are instead taken over the platform libm like the reference takes them -- the policy for that
 * lives in efficient_host.h (host only). */
struct CvMath {
  CV_HD void sincos(double x, double *s, double *c) { cv_sincos(x, s, c); } /* CV_HD carries `static inline` */
  CV_HD double acos(double x) { return cv_acos(x); }
};

/* nalgebra Rotation3::from_axis_angle(&Unit(axis), angle): identity iff angle == 0 (NaN is != 0) */
template <class MT = CvMath>
CV_HD void from_axis_angle(const double *u, double angle, double *m) {
  if (angle != 0.0) {
    const double ux = u[0], uy = u[1], uz = u[2];
    const double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    double sn, cs;
    MT::sincos(angle, &sn, &cs);
    const double omc = 1.0 - cs;
    m[0] = sqx + (1.0 - sqx) * cs;
    m[1] = ux * uy * omc - uz * sn;
    m[2] = ux * uz * omc + uy * sn;
    m[3] = ux * uy * omc + uz * sn;
    m[4] = sqy + (1.0 - sqy) * cs;
    m[5] = uy * uz * omc - ux * sn;
    m[6] = ux * uz * omc - uy * sn;
    m[7] = uy * uz * omc + ux * sn;
    m[8] = sqz + (1.0 - sqz) * cs;
  } else {
    mat3_identity(m);
  }
}

/* nalgebra Rotation3::rotation_between(a, b); false == None (antiparallel) */
template <class MT = CvMath>
CV_HD bool rotation_between(const double *a, const double *b, double *m) {
  const double an = norm3(a), bn = norm3(b);
  if (!(an <= 0.0) && !(bn <= 0.0)) { /* try_normalize(0.0) */
    const double na[3] = {a[0] / an, a[1] / an, a[2] / an};
    const double nb[3] = {b[0] / bn, b[1] / bn, b[2] / bn};
    double c[3];
    cross3(na, nb, c);
    const double sq = dot3(c, c);
    const double eps = 2.220446049250313e-16; /* f64::EPSILON; Unit::try_new: norm^2 > eps^2 */
    if (sq > eps * eps) {
      const double n = CV_SQRT(sq);
      const double axis[3] = {c[0] / n, c[1] / n, c[2] / n};
      from_axis_angle<MT>(axis, MT::acos(dot3(na, nb)) * 1.0, m);
      return true;
    }
    if (dot3(na, nb) < 0.0) return false;
  }
  mat3_identity(m);
  return true;
}

/* src/algebra.rs:92-101: panics (false) when the cross product is exactly zero or nalgebra returns None */
template <class MT = CvMath>
CV_HD bool rotation_from_two_vectors(const double *v1, const double *v2, double *m) {
  double c[3];
  cross3(v1, v2, c);
  if (norm3(c) == 0.0) return false;
  return rotation_between<MT>(v1, v2, m);
}

/* f64::rem_euclid(a, b), b > 0, general */
CV_HD double rem_euclid_general(double a, double b) {
  const double r = (CV_FABS(a) < b || a != a) ? a : fmod(a, b);
  return (r < 0.0) ? r + b : r;
}

/* src/algebra.rs:106-126 */
template <class MT = CvMath>
CV_HD void vector3_from_theta_phi(double theta, double phi, double *v) {
  if (theta < 0.0) {
    theta = CV_FABS(theta);
    phi = phi + CV_PI;
  }
  phi = rem_euclid_general(phi, 2.0 * CV_PI);
  double st, ct, sp, cp;
  MT::sincos(theta, &st, &ct);
  MT::sincos(phi, &sp, &cp);
  v[0] = st * cp;
  v[1] = st * sp;
  v[2] = ct;
}

enum : int { ESC_PANIC = -2 };

/* tail of compute_escape_angle (src/systems.rs:246-259) for an escaped photon:
 * escaped_photon_to_world_direction (:144-187), normalise, angle from (vx, vy).
 * Returns false for the reference's panic (tangent rotation undefined). */
template <int KIND>
CV_HD bool escape_angle_of(const MetricParams &M, const Ray &q, double &angle) {
  double t[3];
  ray_direction<KIND>(M, q, t[0], t[1], t[2]);
  double wpos[3], rot[9];
  vector3_from_theta_phi(q.th, q.ph, wpos);
  const double ex[3] = {1.0, 0.0, 0.0}, ey[3] = {0.0, 1.0, 0.0};
  if (!rotation_from_two_vectors(ex, wpos, rot)) return false;
  double wd[3];
  mat3_vec(rot, t[0], t[1], t[2], wd[0], wd[1], wd[2]);
  const double n = norm3(wd); /* normalize_mut */
  wd[0] = wd[0] / n;
  wd[1] = wd[1] / n;
  wd[2] = wd[2] / n;
  const double vx = dot3(wd, ex);
  const double vy = dot3(wd, ey);
  angle = (vy >= 0.0) ? cv_acos(vx) : 2.0 * CV_PI - cv_acos(vx);
  return true;
}

/* per-frame constants of the efficient renderer, computed once on the host */
struct EfficientFrame {
  double cam_bg[3]; /* vector3_from_theta_phi(theta_cam, phi_cam), src/systems.rs:393-397 */
  double rot_bg[9]; /* rotation_from_two_vectors(x, cam_bg), src/systems.rs:411 (same for every pixel) */
};

/* interp 1.0.3 interp_slice for one query against precomputed slopes m / intercepts c:
 * i = min(prev_index(x, xp), n-2), prev_index = (number of leading x < xp) - 1, saturating.
 * x is strictly increasing here, so a binary search gives the same count as the crate's linear scan. */
CV_HD unsigned interp_index(const double *x, unsigned n, double xp) {
  unsigned lo = 0, hi = n; /* first index with !(x[i] < xp); NaN xp -> 0 */
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (x[mid] < xp)
      lo = mid + 1;
    else
      hi = mid;
  }
  unsigned i = lo ? lo - 1 : 0;
  if (i > n - 2) i = n - 2;
  return i;
}

/* The same index through a bucket grid over the abscissae (the per-pixel kernel: a table has 300 .. 1500 samples, and eleven
 * dependent loads and compares per pixel were 7 % of its instructions).  interp_bucket is monotone non-decreasing in its argument
 * -- a rounded subtraction, a rounded product, a clamp and a truncation all are -- so for a query xp in bucket b every sample in a
 * lower bucket is < xp and every sample in a higher one is > xp: the index sought, #{i : x[i] < xp}, lies between
 * G[b] = #{i : bucket(x[i]) < b} and G[b + 1], and the binary search runs over that span only -- EXACTLY the index of the full
 * search, whatever the rounding of the bucket arithmetic (it only has to be the same function for samples and queries; NaN
 * queries fall into bucket 0 and come out as index 0 like before).  1024 buckets over [-0.1 pi, 1.1 pi], the interval the
 * reference samples (src/systems.rs:437-438); G has kInterpGrid + 1 entries, G[kInterpGrid] = n. */
constexpr unsigned kInterpGrid = 1024;
CV_HD unsigned interp_bucket(double x) {
  const double t = (x - (-0.1 * CV_PI)) * ((double)kInterpGrid / (1.2 * CV_PI));
  if (!(t > 0.0)) return 0u; /* below the interval, NaN */
  if (t >= (double)(kInterpGrid - 1u)) return kInterpGrid - 1u;
  return (unsigned)t;
}
/* the grid entries that sample index i (0 .. n inclusive; x has n entries, strictly increasing) is responsible for:
 * G[b] = i for every b with bucket(x[i - 1]) < b <= bucket(x[i]) (from 0 for i = 0, up to kInterpGrid for i = n) */
CV_HD void interp_grid_fill(const double *x, unsigned n, unsigned i, unsigned *G) {
  const unsigned b0 = i ? interp_bucket(x[i - 1u]) + 1u : 0u;
  const unsigned b1 = i < n ? interp_bucket(x[i]) : kInterpGrid;
  for (unsigned b = b0; b <= b1; ++b) G[b] = i;
}
CV_HD unsigned interp_index_grid(const double *x, unsigned n, double xp, const unsigned *G) {
  const unsigned b = interp_bucket(xp);
  unsigned lo = G[b], hi = G[b + 1u];
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (x[mid] < xp)
      lo = mid + 1;
    else
      hi = mid;
  }
  unsigned i = lo ? lo - 1 : 0;
  if (i > n - 2) i = n - 2;
  return i;
}

/* step 2 of render_image_efficient for one pixel (src/systems.rs:405-433): the angle alpha between the pixel's
 * outward direction and the radial direction, and the rotation axis cam_bg x out_bg */
template <bool SHARED = false>
CV_HD void efficient_pixel_geometry(const CameraParams &C, const EfficientFrame &F, unsigned px, unsigned py, double &alpha,
                                    double *axis, const PixelRecips *R = nullptr) {
  /* outward_vector_on_world_space_from_x_y (src/cameras.rs:150-172) */
  const double h = 0.5 - (SHARED ? div_index<SHARED>((double)py, C.res_y, R->y_res_y) : (double)py / C.res_y);
  const double w = (SHARED ? div_index<SHARED>((double)px, C.res_x, R->y_res_x) : (double)px / C.res_x) - 0.5;
  const double v0[3] = {C.focal * 1.0, -C.sensor_w * w, C.sensor_h * h};
  const double n = sqrt_plain<SHARED>(dot3(v0, v0)); /* norm3 */
  double v[3];
  unit3<SHARED>(v0, n, v);
  double out_tan[3], out_bg[3];
  mat3_vec(C.rot, v[0], v[1], v[2], out_tan[0], out_tan[1], out_tan[2]);
  mat3_vec(F.rot_bg, out_tan[0], out_tan[1], out_tan[2], out_bg[0], out_bg[1], out_bg[2]);
  cross3(F.cam_bg, out_bg, axis);
  const double ex[3] = {1.0, 0.0, 0.0};
  alpha = cv_acos(dot3(out_tan, ex)); /* :431 */
}
/* step 5 (src/systems.rs:498-506): final = from_axis_angle(normalize(axis), escape angle) * cam_bg */
template <bool SHARED = false>
CV_HD void efficient_final_direction(const EfficientFrame &F, const double *axis, double esc, double *fin) {
  const double an = sqrt_plain<SHARED>(dot3(axis, axis)); /* norm3; Unit::new_normalize: 0/0 -> NaN for the centre pixel */
  double u[3];
  unit3<SHARED>(axis, an, u);
  double rot[9];
  from_axis_angle(u, esc, rot);
  mat3_vec(rot, F.cam_bg[0], F.cam_bg[1], F.cam_bg[2], fin[0], fin[1], fin[2]);
}

/* steps 2 + 4 + 5 of render_image_efficient for one pixel: returns the final direction on the background
 * space and the interpolated escape space (1.0 / -1.0 / anything else = black). */
template <bool SHARED = false>
CV_HD void efficient_pixel(const CameraParams &C, const EfficientFrame &F, unsigned px, unsigned py,
                           const double *sx, const double *m_e, const double *c_e, const double *m_s,
                           const double *c_s, unsigned n_samples, double *fin, double &space, const PixelRecips *R = nullptr,
                           const unsigned *grid = nullptr) {
  double alpha, axis[3];
  efficient_pixel_geometry<SHARED>(C, F, px, py, alpha, axis, R);
  double esc;
  if (n_samples == 0) { /* interp_slice on empty tables returns zeros */
    esc = 0.0;
    space = 0.0;
  } else if (n_samples == 1) {
    esc = c_e[0]; /* y[0] */
    space = c_s[0];
  } else {
    const unsigned i = grid ? interp_index_grid(sx, n_samples, alpha, grid) : interp_index(sx, n_samples, alpha); /* both tables share the abscissae */
    esc = m_e[i] * alpha + c_e[i];
    space = m_s[i] * alpha + c_s[i];
  }
  efficient_final_direction<SHARED>(F, axis, esc, fin);
}

}  // namespace cvk
#endif
