/* curvis_oracle.h -- CPU oracle for the CurVis per-pixel geodesic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (fragarriss/CurVis @ 2024-12-18, single-threaded Rust), each function
 * citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; nothing in curvis_amd/ links,
 * imports or calls it, and the product library fails loudly without a GPU.
 *
 * PARITY STATUS: "parity unpinned" for trajectories / pixels.  The reference
 * cannot be built here (no rustc/cargo, crates not vendored) and its own tests pin
 * only: algebra known answers (src/algebra.rs:259-282, 154-176, 200-209) and the
 * photon null/direction test (src/metrics.rs:515-541).  Those ARE checked
 * (tests/test_oracle.py).  Third-party arithmetic (nalgebra 0.33.0, interp 1.0.3,
 * image 0.25.2 -- pinned in Cargo.lock, not under /root/reference) is restated
 * from the published behaviour of those crates; see DESIGN.md section 3.
 * An independent second restatement (tests/ref_python.py, plain Python, written
 * from the reference text) agrees with the libm flavour of this one bit for bit
 * on whole small frames of both renderers (tests/test_ref_python.py); the libm
 * flavour must be built with -fno-builtin-sin/cos (see the Makefile).
 *
 * Math flavours, selected by the `fl` argument of every entry point:
 *   CVO_LIBM (0): glibc libm sin/cos/acos/atan/atan2/log -- what a Linux build of
 *                 the Rust reference calls (Rust f64::sin -> llvm.sin.f64 -> libm),
 *                 every sin and cos a separate libcall.
 *   CVO_LIBM_SINCOS (2), CVO_LIBM_SINCOS_INL (3): the same, with sin and cos of one
 *                 value taken inside one function coming from ONE glibc sincos()
 *                 call -- what LLVM's FSIN+FCOS -> sincos combine emits on
 *                 *-linux-gnu.  rustc cannot be run here; the LLVM of this image does merge them
 *                 for the shape of update_relativistic_object (llvm_sincos_probe.c,
 *                 `make sincos-probe`), g33's sine included once inlined (flavour 3).
 *                 Another LLVM version may decide differently, so all three glibc
 *                 arithmetics are carried and measured.
 *   CVO_CV   (1): curvis_amd/csrc/cv_math.h -- the deterministic, fma-explicit
 *                 functions the gfx950 kernels use; the GPU must match this
 *                 flavour BIT FOR BIT.  Only elementary functions are shared
 *                 with the product; the algorithm below is written independently
 *                 of the kernels.
 */
#ifndef CURVIS_ORACLE_H
#define CURVIS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CVO_LIBM = 0, CVO_CV = 1, CVO_LIBM_SINCOS = 2, CVO_LIBM_SINCOS_INL = 3 };
enum { CVO_ELLIS = 0, CVO_INTERSTELLAR = 1, CVO_FLAT = 2 };
/* escape codes (PhotonEscape, src/systems.rs:39-44) */
enum { CVO_NOT_ESCAPED = 0, CVO_POSITIVE = 1, CVO_NEGATIVE = -1, CVO_PANIC = -2 };

typedef struct {
  int32_t kind;
  int32_t _pad;
  double rho, m, a;
} cvo_metric;

typedef struct {
  double pos[4];     /* (t, l, theta, phi), contravariant */
  double rot[9];     /* camera->world rotation, row-major (Orientation::rotation_matrix) */
  double inv_rot[9]; /* its inverse */
  double fwd[3], up[3];
  double focal, sensor_w, sensor_h;
  uint32_t res_x, res_y;
} cvo_camera;

typedef struct {
  const uint8_t *rgba; /* decoded image as Rgba8, row-major */
  uint32_t w, h;
  double inv_rot[9]; /* Orientation::inverse_rotation_matrix of the image (identity by default) */
} cvo_sky;

typedef struct {
  double x[4];
  double p[4];
  uint32_t steps;
  int32_t code;
  uint32_t tx, ty; /* raw `as u32` texel indices (unclamped); 0 if not escaped */
} cvo_ray_debug;

typedef struct {
  uint64_t rays, steps, n_pos, n_neg, n_none, n_oob;
} cvo_stats;

/* --- algebra (src/algebra.rs) + nalgebra restatements --- */
int cvo_orientation_new(const double fwd[3], const double up[3], double rot[9], double inv_rot[9], double up_out[3]);
void cvo_face_towards(const double dir[3], const double up[3], double m[9]);
/* elementary functions of a flavour over arrays; op: 0 sin, 1 cos, 2 atan, 3 acos, 4 log, 5 atan2(a, b) */
void cvo_math_array(int fl, int op, const double *a, const double *b, double *out, size_t n);
/* compute_photon_trajectory (src/systems.rs:77-92): out[iterations][8] = (x, p_cov) before each Euler step */
void cvo_photon_trajectory(int fl, const cvo_metric *m, const double x0[4], const double p0[4], uint32_t iterations,
                           double delta, double *out);
int cvo_rotation_between(int fl, const double a[3], const double b[3], double m[9]); /* nalgebra; -1 = None */
int cvo_rotation_from_two_vectors(int fl, const double a[3], const double b[3], double m[9]); /* -1 = panic */
void cvo_from_axis_angle(int fl, const double unit_axis[3], double angle, double m[9]);
void cvo_normalize_theta_phi(double theta, double phi, double *to, double *po);
void cvo_vector3_from_theta_phi(int fl, double theta, double phi, double v[3]);
void cvo_theta_phi_from_vector3(int fl, const double v[3], double *theta, double *phi);
void cvo_mat3_vec(const double m[9], const double v[3], double out[3]);

/* --- camera (src/cameras.rs) --- */
int cvo_camera_new(cvo_camera *c, const double pos[4], const double fwd[3], const double up[3], double focal,
                   double diag, uint32_t rx, uint32_t ry);
void cvo_camera_outward_camera_space(const cvo_camera *c, uint32_t px, uint32_t py, double out[3]);
void cvo_camera_outward_world(const cvo_camera *c, uint32_t px, uint32_t py, double out[3]);

/* --- metric + integrator (src/metrics.rs, src/systems.rs:115-139) --- */
double cvo_metric_r(int fl, const cvo_metric *m, double l);
double cvo_metric_r_squared(int fl, const cvo_metric *m, double l);
double cvo_metric_r_derivative(int fl, const cvo_metric *m, double l);
void cvo_new_photon(int fl, const cvo_metric *m, const double pos[4], const double dir[3], double x[4], double p[4]);
void cvo_update(int fl, const cvo_metric *m, double x[4], double p[4], double delta); /* literal: call for call the reference's */
/* the same step with r(l), r_squared(l), r_derivative(l) evaluated once each (pure functions: identical bits); what the render
 * loops use unless cvo_set_metric_memo(0) -- bench.py's cpu_baseline times the literal step */
void cvo_update_memo(int fl, const cvo_metric *m, double x[4], double p[4], double delta);
void cvo_set_metric_memo(int on);
int cvo_get_metric_memo(void);
int cvo_escape_photon(int fl, const cvo_metric *m, double x[4], double p[4], double delta, uint32_t max_iter,
                      double max_radius, uint32_t *steps);
void cvo_vector_to_direction(int fl, const cvo_metric *m, const double p_cov[4], const double x[4], double dir[3]);
double cvo_squared_norm_cov(int fl, const cvo_metric *m, const double p_cov[4], const double x[4]);

/* --- sky lookup (src/images.rs:107-174) --- */
void cvo_sky_indices(int fl, const cvo_sky *s, const double v[3], uint32_t *x, uint32_t *y);
int cvo_sky_pixel(int fl, const cvo_sky *s, const double v[3], uint8_t rgba[4]); /* returns 1 if OOB (clamped) */

/* --- per-pixel renderer (src/systems.rs:307-330) --- */
/* rows row_begin, row_begin+row_step, ... are rendered; others untouched.  dbg (nullable) is W*H entries. */
int cvo_render_image(int fl, const cvo_metric *m, const cvo_camera *c, const cvo_sky *pos, const cvo_sky *neg,
                     uint32_t max_iter, double max_radius, double delta, uint32_t row_begin, uint32_t row_step,
                     uint8_t *rgb, cvo_ray_debug *dbg, cvo_stats *stats);

/* --- efficient renderer (src/systems.rs:144-261, 333-527; src/sampling.rs; interp 1.0.3) --- */
int cvo_compute_escape_angle(int fl, const cvo_metric *m, double l, double alpha, double delta, uint32_t max_iter,
                             double max_radius, double *angle, uint32_t *steps);
typedef struct {
  double *a, *e, *s;
  size_t n;
  uint64_t calls, steps;
  uint32_t rounds;
  int warned_max_iterations;
} cvo_samples;
int cvo_doubly_sample(int fl, const cvo_metric *m, double l, double delta, uint32_t max_iter, double max_radius,
                      double a_min, double a_max, size_t n0, size_t max_iterations, double thr1, double thr2,
                      cvo_samples *out);
void cvo_samples_free(cvo_samples *s);
void cvo_interp_slice(const double *x, const double *y, size_t n, const double *xp, size_t np, double *out);
int cvo_render_image_efficient(int fl, const cvo_metric *m, const cvo_camera *c, const cvo_sky *pos,
                               const cvo_sky *neg, uint32_t max_iter, double max_radius, double delta,
                               uint32_t alpha_nums, uint32_t max_iterations_sampling, double thr1, double thr2,
                               uint8_t *rgb, cvo_samples *samples_out /*nullable*/, cvo_stats *stats);
/* "direct" mode (not a reference function): compute_escape_angle at the alpha of every pixel instead of sampling and
 * interpolating; counterpart of curvis_render_direct */
int cvo_render_image_direct(int fl, const cvo_metric *m, const cvo_camera *c, const cvo_sky *pos, const cvo_sky *neg,
                            uint32_t max_iter, double max_radius, double delta, uint8_t *rgb, cvo_stats *stats);

/* --- camera path (src/csv.rs, src/interpolation.rs, src/rendering.rs:224-238) --- */
typedef struct {
  double *pos; /* n*4 */
  double *fwd; /* n*3 */
  double *up;  /* n*3 */
  size_t n;
} cvo_path;
int cvo_load_path(const char *csv, cvo_path *out);
void cvo_path_free(cvo_path *p);
/* returns 0 ok, -1 panic (t outside range), -2 panic (index out of bounds: the off-by-one at the last segment) */
int cvo_path_camera(const cvo_path *p, double t, double pos[4], double fwd[3], double up[3]);
size_t cvo_times_of_frames(double min_time, double max_time, double frame_rate, double *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
