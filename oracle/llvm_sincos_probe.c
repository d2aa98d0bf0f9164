/* llvm_sincos_probe.c -- TEST INFRASTRUCTURE / evidence, not part of any build product.
 *
 * Question (round-2 review): does rustc/LLVM merge the reference's `theta.sin()` / `theta.cos()` pairs
 * (src/metrics.rs:68, :257, :262) into one sincos() libcall on x86_64-unknown-linux-gnu?  rustc is not in this image, but
 * an LLVM is (the clang inside ROCm).  Rust's f64::sin / f64::cos are the errno-free intrinsics llvm.sin.f64 /
 * llvm.cos.f64; clang emits the same intrinsics for __builtin_sin / __builtin_cos under -fno-math-errno.  This file has
 * the SHAPE of update_relativistic_object: values passed by value (Rust reads them through shared borrows: no aliasing
 * with the outputs), g33's sine in its own small function, the momentum right-hand side with two sines and a cosine.
 * `make -C oracle sincos-probe` compiles it for x86_64-linux-gnu at -O3 and lists the libm calls per function:
 *   update_all_inlined : everything inlined, as rustc's -O3 does with these small trait methods
 *   momentum_not_inlined: object_momentum_diff_cov kept as a function of its own
 * Result with AMD clang 22 (profiles/round3_llvm_sincos_probe.txt): ONE sincos() in each -- the FSIN+FCOS combine fires,
 * and with everything inlined g33's sine is CSE'd into it too.  That is the oracle's CVO_LIBM_SINCOS_INL flavour
 * (CVO_LIBM_SINCOS if object_position_diff_contr is not inlined).  A different LLVM version may decide differently,
 * which is why all three glibc flavours are carried and measured. */
typedef struct { double v[4]; } vec4;
double r_of(double l);
double rd_of(double l);

static inline double g33(double r2, double theta) { double s = __builtin_sin(theta); return r2 * (s * s); }

static inline vec4 position_diff(vec4 x, vec4 p, double r2) { /* src/metrics.rs:223-244 */
  vec4 d = {{p.v[0] * (1.0 / -1.0), p.v[1] * (1.0 / 1.0), p.v[2] * (1.0 / r2), p.v[3] * (1.0 / g33(r2, x.v[2]))}};
  return d;
}
static inline vec4 momentum_diff(vec4 x, vec4 p, double r, double r2, double rd) { /* src/metrics.rs:247-270 */
  double s = __builtin_sin(x.v[2]);
  double b2 = p.v[2] * p.v[2] + (p.v[3] * p.v[3]) / (s * s);
  double s3 = __builtin_sin(x.v[2]);
  s3 = s3 * (s3 * s3);
  vec4 d = {{0.0, b2 * rd / (r * (r * r)), (p.v[3] * p.v[3]) * (__builtin_cos(x.v[2]) / (r2 * s3)), 0.0}};
  return d;
}
void update_all_inlined(vec4 *x, vec4 *p, double delta) { /* src/metrics.rs:283-297 */
  vec4 xv = *x, pv = *p;
  double r = r_of(xv.v[1]), r2 = r * r, rd = rd_of(xv.v[1]);
  vec4 dx = position_diff(xv, pv, r2), dp = momentum_diff(xv, pv, r, r2, rd);
  for (int i = 0; i < 4; ++i) {
    x->v[i] = xv.v[i] + dx.v[i] * delta;
    p->v[i] = pv.v[i] + dp.v[i] * delta;
  }
}
__attribute__((noinline)) vec4 momentum_not_inlined(vec4 x, vec4 p, double r, double r2, double rd) { return momentum_diff(x, p, r, r2, rd); }
