/* cv_device.h -- per-ray device functions of the geodesic hot path (gfx950).
 *
 * One lane owns one ray.  The live state of the forward-Euler loop is five doubles
 * (l, theta, p_l, p_theta and -- only when the caller needs it -- phi) plus the constants
 * p_phi and p_phi^2; t and p_t are dead on this path (dp_t = dp_phi = 0, t is never read:
 * src/metrics.rs:259-264) and are reconstructed on the host for the debug dump.
 *
 * Arithmetic contract: every floating-point operation below is the operation the reference
 * performs, in the reference's order (src/metrics.rs:223-297 for the step), individually
 * rounded; the only fused operations are the explicit fma() calls inside cv_math.h.  The file
 * must be compiled with -ffp-contract=off.  Common sub-expressions are evaluated once
 * (sin/cos(theta), r(l), r^2(l)): they are pure functions of the old state, so the values are
 * identical to the reference's repeated evaluation.
 *
 * Functions are __host__ __device__ so the SAME source can be compiled for x86 by the CPU test
 * suite (tests/host_twin) -- a test vehicle, never a fallback: libcurvis_hip.so only launches
 * the __global__ kernels.
 */
#ifndef CURVIS_CV_DEVICE_H
#define CURVIS_CV_DEVICE_H

#include <math.h>

#include "cv_math.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace cvk {

enum : int { METRIC_ELLIS = 0, METRIC_INTERSTELLAR = 1, METRIC_FLAT = 2 };
enum : int { CODE_NONE = 0, CODE_POS = 1, CODE_NEG = -1 };

struct MetricParams {
  double rho;      /* throat radius */
  double rho2;     /* rho*rho (rho.powi(2)) */
  double m, a;     /* Interstellar mass / half-length */
  double pim;      /* PI*m: denominator of scaled_distance (src/metrics.rs:461) */
  double inv_pim;  /* RN(1/(PI*m)) computed on the host: lets the fast step divide by the constant with the
                      three-instruction Markstein sequence (exact for a correctly rounded reciprocal) */
  double two_o_pi; /* 2.0/PI (src/metrics.rs:481) */
  cv_sc_tab_t T;   /* sin/cos table the per-ray functions read: LDS copy in the hot kernels, cv_sc_table()
                      (host static / device __constant__) everywhere else */
  cv_log_tab_t LT; /* same for cv_log_t and cv_atan_t (Interstellar only) */
  cv_atan_tab_t AT;
};

/* Interstellar metric of the FAST step: x = 2(|l| - a)/(pi m) first, the rest only for x >= 2.
 * Same values as metric_eval below with fewer instructions (each rewrite is exact, not merely close):
 *   2*(|l| - a)       = fma(2, |l|, -2a)          scaling by two commutes with the rounding
 *   xn / (pi m)       Markstein with the host-rounded reciprocal M.inv_pim (exact for a correctly rounded one)
 *   x*at - lg/2       = fma(-0.5, lg, RN(x*at))   lg/2 is exact
 *   (2/pi)*signum(l)*at = copysign(RN(2/pi * at), l)        multiplying by +-1 is exact, at > 0
 * x >= 2 holds for every step outside |l| < a + pi m (a dozen steps of a ray that crosses the throat): it is part of
 * the fast step's guard, so the throat (|l| <= a: r = rho, r' = 0), negative x, NaN and the small-argument forms
 * of atan / log never reach this code -- they take the strict step, i.e. the reference's own branches -- and the
 * fast path carries neither `max(x, 0)` nor a second form of either function. */
CV_HD double interstellar_x(const MetricParams &M, double l) {
  const double xn = CV_FMA(2.0, CV_FABS(l), -2.0 * M.a);
  const double q0 = xn * M.inv_pim;
  return CV_FMA(CV_FMA(-M.pim, q0, xn), M.inv_pim, q0);
}
CV_HD void interstellar_eval_x_ge2(const MetricParams &M, double l, double x, double &r, double &r2, double &rd) {
  const double at = cv_atan_row(cv_div_nr(-1.0, x), M.AT); /* = cv_atan(x) for x >= 2: reciprocal branch, no range test */
  const double lg = cv_log_ge2_t(1.0 + x * x, M.LT);       /* = cv_log(1 + x^2): the k >= 1 formula, no Fast2Sum */
  r = M.rho + M.m * CV_FMA(-0.5, lg, x * at);
  rd = __builtin_copysign(M.two_o_pi * at, l); /* (2/pi * signum(l)) * at: the product by +-1 is exact, and at >= atan 2 > 0
                                                  here, so the sign can be put on afterwards (one v_mul + one v_bfi) */
  r2 = r * r;
}

/* r(l), r^2(l), r'(l): src/metrics.rs:417-421 / 467-485 / 501-505. */
template <int KIND>
CV_HD void metric_eval(const MetricParams &M, double l, double &r, double &r2, double &rd) {
  if (KIND == METRIC_ELLIS) {
    r2 = M.rho2 + l * l;
    r = CV_SQRT(r2);
    rd = l / r;
  } else if (KIND == METRIC_INTERSTELLAR) {
    double al = CV_FABS(l);
    if (al > M.a) {
      const double xn = 2.0 * (al - M.a);
      const double x = xn / M.pim;
      const double at = cv_atan_t(x, M.AT);
      r = M.rho + M.m * (x * at - cv_log_t(1.0 + x * x, M.LT) / 2.0);
      double sg = (cv_bits(l) >> 63) ? -1.0 : 1.0; /* l.signum(), l != NaN-safe below */
      if (l != l) sg = l;
      rd = M.two_o_pi * sg * at;
    } else {
      r = M.rho;
      rd = 0.0;
    }
    r2 = r * r;
  } else {
    r = l;
    r2 = l * l;
    rd = 1.0;
  }
}

/* r(l) only (photon construction) */
template <int KIND>
CV_HD double metric_r(const MetricParams &M, double l) {
  double r, r2, rd;
  metric_eval<KIND>(M, l, r, r2, rd);
  return r;
}

struct Ray {
  double l, th, ph; /* position (contravariant) */
  double p1, p2;    /* covariant momentum: radial, theta */
  double p3, p3sq;  /* covariant phi momentum (constant of motion) and its square */
};

/* One forward-Euler step given sin/cos(theta): src/metrics.rs:283-297 with :223-244 and :247-270
 * inlined, every division and the square root being the compiler's IEEE-754 operations. */
template <int KIND, bool PHI>
CV_HD void ray_step_core(const MetricParams &M, Ray &q, double delta, double s, double c) {
  double r, r2, rd;
  metric_eval<KIND>(M, q.l, r, r2, rd);
  const double ss = s * s;            /* theta.sin().powi(2) */
  const double g22c = 1.0 / r2;       /* g22.powi(-1) */
  const double dx1 = q.p1;            /* p1 * g11.powi(-1) == p1 * 1.0 */
  const double dx2 = q.p2 * g22c;
  const double b2 = q.p2 * q.p2 + q.p3sq / ss;
  const double dp1 = b2 * rd / (r * (r * r));         /* r.powi(3) */
  const double dp2 = q.p3sq * (c / (r2 * (s * ss)));  /* sin.powi(3) = s*(s*s) */
  if (PHI) {
    const double g33c = 1.0 / (r2 * ss);
    q.ph = q.ph + (q.p3 * g33c) * delta;
  }
  q.l = q.l + dx1 * delta;
  q.th = q.th + dx2 * delta;
  q.p1 = q.p1 + dp1 * delta;
  q.p2 = q.p2 + dp2 * delta;
}

/* WIDE_T: M.T is the 256-row form of the sin/cos table (cv_sincos_tw) */
template <int KIND, bool PHI, bool WIDE_T = false>
CV_HD void ray_step(const MetricParams &M, Ray &q, double delta) {
  double s, c;
  if (WIDE_T)
    cv_sincos_tw(q.th, M.T, &s, &c);
  else
    cv_sincos_t(q.th, M.T, &s, &c);
  ray_step_core<KIND, PHI>(M, q, delta, s, c);
}

/* ------------------------------------------------------------------------------------------------
 * Fast step: the same IEEE results with far fewer instructions.
 *
 * The AMDGPU expansion of an f64 division n/d is  y = rcp(d) + 2 Newton steps;  q0 = n*y;
 * rem = fma(-d, q0, n);  q = fma(rem, y, q0)  (plus div_scale / div_fixup for extreme exponents),
 * and that of sqrt(x) is a Goldschmidt iteration on rsq(x) (plus ldexp scaling below 2^-767).  The
 * final fma of either sequence returns the correctly rounded result for ANY y within a few ulp of
 * 1/d (Markstein), so the five divisions of a step do not each need their own rcp + Newton chain:
 * 1/r^2, 1/r^3 and 1/(r^2 sin^3) are products of ONE refined 1/r (a by-product of the sqrt) and
 * ONE refined 1/sin(theta).  The result of each division is still the individually rounded quotient
 * the reference computes -- correctly rounded unless a rounding boundary lies within
 * (|kappa| + 1)^2 2^-106 (relative) below the exact quotient, y = (1 + kappa 2^-53)/d.  MEASURED on
 * the reciprocals the step really forms (probe hook below; tools/gpu_fast_step_rounding.py ->
 * profiles/round5_fast_step_rounding.txt): |kappa| <= 2 .. 9 (Ellis), <= 6 .. 14
 * (Interstellar), and summed over the quotients of a step 1.0e-15 (Ellis) / 1.9e-15 (Interstellar)
 * expected mis-rounded operations per step = 4e-6 per 1080p frame, 1.5e-2 per configs[4] render; an
 * event is ONE quotient one ulp low (tests/test_gpu_fast_step.py drives it there on purpose).
 *
 * Guards: the shortcut skips div_scale/div_fixup, so it is only taken when every operand is finite,
 * non-zero and far from the exponent limits; any lane failing the guard executes ray_step_core
 * (the compiler's IEEE operations) for that step instead. */
CV_HD double rcp_seed(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(x); /* v_rcp_f64 */
#else
  return 1.0 / x;
#endif
}
CV_HD double rsq_seed(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsq(x); /* v_rsq_f64 */
#else
  return 1.0 / CV_SQRT(x);
#endif
}
/* y ~ 1/d to about an ulp: hardware seed (2^-23) + ONE third-order step, y0 (1 + e + e^2) with e = 1 - d y0,
 * i.e. 1/d (1 - e^3): 2^-69 before the final rounding, in three fmas (two Newton steps take four) */
CV_HD double recip_refined(double d) {
  const double y0 = rcp_seed(d);
  const double e = CV_FMA(-d, y0, 1.0);
  const double p = CV_FMA(e, e, e);
  return CV_FMA(y0, p, y0);
}
/* correctly rounded n/d given y ~ 1/d */
CV_HD double div_with_recip(double n, double d, double y) {
  const double q0 = n * y;
  const double rem = CV_FMA(-d, q0, n);
  return CV_FMA(rem, y, q0);
}
/* ---- quotients that share their reciprocal (the per-pixel kernel of the efficient renderer; device only) -------------------------
 * The AMDGPU expansion of an f64 quotient n/d is  y = v_rcp_f64(d) refined by two Newton steps;  q0 = n y;  rem = fma(-d, q0, n);
 * q = fma(rem, y, q0)  -- wrapped in v_div_scale / v_div_fmas / v_div_fixup, which only act when an operand or the quotient is zero,
 * subnormal, infinite, NaN or within 2^-/+768 of the exponent limits, and pass their operands through otherwise.  A pixel forms
 * twelve such quotients, nine of them in groups that share the denominator (v / |v| twice) or divide by a constant of the call (the
 * resolution, pi, 2 pi).  recip_chain is the FIRST HALF of that very sequence -- same instructions, same operand -- and
 * div_with_recip (cv_device.h) its second half, so a quotient formed from a shared y is the quotient the compiler's own expansion
 * returns, operation for operation: nothing is assumed about how close y is to 1/d (unlike the fast Euler step, whose
 * reciprocals are products).  The guards keep every operand inside [2^-300, 2^300), where none of the wrappers acts; a lane outside
 * takes the `/` operator.  The x86 build keeps the operator throughout (its quotients are IEEE by construction); device == twin ==
 * oracle is what the parity tests assert, and tests/test_gpu_fast_step.py puts the helpers themselves on operands whose quotient sits
 * within 2^-106 of a rounding boundary and on every class of special value.  Measured: 914 -> 887 VALU instructions per wave (the
 * guards and their joins give back more than half of what the 36 + 31 + 8 wrapper, Newton and seed instructions save), 0.0559 ->
 * 0.0534 ms per 1080p frame = -4.4 % on one box, interleaved (profiles/round6_eff_pixel_shared_div.txt). */
CV_HD double recip_chain(double d) {
  double y = rcp_seed(d);
  y = CV_FMA(y, CV_FMA(-d, y, 1.0), y);
  return CV_FMA(y, CV_FMA(-d, y, 1.0), y);
}
/* y of the call's constant denominators, formed once per context and resolution by recip_chain ON THE DEVICE (recip_chain_kernel)
 * and handed to the pixel kernel as arguments */
struct PixelRecips {
  double y_res_x, y_res_y, y_pi, y_two_pi;
};
/* v / n for the three components of a vector and its Euclidean norm n (so |v_i| <= n up to rounding: one bound per operand) */
template <bool SHARED>
CV_HD void unit3(const double *v, double n, double *u) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (SHARED) {
    if (n >= 0x1p-300 && n < 0x1p300 && CV_FABS(v[0]) >= 0x1p-300 && CV_FABS(v[1]) >= 0x1p-300 && CV_FABS(v[2]) >= 0x1p-300) {
      const double y = recip_chain(n);
      u[0] = div_with_recip(v[0], n, y);
      u[1] = div_with_recip(v[1], n, y);
      u[2] = div_with_recip(v[2], n, y);
      return;
    }
  }
#endif
  u[0] = v[0] / n;
  u[1] = v[1] / n;
  u[2] = v[2] / n;
}
/* sqrt(x) the same way: the AMDGPU expansion of an f64 square root is v_rsq_f64 and a Goldschmidt / Newton chain of nine
 * operations, wrapped in a scaling by 2^256 for arguments below 2^-767 and a class test that passes zeros and infinities through
 * (v_cmp, v_cndmask, v_ldexp twice, v_cmp_class, two v_cndmask).  Inside [2^-700, 2^700) the wrappers do nothing; the chain below is
 * the compiler's, operation for operation, and a lane outside the range (zero, negative, NaN, ...) takes the operator. */
template <bool SHARED>
CV_HD double sqrt_plain(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (SHARED) {
    if (x >= 0x1p-700 && x < 0x1p700) {
      const double y = rsq_seed(x);
      double g = x * y, h = y * 0.5;
      const double e = CV_FMA(-h, g, 0.5);
      g = CV_FMA(g, e, g);
      h = CV_FMA(h, e, h);
      double d = CV_FMA(-g, g, x);
      g = CV_FMA(d, h, g);
      d = CV_FMA(-g, g, x);
      return CV_FMA(d, h, g);
    }
  }
#endif
  return CV_SQRT(x);
}

/* a / d for a constant d of the call, 2^-300 <= d < 2^300, with y = recip_chain(d):
 * div_index: a is an integer-valued 0 <= a < 2^32 (a pixel index) -- always inside the range, no guard.  (+0: 0 y = +0,
 *            fma(-d, +0, +0) = +0, fma(+0, y, +0) = +0, the IEEE quotient.)
 * div_angle: a is an angle, 0 <= a < 8 or NaN by construction; zero (of either sign), anything below 2^-300 and NaN take the operator. */
template <bool SHARED>
CV_HD double div_index(double a, double d, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (SHARED) return div_with_recip(a, d, y);
#endif
  (void)y;
  return a / d;
}
template <bool SHARED>
CV_HD double div_angle(double a, double d, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (SHARED) {
    if (a >= 0x1p-300 && a < 8.0) return div_with_recip(a, d, y);
  }
#endif
  (void)y;
  return a / d;
}

/* root = sqrt(x) correctly rounded, y ~ 1/sqrt(x) to about an ulp (Goldschmidt on the hardware seed, no
 * scaling).  With g0 = x y0 and E = 1 - g0 y0 (= 1 - x y0^2):  sqrt(x) = g0 (1 - E)^(-1/2) = g0 (1 + E/2 + 3/8 E^2
 * + 5/16 E^3 ...), and the same factor takes y0 to 1/sqrt(x).  One third-order step p = E/2 + 3/8 E^2 brings BOTH
 * to rounding accuracy (seed 2^-23 -> 2^-70 before rounding); the final residual step g + (x - g^2) y/2 then sees
 * the exact residual (fma) and a y good to 2^-53, i.e. a value within ~2^-53 ulp of sqrt(x) before its single
 * rounding.  10 instructions; the compiler's expansion (second-order step, two residual steps with a 2^-45
 * multiplier) plus a reciprocal refinement took 13. */
CV_HD void sqrt_and_rsqrt(double x, double &root, double &y) {
  const double y0 = rsq_seed(x);
  double g = x * y0;
  const double E = CV_FMA(-g, y0, 1.0);
  const double p = E * CV_FMA(0.375, E, 0.5);
  g = CV_FMA(g, p, g);
  y = CV_FMA(y0, p, y0);
  const double d = CV_FMA(-g, g, x);
  root = CV_FMA(d, 0.5 * y, g);
}
/* finite, non-zero, not subnormal (one v_cmp_class_f64) */
CV_HD bool is_normal_number(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_class(v, 0x108); /* -normal | +normal */
#else
  const uint32_t e = (cv_hi(v) >> 20) & 0x7ffu;
  return e != 0u && e != 0x7ffu;
#endif
}
/* lo_hi <= (high word of |v|) < hi_hi : exponent-range test with two integer ops */
CV_HD bool hi_word_in(double v, uint32_t lo_hi, uint32_t hi_hi) {
  return ((cv_hi(v) & 0x7fffffffu) - lo_hi) < (hi_hi - lo_hi);
}
#define CV_HI_2POW(e) ((uint32_t)(1023 + (e)) << 20)

/* per-ray part of the guard.  p_phi^2 is a constant of the motion; the bounds keep b^2 = p_theta^2 + p_phi^2 /
 * sin^2 >= 2^-300 (the dp_l class test relies on it) and the products away from overflow.  A ray with p_phi == 0
 * exactly -- the middle pixel ROW of an axis-aligned camera: 1920 rays of a 1080p frame -- is served as well:
 * with a zero numerator the shared-reciprocal quotients are exact zeros of the IEEE sign (0 * y = 0, fma(-d, 0, 0)
 * = +0, fma(+0, y, 0) = 0), dp_theta = 0 * q = +-0 adds nothing, so p_theta is a constant of the motion too
 * and b^2 = p_theta^2 is bounded once, here.  (p_theta == 0 as well -- the central pixel -- keeps the strict
 * step.)  Before this, that row executed the IEEE step in every iteration, and the 240 waves holding it were
 * among the last to finish. */
CV_HD bool ray_fast_ok(const Ray &q) {
  if (q.p3sq == 0.0) return hi_word_in(q.p2 * q.p2, CV_HI_2POW(-300), CV_HI_2POW(300));
  return hi_word_in(q.p3sq, CV_HI_2POW(-300), CV_HI_2POW(300));
}

/* host-side part of the guard: metric parameters / escape radius far from the exponent limits */
CV_HD bool metric_fast_ok(int kind, const MetricParams &M, double max_radius) {
  bool ok = hi_word_in(max_radius, CV_HI_2POW(-90), CV_HI_2POW(90));
  if (kind == METRIC_ELLIS) ok = ok && hi_word_in(M.rho, CV_HI_2POW(-90), CV_HI_2POW(90));
  if (kind == METRIC_INTERSTELLAR)
    ok = ok && hi_word_in(M.rho, CV_HI_2POW(-90), CV_HI_2POW(90)) && hi_word_in(M.m, CV_HI_2POW(-90), CV_HI_2POW(90)) &&
         hi_word_in(M.a, CV_HI_2POW(-300), CV_HI_2POW(90)) &&
         /* x = 2(|l| - a)/(pi m) stays finite and far below overflow for every |l| <= max_radius (interstellar_eval_x_ge2
          * takes -1/x of it without range handling) */
         2.0 * max_radius * M.inv_pim < 0x1p200;
  return ok;
}

/* Probe hook of the fast step's quotients.  The hot kernels pass none (NoProbe: every call is an empty inline and
 * leaves no instruction); curvis_selftest_fast_step passes a recorder, so that the numerators, denominators and SHARED
 * reciprocals the step really forms -- products of one refined 1/r and one refined 1/sin(theta) -- can be looked at
 * from outside: how far each y is from 1/d decides how often div_with_recip can mis-round.
 * k: 0 r' = l/r (Ellis), 1 1/r^2, 2 p_phi^2/sin^2, 3 dp_l = (b^2 r')/r^3, 4 cos/(r^2 sin^3), 5 1/(r^2 sin^2) (PHI). */
struct NoProbe {
#if defined(__HIPCC__) || defined(__HIP__)
  __host__ __device__
#endif
  inline void rec(int, double, double, double) const {}
};
/* one Newton step on an approximate reciprocal.  For y within an ulp of 1/d the result is RN(1/d) unless d's
 * significand is all ones (Markstein); with y = RN(1/d) div_with_recip is correctly rounded, not just almost always.
 * Only the CV_CERTIFIED_DIV build (an A/B measurement, tools/gpu_ab.py) calls it inside the step. */
CV_HD double recip_newton(double d, double y) {
  const double e = CV_FMA(-d, y, 1.0);
  return CV_FMA(y, e, y);
}
#ifndef CV_CERTIFIED_DIV
#define CV_CERTIFIED_DIV 0
#endif

/* EQ (the sampling kernel of the efficient renderer only): every photon of compute_escape_angle starts at
 * theta = fl(pi/2) with p_theta = 0 (src/systems.rs:221-230) and keeps that theta for ever, so
 * while theta has exactly those bits the step is taken in its equatorial form: sin = 1 and cos = RN(pi/2 - fl(pi/2))
 * are what cv_sincos returns for this argument, and with s = 1 the generic operations collapse EXACTLY --
 * 1/s = 1, s^2 = s^3 = 1, p_phi^2/s^2 = p_phi^2 (zero remainder), r^2 s^3 = r^2, 1/(r^2 s^2) = 1/r^2 -- so
 * the state is the generic step's bit for bit, with ~40 of ~95 instructions less in the dependency chain of the
 * lone waves these launches consist of.  A ray whose theta has moved takes the generic step (never observed). */
template <int KIND, bool PHI, bool WIDE_T = false, bool EQ = false, class PROBE = NoProbe>
CV_HD void ray_step_fast(const MetricParams &M, Ray &q, double delta, bool lane_ok, const PROBE &probe = PROBE()) {
  double s, c;
  int s_ok;
  const bool eq = EQ && cv_bits(q.th) == 0x3FF921FB54442D18ULL;
  if (eq) {
    s = 1.0;
    c = 6.123233995736766036e-17; /* 0x3C91A62633145C07 */
    s_ok = 1;
  } else {
    s_ok = cv_sincos_guarded(q.th, M.T, WIDE_T ? 1 : 0, &s, &c);
  }
  /* guard (branch-free, one compare each): sin(theta) and l non-zero, not NaN and far from the underflow
   * limit.  Upper bounds are implied: |sin| <= 1, and a step is only executed for a ray that has not
   * escaped, |l| <= max_radius < 2^90 (metric_fast_ok; an infinite l has escaped, a NaN fails the compare).
   * The Interstellar step tests x = 2(|l| - a)/(pi m) >= 2 instead of l (false for a NaN l): inside |l| < a + pi m
   * -- the throat and the few steps next to it -- a ray takes the strict step; outside, atan x >= atan 2, so r' is
   * far from zero and from the underflow of the quotient's remainder.
   * cos(theta) needs no test of its own: it is finite iff sin(theta) is, and the cosine of a double is never
   * zero or subnormal. */
  const double x_i = KIND == METRIC_INTERSTELLAR ? interstellar_x(M, q.l) : 0.0;
  const bool ok = (int)lane_ok & s_ok & (int)(KIND == METRIC_INTERSTELLAR ? x_i >= 2.0 : CV_FABS(q.l) > 0x1p-100);
  if (!ok) {
    ray_step_core<KIND, PHI>(M, q, delta, s, c);
    return;
  }
  double r, r2, rd, y_r, y_s;
  if (KIND == METRIC_ELLIS) {
    r2 = M.rho2 + q.l * q.l;
    sqrt_and_rsqrt(r2, r, y_r);
    if (CV_CERTIFIED_DIV) y_r = recip_newton(r, y_r);
    probe.rec(0, q.l, r, y_r);
    rd = div_with_recip(q.l, r, y_r);
    y_s = recip_refined(s);
  } else {
    if (KIND == METRIC_INTERSTELLAR)
      interstellar_eval_x_ge2(M, q.l, x_i, r, r2, rd);
    else
      metric_eval<KIND>(M, q.l, r, r2, rd);
    /* no square root here, so 1/r would need a seed of its own: ONE v_rcp_f64 (a quarter-rate instruction) serves
     * both reciprocals -- y ~ 1/(r s), 1/r ~ y s, 1/s ~ y r (each ~1.5 ulp; the quotients below stay correctly
     * rounded, see div_with_recip).  r s cannot leave the normal range: 2^-90 < r < 2^91 and |s| > 2^-60. */
    if (eq) {
      y_r = recip_refined(r);
      y_s = 1.0;
    } else {
      const double y_rs = recip_refined(r * s);
      y_r = y_rs * s;
      y_s = y_rs * r;
    }
  }
  if (eq) { /* s == 1: see above; same operations as below with the factors that are exactly 1 left out */
    const double y_r2e = y_r * y_r;
    const double g22e = div_with_recip(1.0, r2, y_r2e);
    const double b2e = q.p2 * q.p2 + q.p3sq;
    const double nume = b2e * rd;
    const double r3e = r * (r * r);
    double dp1e = div_with_recip(nume, r3e, y_r2e * y_r);
    if (!is_normal_number(dp1e)) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("; IEEE division for dp_l (equatorial step)");
#endif
      dp1e = nume / r3e;
    }
    const double dp2e = q.p3sq * div_with_recip(c, r2, y_r2e);
    if (PHI) q.ph = q.ph + (q.p3 * g22e) * delta;
    q.l = q.l + q.p1 * delta;
    q.th = q.th + (q.p2 * g22e) * delta;
    q.p1 = q.p1 + dp1e * delta;
    q.p2 = q.p2 + dp2e * delta;
    return;
  }
  double y_r2 = y_r * y_r;
  double y_ss = y_s * y_s;
  const double ss = s * s;
  if (CV_CERTIFIED_DIV) {
    y_r2 = recip_newton(r2, y_r2);
    y_ss = recip_newton(ss, y_ss);
  }
  probe.rec(1, 1.0, r2, y_r2);
  const double g22c = div_with_recip(1.0, r2, y_r2);
  const double dx1 = q.p1;
  const double dx2 = q.p2 * g22c;
  probe.rec(2, q.p3sq, ss, y_ss);
  const double b2 = q.p2 * q.p2 + div_with_recip(q.p3sq, ss, y_ss);
  const double num = b2 * rd;
  const double r3 = r * (r * r);
  double y_r3 = y_r2 * y_r;
  if (CV_CERTIFIED_DIV) y_r3 = recip_newton(r3, y_r3);
  probe.rec(3, num, r3, y_r3);
  /* dp_l = num / r^3: the shared-reciprocal quotient is checked AFTERWARDS with one class test.  Inside the guarded
   * domain a non-zero num has |num| >= 2^-300 |r'| >= 2^-745 (Ellis: |r'| = |l|/r >= 2^-191; Interstellar:
   * |r'| >= (2/pi) atan 2), hence the remainder fma cannot underflow and the true quotient
   * is a normal number (>= 2^-745 / 2^273) or overflows.  Whatever else can happen shows in the result: an
   * overflowing n*y or a non-finite num (|p_theta| exploding at a pole) gives inf or NaN; those are "not a normal
   * number" and take the IEEE division. */
  double dp1 = div_with_recip(num, r3, y_r3);
  if (!is_normal_number(dp1)) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("; IEEE division for dp_l"); /* not speculatable: keeps this a branch around ~25 instructions
                                                 (as a select they would run in every step) */
#endif
    dp1 = num / r3;
  }
  const double r2s3 = r2 * (s * ss);
  double y_r2s3 = y_r2 * (y_ss * y_s);
  if (CV_CERTIFIED_DIV) y_r2s3 = recip_newton(r2s3, y_r2s3);
  probe.rec(4, c, r2s3, y_r2s3);
  const double dp2 = q.p3sq * div_with_recip(c, r2s3, y_r2s3);
  if (PHI) {
    double y_g33 = y_r2 * y_ss;
    if (CV_CERTIFIED_DIV) y_g33 = recip_newton(r2 * ss, y_g33);
    probe.rec(5, 1.0, r2 * ss, y_g33);
    const double g33c = div_with_recip(1.0, r2 * ss, y_g33);
    q.ph = q.ph + (q.p3 * g33c) * delta;
  }
  q.l = q.l + dx1 * delta;
  q.th = q.th + dx2 * delta;
  q.p1 = q.p1 + dp1 * delta;
  q.p2 = q.p2 + dp2 * delta;
}

struct CameraParams {
  double pos[4];
  double rot[9];
  double focal, sensor_w, sensor_h;
  double res_x, res_y; /* as f64 */
};

CV_HD void mat3_vec(const double *m, double v0, double v1, double v2, double &o0, double &o1, double &o2) {
  /* nalgebra gemv: ((m_i0*x0) + m_i1*x1) + m_i2*x2 */
  o0 = (m[0] * v0 + m[1] * v1) + m[2] * v2;
  o1 = (m[3] * v0 + m[4] * v1) + m[5] * v2;
  o2 = (m[6] * v0 + m[7] * v1) + m[8] * v2;
}

/* pixel -> photon: src/cameras.rs:150-172 then src/metrics.rs:301-334 */
template <int KIND>
CV_HD void ray_init(const MetricParams &M, const CameraParams &C, unsigned px, unsigned py, Ray &q) {
  const double h = 0.5 - ((double)py / C.res_y);
  const double w = ((double)px / C.res_x) - 0.5;
  double vx = C.focal * 1.0;
  double vy = -C.sensor_w * w;
  double vz = C.sensor_h * h;
  double n = CV_SQRT(vx * vx + vy * vy + vz * vz); /* Vector3::normalize */
  vx = vx / n;
  vy = vy / n;
  vz = vz / n;
  double d0, d1, d2;
  mat3_vec(C.rot, vx, vy, vz, d0, d1, d2);
  n = CV_SQRT(d0 * d0 + d1 * d1 + d2 * d2); /* direction.normalize() in new_photon */
  d0 = d0 / n;
  d1 = d1 / n;
  d2 = d2 / n;
  const double r = metric_r<KIND>(M, C.pos[1]);
  q.l = C.pos[1];
  q.th = C.pos[2];
  q.ph = C.pos[3];
  q.p1 = d0;
  q.p2 = d1 * r;
  q.p3 = d2 * r * cv_sin_t(C.pos[2], M.T);
  q.p3sq = q.p3 * q.p3;
}

/* photon from an explicit tangent-space direction (compute_escape_angle, src/systems.rs:221-230) */
template <int KIND>
CV_HD void ray_init_dir(const MetricParams &M, const double pos[4], double dx, double dy, double dz, Ray &q) {
  double n = CV_SQRT(dx * dx + dy * dy + dz * dz);
  double d0 = dx / n, d1 = dy / n, d2 = dz / n;
  const double r = metric_r<KIND>(M, pos[1]);
  q.l = pos[1];
  q.th = pos[2];
  q.ph = pos[3];
  q.p1 = d0;
  q.p2 = d1 * r;
  q.p3 = d2 * r * cv_sin_t(pos[2], M.T);
  q.p3sq = q.p3 * q.p3;
}

/* final photon -> tangent-space direction: src/metrics.rs:339-349 (+ :190-203).
 * z uses frame_field_22 (no sin theta), exactly as line 347. */
template <int KIND>
CV_HD void ray_direction(const MetricParams &M, const Ray &q, double &d0, double &d1, double &d2) {
  double r, r2, rd;
  metric_eval<KIND>(M, q.l, r, r2, rd);
  const double s = cv_sin_t(q.th, M.T);
  const double g22c = 1.0 / r2;
  const double g33c = 1.0 / (r2 * (s * s));
  const double v1 = q.p1; /* * 1.0 */
  const double v2 = q.p2 * g22c;
  const double v3 = q.p3 * g33c;
  d0 = v1; /* * frame_field_11 = 1.0 */
  d1 = v2 * r;
  d2 = v3 * r;
}

CV_HD unsigned rust_as_u32(double v) { /* `as u32`: NaN -> 0, saturating */
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned r; /* v_cvt_u32_f64 IS that conversion: truncation, NaN and negatives to 0, 2^32 and above to 0xFFFFFFFF -- one instruction
                 instead of three compares and their selects (tests/test_gpu_fast_step.py checks it on every class of value) */
  asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
  return r;
#endif
  if (!(v == v)) return 0u;
  if (v <= 0.0) return 0u;
  if (v >= 4294967295.0) return 4294967295u;
  return (unsigned)v;
}

/* f64::rem_euclid(a, b) for b > 0.  On this path |a| < b always holds (a is an atan2 result
 * against 2*pi, or 0.5 - phi/(2*pi) against 1), where fmod(a, b) == a exactly; the general
 * branch keeps the function total. */
CV_HD double rem_euclid_pos(double a, double b) {
  double r;
  if (CV_FABS(a) < b || a != a) {
    r = a;
  } else {
    r = fmod(a, b);
  }
  return (r < 0.0) ? r + b : r;
}

struct SkyParams {
  const unsigned *texels; /* RGBA8 packed little-endian, row-major */
  unsigned w, h;
  double inv_rot[9];
};

/* direction -> texel indices: src/images.rs:132-142 -> src/algebra.rs:128-134 -> :106-116 ->
 * src/images.rs:115-121.  Returns raw `as u32` indices (may equal w / h: reference panics). */
template <bool SHARED = false> /* SHARED (device, the efficient renderer's pixel kernel): theta / pi and phi / 2 pi through the call's
                                  precomputed reciprocals -- div_angle above */
CV_HD void sky_indices(const SkyParams &S, double d0, double d1, double d2, unsigned &tx, unsigned &ty, double y_pi = 0.0,
                       double y_two_pi = 0.0) {
  double w0, w1, w2;
  mat3_vec(S.inv_rot, d0, d1, d2, w0, w1, w2);
  const double rn = sqrt_plain<SHARED>(w0 * w0 + w1 * w1 + w2 * w2);
  double theta = cv_acos(w2 / rn);
  double phi = cv_atan2(w1, w0);
  const double TWO_PI = 2.0 * CV_PI;
  /* normalize_theta_phi #1 (theta_phi_from_vector3) and #2 (pixel_indexes...) */
  for (int k = 0; k < 2; ++k) {
    if (theta < 0.0) {
      theta = CV_FABS(theta);
      phi = phi + CV_PI;
    }
    phi = rem_euclid_pos(phi, TWO_PI);
  }
  ty = rust_as_u32(div_angle<SHARED>(theta, CV_PI, y_pi) * (double)S.h);
  tx = rust_as_u32(rem_euclid_pos(0.5 - div_angle<SHARED>(phi, TWO_PI, y_two_pi), 1.0) * (double)S.w);
}

}  // namespace cvk

#endif /* CURVIS_CV_DEVICE_H */
