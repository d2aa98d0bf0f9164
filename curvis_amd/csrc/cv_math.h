/* cv_math.h -- deterministic double-precision elementary functions shared by the
 * gfx950 kernels (device) and the host-side code of curvis_amd.
 *
 * Why this exists
 * ---------------
 * The reference (fragarriss/CurVis, single-threaded Rust) calls f64::sin / cos /
 * acos / atan / atan2 / ln (src/metrics.rs:68,257,262,470,481; src/algebra.rs:130-131;
 * src/systems.rs:221,252), which lower to whatever libm the platform provides.
 * Their last-bit behaviour is therefore platform-defined.  A null geodesic is
 * integrated with ~2000 dependent Euler steps, so a GPU renderer can only be
 * compared bit-for-bit with a CPU restatement if both sides evaluate *the same*
 * sequence of IEEE-754 operations.  This header is that sequence: every function
 * below is written with explicit, individually rounded +,-,*,/ and fma() only
 * (compile with -ffp-contract=off), integer bit manipulation, and no table
 * lookups that depend on the platform.  The same source compiled by gcc for
 * x86-64 (with hardware FMA) and by hipcc for gfx950 returns bit-identical
 * results; tests/test_cv_math.py checks that on the GPU and checks the accuracy
 * (< 1 ulp) against mpmath.
 *
 * Algorithms: sin/cos are table-driven (pi/64 grid of double-double values +
 * degree-3 Taylor kernels, three-piece Cody-Waite reduction); atan/acos/log and the
 * pi/2 reduction of unusual arguments follow the classical Sun fdlibm / FreeBSD msun
 * designs (published minimax coefficients) evaluated with fma-Horner; huge-argument
 * reduction is a 192-bit integer Payne-Hanek written for 64-bit multiplies.
 *
 * The header is C99 / C++ / HIP clean.  All functions are `static inline`.
 */
#ifndef CURVIS_CV_MATH_H
#define CURVIS_CV_MATH_H

#include <stdint.h>

#include "cv_sincos_table.h"
#include "cv_log_table.h"
#include "cv_atan_table.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define CV_HD __host__ __device__ static inline
#else
#define CV_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#define CV_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define CV_FABS(a) __builtin_fabs((a))
#define CV_SQRT(a) __builtin_sqrt((a))
#define CV_RINT(a) __builtin_rint((a))
/* 1.5 * 2^52: fma(x, c, MAGIC) - MAGIC is the integer nearest to the exact product x*c (|x*c| < 2^31), and the
 * low word of the biased sum is that integer in two's complement -- rounding and int conversion in two ops */
/* an empty, non-speculatable statement: keeps the assignments of a rarely taken branch inside that branch (the
 * compiler otherwise hoists them, as v_mov_b64, in front of it -- into the hot path) */
#if defined(__HIP_DEVICE_COMPILE__)
#define CV_KEEP_BRANCH() asm volatile("")
#else
#define CV_KEEP_BRANCH() ((void)0)
#endif
#define CV_RND_MAGIC 6755399441055744.0
#define CV_RND_MAGIC_128TH 52776558133248.0 /* 1.5 * 2^45: ulp 2^-7 */
#define CV_RND_MAGIC_256TH 26388279066624.0 /* 1.5 * 2^44: ulp 2^-8 */

#define CV_PI 3.14159265358979311600e+00 /* 0x400921FB54442D18 = Rust std::f64::consts::PI */

CV_HD uint64_t cv_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, sizeof u);
  return u;
}
CV_HD double cv_from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, sizeof x);
  return x;
}
CV_HD uint32_t cv_hi(double x) { return (uint32_t)(cv_bits(x) >> 32); }
CV_HD uint32_t cv_lo(double x) { return (uint32_t)cv_bits(x); }

/* 64x64 -> 128 multiply pieces */
CV_HD uint64_t cv_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}
CV_HD int cv_clz64(uint64_t a) { /* a != 0 */
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)a);
#else
  return __builtin_clzll(a);
#endif
}

/* Horner step acc*z + C with a loop-invariant constant C.  On gfx950 LLVM lowers fma(z, acc, C)
 * to v_mov_b64 tmp, C ; v_fmac_f64 tmp, z, acc (two-address form, one extra full-rate move per
 * coefficient); the three-address VOP3 form needs no copy.  Same IEEE operation either way. */
CV_HD double cv_fma_c(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CV_NO_ASM_FMA)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
#else
  return CV_FMA(a, b, c);
#endif
}

/* same with the addend in a scalar register pair: for a loop-invariant constant this frees the VGPR pair the
 * "v" form pins it in (one constant-bus operand per VOP3 instruction is allowed on gfx9) */
CV_HD double cv_fma_ks(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CV_NO_ASM_FMA)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
#else
  return CV_FMA(a, b, c);
#endif
}

/* n/d for operands and quotient well inside the normal range (as inside atan/log below).  Host: the IEEE
 * operator.  Device: the shape of the AMDGPU fdiv expansion (v_rcp_f64, refinement of the reciprocal
 * to ~1 ulp -- one third-order step instead of the compiler's two Newton steps --, quotient, exact remainder,
 * final fma) without its div_scale/div_fixup range handling.  The final fma rounds q + rem/d computed to
 * ~2^-100 relative, i.e. the correctly rounded quotient (Markstein); device == host is asserted on sweeps. */
CV_HD double cv_div_nr(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y0 = __builtin_amdgcn_rcp(d);
  const double e = CV_FMA(-d, y0, 1.0);
  const double y = CV_FMA(y0, CV_FMA(e, e, e), y0); /* y0 (1 + e + e^2) = (1 - e^3)/d: one third-order step */
  const double q = n * y;
  const double r = CV_FMA(-d, q, n);
  return CV_FMA(r, y, q);
#else
  return n / d;
#endif
}

/* ------------------------------------------------------------------------- */
/* sin / cos                                                                  */
/* ------------------------------------------------------------------------- */

/* 2/pi: 64 zero bits, then 1216 fractional bits (tools/gen_math_tables.py). */
#if defined(__HIPCC__) || defined(__HIP__)
__device__ __constant__ static const uint64_t cv_two_over_pi_dev[20] = {
    0x0000000000000000ULL, 0xA2F9836E4E441529ULL, 0xFC2757D1F534DDC0ULL, 0xDB6295993C439041ULL,
    0xFE5163ABDEBBC561ULL, 0xB7246E3A424DD2E0ULL, 0x06492EEA09D1921CULL, 0xFE1DEB1CB129A73EULL,
    0xE88235F52EBB4484ULL, 0xE99C7026B45F7E41ULL, 0x3991D639835339F4ULL, 0x9C845F8BBDF9283BULL,
    0x1FF897FFDE05980FULL, 0xEF2F118B5A0A6D1FULL, 0x6D367ECF27CB09B7ULL, 0x4F463F669E5FEA2DULL,
    0x7527BAC7EBE5F17BULL, 0x3D0739F78A5292EAULL, 0x6BFB5FB11F8D5D08ULL, 0x56033046FC7B6BABULL};
#endif
static const uint64_t cv_two_over_pi_host[20] = {
    0x0000000000000000ULL, 0xA2F9836E4E441529ULL, 0xFC2757D1F534DDC0ULL, 0xDB6295993C439041ULL,
    0xFE5163ABDEBBC561ULL, 0xB7246E3A424DD2E0ULL, 0x06492EEA09D1921CULL, 0xFE1DEB1CB129A73EULL,
    0xE88235F52EBB4484ULL, 0xE99C7026B45F7E41ULL, 0x3991D639835339F4ULL, 0x9C845F8BBDF9283BULL,
    0x1FF897FFDE05980FULL, 0xEF2F118B5A0A6D1FULL, 0x6D367ECF27CB09B7ULL, 0x4F463F669E5FEA2DULL,
    0x7527BAC7EBE5F17BULL, 0x3D0739F78A5292EAULL, 0x6BFB5FB11F8D5D08ULL, 0x56033046FC7B6BABULL};

CV_HD uint64_t cv_two_over_pi_word(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return cv_two_over_pi_dev[i];
#else
  return cv_two_over_pi_host[i];
#endif
}

/* Payne-Hanek for finite |x| >= 2^20: returns quadrant (mod 4) and y0+y1 = x - n*pi/2,
 * |y0| <= pi/4 (+ rounding).  Pure 64-bit integer arithmetic up to the final
 * conversion, so host and device agree bit for bit. */
CV_HD int cv_rem_pio2_large(double x, double *y0, double *y1) {
  const uint64_t ux = cv_bits(x);
  const int neg = (int)(ux >> 63);
  const int bexp = (int)((ux >> 52) & 0x7ff);
  const uint64_t M = (ux & 0x000fffffffffffffULL) | 0x0010000000000000ULL; /* 53-bit integer */
  const int E = bexp - 1075;                                             /* |x| = M * 2^E, E >= -32 here */
  /* window of 192 bits of 2/pi starting at fractional bit k0 = E-1 (bits k <= 0 are zero);
   * table bit index i <-> k = i - 63 */
  const int i0 = E + 62;
  const int wd = i0 >> 6, off = i0 & 63;
  uint64_t t0 = cv_two_over_pi_word(wd), t1 = cv_two_over_pi_word(wd + 1), t2 = cv_two_over_pi_word(wd + 2),
           t3 = cv_two_over_pi_word(wd + 3);
  uint64_t w0, w1, w2;
  if (off) {
    w0 = (t0 << off) | (t1 >> (64 - off));
    w1 = (t1 << off) | (t2 >> (64 - off));
    w2 = (t2 << off) | (t3 >> (64 - off));
  } else {
    w0 = t0;
    w1 = t1;
    w2 = t2;
  }
  /* P = M * (w0:w1:w2) mod 2^192, value = P * 2^-190 quarter turns */
  uint64_t p3 = M * w2;
  uint64_t h2 = cv_mulhi64(M, w2);
  uint64_t l1 = M * w1;
  uint64_t h1 = cv_mulhi64(M, w1);
  uint64_t l0 = M * w0;
  uint64_t p2 = l1 + h2;
  uint64_t c1 = (p2 < l1) ? 1u : 0u;
  uint64_t p1 = l0 + h1 + c1;
  int q = (int)(p1 >> 62);
  /* fraction aligned to the top of a 192-bit word */
  uint64_t A = (p1 << 2) | (p2 >> 62);
  uint64_t B = (p2 << 2) | (p3 >> 62);
  uint64_t C = (p3 << 2);
  int fneg = 0;
  if (A >> 63) { /* fraction >= 1/2: round to nearest integer, fraction becomes negative */
    q += 1;
    fneg = 1;
    /* two's complement of A:B:C */
    C = ~C + 1;
    uint64_t cb = (C == 0) ? 1u : 0u;
    B = ~B + cb;
    uint64_t ca = (cb && B == 0) ? 1u : 0u;
    A = ~A + ca;
  }
  /* normalise the magnitude A:B:C / 2^192 */
  int sh = 0;
  if (A == 0) {
    A = B;
    B = C;
    C = 0;
    sh = 64;
  }
  double fh, fl;
  if (A == 0) {
    fh = 0.0;
    fl = 0.0;
  } else {
    int lz = cv_clz64(A);
    if (lz) {
      A = (A << lz) | (B >> (64 - lz));
      B = (B << lz) | (C >> (64 - lz));
    }
    sh += lz;
    /* top 53 bits -> fh, next 53 bits -> fl */
    uint64_t hi53 = A >> 11;
    uint64_t lo53 = ((A & 0x7ffULL) << 42) | (B >> 22);
    double sc_h = cv_from_bits((uint64_t)(1023 - 53 - sh) << 52);  /* 2^(-53-sh) */
    double sc_l = cv_from_bits((uint64_t)(1023 - 106 - sh) << 52); /* 2^(-106-sh) */
    fh = (double)hi53 * sc_h;
    fl = (double)lo53 * sc_l;
  }
  /* (fh+fl) * pi/2 */
  const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  double r0 = fh * PIO2_HI;
  double e0 = CV_FMA(fh, PIO2_HI, -r0);
  double r1 = e0 + CV_FMA(fh, PIO2_LO, fl * PIO2_HI);
  if (fneg) {
    r0 = -r0;
    r1 = -r1;
  }
  if (neg) {
    q = -q;
    r0 = -r0;
    r1 = -r1;
  }
  *y0 = r0;
  *y1 = r1;
  return q & 3;
}

/* x - n*pi/2 as head/tail; returns n mod 4.  x finite, |x| > pi/4. */
CV_HD int cv_rem_pio2(double x, double *y0, double *y1) {
  const double INVPIO2 = 6.36619772367581382433e-01, /* 0x3FE45F306DC9C883 */
      PIO2_1 = 1.57079632673412561417e+00,           /* 0x3FF921FB54400000: first 33 bits of pi/2 */
      PIO2_1T = 6.07710050650619224932e-11,          /* 0x3DD0B4611A626331: pi/2 - PIO2_1 */
      PIO2_2 = 6.07710050630396597660e-11,           /* 0x3DD0B4611A600000: second 33 bits */
      PIO2_2T = 2.02226624879595063154e-21,          /* 0x3BA3198A2E037073 */
      PIO2_3 = 2.02226624871116645580e-21,           /* 0x3BA3198A2E000000: third 33 bits */
      PIO2_3T = 8.47842766036889956997e-32;          /* 0x397B839A252049C1 */
  const uint32_t ix = cv_hi(x) & 0x7fffffffu;
  if (ix >= 0x41300000u) /* |x| >= 2^20 */
    return cv_rem_pio2_large(x, y0, y1);
  double fn = CV_RINT(x * INVPIO2);
  double r = CV_FMA(-fn, PIO2_1, x); /* fn*PIO2_1 is exact (33+20 bits) */
  double w = fn * PIO2_1T;
  double a = r - w;
  int ediff = (int)(ix >> 20) - (int)((cv_hi(a) >> 20) & 0x7ff);
  if (ediff > 16) { /* 2nd iteration: good to 118 bits */
    double t = r;
    w = fn * PIO2_2;
    r = t - w;
    w = CV_FMA(fn, PIO2_2T, -((t - r) - w));
    a = r - w;
    ediff = (int)(ix >> 20) - (int)((cv_hi(a) >> 20) & 0x7ff);
    if (ediff > 49) { /* 3rd iteration: 151 bits, covers every double */
      t = r;
      w = fn * PIO2_3;
      r = t - w;
      w = CV_FMA(fn, PIO2_3T, -((t - r) - w));
      a = r - w;
    }
  }
  *y0 = a;
  *y1 = (r - a) - w;
  return ((int)fn) & 3;
}

/* sin(K*pi/64), cos(K*pi/64), K = 0..127, as double-double {S_hi, S_lo, C_hi, C_lo} (cv_sincos_table.h).
 * Host code reads the static copy; device code reads the __constant__ copy unless the caller passes its
 * own pointer (the hot kernels keep a copy in LDS: 4 KiB per workgroup, two ds_read_b128 per evaluation). */
typedef const double (*cv_sc_tab_t)[4];
#if defined(__HIPCC__) || defined(__HIP__)
__device__ __constant__ static const double cv_sc_table_dev[128][4] = {CV_SC_TABLE_ROWS};
#endif
static const double cv_sc_table_host[128][4] = {CV_SC_TABLE_ROWS};
CV_HD cv_sc_tab_t cv_sc_table(void) {
#if defined(__HIP_DEVICE_COMPILE__)
  return cv_sc_table_dev;
#else
  return cv_sc_table_host;
#endif
}

/* sin/cos of K*pi/64 + (y + yl), |y| <= pi/128:
 *   sin = S + [ C*y + ( S_lo + S*(cos y - 1) + C*(sin y - y) ) ],   cos = C + [ -S*y + ( C_lo + C*(cos y - 1) - S*(sin y - y) ) ]
 * with sin y - y = y^3 (s1 + s2 z + s3 z^2) + yl and cos y - 1 = z (c1 + c2 z + c3 z^2), z = y^2 (Taylor
 * coefficients: truncation < 2^-61 on this interval; the term -y*yl of cos(y + yl) is below 2^-63 and left out).  The bracket is evaluated with one fma, so each
 * result carries two roundings; the error is < 0.6 ulp except where the result is much smaller than pi/64
 * times its partner (next to the zeros of sin resp. cos), where it stays below 1 ulp. */
CV_HD void cv_sincos_core(int K, double y, double yl, cv_sc_tab_t T, double *sn, double *cs) {
  const double Sh = T[K][0], Sl = T[K][1], Ch = T[K][2], Cl = T[K][3];
  const double z = y * y;
  const double ps = cv_fma_ks(z, cv_fma_ks(z, -1.98412698412698412698e-04, 8.33333333333333321769e-03), -1.66666666666666657415e-01);
  const double pc = cv_fma_ks(z, cv_fma_ks(z, -1.38888888888888894189e-03, 4.16666666666666643537e-02), -0.5);
  const double yz = y * z;
  const double sl = CV_FMA(yz, ps, yl);
  const double cm = z * pc;
  double e = CV_FMA(Sh, cm, Sl);
  e = CV_FMA(Ch, sl, e);
  *sn = Sh + CV_FMA(Ch, y, e);
  double f = CV_FMA(Ch, cm, Cl);
  f = CV_FMA(-Sh, sl, f);
  *cs = Ch + CV_FMA(-Sh, y, f);
}

/* sin and cos of x together, table-driven.
 *
 * Main path (|x| < 6.25 and x not within 2^-20 of a multiple of pi/64), branch-free:
 *   k  = nearest integer to x * 64/pi        |k| <= 127 (magic-number rounding, see CV_RND_MAGIC)
 *   r1 = fma(-k, A, x)                       exact: A has 44 bits, k A is a double, the difference fits 53 bits
 *   t  = fma(-k, B, r1)                      head of the reduced argument, |t| <= pi/128; k B is a double too
 *   u  = r1 - t                              exact (|u| < 2^-39, a multiple of ulp(t) >= 2^-72)
 *   tl = fma(-k, B, u)                       exact: the rounding error of t
 * with pi/64 = A + B + 1.1e-28 (two 44-bit pieces): what is left out is |k| 1.1e-28 < 2^-86, below 2^-66 of
 * |t| >= 2^-20.  Every other argument (larger, tiny, non-finite, or deeply cancelling
 * such as theta == fl(pi/2), which the equatorial rays hold for ever) is first reduced modulo pi/2 by
 * cv_rem_pio2 (fdlibm-style iterations / 192-bit Payne-Hanek) and then by a three-piece step (38 + 38 + 53 bits
 * of pi/64, to 2^-134) with |k| <= 17.  Which path an argument takes is a function of the argument alone, and both paths end in
 * cv_sincos_core, so host and device agree bit for bit. */
/* WIDE (a compile-time constant at every call): T has 256 rows, the 128 rows twice, and the main path indexes it
 * with k + 128 = 1..255 straight from the low word of the rounding sum -- no "& 127".  Same values either way. */
/* main path: returns 1 with the arguments of cv_sincos_core, or 0 when x belongs to the other path */
CV_HD int cv_sincos_main_args(double x, int wide, int *K, double *y, double *yl) {
  const double magic = wide ? CV_RND_MAGIC + 128.0 : CV_RND_MAGIC;
  const double kb = cv_fma_ks(x, CV_64OPI, magic); /* addend from a scalar pair: as a plain fma the compiler emits
                                                      v_mov_b64 + v_fmac_f64 to keep the constant's VGPR pair */
  const double k = kb - magic;
  const double r1 = CV_FMA(-k, CV_PIO64_A, x);
  const double t = CV_FMA(-k, CV_PIO64_B, r1);
  if (!(CV_FABS(x) < 6.25 && CV_FABS(t) >= 9.5367431640625e-07 /* 2^-20 */)) return 0;
  const double u = r1 - t;
  *y = t;
  *yl = CV_FMA(-k, CV_PIO64_B, u);
  *K = wide ? (int)cv_lo(kb) : ((int)cv_lo(kb) & 127);
  return 1;
}
/* every other argument: returns 1 with the arguments of cv_sincos_core (K in 0..127), or 0 with *sn, *cs final */
CV_HD int cv_sincos_other_args(double x, int *K, double *y, double *yl, double *sn, double *cs) {
  /* theta == fl(pi/2), bit for bit: the rays of the equatorial plane (p_theta == 0) keep this value for ever and
   * come here in every step.  The general route below gives sin = 1 and cos = RN(pi/2 - fl(pi/2)) for it
   * (tests/test_cv_math.py checks the shortcut against that route); returning them at once keeps the waves that
   * hold such rays from paying ~100 instructions per step -- they are among the last to finish in every frame (the default camera's middle pixel column). */
  if (cv_bits(x) == 0x3FF921FB54442D18ULL) {
    *sn = 1.0;
    *cs = 6.123233995736766036e-17; /* 0x3C91A62633145C07 */
    return 0;
  }
  const uint32_t ix = cv_hi(x) & 0x7fffffffu;
  if (ix >= 0x7ff00000u) { /* inf / nan */
    *sn = *cs = x - x;
    return 0;
  }
  if (ix < 0x3e400000u) { /* |x| < 2^-27: sin x = x (keeps the sign of zero), cos x = 1 */
    *sn = x;
    *cs = 1.0;
    return 0;
  }
  double y0, y1;
  int n = 0;
  if (ix <= 0x3fe921fbu) { /* |x| <= ~pi/4 */
    y0 = x;
    y1 = 0.0;
  } else {
    n = cv_rem_pio2(x, &y0, &y1);
  }
  const double k2 = CV_RINT(y0 * CV_64OPI);
  const double a1 = CV_FMA(-k2, CV_PIO64_1, y0);
  const double t2 = CV_FMA(-k2, CV_PIO64_2, a1);
  const double u2 = a1 - t2;
  *y = t2;
  *yl = CV_FMA(-k2, CV_PIO64_3, CV_FMA(-k2, CV_PIO64_2, u2)) + y1;
  *K = (n * 32 + (int)k2) & 127;
  return 1;
}
CV_HD void cv_sincos_other(double x, cv_sc_tab_t T, double *sn, double *cs) {
  double y, yl;
  int K;
  if (cv_sincos_other_args(x, &K, &y, &yl, sn, cs)) cv_sincos_core(K, y, yl, T, sn, cs);
}
CV_HD void cv_sincos_impl(double x, cv_sc_tab_t T, int wide, double *sn, double *cs) {
  double y, yl;
  int K;
  if (cv_sincos_main_args(x, wide, &K, &y, &yl))
    cv_sincos_core(K, y, yl, T, sn, cs);
  else
    cv_sincos_other(x, T, sn, cs);
}
/* sin and cos as above, and whether |sin x| > 2^-60 (the fast Euler step's guard on the sine).  On the main path
 * that is implied -- x = K pi/64 + y with 2^-20 <= |y| <= pi/128: for K = 0 (mod 64) |sin x| = |sin y| >= 2^-21,
 * otherwise |sin x| >= sin(pi/64) cos y - |sin y| > 0.024 -- so the compare is only executed on the other path. */
CV_HD int cv_sincos_guarded(double x, cv_sc_tab_t T, int wide, double *sn, double *cs) {
  double y, yl;
  int K;
  if (cv_sincos_main_args(x, wide, &K, &y, &yl)) {
    cv_sincos_core(K, y, yl, T, sn, cs);
    return 1;
  }
  cv_sincos_other(x, T, sn, cs);
  return CV_FABS(*sn) > 0x1p-60;
}
CV_HD void cv_sincos_t(double x, cv_sc_tab_t T, double *sn, double *cs) { cv_sincos_impl(x, T, 0, sn, cs); }
/* T = the 128 rows twice (the hot kernels' LDS copy where there is room for it) */
CV_HD void cv_sincos_tw(double x, cv_sc_tab_t T, double *sn, double *cs) { cv_sincos_impl(x, T, 1, sn, cs); }

CV_HD void cv_sincos(double x, double *sn, double *cs) { cv_sincos_t(x, cv_sc_table(), sn, cs); }

CV_HD double cv_sin_t(double x, cv_sc_tab_t T) {
  double s, c;
  cv_sincos_t(x, T, &s, &c);
  return s;
}
CV_HD double cv_sin(double x) {
  double s, c;
  cv_sincos(x, &s, &c);
  return s;
}
CV_HD double cv_cos(double x) {
  double s, c;
  cv_sincos(x, &s, &c);
  return c;
}

/* ------------------------------------------------------------------------- */
/* atan / atan2                                                               */
/* ------------------------------------------------------------------------- */

/* Two tables (cv_atan_table.h).  cv_atan_table(): the RECIPROCAL branch, rows j = -128..0 at c = j/256,
 * {X_hi, X_lo, q1..q5, 0} (8.1 KiB) -- the one the Euler loop reads in every step; host: static copy; device:
 * __constant__ copy unless the caller passes its own (the Interstellar kernels keep it in LDS).
 * cv_atan_dtable(): the DIRECT branch, rows j = 56..256 at c = j/128, {X_hi, X_lo, q1..q6} (12.6 KiB); only
 * arguments in [0.4375, 2) read it (the throat's neighbourhood, the sky lookup), always from the static /
 * __constant__ copy. */
typedef const double (*cv_atan_tab_t)[8];
#if defined(__HIPCC__) || defined(__HIP__)
__device__ __constant__ static const double cv_atan_table_dev[CV_ATAN_TABLE_N][8] __attribute__((aligned(16))) = {CV_ATAN_TABLE_ROWS};
__device__ __constant__ static const double cv_atan_dtable_dev[CV_ATAN_DTABLE_N][8] __attribute__((aligned(16))) = {CV_ATAN_DTABLE_ROWS};
#endif
static const double cv_atan_table_host[CV_ATAN_TABLE_N][8] = {CV_ATAN_TABLE_ROWS};
static const double cv_atan_dtable_host[CV_ATAN_DTABLE_N][8] = {CV_ATAN_DTABLE_ROWS};
CV_HD cv_atan_tab_t cv_atan_table(void) {
#if defined(__HIP_DEVICE_COMPILE__)
  return cv_atan_table_dev;
#else
  return cv_atan_table_host;
#endif
}
CV_HD cv_atan_tab_t cv_atan_dtable(void) {
#if defined(__HIP_DEVICE_COMPILE__)
  return cv_atan_dtable_dev;
#else
  return cv_atan_dtable_host;
#endif
}

/* |x| < 0.4375 (odd minimax polynomial, fdlibm coefficients), |x| >= 2^66, NaN */
CV_HD double cv_atan_edge(double x) {
  const double aT0 = 3.33333333333329318027e-01, /* 0x3FD555555555550D */
      aT1 = -1.99999999998764832476e-01,         /* 0xBFC999999998EBC4 */
      aT2 = 1.42857142725034663711e-01,          /* 0x3FC24924920083FF */
      aT3 = -1.11111104054623557880e-01,         /* 0xBFBC71C6FE231671 */
      aT4 = 9.09088713343650656196e-02,          /* 0x3FB745CDC54C206E */
      aT5 = -7.69187620504482999495e-02,         /* 0xBFB3B0F2AF749A6D */
      aT6 = 6.66107313738753120669e-02,          /* 0x3FB10D66A0D03D51 */
      aT7 = -5.83357013379057348645e-02,         /* 0xBFADDE2D52DEFD9A */
      aT8 = 4.97687799461593236017e-02,          /* 0x3FA97B4B24760DEB */
      aT9 = -3.65315727442169155270e-02,         /* 0xBFA2B4442C6A6C2F */
      aT10 = 1.62858201153657823623e-02;         /* 0x3F90AD3AE322DA11 */
  const uint32_t hx = cv_hi(x);
  const uint32_t ix = hx & 0x7fffffffu;
  if (ix >= 0x44100000u) { /* |x| >= 2^66 */
    if (ix > 0x7ff00000u || (ix == 0x7ff00000u && cv_lo(x) != 0)) return x + x; /* nan */
    const double z = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return (hx >> 31) ? -z : z;
  }
  if (ix < 0x3e400000u) { /* |x| < 2^-27 */
    CV_KEEP_BRANCH();
    return x;
  }
  const double z = x * x;
  const double w = z * z;
  const double s1 = z * cv_fma_c(w, cv_fma_c(w, cv_fma_c(w, cv_fma_c(w, cv_fma_c(w, aT10, aT8), aT6), aT4), aT2), aT0);
  const double s2 = w * cv_fma_c(w, cv_fma_c(w, cv_fma_c(w, cv_fma_c(w, aT9, aT7), aT5), aT3), aT1);
  return x - x * (s1 + s2);
}
/* the unrounded pair behind cv_atan(ax) for a finite ax >= 2^-27 and < 2^66 (atan2 adds its own corrections before
 * the one final rounding): atan(ax) = *hi + *lo with |*lo| << |*hi|; hi + lo rounded is exactly cv_atan(ax) */
CV_HD void cv_atan_pair(double ax, cv_atan_tab_t T, double *hi, double *lo) {
  const uint32_t ix = cv_hi(ax);
  if (ix < 0x3fdc0000u) { /* the odd polynomial of cv_atan_edge */
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    const double z = ax * ax;
    const double w = z * z;
    const double s1 = z * CV_FMA(w, CV_FMA(w, CV_FMA(w, CV_FMA(w, CV_FMA(w, aT10, aT8), aT6), aT4), aT2), aT0);
    const double s2 = w * CV_FMA(w, CV_FMA(w, CV_FMA(w, CV_FMA(w, aT9, aT7), aT5), aT3), aT1);
    *hi = ax;
    *lo = -(ax * (s1 + s2));
    return;
  }
  if (ix >= 0x40000000u) { /* reciprocal branch: same operations as cv_atan_row */
    const double u = cv_div_nr(-1.0, ax);
    const double jb = u + CV_RND_MAGIC_256TH;
    const double h = u - (jb - CV_RND_MAGIC_256TH);
    const double *R = T[(int)cv_lo(jb) + 128];
    const double Q = CV_FMA(h, CV_FMA(h, CV_FMA(h, CV_FMA(h, R[6], R[5]), R[4]), R[3]), R[2]);
    *hi = R[0];
    *lo = CV_FMA(h, Q, R[1]);
    return;
  }
  const double jb = ax + CV_RND_MAGIC_128TH; /* direct branch: same operations as cv_atan_row_direct */
  const double h = ax - (jb - CV_RND_MAGIC_128TH);
  const double *R = cv_atan_dtable()[(int)cv_lo(jb) - 56];
  const double Q = CV_FMA(h, CV_FMA(h, CV_FMA(h, CV_FMA(h, CV_FMA(h, R[7], R[6]), R[5]), R[4]), R[3]), R[2]);
  *hi = R[0];
  *lo = CV_FMA(h, Q, R[1]);
}

/* atan x, table-driven for 0.4375 <= |x| < 2^66:
 *   |x| >= 2:  u = -1/|x|  (correctly rounded reciprocal; u in [-1/2, 0]),  j = rint(256 u), h = u - j/256 exact,
 *              |h| <= 2^-9:   atan|x| = X_j + h (q1 + q2 h + .. + q5 h^4),   X_j = pi/2 + atan(j/256)
 *   |x| <  2:  u = |x|,  j = rint(128 u), |h| <= 2^-8:   atan|x| = X_j + h (q1 + .. + q6 h^5),  X_j = atan(j/128)
 * (magic-number rounding to multiples of 2^-8 resp. 2^-7), evaluated as X_hi + fma(h, Q, X_lo).  Error: 0.5 ulp of
 * the last addition + the rounding of the reciprocal (<= 2^-53 u/(1+u^2) <= 0.2 ulp of a result >= atan 2) +
 * Taylor truncation (h^6/6 <= 2^-56.6 resp. h^7/7 <= 2^-58.8, < 0.05 ulp): < 0.75 ulp (measured < 0.62).
 * The reciprocal branch is the one the Interstellar Euler step takes (x >= 2 outside |l| < a + pi m): its rows
 * are twice as fine as the direct branch's so that its polynomial is one term shorter. */
/* the table step of the reciprocal branch for a given u in [-1/2, 0] */
CV_HD double cv_atan_row(double u, cv_atan_tab_t T) {
  /* u + 1.5 2^44 rounds u to the nearest multiple of 2^-8 (ties to even j, like rint(256 u)) and leaves j in the
   * low mantissa bits; plain additions, so the constant can sit in a scalar register */
  const double jb = u + CV_RND_MAGIC_256TH;
  const double jq = jb - CV_RND_MAGIC_256TH; /* j / 256 */
  const double h = u - jq;                   /* exact */
#if defined(__HIP_DEVICE_COMPILE__)
  /* the 64-byte row as four 16-byte loads (tables are 16-byte aligned): from LDS that is four ds_read_b128
   * off ONE address register (their offset field reaches the whole LDS; ds_read2_b64's does not) */
  typedef double cv_f64x2 __attribute__((ext_vector_type(2)));
  const cv_f64x2 *R = (const cv_f64x2 *)T[(int)cv_lo(jb) + 128];
  const cv_f64x2 r01 = R[0], r23 = R[1], r45 = R[2], r67 = R[3];
  const double Q = cv_fma_c(h, cv_fma_c(h, cv_fma_c(h, cv_fma_c(h, r67.x, r45.y), r45.x), r23.y), r23.x);
  return r01.x + CV_FMA(h, Q, r01.y);
#else
  const double *R = T[(int)cv_lo(jb) + 128];
  const double Q = cv_fma_c(h, cv_fma_c(h, cv_fma_c(h, cv_fma_c(h, R[6], R[5]), R[4]), R[3]), R[2]);
  return R[0] + CV_FMA(h, Q, R[1]);
#endif
}
/* the table step of the direct branch for u = |x| in [0.4375, 2) */
CV_HD double cv_atan_row_direct(double u) {
  const double jb = u + CV_RND_MAGIC_128TH;
  const double h = u - (jb - CV_RND_MAGIC_128TH);
  const double *R = cv_atan_dtable()[(int)cv_lo(jb) - 56];
  const double Q = cv_fma_c(h, cv_fma_c(h, cv_fma_c(h, cv_fma_c(h, cv_fma_c(h, R[7], R[6]), R[5]), R[4]), R[3]), R[2]);
  return R[0] + CV_FMA(h, Q, R[1]);
}
/* table path for 0.4375 <= ax < 2^66 (ix = high word of ax): atan(ax) */
CV_HD double cv_atan_main(double ax, uint32_t ix, cv_atan_tab_t T) {
  if (ix >= 0x40000000u) return cv_atan_row(cv_div_nr(-1.0, ax), T);
  return cv_atan_row_direct(ax);
}

CV_HD double cv_atan_t(double x, cv_atan_tab_t T) {
  const uint64_t ux = cv_bits(x);
  const uint32_t hx = (uint32_t)(ux >> 32);
  const uint32_t ix = hx & 0x7fffffffu;
  if (ix - 0x3fdc0000u >= 0x44100000u - 0x3fdc0000u) return cv_atan_edge(x);
  const double r = cv_atan_main(cv_from_bits(ux & 0x7fffffffffffffffULL), ix, T);
  return cv_from_bits(cv_bits(r) | (ux & 0x8000000000000000ULL));
}

CV_HD double cv_atan(double x) { return cv_atan_t(x, cv_atan_table()); }

CV_HD double cv_atan2(double y, double x) {
  const double PI_LO = 1.2246467991473531772E-16; /* 0x3CA1A62633145C07 */
  const double PI_O_4 = 7.8539816339744827900E-01, PI_O_2 = 1.5707963267948965580E+00;
  const double TINY = 1.0e-300;
  const uint32_t hx = cv_hi(x), lx = cv_lo(x), hy = cv_hi(y), ly = cv_lo(y);
  const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  /* both operands finite, non-zero and not subnormal (exponent field 1 .. 0x7fe): none of fdlibm's special cases below can apply,
   * and the per-pixel caller's lanes skip their dozen compares with two */
  const int plain = (ix - 0x00100000u) < 0x7fe00000u && (iy - 0x00100000u) < 0x7fe00000u;
  if (!plain && (ix > 0x7ff00000u || (ix == 0x7ff00000u && lx != 0) || iy > 0x7ff00000u || (iy == 0x7ff00000u && ly != 0)))
    return x + y;                                        /* nan */
  if (hx == 0x3ff00000u && lx == 0) return cv_atan(y);   /* x == 1 */
  const int m = (int)((hy >> 31) & 1) | (int)((hx >> 30) & 2); /* 2*sign(x) + sign(y) */
  if (!plain) {
    if ((iy | ly) == 0) { /* y == 0 */
      switch (m) {
        case 0:
        case 1:
          return y;
        case 2:
          return CV_PI + TINY;
        default:
          return -CV_PI - TINY;
      }
    }
    if ((ix | lx) == 0) return (hy >> 31) ? -PI_O_2 - TINY : PI_O_2 + TINY; /* x == 0 */
    if (ix == 0x7ff00000u) { /* x inf */
      if (iy == 0x7ff00000u) {
        switch (m) {
          case 0:
            return PI_O_4 + TINY;
          case 1:
            return -PI_O_4 - TINY;
          case 2:
            return 3.0 * PI_O_4 + TINY;
          default:
            return -3.0 * PI_O_4 - TINY;
        }
      } else {
        switch (m) {
          case 0:
            return 0.0;
          case 1:
            return -0.0;
          case 2:
            return CV_PI + TINY;
          default:
            return -CV_PI - TINY;
        }
      }
    }
    if (iy == 0x7ff00000u) return (hy >> 31) ? -PI_O_2 - TINY : PI_O_2 + TINY; /* y inf */
  }
  const int k = (int)(iy >> 20) - (int)(ix >> 20);
  double zh, zl; /* atan(|y/x|) = zh + zl, unrounded */
  if (k > 60) { /* |y/x| > 2^60 */
    zh = PI_O_2;
    zl = 0.5 * PI_LO;
  } else if ((hx >> 31) && k < -60) {
    zh = zl = 0.0; /* |y|/x < -2^-60 with x < 0 */
  } else {
    /* q = RN(|y/x|) is off by up to half an ulp, which alone would cost up to 0.5 ulp of the result (fdlibm's
     * atan2, 1.45 ulp measured on the sky-lookup arguments): the exact remainder rem = |y| - q |x| (one fma) puts
     * it back, atan(q + rem/|x|) = atan q + (rem/|x|)/(1 + q^2) + O(2^-106).  The pair (zh, zl) keeps what cv_atan
     * would round away, and pi - z is formed with a Fast2Sum, so the result is rounded ONCE: < 0.65 ulp measured. */
    const double ay = CV_FABS(y), ax = CV_FABS(x);
    const double q = ay / ax;
    const uint32_t iq = cv_hi(q);
    if (iq < 0x3e400000u || iq >= 0x44100000u || !(ay - ay == 0.0)) { /* tiny (subnormal remainder), huge, overflowed quotient */
      zh = cv_atan(q);
      zl = 0.0;
    } else {
      const double rem = CV_FMA(-q, ax, ay);
      cv_atan_pair(q, cv_atan_table(), &zh, &zl);
      zl = zl + (rem / ax) / CV_FMA(q, q, 1.0);
    }
  }
  switch (m) {
    case 0:
      return zh + zl;
    case 1:
      return -(zh + zl);
    default: { /* pi - z with z <= pi/2 < pi: Fast2Sum of the leading parts, the tails added in one go */
      const double s = CV_PI - zh;
      const double e = (CV_PI - s) - zh;
      const double r = s + ((e + PI_LO) - zl);
      return m == 2 ? r : -r;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* acos                                                                       */
/* ------------------------------------------------------------------------- */

CV_HD double cv_acos_R(double z) { /* (asin(x)-x)/x^3 as p(z)/q(z), z = x^2 */
  const double pS0 = 1.66666666666666657415e-01, /* 0x3FC5555555555555 */
      pS1 = -3.25565818622400915405e-01,         /* 0xBFD4D61203EB6F7D */
      pS2 = 2.01212532134862925881e-01,          /* 0x3FC9C1550E884455 */
      pS3 = -4.00555345006794114027e-02,         /* 0xBFA48228B5688F3B */
      pS4 = 7.91534994289814532176e-04,          /* 0x3F49EFE07501B288 */
      pS5 = 3.47933107596021167570e-05,          /* 0x3F023DE10DFDF709 */
      qS1 = -2.40339491173441421878e+00,         /* 0xC0033A271C8A2D4B */
      qS2 = 2.02094576023350569471e+00,          /* 0x40002AE59C598AC8 */
      qS3 = -6.88283971605453293030e-01,         /* 0xBFE6066C1B8D0159 */
      qS4 = 7.70381505559019352791e-02;          /* 0x3FB3B8C5B12E9282 */
  double p = z * CV_FMA(z, CV_FMA(z, CV_FMA(z, CV_FMA(z, CV_FMA(z, pS5, pS4), pS3), pS2), pS1), pS0);
  double q = CV_FMA(z, CV_FMA(z, CV_FMA(z, CV_FMA(z, qS4, qS3), qS2), qS1), 1.0);
  return p / q;
}

CV_HD double cv_acos(double x) {
  const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  const uint32_t hx = cv_hi(x);
  const uint32_t ix = hx & 0x7fffffffu;
  if (ix >= 0x3ff00000u) { /* |x| >= 1 */
    if (((ix - 0x3ff00000u) | cv_lo(x)) == 0) {
      if (hx >> 31) return 2.0 * PIO2_HI + 2.0 * PIO2_LO; /* acos(-1) = pi */
      return 0.0;                                         /* acos(1) = 0 */
    }
    return (x - x) / (x - x); /* |x| > 1 or nan: NaN */
  }
  if (ix < 0x3fe00000u) {                                 /* |x| < 0.5 */
    if (ix <= 0x3c600000u) return PIO2_HI + PIO2_LO;      /* |x| < 2^-57 */
    double r = cv_acos_R(x * x);
    return PIO2_HI - (x - (PIO2_LO - x * r));
  }
  if (hx >> 31) { /* x < -0.5 */
    double z = (1.0 + x) * 0.5;
    double s = CV_SQRT(z);
    double r = cv_acos_R(z);
    double w = CV_FMA(r, s, -PIO2_LO);
    return CV_PI - 2.0 * (s + w);
  }
  /* x > 0.5 */
  double z = (1.0 - x) * 0.5;
  double s = CV_SQRT(z);
  double df = cv_from_bits(cv_bits(s) & 0xffffffff00000000ULL);
  double c = CV_FMA(-df, df, z) / (s + df);
  double r = cv_acos_R(z);
  double w = CV_FMA(r, s, c);
  return 2.0 * (df + w);
}

/* ------------------------------------------------------------------------- */
/* natural logarithm                                                          */
/* ------------------------------------------------------------------------- */

/* {2 invc, logc_hi - LN2_HI, logc_lo - LN2_LO} per 1/512-wide slice of [1, 2) (cv_log_table.h).  Host code reads the
 * static copy, device code the __constant__ copy unless the caller passes its own (the Interstellar kernels keep
 * one in LDS: 12 KiB per workgroup). */
typedef const double (*cv_log_tab_t)[3];
#if defined(__HIPCC__) || defined(__HIP__)
__device__ __constant__ static const double cv_log_table_dev[CV_LOG_TABLE_N][3] = {CV_LOG_TABLE_ROWS};
#endif
static const double cv_log_table_host[CV_LOG_TABLE_N][3] = {CV_LOG_TABLE_ROWS};
CV_HD cv_log_tab_t cv_log_table(void) {
#if defined(__HIP_DEVICE_COMPILE__)
  return cv_log_table_dev;
#else
  return cv_log_table_host;
#endif
}

/* log x, table-driven.  x = 2^e g with g in [1/2, 1) -- what v_frexp_exp_i32_f64 / v_frexp_mant_f64 deliver, one
 * instruction each -- i.e. x = 2^k z, z = 2g in [1, 2), k = e - 1; i = top 9 mantissa bits; c = 1/invc_i is close to z:
 *   r  = fma(g, 2 invc, -1)           = z invc - 1, exact (invc is a multiple of 2^-10 and |r| <= 2^-9); the table holds 2 invc
 *   w  = k*LN2_HI + logc_hi           exact (both multiples of 2^-32); computed as e*LN2_HI + (logc_hi - LN2_HI)
 *                                     (the table holds logc_hi - LN2_HI, exactly, and logc_lo - LN2_LO)
 *   hi + lo = w + r                   Fast2Sum (w == 0 or |w| >= |r|, checked by the table generator)
 *   log x = hi + (lo + k*LN2_LO + logc_lo + r^2 (-1/2 + r/3 - ... + r^5/7))
 * Taylor truncation < 2^-59 relative even on the slice next to 1 (invc = 1, w = 0, log x = r + ...), so the
 * error is 0.5 ulp of the final addition plus ~0.02 ulp.
 * Arguments outside [1/2, 2) (k >= 1 or k <= -2) have |w| >= 0.69 while |r| <= 2^-9, so w needs no help from r:
 *   log x = w + v,   v = t + r (1 + r (-1/2 + r (1/3 + r (-1/4 + r/5)))),   t = k*LN2_LO + logc_lo
 * -- one Horner chain of five fmas ending in the small term (|v| < 2^-8: its own rounding and that of the inner
 * factor ~1 are below 2^-61), no Fast2Sum and two Taylor terms less (r^6/6 <= 2^-56.6 absolute against ulp(0.69) =
 * 2^-53, on the two slices next to 1 only; 2^-62.6 elsewhere): the result carries 0.5 ulp + 0.08.  This is the form
 * every Interstellar Euler step takes (its argument 1 + x^2 is >= 5 whenever x >= 2): 16 instructions with the
 * argument's construction (round 2: r^2 P3(r) + (r + t) and a bit-field insert for z, 18).  WHICH formula applies is
 * a function of the argument alone (its exponent), so cv_log stays one function with one value per argument on host
 * and device. */
/* the k >= 1 / k <= -2 formula: e = k + 1 as a double, r, row of the table */
CV_HD double cv_log_far(double kd, double r, double lch, double lcl) {
  const double LN2_HI = 6.93147180369123816490e-01, /* 0x3FE62E42FEE00000 */
      LN2_LO = 1.90821492927058770002e-10;          /* 0x3DEA39EF35793C76 */
  const double t = CV_FMA(kd, LN2_LO, lcl);
  const double w = CV_FMA(kd, LN2_HI, lch);
  /* -0.5 and 1.0 are inline constants of the VOP3 encoding: plain fmas; the other two addends come from scalar pairs */
  const double c1 = CV_FMA(r, CV_FMA(r, cv_fma_ks(r, cv_fma_ks(r, 0.2, -0.25), 3.33333333333333314830e-01), -0.5), 1.0);
  return w + CV_FMA(r, c1, t);
}
/* log of the normal positive double with bits ux (hx = high word), plus k0 * ln 2 */
CV_HD double cv_log_main(uint64_t ux, uint32_t hx, int k0, cv_log_tab_t T) {
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const int e = k0 + (int)(hx >> 20) - 0x3fe; /* k + 1 */
  const unsigned i = (hx >> 11) & 0x1ffu;
  const uint32_t gh = (hx & 0x000fffffu) | 0x3fe00000u;
  const double g = cv_from_bits(((uint64_t)gh << 32) | (ux & 0xffffffffULL)); /* mantissa in [1/2, 1) */
  const double invc2 = T[i][0], lch = T[i][1], lcl = T[i][2];
  const double r = CV_FMA(g, invc2, -1.0);
  const double kd = (double)e;
  if ((unsigned)e >= 2u) return cv_log_far(kd, r, lch, lcl); /* k >= 1 or k <= -2: |w| >= 0.69 */
  const double w = CV_FMA(kd, LN2_HI, lch);
  const double hi = w + r;
  const double lo = ((w - hi) + r) + CV_FMA(kd, LN2_LO, lcl);
  const double r2 = r * r;
  const double p = cv_fma_ks(
      r, cv_fma_ks(r, cv_fma_ks(r, cv_fma_ks(r, cv_fma_ks(r, 1.42857142857142849213e-01, -1.66666666666666657415e-01), 0.2), -0.25),
                   3.33333333333333314830e-01),
      -0.5);
  return hi + CV_FMA(r2, p, lo);
}

CV_HD double cv_log_t(double x, cv_log_tab_t T) {
  uint64_t ux = cv_bits(x);
  uint32_t hx = (uint32_t)(ux >> 32);
  int k = 0;
  if (hx < 0x00100000u || (hx >> 31)) {          /* x < 2^-1022, zero, or negative */
    if ((ux << 1) == 0) return -1.0 / (x * x);   /* log(+-0) = -inf */
    if (hx >> 31) return (x - x) / 0.0;          /* log(-#) = NaN */
    k -= 54;
    x *= 1.80143985094819840000e+16; /* 2^54: scale up subnormal */
    ux = cv_bits(x);
    hx = (uint32_t)(ux >> 32);
  } else if (hx >= 0x7ff00000u) {
    return x + x; /* inf or nan */
  }
  return cv_log_main(ux, hx, k, T);
}

/* the same value for a finite argument >= 2 (1 + x^2 with x >= 2: every Euler step outside |l| < a + pi m): only the
 * k >= 1 formula of cv_log_main, no test of k; exponent and mantissa by v_frexp_exp_i32_f64 / v_frexp_mant_f64 */
CV_HD double cv_log_ge2_t(double x, cv_log_tab_t T) {
  const uint32_t hx = cv_hi(x);
  const unsigned i = (hx >> 11) & 0x1ffu;
#if defined(__HIP_DEVICE_COMPILE__)
  const int e = __builtin_amdgcn_frexp_exp(x); /* k + 1 for a normal number */
  const double g = __builtin_amdgcn_frexp_mant(x);
#else
  const int e = (int)(hx >> 20) - 0x3fe;
  const double g = cv_from_bits((((uint64_t)((hx & 0x000fffffu) | 0x3fe00000u)) << 32) | (cv_bits(x) & 0xffffffffULL));
#endif
  const double r = CV_FMA(g, T[i][0], -1.0);
  return cv_log_far((double)e, r, T[i][1], T[i][2]);
}

CV_HD double cv_log(double x) { return cv_log_t(x, cv_log_table()); }

#endif /* CURVIS_CV_MATH_H */
