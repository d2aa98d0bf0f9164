/* curvis_quad.c -- TEST INFRASTRUCTURE: error of an elementary-function result against the exact value,
 * computed in IEEE binary128 with gcc's libquadmath (sinq, cosq, atanq, acosq, logq, atan2q: independent of both
 * glibc's double functions and curvis_amd/csrc/cv_math.h).  Used by tests/test_cv_math.py and
 * tools/gpu_libm_parity.py to measure, on the arguments the Euler loop really produces, how far cv_math.h and
 * glibc are from the truth -- the bit-exact GPU-vs-oracle tests share cv_math.h between both sides and cannot
 * see an error in it.  Kept in its own small library so that the oracle proper has no dependency on libquadmath.
 */
#include <quadmath.h>
#include <stddef.h>
#include <stdint.h>

/* op: 0 sin, 1 cos, 2 atan, 3 acos, 4 log, 5 atan2(a, b).  err_ulp[i] = |got[i] - f(a[i])| in units of the last
 * place of a double at the exact value (2^(e-52) for |f| in [2^e, 2^(e+1)), the subnormal spacing below 2^-1022);
 * 0.5 is the best any double can do.  Non-finite exact values (or got) give 0 when both agree in class, else inf. */
void cvq_ulp_errors(int op, const double *a, const double *b, const double *got, double *err_ulp, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const __float128 x = a[i], y = b ? (__float128)b[i] : 0;
    __float128 f;
    switch (op) {
      case 0: f = sinq(x); break;
      case 1: f = cosq(x); break;
      case 2: f = atanq(x); break;
      case 3: f = acosq(x); break;
      case 4: f = logq(x); break;
      default: f = atan2q(x, y); break;
    }
    const double g = got[i];
    if (isnanq(f) || isinfq(f) || g != g || g - g != 0.0) {
      const int same = (isnanq(f) && g != g) || (isinfq(f) && g - g != 0.0 && g == g && ((f > 0) == (g > 0)));
      err_ulp[i] = same ? 0.0 : __builtin_inf();
      continue;
    }
    int e;
    (void)frexpq(fabsq(f), &e); /* |f| = m 2^e, m in [0.5, 1) */
    int ulp_exp = e - 53;        /* ulp = 2^(e-1-52) */
    if (ulp_exp < -1074) ulp_exp = -1074;
    if (f == 0) ulp_exp = -1074;
    err_ulp[i] = (double)(fabsq((__float128)g - f) * scalbnq(1, -ulp_exp));
  }
}
