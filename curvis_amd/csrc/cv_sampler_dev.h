/* cv_sampler_dev.h -- the reference's adaptive sampler (doubly_sample_function, src/sampling.rs:46-124; evaluate_denser_bipoints
 * :144-195; clean_bipoints :21-31) as the job of ONE WORKGROUP: the table lives in LDS, a round is planned by one lane, its new
 * points are integrated by all lanes, assembled, and the loop goes round again -- no host round trip per refinement round
 * (kernels_efficient.h sampler_kernel).  The control flow is the host sampler's (cv_sampler.h, the checker and the fall-back),
 * restated over fixed arrays; what decides a refinement -- the two shoelace areas -- is ONE function shared by both.
 * __host__ __device__: the x86 twin of the tests runs the same plan / consume code around a serial evaluation
 * (tests/host_twin/twin.cpp twin_sampler_dev) and must produce the host sampler's table bit for bit. */
#ifndef CURVIS_CV_SAMPLER_DEV_H
#define CURVIS_CV_SAMPLER_DEV_H

#include "cv_math.h" /* CV_HD */

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace cvk {

constexpr unsigned kSamplerCap = 1536;     /* samples a table may hold (defaults end at ~500-700; the fly-through's throat poses at ~900) */
constexpr unsigned kSamplerPendCap = 1024; /* evaluations one round may ask for */

enum : int { SAMPLER_OK = 0, SAMPLER_PANIC = 1, SAMPLER_OVERFLOW = 2 };

struct SamplerResult { /* one per job, written by the kernel */
  unsigned n, rounds;
  unsigned long long calls, steps;
  int warned, status;
};

/* src/sampling.rs:169-177: refine the triple when the area of (alpha, escape angle) exceeds thr1 or the area of (alpha, escape
 * space) exceeds thr2 -- the shoelace formula in the reference's order of operations */
CV_HD bool sampler_refine(double a1, double e1, double s1, double a2, double e2, double s2, double a3, double e3, double s3,
                          double thr1, double thr2) {
  const double area1 = CV_FABS((a1 * e2 + a2 * e3 + a3 * e1) - (e1 * a2 + e2 * a3 + e3 * a1));
  const double area2 = CV_FABS((a1 * s2 + a2 * s3 + a3 * s1) - (s1 * a2 + s2 * a3 + s3 * a1));
  return area1 > thr1 || area2 > thr2;
}
CV_HD bool sampler_finite(double v) { return (v - v) == 0.0; } /* false for NaN and +-inf */

struct SamplerState {
  double a[2][kSamplerCap], e[2][kSamplerCap], s[2][kSamplerCap]; /* the table and the one being assembled (ping-pong) */
  double pend_a[kSamplerPendCap];         /* alphas of the round, in the reference's evaluation order */
  unsigned short pend_out[kSamplerPendCap]; /* where each lands in the table being assembled */
  unsigned cur, n, n_new, n_pend, iteration, prev_amount, rounds;
  int started, finished, panicked, warned, overflow, go;
  unsigned long long calls, steps;
};

CV_HD void sampler_reset(SamplerState &S) {
  S.cur = S.n = S.n_new = S.n_pend = S.iteration = S.prev_amount = S.rounds = 0u;
  S.started = S.finished = S.panicked = S.warned = S.overflow = S.go = 0;
  S.calls = S.steps = 0ull;
}
CV_HD void sampler_finish(SamplerState &S, unsigned max_iterations) {
  if (S.iteration == max_iterations) S.warned = 1; /* "Warning: maximum number of iterations ..." */
  S.finished = 1;
}

/* ONE lane: plan the next round (cv_sampler.h Sampler::plan) and put the points that stay where they will sit in the new table.
 * Returns false when the sampler has finished, panicked or run out of room. */
CV_HD bool sampler_plan(SamplerState &S, unsigned n0, unsigned max_iterations, double a_min, double a_max, double thr1, double thr2) {
  S.n_pend = 0u;
  if (S.finished) return false;
  const unsigned dst = S.cur ^ 1u;
  if (!S.started) { /* compute_uniform_range */
    if (n0 > kSamplerCap || n0 > kSamplerPendCap) {
      S.overflow = S.finished = 1;
      return false;
    }
    const double step = (a_max - a_min) / (double)(n0 - 1u);
    for (unsigned i = 0; i < n0; ++i) {
      S.pend_a[i] = a_min + (double)i * step;
      S.pend_out[i] = (unsigned short)i;
    }
    S.n_pend = S.n_new = n0;
    return true;
  }
  if (!(S.iteration < max_iterations)) {
    sampler_finish(S, max_iterations);
    return false;
  }
  S.prev_amount = S.n; /* (the table is clean: every assembly ends with clean_bipoints) */
  if (S.n < 3u) {
    S.panicked = S.finished = 1; /* "bipoints list has length < 3" */
    return false;
  }
  const double *A = S.a[S.cur], *E = S.e[S.cur], *Z = S.s[S.cur];
  double *NA = S.a[dst], *NE = S.e[dst], *NZ = S.s[dst];
  const unsigned n = S.n;
  unsigned i = 0u, out = 0u, k = 0u;
  while (i < n - 2u) {
    if (out + 4u > kSamplerCap || k + 2u > kSamplerPendCap) {
      S.overflow = S.finished = 1;
      return false;
    }
    NA[out] = A[i], NE[out] = E[i], NZ[out] = Z[i];
    if (!sampler_refine(A[i], E[i], Z[i], A[i + 1u], E[i + 1u], Z[i + 1u], A[i + 2u], E[i + 2u], Z[i + 2u], thr1, thr2)) {
      out += 1u;
      i += 1u;
    } else {
      S.pend_a[k] = (A[i] + A[i + 1u]) / 2.0;
      S.pend_out[k] = (unsigned short)(out + 1u);
      NA[out + 2u] = A[i + 1u], NE[out + 2u] = E[i + 1u], NZ[out + 2u] = Z[i + 1u];
      S.pend_a[k + 1u] = (A[i + 1u] + A[i + 2u]) / 2.0;
      S.pend_out[k + 1u] = (unsigned short)(out + 3u);
      out += 4u;
      k += 2u;
      i += 2u;
    }
  }
  S.n_new = out;
  S.n_pend = k;
  return true; /* possibly with nothing to evaluate: the round is assembled all the same */
}

/* ANY lane: the value of pending point t goes straight to its place in the table being assembled */
CV_HD void sampler_store(SamplerState &S, unsigned t, double e, double s) {
  const unsigned dst = S.cur ^ 1u, o = S.pend_out[t];
  S.a[dst][o] = S.pend_a[t];
  S.e[dst][o] = e;
  S.s[dst][o] = s;
}

/* ONE lane, after every pending point has been stored (and S.steps += their steps): Sampler::consume */
CV_HD void sampler_consume(SamplerState &S, unsigned max_iterations) {
  S.calls += S.n_pend;
  const unsigned dst = S.cur ^ 1u;
  double *NA = S.a[dst], *NE = S.e[dst], *NZ = S.s[dst];
  unsigned k = 0u; /* clean_bipoints: drop every point with a non-finite member (NotEscaped -> (NaN, NaN)) */
  for (unsigned i = 0; i < S.n_new; ++i)
    if (sampler_finite(NA[i]) && sampler_finite(NE[i]) && sampler_finite(NZ[i])) {
      if (k != i) NA[k] = NA[i], NE[k] = NE[i], NZ[k] = NZ[i];
      ++k;
    }
  S.n = k;
  S.cur = dst;
  if (!S.started) {
    S.started = 1;
    return;
  }
  S.rounds++;
  if (S.n <= S.prev_amount) {
    sampler_finish(S, max_iterations);
    return;
  }
  S.iteration += 1u;
}

}  // namespace cvk
#endif
