/* cv_sampler.h -- host-side control flow of the reference's adaptive 1-D sampler
 * (doubly_sample_function, src/sampling.rs:46-124; evaluate_denser_bipoints :144-195;
 * evaluate_convergence_scores :198-245; compute_uniform_range :129-140; clean_bipoints :21-31),
 * restructured so that the "expensive function" (compute_escape_angle, the closure at
 * src/systems.rs:473-485) is evaluated in BATCHES on the GPU: within one refinement round the set of new
 * sample points depends only on values of previous rounds, so a round is planned first, evaluated with
 * one kernel launch, then assembled -- the resulting table is identical to the sequential algorithm's.
 * Several frames (independent samplers) advance in lock step and share each launch.
 */
#ifndef CURVIS_CV_SAMPLER_H
#define CURVIS_CV_SAMPLER_H
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "cv_sampler_dev.h"

namespace cvs {

struct BiPoint {
  double a, e, s; /* alpha, escape angle, escape space (+1 / -1) */
};

inline bool finite3(const BiPoint &b) { return std::isfinite(b.a) && std::isfinite(b.e) && std::isfinite(b.s); }
inline void clean(std::vector<BiPoint> &v) { /* clean_bipoints */
  size_t k = 0;
  for (size_t i = 0; i < v.size(); ++i)
    if (finite3(v[i])) v[k++] = v[i];
  v.resize(k);
}

struct Sampler {
  /* inputs */
  double a_min = 0, a_max = 0, thr1 = 0, thr2 = 0;
  size_t n0 = 0, max_iterations = 0;
  /* state */
  std::vector<BiPoint> pts;
  size_t iteration = 0, prev_amount = 0;
  bool started = false, finished = false, panicked = false, warned = false;
  uint64_t calls = 0, steps = 0;
  uint32_t rounds = 0;
  /* plan of the current round */
  std::vector<double> pending;  /* alphas to evaluate, in the reference's evaluation order */
  std::vector<double> pend_lo, pend_hi; /* the adjacent samples each pending alpha is the midpoint of
                                           (NaN for the initial uniform grid) -- used for speculation */
  std::vector<uint8_t> refined; /* per visited triple start: 1 = refined (consumes two evaluations) */
  std::vector<size_t> visit;    /* triple start indices in visiting order */

  /* plan the next batch; returns false when the sampler has finished (or panicked) */
  bool plan() {
    pending.clear();
    pend_lo.clear();
    pend_hi.clear();
    if (finished) return false;
    if (!started) { /* compute_uniform_range */
      const double step = (a_max - a_min) / (double)(n0 - 1);
      for (size_t i = 0; i < n0; ++i) {
        pending.push_back(a_min + (double)i * step);
        pend_lo.push_back(std::nan(""));
        pend_hi.push_back(std::nan(""));
      }
      return true;
    }
    if (!(iteration < max_iterations)) {
      finish();
      return false;
    }
    prev_amount = pts.size();
    clean(pts); /* evaluate_denser_bipoints cleans its input */
    if (pts.size() < 3) {
      panicked = true; /* "bipoints list has length < 3" */
      finished = true;
      return false;
    }
    refined.clear();
    visit.clear();
    const size_t n = pts.size();
    size_t i = 0;
    while (i < n - 2) {
      const BiPoint &b1 = pts[i], &b2 = pts[(i + 1) % n], &b3 = pts[(i + 2) % n];
      visit.push_back(i);
      /* the two shoelace areas: cvk::sampler_refine, shared with the device-resident sampler (cv_sampler_dev.h) */
      if (!cvk::sampler_refine(b1.a, b1.e, b1.s, b2.a, b2.e, b2.s, b3.a, b3.e, b3.s, thr1, thr2)) {
        refined.push_back(0);
        i += 1;
      } else {
        refined.push_back(1);
        pending.push_back((b1.a + b2.a) / 2.0);
        pend_lo.push_back(b1.a);
        pend_hi.push_back(b2.a);
        pending.push_back((b2.a + b3.a) / 2.0);
        pend_lo.push_back(b2.a);
        pend_hi.push_back(b3.a);
        i += 2;
      }
    }
    return true; /* possibly with an empty batch: the round still has to be assembled */
  }

  /* consume the results of the planned batch (e/s/steps arrays parallel to `pending`) */
  void consume(const double *e, const double *s, const uint32_t *st) {
    calls += pending.size();
    for (size_t k = 0; k < pending.size(); ++k) steps += st[k];
    if (!started) {
      pts.resize(pending.size());
      for (size_t k = 0; k < pending.size(); ++k) pts[k] = BiPoint{pending[k], e[k], s[k]};
      clean(pts);
      started = true;
      return;
    }
    std::vector<BiPoint> nb;
    nb.reserve(2 * pts.size() + 4);
    const size_t n = pts.size();
    size_t k = 0;
    for (size_t v = 0; v < visit.size(); ++v) {
      const size_t i = visit[v];
      if (!refined[v]) {
        nb.push_back(pts[i]);
      } else {
        nb.push_back(pts[i]);
        nb.push_back(BiPoint{pending[k], e[k], s[k]});
        nb.push_back(pts[(i + 1) % n]);
        nb.push_back(BiPoint{pending[k + 1], e[k + 1], s[k + 1]});
        k += 2;
      }
    }
    clean(nb);
    pts.swap(nb);
    rounds++;
    if (pts.size() < prev_amount || pts.size() == prev_amount) {
      finish();
      return;
    }
    iteration += 1;
  }

  void finish() {
    if (iteration == max_iterations) warned = true; /* "Warning: maximum number of iterations ..." */
    finished = true;
  }
};

/* interp 1.0.3 tables for interp_slice(x, y, .): m_i = dy/dx, c_i = y_i - x_i*m_i (n-1 entries);
 * for n == 1 the single "intercept" is y[0]; n == 0 leaves the tables empty (result 0). */
inline void interp_tables(const std::vector<double> &x, const std::vector<double> &y, std::vector<double> &m,
                          std::vector<double> &c) {
  const size_t n = x.size();
  m.clear();
  c.clear();
  if (n == 1) {
    m.push_back(0.0);
    c.push_back(y[0]);
    return;
  }
  for (size_t i = 0; i + 1 < n; ++i) {
    const double dx = x[i + 1] - x[i], dy = y[i + 1] - y[i];
    const double mi = dy / dx;
    m.push_back(mi);
    c.push_back(y[i] - x[i] * mi);
  }
}

}  // namespace cvs
#endif
