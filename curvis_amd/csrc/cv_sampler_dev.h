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
  unsigned eval_phases, evaluated; /* diagnostics: Euler chains the job waited for, points integrated (speculative ones included) */
};

/* ---- evaluation cache + speculation ------------------------------------------------------------------------------------
 * A refinement round costs one Euler chain (~2000 dependent steps) however few points it asks for, and a job has ~17 rounds.
 * Every point a later round can ask for is the midpoint of two samples adjacent at that time, i.e. a node of the dyadic tree
 * below an interval of the current table, computed by the same (lo + hi) / 2.0.  So a round that has to integrate anything also
 * integrates, on the lanes that would otherwise idle, the subtree below every interval it refines -- as deep as fits into ONE
 * chain of the workgroup -- and keeps the values in a per-job hash table in HBM keyed by the bit pattern of alpha; the next
 * rounds are then assembled from the table without integrating at all.  The sampler consumes exactly the values the sequential
 * algorithm would compute (compute_escape_angle is a pure function of alpha), and calls / steps are counted at consumption, so
 * the bookkeeping is the reference's.  (Same idea as the host-paced sampler's cache in efficient_host.h, which is its checker.) */
constexpr unsigned kSpecSlots = 8192;       /* hash slots per job (open addressing; speculation stops at half full) */
constexpr unsigned kSpecEvalCap = 2048;     /* room in a round's integration list: every pending point (<= kSamplerPendCap) always fits;
                                               speculative ones are added while it holds fewer than kSpecEvalSoft */
constexpr unsigned kSpecEvalSoft = 1024;    /* (checked by every lane before every insertion: overshoot < the lanes of a workgroup) */
constexpr unsigned long long kSpecEmpty = ~0ull; /* a NaN pattern: never the bits of an alpha */

struct SpecTable { /* views into the job's slice of the scratch */
  unsigned long long *key;
  double *e, *s;
  unsigned *steps;
  int *status;
};

#if defined(__HIP_DEVICE_COMPILE__)
/* the table is read and written by several waves of ONE workgroup between barriers; its keys are claimed with atomics, which
 * are performed in L2 -- every access goes there (agent scope), not through the CU's L1 */
#define CV_SPEC_CAS(p, expect, val) atomicCAS((p), (expect), (val))
#define CV_SPEC_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CV_SPEC_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
static inline unsigned long long cv_spec_cas_host(unsigned long long *p, unsigned long long expect, unsigned long long val) {
  const unsigned long long old = *p;
  if (old == expect) *p = val;
  return old;
}
#define CV_SPEC_CAS(p, expect, val) cv_spec_cas_host((p), (expect), (val))
#define CV_SPEC_LD(p) (*(p))
#define CV_SPEC_ST(p, v) (*(p) = (v))
#endif

CV_HD unsigned long long spec_key(double a) {
  unsigned long long k;
  __builtin_memcpy(&k, &a, sizeof k);
  return k;
}
CV_HD unsigned spec_hash(unsigned long long k) { return (unsigned)((k * 0x9E3779B97F4A7C15ull) >> 44) & (kSpecSlots - 1u); }
/* slot of key k, inserting it if absent (fresh = this call inserted it); concurrent callers get one winner per key */
CV_HD unsigned spec_claim(const SpecTable &T, unsigned long long k, bool &fresh) {
  unsigned i = spec_hash(k);
  for (;;) {
    const unsigned long long old = CV_SPEC_CAS(&T.key[i], kSpecEmpty, k);
    if (old == kSpecEmpty) {
      fresh = true;
      return i;
    }
    if (old == k) {
      fresh = false;
      return i;
    }
    i = (i + 1u) & (kSpecSlots - 1u);
  }
}
/* slot of key k or -1 (never called while keys are being claimed) */
CV_HD int spec_find(const SpecTable &T, unsigned long long k) {
  unsigned i = spec_hash(k);
  for (;;) {
    const unsigned long long cur = CV_SPEC_LD(&T.key[i]);
    if (cur == k) return (int)i;
    if (cur == kSpecEmpty) return -1;
    i = (i + 1u) & (kSpecSlots - 1u);
  }
}

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
  unsigned char flag[kSamplerCap];          /* plan: triple i asks for refinement */
  unsigned short vis[kSamplerCap], vis_out[kSamplerCap]; /* plan: the visits of the walk (sampler_walk) */
  unsigned cur, n, n_new, n_pend, n_vis, iteration, prev_amount, rounds;
  int started, finished, panicked, warned, overflow, go, nonfinite;
  unsigned long long calls, steps;
  /* speculation (see above): this round's pending points that the table does not hold yet, and what is integrated for them */
  unsigned short miss[kSamplerPendCap];
  double eval_a[kSpecEvalCap];
  unsigned short eval_slot[kSpecEvalCap];
  unsigned n_miss, n_eval, n_cached /* keys in the table */, eval_phases, evaluated;
};

CV_HD void sampler_reset(SamplerState &S) {
  S.cur = S.n = S.n_new = S.n_pend = S.n_vis = S.iteration = S.prev_amount = S.rounds = 0u;
  S.started = S.finished = S.panicked = S.warned = S.overflow = S.go = S.nonfinite = 0;
  S.calls = S.steps = 0ull;
  S.n_miss = S.n_eval = S.n_cached = S.eval_phases = S.evaluated = 0u;
}

/* ---- the phases of a round between plan and consume; every function is what ONE lane does for ONE item ---- */

/* phase 1, pending point t: taken from the table if it is there (returns true), else noted as missing */
#if defined(__HIP_DEVICE_COMPILE__)
#define CV_LDS_INC(p) atomicAdd((p), 1u)
#define CV_LDS_ADD64(p, v) atomicAdd((p), (unsigned long long)(v))
#else
#define CV_LDS_INC(p) ((*(p))++)
#define CV_LDS_ADD64(p, v) (*(p) += (unsigned long long)(v))
#endif
struct SamplerState;
CV_HD void sampler_store(SamplerState &S, unsigned t, double e, double s);

CV_HD void sampler_finish(SamplerState &S, unsigned max_iterations) {
  if (S.iteration == max_iterations) S.warned = 1; /* "Warning: maximum number of iterations ..." */
  S.finished = 1;
}

/* ---- planning a round (cv_sampler.h Sampler::plan), in four pieces so that only the WALK is sequential ----
 * begin (one lane): is there a round at all?  flag (any lane, one triple each): does triple i ask for refinement?  walk (one
 * lane): which triples are visited -- `i += 1` or `i += 2` depending on the triple just seen -- and where their points land in the
 * new table; place (any lane, one visit each): copy the points that stay, compute the midpoints that are asked for.
 * sampler_plan below is their serial composition (the x86 twin runs that; the kernel runs the pieces on all lanes). */
enum : int { PLAN_STOP = 0, PLAN_GRID = 1, PLAN_REFINE = 2 };

CV_HD int sampler_plan_begin(SamplerState &S, unsigned n0, unsigned max_iterations) {
  S.n_pend = 0u;
  if (S.finished) return PLAN_STOP;
  if (!S.started) { /* compute_uniform_range */
    if (n0 > kSamplerCap || n0 > kSamplerPendCap) {
      S.overflow = S.finished = 1;
      return PLAN_STOP;
    }
    S.n_pend = S.n_new = n0;
    return PLAN_GRID;
  }
  if (!(S.iteration < max_iterations)) {
    sampler_finish(S, max_iterations);
    return PLAN_STOP;
  }
  S.prev_amount = S.n; /* (the table is clean: every assembly ends with clean_bipoints) */
  if (S.n < 3u) {
    S.panicked = S.finished = 1; /* "bipoints list has length < 3" */
    return PLAN_STOP;
  }
  return PLAN_REFINE;
}
/* grid point i of the first round */
CV_HD void sampler_grid_point(SamplerState &S, unsigned i, unsigned n0, double a_min, double a_max) {
  const double step = (a_max - a_min) / (double)(n0 - 1u);
  S.pend_a[i] = a_min + (double)i * step;
  S.pend_out[i] = (unsigned short)i;
}
/* triple i (0 <= i < n - 2) */
CV_HD void sampler_flag(SamplerState &S, unsigned i, double thr1, double thr2) {
  const double *A = S.a[S.cur], *E = S.e[S.cur], *Z = S.s[S.cur];
  S.flag[i] = sampler_refine(A[i], E[i], Z[i], A[i + 1u], E[i + 1u], Z[i + 1u], A[i + 2u], E[i + 2u], Z[i + 2u], thr1, thr2) ? 1 : 0;
}
/* the walk: visit v starts at sample vis[v] & 0x7fff (bit 15: refined) and puts it at vis_out[v] of the new table; a refined
 * visit takes four places (sample, midpoint, next sample, midpoint) and two pending points, the others one place */
CV_HD bool sampler_walk(SamplerState &S) {
  const unsigned n = S.n;
  unsigned i = 0u, out = 0u, k = 0u, v = 0u;
  while (i < n - 2u) {
    if (out + 5u > kSamplerCap || k + 2u > kSamplerPendCap) {
      S.overflow = S.finished = 1;
      return false;
    }
    const unsigned f = S.flag[i];
    S.vis[v] = (unsigned short)(i | (f << 15));
    S.vis_out[v] = (unsigned short)out;
    ++v;
    if (f) {
      out += 4u, k += 2u, i += 2u;
    } else {
      out += 1u, i += 1u;
    }
  }
  S.n_vis = v;
  S.n_new = out;
  S.n_pend = k;
  return true; /* possibly with nothing to evaluate: the round is assembled all the same */
}
/* visit v */
CV_HD void sampler_place(SamplerState &S, unsigned v) {
  const unsigned dst = S.cur ^ 1u;
  const double *A = S.a[S.cur], *E = S.e[S.cur], *Z = S.s[S.cur];
  double *NA = S.a[dst], *NE = S.e[dst], *NZ = S.s[dst];
  const unsigned i = S.vis[v] & 0x7fffu, out = S.vis_out[v];
  NA[out] = A[i], NE[out] = E[i], NZ[out] = Z[i];
  if (S.vis[v] >> 15) {
    const unsigned k = 2u * ((out - v) / 3u); /* out = (plain visits before) + 4 (refined visits before), v = plain + refined */
    S.pend_a[k] = (A[i] + A[i + 1u]) / 2.0;
    S.pend_out[k] = (unsigned short)(out + 1u);
    NA[out + 2u] = A[i + 1u], NE[out + 2u] = E[i + 1u], NZ[out + 2u] = Z[i + 1u];
    S.pend_a[k + 1u] = (A[i + 1u] + A[i + 2u]) / 2.0;
    S.pend_out[k + 1u] = (unsigned short)(out + 3u);
    /* the sample on the far side of the second midpoint: the next visit writes the same values here, unless the walk ended
     * (then n_new stops short of it) -- sampler_speculate reads the neighbours of every pending point from this table */
    NA[out + 4u] = A[i + 2u], NE[out + 4u] = E[i + 2u], NZ[out + 4u] = Z[i + 2u];
  }
}
/* all of it on one lane.  Returns false when the sampler has finished, panicked or run out of room. */
CV_HD bool sampler_plan(SamplerState &S, unsigned n0, unsigned max_iterations, double a_min, double a_max, double thr1, double thr2) {
  const int mode = sampler_plan_begin(S, n0, max_iterations);
  if (mode == PLAN_STOP) return false;
  if (mode == PLAN_GRID) {
    for (unsigned i = 0; i < n0; ++i) sampler_grid_point(S, i, n0, a_min, a_max);
    return true;
  }
  for (unsigned i = 0; i + 2u < S.n; ++i) sampler_flag(S, i, thr1, thr2);
  if (!sampler_walk(S)) return false;
  for (unsigned v = 0; v < S.n_vis; ++v) sampler_place(S, v);
  return true;
}

/* ANY lane: the value of pending point t goes straight to its place in the table being assembled */
CV_HD void sampler_store(SamplerState &S, unsigned t, double e, double s) {
  const unsigned dst = S.cur ^ 1u, o = S.pend_out[t];
  S.a[dst][o] = S.pend_a[t];
  S.e[dst][o] = e;
  S.s[dst][o] = s;
  if (!(sampler_finite(e) && sampler_finite(s))) S.nonfinite = 1; /* NotEscaped -> (NaN, NaN): clean_bipoints has work to do */
}

/* pending point t from the table into the table being assembled; false = not there.  panic: a consumed point whose tangent
 * rotation is undefined (ESC_PANIC = -2) -- the reference's panic counts only for points the sampler really uses */
CV_HD bool sampler_take(SamplerState &S, const SpecTable &T, unsigned t, int &panic) {
  const int slot = spec_find(T, spec_key(S.pend_a[t]));
  if (slot < 0) return false;
  sampler_store(S, t, CV_SPEC_LD(&T.e[slot]), CV_SPEC_LD(&T.s[slot]));
  CV_LDS_ADD64(&S.steps, CV_SPEC_LD(&T.steps[slot]));
  if (CV_SPEC_LD(&T.status[slot]) == -2) panic = 1;
  return true;
}
/* queue alpha for integration unless the table has it (or has it queued) already; false = no room left */
CV_HD bool sampler_want(SamplerState &S, const SpecTable &T, double alpha, bool must) {
  /* soft limits, read without synchronisation by every lane before every insertion: they overshoot by less than one insertion
   * per lane of the workgroup, which the hard sizes leave room for */
  if (!must && (S.n_cached >= kSpecSlots / 2u || S.n_eval >= kSpecEvalSoft)) return false;
  if (S.n_cached >= kSpecSlots - 1024u || S.n_eval >= kSpecEvalCap - 512u) { /* (the probe loops need empty slots) */
    if (must) S.overflow = 1;
    return false;
  }
  bool fresh;
  const unsigned slot = spec_claim(T, spec_key(alpha), fresh);
  if (!fresh) return true; /* in the table, or queued by another lane this round */
  CV_LDS_INC(&S.n_cached);
  const unsigned k = CV_LDS_INC(&S.n_eval);
  S.eval_a[k] = alpha;
  S.eval_slot[k] = (unsigned short)slot;
  return true;
}
/* nodes of the dyadic tree below (lo, hi), `depth` levels of them (depth 1 = the midpoint alone), level by level; node i of level
 * j is reached from the root by the bits of i -- every midpoint is the (a + b) / 2.0 the sampler itself would compute for two
 * adjacent samples a, b.  (No stack: indexed local arrays would live in scratch memory.) */
CV_HD void sampler_subtree(SamplerState &S, const SpecTable &T, double lo, double hi, unsigned depth) {
  if (depth > 7u) depth = 7u;
  for (unsigned level = 0u; level < depth; ++level)
    for (unsigned i = 0u; i < (1u << level); ++i) {
      double a = lo, b = hi, mid = (a + b) / 2.0;
      bool ok = mid > a && mid < b; /* false: interval exhausted in double precision */
      for (unsigned bit = level; ok && bit-- > 0u;) {
        if ((i >> bit) & 1u)
          a = mid;
        else
          b = mid;
        mid = (a + b) / 2.0;
        ok = mid > a && mid < b;
      }
      if (ok && !sampler_want(S, T, mid, false)) return;
    }
}
/* levels of dyadic subtree (1 = the midpoint alone) that fit `budget` integrations when `n` intervals get one each:
 * the largest d with n (2^d - 1) <= budget; 0 when not even the midpoints fit */
CV_HD unsigned sampler_spec_depth(unsigned n, unsigned budget) {
  unsigned d = 0u;
  while (d < 7u && (unsigned long long)n * ((2ull << d) - 1ull) <= budget) ++d;
  return d;
}
/* phase 2, missing pending point number m of the round (index into S.miss): itself, and the subtree below its interval */
CV_HD void sampler_speculate(SamplerState &S, const SpecTable &T, unsigned m, unsigned depth, unsigned grid_depth) {
  const unsigned t = S.miss[m];
  if (!S.started) { /* the uniform grid: the interval to the next grid point */
    if (grid_depth && t + 1u < S.n_pend) sampler_subtree(S, T, S.pend_a[t], S.pend_a[t + 1u], grid_depth);
    return;
  }
  const unsigned dst = S.cur ^ 1u, o = S.pend_out[t];
  /* the neighbours in the table being assembled are the samples this point is the midpoint of (sampler_plan put them there);
   * the point itself is the root of the subtree: depth - 1 levels below it, on either side */
  if (depth > 1u) {
    sampler_subtree(S, T, S.a[dst][o - 1u], S.pend_a[t], depth - 1u);
    sampler_subtree(S, T, S.pend_a[t], S.a[dst][o + 1u], depth - 1u);
  }
}

/* ONE lane, after every pending point has been stored (and S.steps += their steps): Sampler::consume */
CV_HD void sampler_consume(SamplerState &S, unsigned max_iterations) {
  S.calls += S.n_pend;
  const unsigned dst = S.cur ^ 1u;
  double *NA = S.a[dst], *NE = S.e[dst], *NZ = S.s[dst];
  unsigned k = S.n_new;
  if (S.nonfinite) { /* clean_bipoints: drop every point with a non-finite member (NotEscaped -> (NaN, NaN)); the points that were
                        kept are finite (they survived an earlier cleaning), so only a round that stored such a value needs it */
    k = 0u;
    for (unsigned i = 0; i < S.n_new; ++i)
      if (sampler_finite(NA[i]) && sampler_finite(NE[i]) && sampler_finite(NZ[i])) {
        if (k != i) NA[k] = NA[i], NE[k] = NE[i], NZ[k] = NZ[i];
        ++k;
      }
    S.nonfinite = 0;
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
