// Host twin of the device per-ray functions (curvis_amd/csrc/cv_device.h, cv_math.h) compiled for
// x86-64.  TEST VEHICLE ONLY: lets the CPU test-suite (no GPU in the build container) check that the
// kernel source is bit-identical to the oracle's CVO_CV flavour.  Nothing in curvis_amd/ loads this.
#include <cstdint>
#include <cstring>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../../curvis_amd/csrc/cv_device.h"
#include "../../curvis_amd/csrc/cv_efficient.h"
#include "../../curvis_amd/csrc/cv_frame_host.h"
#include "../../curvis_amd/csrc/cv_sampler.h"
#include "../../include/curvis_hip.h"

template <int KIND>
static void render_kind(const cvk::MetricParams &M, const cvk::CameraParams &C, const cvk::SkyParams sky[2],
                        unsigned W, unsigned H, unsigned max_iter, double R, double delta, uint8_t *rgb,
                        curvis_ray_debug *dbg, int fast) {
  const bool metric_ok = cvk::metric_fast_ok(KIND, M, R);
  for (unsigned py = 0; py < H; ++py)
    for (unsigned px = 0; px < W; ++px) {
      cvk::Ray q;
      cvk::ray_init<KIND>(M, C, px, py, q);
      unsigned steps = 0;
      int code = cvk::CODE_NONE;
      const bool lane_ok = fast && metric_ok && cvk::ray_fast_ok(q);
      while (steps < max_iter) {
        if (fast)
          cvk::ray_step_fast<KIND, true>(M, q, delta, lane_ok);
        else
          cvk::ray_step<KIND, true>(M, q, delta);
        ++steps;
        if (q.l > R) { code = cvk::CODE_POS; break; }
        else if (q.l < -R) { code = cvk::CODE_NEG; break; }
      }
      unsigned texel = 0xFF000000u, tx = 0, ty = 0;
      if (code != cvk::CODE_NONE) {
        double d0, d1, d2;
        cvk::ray_direction<KIND>(M, q, d0, d1, d2);
        const cvk::SkyParams &S = sky[code == cvk::CODE_POS ? 0 : 1];
        cvk::sky_indices(S, d0, d1, d2, tx, ty);
        unsigned cx = tx >= S.w ? S.w - 1 : tx, cy = ty >= S.h ? S.h - 1 : ty;
        texel = S.texels[(size_t)cy * S.w + cx];
      }
      size_t o = (size_t)py * W + px;
      rgb[o * 3 + 0] = texel & 0xFF;
      rgb[o * 3 + 1] = (texel >> 8) & 0xFF;
      rgb[o * 3 + 2] = (texel >> 16) & 0xFF;
      if (dbg) {
        curvis_ray_debug &d = dbg[o];
        d.x[0] = 0.0; d.x[1] = q.l; d.x[2] = q.th; d.x[3] = q.ph;
        d.p[0] = 1.0; d.p[1] = q.p1; d.p[2] = q.p2; d.p[3] = steps ? q.p3 + 0.0 : q.p3;
        d.steps = steps; d.code = code; d.tx = tx; d.ty = ty;
      }
    }
}


extern "C" {

double twin_math(int op, double x, double y) {
  switch (op) {
    case 0: return cv_sin(x);
    case 1: return cv_cos(x);
    case 2: return cv_atan(x);
    case 3: return cv_acos(x);
    case 4: return cv_log(x);
    case 5: return cv_atan2(x, y);
    case 6: return x / y;
    case 7: return CV_SQRT(x);
    default: return CV_FMA(x, y, x);
  }
}
// the fast step's primitives on three inputs: the same ops as curvis_selftest_math3 (include/curvis_hip.h).  On x86
// the hardware seeds are exact quotients (cv_device.h rcp_seed / rsq_seed), so ops 1, 2, 4 differ from the device in the
// seed; ops 0, 3, 6 are pure fma sequences and must agree bit for bit.
void twin_math3_array(int op, const double *a, const double *b, const double *c, double *out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const double x = a[i], y = b ? b[i] : 0.0, z = c ? c[i] : 0.0;
    double r, t;
    switch (op) {
      case 0: r = cvk::div_with_recip(x, y, z); break;
      case 1: cvk::sqrt_and_rsqrt(x, r, t); break;
      case 2: cvk::sqrt_and_rsqrt(x, t, r); break;
      case 3: r = CV_FMA(CV_FMA(-y, y, x), 0.5 * z, y); break;
      case 4: r = cvk::recip_refined(x); break;
      case 5: r = cv_div_nr(x, y); break;
      default: r = cvk::recip_newton(x, y); break;
    }
    out[i] = r;
  }
}
void twin_math_array(int op, const double *a, const double *b, double *out, size_t n) {
  for (size_t i = 0; i < n; ++i) out[i] = twin_math(op, a[i], b ? b[i] : 0.0);
}

void twin_render(const curvis_metric *m, const curvis_camera *c, const uint8_t *sky_pos, unsigned wp, unsigned hp,
                 const uint8_t *sky_neg, unsigned wn, unsigned hn, unsigned max_iter, double R, double delta,
                 uint8_t *rgb, curvis_ray_debug *dbg, int fast) {
  cvk::MetricParams M;
  M.rho = m->rho; M.rho2 = m->rho * m->rho; M.m = m->m; M.a = m->a; M.pim = CV_PI * m->m; M.inv_pim = 1.0 / M.pim; M.two_o_pi = 2.0 / CV_PI; M.T = cv_sc_table(); M.LT = cv_log_table(); M.AT = cv_atan_table();
  cvk::CameraParams C;
  for (int i = 0; i < 4; ++i) C.pos[i] = c->pos[i];
  for (int i = 0; i < 9; ++i) C.rot[i] = c->rot[i];
  C.focal = c->focal; C.sensor_w = c->sensor_w; C.sensor_h = c->sensor_h;
  C.res_x = (double)c->res_x; C.res_y = (double)c->res_y;
  cvk::SkyParams sky[2];
  sky[0].texels = (const unsigned *)sky_pos; sky[0].w = wp; sky[0].h = hp;
  sky[1].texels = (const unsigned *)sky_neg; sky[1].w = wn; sky[1].h = hn;
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 9; ++i) sky[s].inv_rot[i] = (i % 4 == 0) ? 1.0 : 0.0;
  switch (m->kind) {
    case 0: render_kind<0>(M, C, sky, c->res_x, c->res_y, max_iter, R, delta, rgb, dbg, fast); break;
    case 1: render_kind<1>(M, C, sky, c->res_x, c->res_y, max_iter, R, delta, rgb, dbg, fast); break;
    default: render_kind<2>(M, C, sky, c->res_x, c->res_y, max_iter, R, delta, rgb, dbg, fast); break;
  }
}
}

// ---- efficient renderer on the host: cvs::Sampler + cv_efficient.h (the code the GPU path runs, on x86)
template <int KIND>
static int escape_angle_host(const cvk::MetricParams &M, double l, double alpha, double delta, unsigned max_iter,
                             double R, int fast, double &angle, double &space, uint32_t &steps) {
  double sa, ca;
  cv_sincos(alpha, &sa, &ca);
  const double pos[4] = {0.0, l, CV_PI / 2.0, 0.0};
  cvk::Ray q;
  cvk::ray_init_dir<KIND>(M, pos, ca, 0.0, sa, q);
  const bool lane_ok = fast && cvk::metric_fast_ok(KIND, M, R) && cvk::ray_fast_ok(q);
  steps = 0;
  int code = cvk::CODE_NONE;
  while (steps < max_iter) {
    if (fast) cvk::ray_step_fast<KIND, true, false, true>(M, q, delta, lane_ok); /* EQ: the sampling kernel's equatorial form */
    else cvk::ray_step<KIND, true>(M, q, delta);
    ++steps;
    if (q.l > R) { code = cvk::CODE_POS; break; }
    else if (q.l < -R) { code = cvk::CODE_NEG; break; }
  }
  angle = space = __builtin_nan("");
  if (code == cvk::CODE_NONE) return code;
  if (!cvk::escape_angle_of<KIND>(M, q, angle)) { angle = __builtin_nan(""); return cvk::ESC_PANIC; }
  space = (code == cvk::CODE_POS) ? 1.0 : -1.0;
  return code;
}

static int escape_angle_any(int kind, const cvk::MetricParams &M, double l, double alpha, double delta,
                            unsigned max_iter, double R, int fast, double &angle, double &space, uint32_t &steps) {
  switch (kind) {
    case 0: return escape_angle_host<0>(M, l, alpha, delta, max_iter, R, fast, angle, space, steps);
    case 1: return escape_angle_host<1>(M, l, alpha, delta, max_iter, R, fast, angle, space, steps);
    default: return escape_angle_host<2>(M, l, alpha, delta, max_iter, R, fast, angle, space, steps);
  }
}

extern "C" int twin_render_efficient(const curvis_metric *m, const curvis_camera *c, const uint8_t *sky_pos, unsigned wp,
                                     unsigned hp, const uint8_t *sky_neg, unsigned wn, unsigned hn, unsigned max_iter,
                                     double R, double delta, unsigned alpha_nums, unsigned max_it_sampling, double thr1,
                                     double thr2, uint8_t *rgb, double *sa, double *se, double *ss, size_t cap,
                                     size_t *n_out, uint64_t *calls, uint64_t *steps_out, int fast, int dev_sampler) {
  cvk::MetricParams M;
  M.rho = m->rho; M.rho2 = m->rho * m->rho; M.m = m->m; M.a = m->a; M.pim = CV_PI * m->m; M.inv_pim = 1.0 / M.pim; M.two_o_pi = 2.0 / CV_PI; M.T = cv_sc_table(); M.LT = cv_log_table(); M.AT = cv_atan_table();
  cvk::CameraParams C;
  for (int i = 0; i < 4; ++i) C.pos[i] = c->pos[i];
  for (int i = 0; i < 9; ++i) C.rot[i] = c->rot[i];
  C.focal = c->focal; C.sensor_w = c->sensor_w; C.sensor_h = c->sensor_h;
  C.res_x = (double)c->res_x; C.res_y = (double)c->res_y;
  cvk::SkyParams sky[2];
  sky[0].texels = (const unsigned *)sky_pos; sky[0].w = wp; sky[0].h = hp;
  sky[1].texels = (const unsigned *)sky_neg; sky[1].w = wn; sky[1].h = hn;
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 9; ++i) sky[s].inv_rot[i] = (i % 4 == 0) ? 1.0 : 0.0;
  cvk::EfficientFrame F;
  if (!cvk::efficient_frame_pose(c->pos[2], c->pos[3], F)) return -2; /* host values over the platform libm, as the product */
  cvs::Sampler S;
  S.a_min = -0.1 * CV_PI; S.a_max = 1.1 * CV_PI; S.n0 = alpha_nums; S.max_iterations = max_it_sampling;
  S.thr1 = thr1; S.thr2 = thr2;
  bool panic = false;
  std::vector<double> e, s;
  std::vector<uint32_t> st;
  if (dev_sampler) {
    /* the DEVICE-resident sampler's control flow (cv_sampler_dev.h: what lane 0 of sampler_kernel runs) around a serial
     * evaluation; its result is poured into S so that everything downstream is shared */
    std::unique_ptr<cvk::SamplerState> D(new cvk::SamplerState);
    cvk::sampler_reset(*D);
    /* the job's evaluation cache, and the phases of sampler_kernel with the lanes run one after the other (dev_sampler == 2:
     * with speculation, as the kernel does by default) */
    const unsigned LANES = 512;
    std::vector<unsigned long long> hk(cvk::kSpecSlots, cvk::kSpecEmpty);
    std::vector<double> he(cvk::kSpecSlots), hs(cvk::kSpecSlots);
    std::vector<unsigned> hst(cvk::kSpecSlots);
    std::vector<int> hu(cvk::kSpecSlots);
    cvk::SpecTable T{hk.data(), he.data(), hs.data(), hst.data(), hu.data()};
    int spanic = 0;
    for (;;) {
      const bool go = cvk::sampler_plan(*D, alpha_nums, max_it_sampling, S.a_min, S.a_max, thr1, thr2);
      D->n_miss = D->n_eval = 0;
      if (!go) break;
      for (unsigned t = 0; t < D->n_pend; ++t)
        if (!cvk::sampler_take(*D, T, t, spanic)) D->miss[D->n_miss++] = (unsigned short)t;
      const unsigned nm = D->n_miss;
      if (nm) {
        for (unsigned mi = 0; mi < nm; ++mi) cvk::sampler_want(*D, T, D->pend_a[D->miss[mi]], true);
        if (dev_sampler == 2 && !D->overflow) {
          const unsigned depth = D->started ? cvk::sampler_spec_depth(nm, LANES) : 0u;
          const unsigned grid_depth = (!D->started && LANES > nm) ? cvk::sampler_spec_depth(nm - 1u, LANES - nm) : 0u;
          for (unsigned mi = 0; mi < nm; ++mi) cvk::sampler_speculate(*D, T, mi, depth, grid_depth);
        }
        const unsigned ne = D->n_eval < cvk::kSpecEvalCap ? D->n_eval : cvk::kSpecEvalCap;
        for (unsigned k = 0; k < ne; ++k) {
          double ee, sp_;
          uint32_t stp;
          const int status = escape_angle_any(m->kind, M, c->pos[1], D->eval_a[k], delta, max_iter, R, fast, ee, sp_, stp);
          const unsigned slot = D->eval_slot[k];
          he[slot] = ee; hs[slot] = sp_; hst[slot] = stp; hu[slot] = status;
        }
        D->eval_phases += (ne + LANES - 1) / LANES;
        D->evaluated += ne;
        for (unsigned mi = 0; mi < nm; ++mi)
          if (!cvk::sampler_take(*D, T, D->miss[mi], spanic)) D->overflow = 1;
      }
      if (D->overflow) D->finished = 1;
      cvk::sampler_consume(*D, max_it_sampling);
    }
    if (spanic) panic = true;
    if (getenv("TWIN_SAMPLER_DIAG"))
      fprintf(stderr, "twin device sampler (speculation %d): l = %g: %u samples, %u rounds, %llu calls; %u Euler chains, %u points integrated, %u keys\n",
              dev_sampler == 2, c->pos[1], D->n, D->rounds, (unsigned long long)D->calls, D->eval_phases, D->evaluated, D->n_cached);
    if (D->overflow) return -4;
    S.pts.resize(D->n);
    for (unsigned i = 0; i < D->n; ++i) S.pts[i] = cvs::BiPoint{D->a[D->cur][i], D->e[D->cur][i], D->s[D->cur][i]};
    S.calls = D->calls; S.steps = D->steps; S.rounds = D->rounds; S.panicked = D->panicked != 0; S.warned = D->warned != 0;
  } else {
    while (S.plan()) {
      const size_t n = S.pending.size();
      e.resize(n); s.resize(n); st.resize(n);
      for (size_t k = 0; k < n; ++k)
        if (escape_angle_any(m->kind, M, c->pos[1], S.pending[k], delta, max_iter, R, fast, e[k], s[k], st[k]) == cvk::ESC_PANIC)
          panic = true;
      S.consume(e.data(), s.data(), st.data());
    }
  }
  if (panic || S.panicked) return -2;
  *n_out = S.pts.size();
  *calls = S.calls;
  *steps_out = S.steps;
  if (S.pts.size() > cap) return -3;
  std::vector<double> x, ye, ys, m_e, c_e, m_s, c_s;
  for (size_t i = 0; i < S.pts.size(); ++i) {
    sa[i] = S.pts[i].a; se[i] = S.pts[i].e; ss[i] = S.pts[i].s;
    x.push_back(S.pts[i].a); ye.push_back(S.pts[i].e); ys.push_back(S.pts[i].s);
  }
  cvs::interp_tables(x, ye, m_e, c_e);
  cvs::interp_tables(x, ys, m_s, c_s);
  const unsigned W = c->res_x, H = c->res_y;
  for (unsigned py = 0; py < H; ++py)
    for (unsigned px = 0; px < W; ++px) {
      double fin[3], space;
      cvk::efficient_pixel(C, F, px, py, x.data(), m_e.data(), c_e.data(), m_s.data(), c_s.data(), (unsigned)x.size(), fin, space);
      unsigned texel = 0xFF000000u;
      if (space == 1.0 || space == -1.0) {
        const cvk::SkyParams &Sk = sky[space == 1.0 ? 0 : 1];
        unsigned tx, ty;
        cvk::sky_indices(Sk, fin[0], fin[1], fin[2], tx, ty);
        if (tx >= Sk.w) tx = Sk.w - 1;
        if (ty >= Sk.h) ty = Sk.h - 1;
        texel = Sk.texels[(size_t)ty * Sk.w + tx];
      }
      size_t o = (size_t)py * W + px;
      rgb[o * 3 + 0] = texel & 0xFF; rgb[o * 3 + 1] = (texel >> 8) & 0xFF; rgb[o * 3 + 2] = (texel >> 16) & 0xFF;
    }
  return 0;
}

/* inflate_fast.h (the PNG backgrounds' inflater) for tests/test_inflate_host.py: zlib stream in, bytes out.  The input is copied
 * into a buffer with the padding the decoder asks for; returns its status, *out_len = bytes produced, *adler = stored Adler-32 */
#include "../../curvis_amd/csrc/host/inflate_fast.h"
#include <memory>
extern "C" int twin_inflate_zlib(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *adler) {
  std::unique_ptr<uint8_t[]> buf(new uint8_t[in_len + cvinflate::kInputPadding]);
  std::memcpy(buf.get(), in, in_len);
  std::memset(buf.get() + in_len, 0, cvinflate::kInputPadding);
  return cvinflate::inflate_zlib(buf.get(), in_len, out, out_cap, out_len, adler);
}


/* cv_efficient.h interp_index_grid against interp_index: the grid is built over x[0..n) the way the kernels build it (one call of
 * interp_grid_fill per index 0..n), then every query is looked up both ways; returns the number of queries whose indices differ
 * and leaves the grid in G (kInterpGrid + 1 entries) */
extern "C" size_t twin_interp_grid_check(const double *x, unsigned n, const double *q, size_t m, unsigned *G) {
  for (unsigned b = 0; b <= cvk::kInterpGrid; ++b) G[b] = 0xFFFFFFFFu; /* every entry must be written */
  for (unsigned i = 0; i <= n; ++i) cvk::interp_grid_fill(x, n, i, G);
  size_t bad = 0;
  for (size_t k = 0; k < m; ++k)
    if (cvk::interp_index_grid(x, n, q[k], G) != cvk::interp_index(x, n, q[k])) ++bad;
  return bad;
}

/* png_codes.h, the code construction shared by the host PNG writer and the device's png_codes_kernel: lengths for freq[0..n)
 * exactly the way pngio::huffman_lengths wraps it (symbols in (count, index) order, zero counts get no code) */
#include "../../curvis_amd/csrc/png_codes.h"
#include <algorithm>
extern "C" void twin_huffman_lengths(const uint32_t *freq, int n, int maxlen, uint8_t *len) {
  uint64_t key[pngcodes::kMaxLeaves];
  int m = 0;
  for (int i = 0; i < n; ++i) {
    len[i] = 0;
    if (freq[i]) key[m++] = ((uint64_t)freq[i] << 9) | (uint64_t)i;
  }
  if (m == 0) return;
  std::sort(key, key + m);
  uint32_t w[pngcodes::kMaxLeaves];
  uint8_t ls[pngcodes::kMaxLeaves];
  for (int k = 0; k < m; ++k) w[k] = (uint32_t)(key[k] >> 9);
  pngcodes::pm_lengths_sorted(w, m, maxlen, ls);
  for (int k = 0; k < m; ++k) len[key[k] & 511u] = ls[k];
}
extern "C" void twin_length_symbol(int length, int *sym, int *ebits, int *eval) { pngcodes::length_symbol(length, *sym, *ebits, *eval); }
