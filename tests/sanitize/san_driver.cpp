// san_driver.cpp -- TEST INFRASTRUCTURE: the CPU oracle (oracle/curvis_oracle.c) and the x86 twin of the device
// functions (tests/host_twin/twin.cpp) exercised under AddressSanitizer + UndefinedBehaviorSanitizer
// (tests/sanitize/Makefile, run by tests/test_sanitize.py).  Both renderers, both math flavours, all three
// metrics, ragged frame sizes, adversarial cameras (inside the throat, next to a pole, in the -l space), the
// camera-path reader incl. its off-by-one, and the sampler.  Exit status 0 = no sanitizer report and oracle(cv)
// == twin on every frame; the sanitizers abort the process on the first finding (-fno-sanitize-recover).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/curvis_hip.h"
extern "C" {
#include "../../oracle/curvis_oracle.h"
void twin_render(const curvis_metric *m, const curvis_camera *c, const uint8_t *sky_pos, unsigned wp, unsigned hp,
                 const uint8_t *sky_neg, unsigned wn, unsigned hn, unsigned max_iter, double R, double delta, uint8_t *rgb,
                 curvis_ray_debug *dbg, int fast);
int twin_render_efficient(const curvis_metric *m, const curvis_camera *c, const uint8_t *sky_pos, unsigned wp, unsigned hp,
                          const uint8_t *sky_neg, unsigned wn, unsigned hn, unsigned max_iter, double R, double delta,
                          unsigned alpha_nums, unsigned max_it_sampling, double thr1, double thr2, uint8_t *rgb, double *sa,
                          double *se, double *ss, size_t cap, size_t *n_out, uint64_t *calls, uint64_t *steps_out, int fast);
}

static int failures = 0;
#define CHECK(cond, ...)                         \
  do {                                           \
    if (!(cond)) {                               \
      std::fprintf(stderr, "FAIL: " __VA_ARGS__); \
      std::fprintf(stderr, "\n");                \
      ++failures;                                \
    }                                            \
  } while (0)

static std::vector<uint8_t> make_sky(unsigned w, unsigned h, unsigned seed) {
  std::vector<uint8_t> s((size_t)w * h * 4);
  unsigned x = seed * 2654435761u + 1u;
  for (size_t i = 0; i < s.size(); ++i) {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    s[i] = (i % 4 == 3) ? 255 : (uint8_t)(x >> 24);
  }
  return s;
}

struct Scene {
  int kind;
  double rho, m, a;
  double pos[4], fwd[3], up[3];
  unsigned w, h, cap;
};

int main() {
  const unsigned SW = 64, SH = 32;
  const std::vector<uint8_t> sp = make_sky(SW, SH, 1), sn = make_sky(SW, SH, 2);
  cvo_sky osp, osn;
  std::memset(&osp, 0, sizeof osp);
  std::memset(&osn, 0, sizeof osn);
  osp.rgba = sp.data(); osp.w = SW; osp.h = SH;
  osn.rgba = sn.data(); osn.w = SW; osn.h = SH;
  for (int i = 0; i < 9; ++i) osp.inv_rot[i] = osn.inv_rot[i] = (i % 4 == 0) ? 1.0 : 0.0;
  const double HP = 1.5707963267948966;
  const Scene scenes[] = {
      {CVO_ELLIS, 1.0, 0, 0, {0, 5, HP, 0}, {-1, 0, 0}, {0, 0, 1}, 13, 7, 2600},
      {CVO_INTERSTELLAR, 1.0, 0.1, 1e-4, {0, 5, HP, 0}, {-1, 0, 0}, {0, 0, 1}, 9, 9, 2600},
      {CVO_INTERSTELLAR, 1.0, 0.1, 1e-4, {0, 5e-5, 1.0, 2.0}, {1, 0.2, -0.1}, {0, 0, 1}, 8, 5, 3000},  /* inside the throat */
      {CVO_ELLIS, 1.0, 0, 0, {0, -3, 0.05, 4.0}, {1, 0.3, 0.2}, {0.1, 0, 1}, 7, 6, 900},                 /* next to a pole, -l, cap binds */
      {CVO_FLAT, 0, 0, 0, {0, 5, 1.0, 0.5}, {1, 0.3, 0.2}, {0, 0, 1}, 6, 4, 4096},
      {CVO_ELLIS, 0.3, 0, 0, {0, 0.0, HP, 0}, {0, 1, 0}, {0, 0, 1}, 5, 5, 1500},                          /* l = 0 exactly */
  };
  for (const Scene &S : scenes) {
    cvo_metric om = {S.kind, 0, S.rho, S.m, S.a};
    curvis_metric pm = {S.kind, 0, S.rho, S.m, S.a};
    cvo_camera oc;
    CHECK(cvo_camera_new(&oc, S.pos, S.fwd, S.up, 15.0, 43.0, S.w, S.h) == 0, "camera");
    curvis_camera pc;
    std::memcpy(pc.pos, oc.pos, sizeof pc.pos);
    std::memcpy(pc.rot, oc.rot, sizeof pc.rot);
    pc.focal = oc.focal; pc.sensor_w = oc.sensor_w; pc.sensor_h = oc.sensor_h; pc.res_x = S.w; pc.res_y = S.h;
    const size_t n = (size_t)S.w * S.h;
    for (int fl = 0; fl < 2; ++fl) {
      std::vector<uint8_t> rgb(n * 3);
      std::vector<cvo_ray_debug> dbg(n);
      cvo_stats st;
      std::memset(&st, 0, sizeof st); /* the oracle accumulates into the caller's counters */
      CHECK(cvo_render_image(fl, &om, &oc, &osp, &osn, S.cap, 100.0, 0.05, 0, 1, rgb.data(), dbg.data(), &st) == 0, "oracle render");
      CHECK(st.rays == n && st.n_pos + st.n_neg + st.n_none == n, "oracle stats");
      if (fl == CVO_CV) {
        for (int fast = 0; fast < 2; ++fast) {
          std::vector<uint8_t> trgb(n * 3);
          std::vector<curvis_ray_debug> tdbg(n);
          twin_render(&pm, &pc, sp.data(), SW, SH, sn.data(), SW, SH, S.cap, 100.0, 0.05, trgb.data(), tdbg.data(), fast);
          CHECK(std::memcmp(trgb.data(), rgb.data(), rgb.size()) == 0, "twin pixels differ (kind %d fast %d)", S.kind, fast);
          for (size_t i = 0; i < n; ++i)
            CHECK(tdbg[i].steps == dbg[i].steps && tdbg[i].code == dbg[i].code &&
                      std::memcmp(&tdbg[i].x[1], &dbg[i].x[1], 3 * sizeof(double)) == 0 &&
                      std::memcmp(tdbg[i].p, dbg[i].p, sizeof dbg[i].p) == 0,
                  "twin ray %zu differs (kind %d fast %d)", i, S.kind, fast);
        }
      }
      if (S.kind != CVO_FLAT && std::fabs(S.pos[1]) > 1.0 && S.pos[2] == HP) { /* efficient renderer */
        std::vector<uint8_t> ergb(n * 3);
        cvo_samples smp;
        cvo_stats est;
        std::memset(&est, 0, sizeof est);
        const int rc = cvo_render_image_efficient(fl, &om, &oc, &osp, &osn, S.cap, 100.0, 0.05, 40, 12, 1e-4, 1e-4, ergb.data(), &smp, &est);
        CHECK(rc == 0, "oracle efficient render rc %d", rc);
        if (rc == 0 && fl == CVO_CV) {
          std::vector<uint8_t> trgb(n * 3);
          std::vector<double> a(4096), e(4096), s(4096);
          size_t tn = 0;
          uint64_t calls = 0, steps = 0;
          const int trc = twin_render_efficient(&pm, &pc, sp.data(), SW, SH, sn.data(), SW, SH, S.cap, 100.0, 0.05, 40, 12, 1e-4, 1e-4,
                                                trgb.data(), a.data(), e.data(), s.data(), a.size(), &tn, &calls, &steps, 1);
          CHECK(trc == 0 && tn == smp.n && calls == smp.calls && steps == smp.steps, "twin efficient bookkeeping");
          CHECK(trc == 0 && std::memcmp(trgb.data(), ergb.data(), ergb.size()) == 0, "twin efficient pixels differ");
        }
        if (rc == 0) cvo_samples_free(&smp);
      }
    }
  }
  /* camera path: CRLF file, short last line, the off-by-one at the last segment */
  {
    const char *csv = "/tmp/curvis_san_path.csv";
    FILE *f = std::fopen(csv, "wb");
    std::fputs("t,l,theta,phi,fx,fy,fz,upx,upy,upz\r\n", f);
    for (int i = 0; i < 5; ++i) std::fprintf(f, "%d.0,3.0,1.5707963267948966,%g,-1.0,0.0,0.0,0.0,0.0,1.0\r\n", i, 0.1 * i);
    std::fclose(f);
    cvo_path p;
    CHECK(cvo_load_path(csv, &p) == 0 && p.n == 5, "load_path");
    double pos[4], fwd[3], up[3];
    CHECK(cvo_path_camera(&p, 1.5, pos, fwd, up) == 0, "path_camera inside");
    CHECK(cvo_path_camera(&p, 3.5, pos, fwd, up) == -2, "path_camera off-by-one must report the out-of-bounds read, not perform it");
    CHECK(cvo_path_camera(&p, 9.0, pos, fwd, up) == -1, "path_camera beyond");
    std::vector<double> times(64);
    CHECK(cvo_times_of_frames(0.0, 4.0, 4.0, times.data(), times.size()) == 16, "times_of_frames");
    cvo_path_free(&p);
    std::remove(csv);
  }
  /* elementary functions of both flavours on special values */
  {
    const double xs[] = {0.0, -0.0, 1.0, -1.0, 1e-310, 5e-324, 1e300, -1e300, HUGE_VAL, -HUGE_VAL, NAN, 6.25, 3.141592653589793};
    const size_t n = sizeof xs / sizeof xs[0];
    std::vector<double> out(n);
    for (int fl = 0; fl < 2; ++fl)
      for (int op = 0; op < 6; ++op) cvo_math_array(fl, op, xs, xs, out.data(), n);
  }
  if (failures) {
    std::fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  std::puts("sanitize ok");
  return 0;
}
