/* cli_args.h -- part of the `curvis` binary (host/curvis_cli.cpp includes the parts in order; one translation unit):
 * command line (src/cli.rs:35-122), the inputs every subcommand shares, context creation, the two render calls. */
#ifndef CURVIS_CLI_ARGS_H
#define CURVIS_CLI_ARGS_H

namespace {

/* ------------------------------------------------------------------ command line */
struct Args {
  std::string sub, bg1, bg2, out, image_toml, video_toml, metric_toml, camera_toml, sim_toml, mode = "efficient", stats,
      sky_broadcast = "rccl";
  bool sky_broadcast_explicit = false, resume = false;
  bool contexts_auto = false; /* --contexts-per-device 0 / not given: video_main may lower the count for short videos */
  int contexts = 0; /* 0 = automatic (4 in --mode efficient, else 1); video: contexts (= host worker threads) per device: while one waits on the host-side sampler or the D2H copy another's kernels run */
  int devices = 1, device = 0, batch = 0, writers = 0; /* batch 0 = automatic (video: 8 frames per launch, in --mode efficient up to 32: cli_video.h); writers 0 = automatic: a quarter of the host's threads, 4..64 */
  int png_level = -1; /* -1 = the fast PNG writer (png_io.h; the reference's image crate also saves with its fast setting), 0..9 = zlib */
  int encode_bench = 0; /* video, diagnostics: every rendered frame is encoded this many extra times into a scratch file */
  std::string gpu_png = "auto"; /* video: PNG front end on the device (curvis_ctx_deflate_frames): auto = with the fast writer, on, off */
};
[[noreturn]] void die(const std::string &msg, int code = 1) {
  std::fprintf(stderr, "%s\n", msg.c_str());
  std::exit(code);
}
void usage() {
  std::printf(
      "Usage: curvis <COMMAND>\n\nCommands:\n  image   renders a single image frame\n  video   renders a video\n"
      "  custom  runs the custom script\n\n"
      "curvis image <IMAGE FILE 1> <IMAGE FILE 2> [OUTPUT FOLDER] [-i|--image-settings <TOML FILE>]\n"
      "curvis video <IMAGE FILE 1> <IMAGE FILE 2> [OUTPUT FOLDER] [-v|--video-settings <TOML FILE>]\n"
      "  common: [-m|--metric-settings <TOML FILE>] [-c|--camera-settings <TOML FILE>] [-s|--simulation-settings <TOML FILE>]\n"
      "  extensions: [--mode efficient|brute|direct] [--device N] [--devices N] [--batch B] [--stats FILE]\n"
      "              [--sky-broadcast rccl|upload] [--writers T] [--resume] [--png-level -1..9] [--gpu-png auto|on|off]\n"
      "              [--contexts-per-device C]\n");
}
Args parse_args(int argc, char **argv) {
  Args a;
  if (argc < 2) {
    usage();
    die("Subcommand not found");
  }
  a.sub = argv[1];
  if (a.sub == "selftest-png") return a;
  if (a.sub == "-h" || a.sub == "--help") {
    usage();
    std::exit(0);
  }
  std::vector<std::string> pos;
  for (int i = 2; i < argc; ++i) {
    std::string s = argv[i], val;
    auto take = [&](std::string &dst) {
      const size_t eq = s.find('=');
      if (s.rfind("--", 0) == 0 && eq != std::string::npos)
        dst = s.substr(eq + 1);
      else if (i + 1 < argc)
        dst = argv[++i];
      else
        die("error: a value is required for '" + s + "' but none was supplied", 2);
    };
    const std::string key = s.substr(0, s.find('='));
    if (key == "-i" || key == "--image-settings") take(a.image_toml);
    else if (key == "-v" || key == "--video-settings") take(a.video_toml);
    else if (key == "-m" || key == "--metric-settings") take(a.metric_toml);
    else if (key == "-c" || key == "--camera-settings") take(a.camera_toml);
    else if (key == "-s" || key == "--simulation-settings") take(a.sim_toml);
    else if (key == "--mode") take(a.mode);
    else if (key == "--stats") take(a.stats);
    else if (key == "--sky-broadcast") { take(a.sky_broadcast); a.sky_broadcast_explicit = true; }
    else if (key == "--resume") a.resume = true;
    else if (key == "--devices") { take(val); a.devices = std::atoi(val.c_str()); }
    else if (key == "--contexts-per-device") { take(val); a.contexts = std::max(0, std::min(8, std::atoi(val.c_str()))); }
    else if (key == "--device") { take(val); a.device = std::atoi(val.c_str()); }
    else if (key == "--batch") { take(val); a.batch = std::atoi(val.c_str()); }
    else if (key == "--writers") { take(val); a.writers = std::atoi(val.c_str()); }
    else if (key == "--png-level") { take(val); a.png_level = std::max(-1, std::min(9, std::atoi(val.c_str()))); }
    else if (key == "--encode-bench") { take(val); a.encode_bench = std::max(0, std::atoi(val.c_str())); }
    else if (key == "--gpu-png") take(a.gpu_png);
    else if (key == "-h" || key == "--help") { usage(); std::exit(0); }
    else if (!s.empty() && s[0] == '-') die("error: unexpected argument '" + s + "' found", 2);
    else pos.push_back(s);
  }
  if (a.sub == "image" || a.sub == "video") {
    if (pos.size() < 2) die("error: the following required arguments were not provided:\n  <IMAGE FILE 1>\n  <IMAGE FILE 2>", 2);
    if (pos.size() > 3) die("error: unexpected argument '" + pos[3] + "' found", 2);
    a.bg1 = pos[0];
    a.bg2 = pos[1];
    if (pos.size() == 3) a.out = pos[2];
    if (a.sub == "image" && !a.video_toml.empty()) die("error: unexpected argument '-v' found", 2);
    if (a.sub == "video" && !a.image_toml.empty()) die("error: unexpected argument '-i' found", 2);
  }
  if (a.mode != "efficient" && a.mode != "brute" && a.mode != "direct") die("error: --mode must be efficient, brute or direct", 2);
  if (a.sky_broadcast != "rccl" && a.sky_broadcast != "upload") die("error: --sky-broadcast must be rccl or upload", 2);
  if (a.gpu_png != "auto" && a.gpu_png != "on" && a.gpu_png != "off") die("error: --gpu-png must be auto, on or off", 2);
  if (a.devices < 1) a.devices = 1;
  /* --mode efficient spends more than half of a frame's render call on the host (the adaptive sampler between its launches of
   * lone waves) and its kernels do not fill the GPU: several contexts per GPU overlap one's host work and another's kernels, and
   * the kernels of different contexts run side by side.  Measured at 30 000 frames per cell on one MI355X, 32 frames per launch
   * (profiles/round5_eff_contexts_sweep.txt): 2 757 / 4 058 / 5 488 / 6 116 / 5 936 1080p frames/s with 1 / 2 / 3 / 4 / 6 contexts.
   * The per-pixel modes keep the GPU busy by themselves. */
  /* CPUs this process may really use: a container may see every CPU of the host behind a much smaller cgroup quota
   * ("1600000 100000" = 16 CPUs) */
  unsigned hw = std::max(1u, std::thread::hardware_concurrency()), cpus = hw;
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    double quota = 0.0, period = 0.0;
    if (std::fscanf(f, "%lf %lf", &quota, &period) == 2 && quota > 0.0 && period > 0.0) {
      cpus = std::max(1u, std::min(hw, (unsigned)(quota / period + 0.5)));
      hw = std::min(hw, cpus * 4u); /* the quota itself: this many writers at most */
    }
    std::fclose(f);
  }
  /* up to six contexts per GPU in --mode efficient (four below 12 000 frames per GPU, see video_main) -- as far as the host has two CPUs per worker thread (each context's thread runs
   * the sampler's host side; 8 GPUs behind a 16-CPU quota get one context each, not 32 threads fighting the writers) */
  a.contexts_auto = a.contexts == 0;
  if (a.contexts == 0) a.contexts = a.mode == "efficient" ? (int)std::max(1u, std::min(6u, cpus / (2u * (unsigned)a.devices))) : 1; /* video_main: fewer for shorter videos */
  if (a.batch < 0) a.batch = 0;
  if (a.writers < 1) /* encoding a 1080p frame costs 5-25 ms of a host thread (zlib: 20-230): the GPU renders one in 0.4-10 ms */
    a.writers = (int)std::min(64u, std::max(4u, hw / 4u));
  return a;
}

/* CURVIS_DEBUG_TIMING=1: the phases of a run on stderr (settings, background images, contexts, render, save) */
struct PhaseClock {
  const bool on = std::getenv("CURVIS_DEBUG_TIMING") != nullptr;
  double t0 = pngio::now_s(), last = t0;
  void mark(const char *what) {
    if (!on) return;
    const double t = pngio::now_s();
    std::fprintf(stderr, "[curvis timing] %-34s %8.1f ms  (at %8.1f ms)\n", what, (t - last) * 1e3, (t - t0) * 1e3);
    last = t;
  }
};

struct Common {
  curvis_metric metric{CURVIS_METRIC_ELLIS, 0, 1.0, 0.0, 0.0}; /* default: Ellis rho = 1 (ellis_metric_settings.toml) */
  CameraSettings cam;
  SimulationSettings sim;
  pngio::Image sky1, sky2;
  std::string out;
};

void load_common(const Args &a, Common &c, const char *what) {
  std::string err;
  auto need = [&](const std::string &f, const char *label) {
    if (!path_exists(f)) die(std::string("Error with ") + label + ": File \"" + f + "\" not found.");
  };
  need(a.bg1, "background image 1");
  need(a.bg2, "background image 2");
  if (a.out.empty()) {
    char cwd[4096];
    if (!::getcwd(cwd, sizeof cwd)) die("Error with output folder: Could not get current working directory.");
    c.out = cwd;
  } else {
    if (!path_exists(a.out)) die("Error with output folder: File \"" + a.out + "\" not found.");
    if (!is_dir(a.out)) die("Error with output folder: \"" + a.out + "\" is not a folder.");
    c.out = a.out;
  }
  if (!a.metric_toml.empty()) {
    need(a.metric_toml, "metric settings");
    if (!metric_from_toml(a.metric_toml, c.metric, err)) die("Error with metric settings: " + err);
  }
  if (!a.camera_toml.empty()) {
    need(a.camera_toml, "camera settings");
    if (!from_toml(a.camera_toml, c.cam, err)) die("Error with camera settings: " + err);
  }
  if (!a.sim_toml.empty()) {
    need(a.sim_toml, "simulation settings");
    if (!from_toml(a.sim_toml, c.sim, err)) die("Error with simulation settings: " + err);
  }
  /* instantiate_metric (src/main.rs:114-132): constructor panics */
  if (curvis_metric_validate(&c.metric) != CURVIS_OK)
    die(std::string("Error in rendering ") + what + ": metric parameters must be positive (src/metrics.rs:409-456)", 101);
  if (!validate(c.cam, err) || !validate(c.sim, err)) die(std::string("Error in rendering ") + what + ": " + err);
  /* both backgrounds are decoded at the same time (an 8192x4096 PNG takes the better part of a second to inflate and
   * unfilter); errors are reported in the reference's order, image 1 first */
  std::string err2;
  bool ok2 = false;
  /* the HIP runtime comes up (hipInit, behind curvis_device_count) while the backgrounds are decoded: the two slow each other
   * down -- the decode 90 -> 150 ms, the runtime's start-up 70 -> ~110 -- and still finish 10-25 ms sooner than one after the
   * other (251 -> 235 ms per `curvis image`; CURVIS_NO_EARLY_INIT=1 switches it off) */
  std::thread early_init;
  if (!std::getenv("CURVIS_NO_EARLY_INIT")) {
    try {
      early_init = std::thread([] { (void)curvis_device_count(); });
    } catch (const std::exception &) {
    }
  }
  struct JoinEarly {
    std::thread &t;
    ~JoinEarly() { if (t.joinable()) t.join(); }
  } join_early{early_init};
  std::thread second([&] { ok2 = jpegio::load_image(a.bg2, c.sky2, err2); });
  const bool ok1 = jpegio::load_image(a.bg1, c.sky1, err);
  second.join();
  /* die() is std::exit(): it does not unwind, so JoinEarly would never run, and exit()'s handlers / the static destructors of
   * libamdhip64 would tear the runtime down while hipInit is still executing on the helper thread (a decode error returns in
   * microseconds, hipInit takes ~100 ms) -- join it first, the error path is not the fast path */
  if ((!ok1 || !ok2) && early_init.joinable()) early_init.join();
  if (!ok1) die(std::string("Error in rendering ") + what + ": background image 1: " + err);
  if (!ok2) die(std::string("Error in rendering ") + what + ": background image 2: " + err2);
}

void check(int rc, curvis_ctx *ctx, const char *what) {
  if (rc == CURVIS_OK) return;
  const char *msg = curvis_last_error(ctx);
  die(std::string("Error in rendering ") + what + ": " + (msg ? msg : "") + " (code " + std::to_string(rc) + ")",
      (rc == CURVIS_E_CAMERA_OUTSIDE || rc == CURVIS_E_PARALLEL || rc == CURVIS_E_SAMPLING) ? 101 : 1);
}

curvis_ctx *make_ctx_bare(int device, const char *what) {
  curvis_ctx *ctx = nullptr;
  int rc = curvis_ctx_create(device, &ctx);
  if (rc != CURVIS_OK) die(std::string("Error in rendering ") + what + ": " + curvis_last_error(nullptr));
  /* tuning knobs of the library for experiments (tools/gpu_eff_speculation_sweep.py): CURVIS_CTX_OPTIONS="key=value,key=value"
   * is applied to every context of the run; an unknown key is an error, not a silent no-op */
  if (const char *opts = std::getenv("CURVIS_CTX_OPTIONS")) {
    std::stringstream ss(opts);
    std::string kv;
    while (std::getline(ss, kv, ',')) {
      const size_t eq = kv.find('=');
      if (eq == std::string::npos || curvis_ctx_set_option(ctx, kv.substr(0, eq).c_str(), std::atoll(kv.c_str() + eq + 1)) != CURVIS_OK)
        die("error: CURVIS_CTX_OPTIONS: cannot set `" + kv + "`", 2);
    }
  }
  return ctx;
}
void upload_skies(curvis_ctx *ctx, const Common &c, const char *what) {
  check(curvis_ctx_set_sky(ctx, 0, c.sky1.rgba.data(), c.sky1.w, c.sky1.h), ctx, what);
  check(curvis_ctx_set_sky(ctx, 1, c.sky2.rgba.data(), c.sky2.w, c.sky2.h), ctx, what);
}
curvis_ctx *make_ctx(int device, const Common &c, const char *what) {
  curvis_ctx *ctx = make_ctx_bare(device, what);
  upload_skies(ctx, c, what);
  return ctx;
}

/* the sampler of a batch that render_frames will render later, started now (efficient mode; curvis_ctx_prefetch_efficient is a
 * no-op where the library would not sample on the device).  Failures are not fatal: the render call then samples itself. */
void prefetch_frames(curvis_ctx *ctx, const Args &a, const Common &c, const curvis_camera *cams, uint32_t n, double thr2) {
  if (a.mode != "efficient" || n == 0) return;
  (void)curvis_ctx_prefetch_efficient(ctx, &c.metric, cams, n, c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                      c.sim.ray_integration_step, c.sim.sampling_initial_nums, c.sim.sampling_initial_nums,
                                      c.sim.sampling_convergence_threshold_1, thr2);
}

/* per-frame statistics of the last render_frames call of this thread in "direct" mode (one render call per frame there;
 * the batch calls of the other modes keep theirs inside the context: curvis_ctx_frame_stats) */
thread_local std::vector<curvis_stats> g_direct_frame_stats;

int render_frames(curvis_ctx *ctx, const Args &a, const Common &c, const curvis_camera *cams, uint32_t n, double thr2,
                  uint8_t *rgb, curvis_stats *st) {
  if (a.mode == "brute")
    return curvis_render_brute_batch(ctx, &c.metric, cams, n, c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                     c.sim.ray_integration_step, rgb, st);
  if (a.mode == "direct") { /* extension: compute_escape_angle for every pixel, no sampling / interpolation; frame by frame */
    curvis_stats tot;
    std::memset(&tot, 0, sizeof tot);
    const size_t fbytes = (size_t)cams[0].res_x * cams[0].res_y * 3;
    g_direct_frame_stats.clear();
    for (uint32_t f = 0; f < n; ++f) {
      curvis_stats one;
      const int rc = curvis_render_direct(ctx, &c.metric, cams + f, c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                          c.sim.ray_integration_step, rgb + (size_t)f * fbytes, &one);
      if (rc != CURVIS_OK) return rc;
      g_direct_frame_stats.push_back(one);
      tot.rays += one.rays;
      tot.steps += one.steps;
      tot.n_pos += one.n_pos;
      tot.n_neg += one.n_neg;
      tot.n_none += one.n_none;
      tot.n_oob += one.n_oob;
      tot.kernel_ms += one.kernel_ms;
      tot.integrate_ms += one.integrate_ms;
      tot.total_ms += one.total_ms;
    }
    if (st) *st = tot;
    return CURVIS_OK;
  }
  /* src/main.rs:46-47 / :106-107: alphas_num AND max_iterations_sampling both take sampling_initial_nums */
  return curvis_render_efficient_batch(ctx, &c.metric, cams, n, c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                       c.sim.ray_integration_step, c.sim.sampling_initial_nums, c.sim.sampling_initial_nums,
                                       c.sim.sampling_convergence_threshold_1, thr2, rgb, st);
}

}  // namespace

#endif /* CURVIS_CLI_ARGS_H */
