/* cli_image.h -- part of the `curvis` binary (host/curvis_cli.cpp includes the parts in order; one translation unit):
 * `curvis image` (src/main.rs:171-205, src/rendering.rs:85-117). */
#ifndef CURVIS_CLI_IMAGE_H
#define CURVIS_CLI_IMAGE_H

namespace {

int image_main(const Args &a) {
  std::printf("Image rendering\n");
  Common c;
  ImageSettings is;
  std::string err;
  if (!a.image_toml.empty()) {
    if (!path_exists(a.image_toml)) die("Error with image settings: File \"" + a.image_toml + "\" not found.");
    if (!from_toml(a.image_toml, is, err)) die("Error with image settings: " + err);
  }
  PhaseClock clk;
  load_common(a, c, "image");
  clk.mark("settings + background images");
  if (is.image_name.empty()) die("Error in rendering image: Image name cannot be an empty string.");
  const double pos[4] = {is.t, is.l, is.theta, is.phi}, fwd[3] = {is.forward_x, is.forward_y, is.forward_z},
               up[3] = {is.up_x, is.up_y, is.up_z};
  curvis_camera cam;
  int rc = curvis_camera_init(&cam, pos, fwd, up, c.cam.focal_length, c.cam.diagonal, c.cam.resolution_x, c.cam.resolution_y);
  if (rc == CURVIS_E_PARALLEL) die("Error in rendering image: Forward and up vectors must not be parallel", 101);
  if (rc != CURVIS_OK) die("Error in rendering image: invalid camera settings");
  curvis_ctx *ctx = make_ctx(a.device, c, "image");
  clk.mark("context + sky upload");
  if (!path_exists(c.out) && ::mkdir(c.out.c_str(), 0777) != 0)
    die("Error in rendering image: Could not create video output folder \"" + c.out + "\"");
  std::vector<uint8_t> rgb((size_t)cam.res_x * cam.res_y * 3);
  curvis_stats st;
  std::memset(&st, 0, sizeof st);
  if (a.mode == "brute" && a.devices > 1) {
    /* --mode brute --devices N: the rows of the ONE frame are split over N GPUs (rays are independent,
     * src/systems.rs:316-326); one host thread + one context per GPU, the bands land in place in `rgb` */
    std::vector<curvis_ctx *> ctxs((size_t)a.devices, nullptr);
    std::vector<curvis_stats> sts((size_t)a.devices);
    std::vector<int> rcs((size_t)a.devices, CURVIS_OK);
    ctxs[0] = ctx;
    const bool share = std::getenv("CURVIS_TEST_SHARE_DEVICE") != nullptr; /* test hook: all bands on one GPU */
    for (int r = 1; r < a.devices; ++r) ctxs[(size_t)r] = make_ctx(share ? a.device : a.device + r, c, "image");
    const uint32_t H = cam.res_y, base = H / (uint32_t)a.devices, extra = H % (uint32_t)a.devices;
    std::vector<std::thread> th;
    for (int r = 0; r < a.devices; ++r)
      th.emplace_back([&, r] {
        const uint32_t begin = (uint32_t)r * base + std::min<uint32_t>((uint32_t)r, extra);
        const uint32_t count = base + ((uint32_t)r < extra ? 1u : 0u);
        std::memset(&sts[(size_t)r], 0, sizeof(curvis_stats));
        if (count == 0) return;
        rcs[(size_t)r] = curvis_render_brute_rows(ctxs[(size_t)r], &c.metric, &cam, begin, count,
                                                  c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                                  c.sim.ray_integration_step, rgb.data() + (size_t)begin * cam.res_x * 3,
                                                  &sts[(size_t)r]);
      });
    for (auto &t : th) t.join();
    for (int r = 0; r < a.devices; ++r) {
      check(rcs[(size_t)r], ctxs[(size_t)r], "image");
      st.rays += sts[(size_t)r].rays;
      st.steps += sts[(size_t)r].steps;
      st.n_pos += sts[(size_t)r].n_pos;
      st.n_neg += sts[(size_t)r].n_neg;
      st.n_none += sts[(size_t)r].n_none;
      st.n_oob += sts[(size_t)r].n_oob;
      st.kernel_ms = std::max(st.kernel_ms, sts[(size_t)r].kernel_ms);
      if (r > 0) curvis_ctx_destroy(ctxs[(size_t)r]);
    }
  } else {
    check(render_frames(ctx, a, c, &cam, 1, c.sim.sampling_convergence_threshold_2, rgb.data(), &st), ctx, "image");
  }
  clk.mark("render");
  /* PathBuf::join(image_name).with_extension("png") (src/rendering.rs:108): an existing extension is REPLACED */
  const std::string file = (std::filesystem::path(c.out) / std::filesystem::path(is.image_name).replace_extension("png")).string();
  if (!pngio::save_rgb8(file, rgb.data(), cam.res_x, cam.res_y, err, a.png_level))
    die("Error in rendering image: Could not save image frame \"" + file + "\" due to error: " + err);
  clk.mark("PNG encode + write");
  if (!a.stats.empty()) {
    FILE *f = std::fopen(a.stats.c_str(), "w");
    if (f) {
      std::fprintf(f, "{\"frame\": 0, \"mode\": \"%s\", \"rays\": %llu, \"steps\": %llu, \"n_pos\": %llu, \"n_neg\": %llu, \"n_none\": %llu, \"n_oob\": %llu, \"kernel_ms\": %.4f, \"mray_steps_per_s\": %.1f}\n",
                   a.mode.c_str(), (unsigned long long)st.rays, (unsigned long long)st.steps, (unsigned long long)st.n_pos,
                   (unsigned long long)st.n_neg, (unsigned long long)st.n_none, (unsigned long long)st.n_oob, st.kernel_ms,
                   st.kernel_ms > 0.0 ? (double)st.steps / st.kernel_ms / 1e3 : 0.0);
      std::fclose(f);
    }
  }
  curvis_ctx_destroy(ctx);
  clk.mark("context destroyed");
  return 0;
}

}  // namespace

#endif /* CURVIS_CLI_IMAGE_H */
