/* efficient_host.h -- host side of render_image_efficient (E1-E4): batched escape-angle evaluation, the adaptive sampler
 * driver with speculation, per-pixel launch; direct mode; trajectories.
 * Part of the ONE translation unit curvis_hip.hip (included there, nowhere else). */
#pragma once

namespace {

/* ---- efficient mode ------------------------------------------------------------------------- */

template <int KIND>
int launch_escape_kind(curvis_ctx *ctx, bool fast, const EscapeAngleParams &P) {
  const unsigned blocks = (P.n + 63u) / 64u;
  if (fast)
    hipLaunchKernelGGL((escape_angle_kernel<KIND, true>), dim3(blocks), dim3(64), 0, ctx->stream, P);
  else
    hipLaunchKernelGGL((escape_angle_kernel<KIND, false>), dim3(blocks), dim3(64), 0, ctx->stream, P);
  HIP_TRY(ctx, hipGetLastError());
  return CURVIS_OK;
}

/* evaluate compute_escape_angle for a batch on the GPU */
int eval_escape_batch(curvis_ctx *ctx, const curvis_metric *metric, const cvk::MetricParams &MP,
                      const std::vector<double> &alpha, const std::vector<double> &lcam, uint32_t max_iter,
                      double max_radius, double delta, std::vector<double> &angle, std::vector<double> &space,
                      std::vector<uint32_t> &steps, std::vector<int> &status, double *ms_acc) {
  const size_t n = alpha.size();
  angle.resize(n);
  space.resize(n);
  steps.resize(n);
  status.resize(n);
  if (n == 0) return CURVIS_OK;
  /* layout: alpha | l | angle | space (f64) | steps (u32) | status (i32) */
  const size_t bytes = n * (4 * sizeof(double) + sizeof(unsigned) + sizeof(int));
  int rc = ensure_device(ctx, ctx->d_eff, ctx->eff_cap, bytes);
  if (rc) return rc;
  double *d_alpha = (double *)ctx->d_eff, *d_l = d_alpha + n, *d_angle = d_l + n, *d_space = d_angle + n;
  unsigned *d_steps = (unsigned *)(d_space + n);
  int *d_status = (int *)(d_steps + n);
  /* one pinned staging buffer, one copy in and one copy out per launch: pageable hipMemcpyAsync of more than
   * 1 MiB takes a path that costs ~10 ms per array on this stack (a 262 144-point launch took 20-30 ms instead of
   * 3), and six small pageable copies per launch cost more host time than the kernel of a small launch */
  if (ctx->h_eff_cap < bytes) {
    if (ctx->h_eff) HIP_TRY(ctx, hipHostFree(ctx->h_eff));
    ctx->h_eff = nullptr;
    ctx->h_eff_cap = 0;
    const size_t cap = bytes + bytes / 2;
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_eff, cap));
    ctx->h_eff_cap = cap;
  }
  double *h_alpha = (double *)ctx->h_eff, *h_l = h_alpha + n;
  std::memcpy(h_alpha, alpha.data(), n * sizeof(double));
  std::memcpy(h_l, lcam.data(), n * sizeof(double));
  HIP_TRY(ctx, hipMemcpyAsync(d_alpha, h_alpha, 2 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  EscapeAngleParams P;
  P.metric = MP;
  P.alpha = d_alpha;
  P.l_cam = d_l;
  P.angle = d_angle;
  P.space = d_space;
  P.steps = d_steps;
  P.status = d_status;
  P.n = (unsigned)n;
  P.max_iter = max_iter;
  P.max_radius = max_radius;
  P.delta = delta;
  P.fast_ok = cvk::metric_fast_ok(metric->kind, MP, max_radius) ? 1 : 0;
  const bool fast = ctx->fast_math != 0;
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  switch (metric->kind) {
    case CURVIS_METRIC_ELLIS:
      rc = launch_escape_kind<cvk::METRIC_ELLIS>(ctx, fast, P);
      break;
    case CURVIS_METRIC_INTERSTELLAR:
      rc = launch_escape_kind<cvk::METRIC_INTERSTELLAR>(ctx, fast, P);
      break;
    default:
      rc = launch_escape_kind<cvk::METRIC_FLAT>(ctx, fast, P);
      break;
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  const size_t out_bytes = n * (2 * sizeof(double) + sizeof(unsigned) + sizeof(int));
  unsigned char *h_out = ctx->h_eff + 2 * n * sizeof(double);
  HIP_TRY(ctx, hipMemcpyAsync(h_out, d_angle, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(angle.data(), h_out, n * sizeof(double));
  std::memcpy(space.data(), h_out + n * sizeof(double), n * sizeof(double));
  std::memcpy(steps.data(), h_out + 2 * n * sizeof(double), n * sizeof(unsigned));
  std::memcpy(status.data(), h_out + 2 * n * sizeof(double) + n * sizeof(unsigned), n * sizeof(int));
  float ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  if (ms_acc) *ms_acc += ms;
  return CURVIS_OK;
}


/* The per-pixel kernel divides by four constants of the call -- the resolution (twice), pi and 2 pi -- and takes the first half of
 * those divisions, y = v_rcp_f64 + two Newton steps, from here: formed ONCE per context and resolution, on the device, by the same
 * instructions the compiler's own expansion of `/` uses (cv_device.h recip_chain), so that the kernel's quotients stay that expansion's
 * quotients bit for bit.  ~50 us, once. */
int ensure_pixel_recips(curvis_ctx *ctx, double res_x, double res_y, cvk::PixelRecips &out) {
  curvis_ctx::PixRecips &R = ctx->pix_recips;
  if (!R.valid || std::memcmp(&R.res_x, &res_x, sizeof res_x) != 0 || std::memcmp(&R.res_y, &res_y, sizeof res_y) != 0) {
    const double d[4] = {res_x, res_y, CV_PI, 2.0 * CV_PI};
    double y[4] = {0, 0, 0, 0}, *dev = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&dev, sizeof d + sizeof y));
    hipError_t e = hipMemcpyAsync(dev, d, sizeof d, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(recip_chain_kernel, dim3(1), dim3(64), 0, ctx->stream, dev, dev + 4, 4u);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(y, dev + 4, sizeof y, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(ctx, CURVIS_E_HIP, std::string("reciprocals of the pixel kernel's constants: ") + hipGetErrorString(e));
    for (int k = 0; k < 4; ++k) /* a resolution of 1 .. 2^32 and pi: nothing else can come out */
      if (!(y[k] > 0.0) || !std::isfinite(y[k])) return fail(ctx, CURVIS_E_HIP, "reciprocals of the pixel kernel's constants: not finite");
    R.res_x = res_x;
    R.res_y = res_y;
    R.y.y_res_x = y[0];
    R.y.y_res_y = y[1];
    R.y.y_pi = y[2];
    R.y.y_two_pi = y[3];
    R.valid = true;
  }
  out = R.y;
  return CURVIS_OK;
}

/* ---- efficient mode with the DEVICE-RESIDENT sampler (kernels_efficient.h sampler_kernel) ------------------------------------
 * One launch samples every frame of the call -- a workgroup per distinct camera radius, rounds and all --, the per-pixel kernel
 * follows on the same stream and reads the tables where the sampler left them: no host round trip per refinement round, no
 * evaluation cache, no table building on the host.  What comes back is 32 bytes per job (counts and status) and the frame
 * counters; the sample tables themselves are fetched only if curvis_ctx_samples asks for them.
 * Returns CURVIS_OK, an error, or kSamplerFallback: a table outgrew the kernel's fixed arrays (cv_sampler_dev.h kSamplerCap) --
 * the caller then runs the host-paced sampler, which has no such bound. */
constexpr int kSamplerFallback = 1;

template <int KIND>
int launch_sampler_kind(curvis_ctx *ctx, hipStream_t stream, bool fast, const SamplerParams &P) {
  if (fast)
    hipLaunchKernelGGL((sampler_kernel<KIND, true>), dim3(P.n_jobs), dim3(kSamplerThreads), 0, stream, P);
  else
    hipLaunchKernelGGL((sampler_kernel<KIND, false>), dim3(P.n_jobs), dim3(kSamplerThreads), 0, stream, P);
  HIP_TRY(ctx, hipGetLastError());
  return CURVIS_OK;
}

/* what a sampler launch depends on: two launches with equal keys produce equal tables */
bool sampler_key_equal(const curvis_ctx::SamplerSlot &S, const curvis_metric &metric, const curvis_camera *cams, uint32_t n_frames, uint32_t max_iter,
                       double max_radius, double delta, uint32_t alpha_nums, uint32_t max_iterations_sampling, double thr1, double thr2, int fast,
                       int speculate) {
  /* (field by field: the struct has padding, and a copy need not carry it) */
  const bool same_metric = S.metric.kind == metric.kind && std::memcmp(&S.metric.rho, &metric.rho, sizeof(double)) == 0 &&
                           std::memcmp(&S.metric.m, &metric.m, sizeof(double)) == 0 && std::memcmp(&S.metric.a, &metric.a, sizeof(double)) == 0;
  if (!S.valid || S.n_frames != n_frames || !same_metric || S.max_iter != max_iter ||
      S.alpha_nums != alpha_nums || S.max_iterations_sampling != max_iterations_sampling || S.fast != fast || S.speculate != speculate)
    return false;
  const double p[4] = {max_radius, delta, thr1, thr2};
  if (std::memcmp(S.params, p, sizeof p) != 0) return false;
  for (uint32_t f = 0; f < n_frames; ++f)
    if (std::memcmp(&S.l_frame[f], &cams[f].pos[1], sizeof(double)) != 0) return false;
  return true;
}

/* Sample the frames of a call on `stream` into slot `slot` of the context (device buffer + page-locked mirror of its own): jobs,
 * staging, the sampler kernel, the jobs' results on their way back, an event when all of that is done.  Nothing waits here. */
int sampler_submit(curvis_ctx *ctx, unsigned slot, hipStream_t stream, const curvis_metric *metric, const cvk::MetricParams &MP,
                   const curvis_camera *cams, uint32_t n_frames, uint32_t max_iter, double max_radius, double delta, uint32_t alpha_nums,
                   uint32_t max_iterations_sampling, double thr1, double thr2) {
  curvis_ctx::SamplerSlot &S = ctx->samp[slot];
  /* the slot's previous occupant may have been submitted on the OTHER stream (a prefetch nobody consumed, then a call that samples
   * itself, or the reverse): its kernel, its staging copy and its read-back must be over before the buffers are touched again.  It
   * was submitted two submissions ago, so this wait is over before it starts. */
  if (S.seq != 0 && S.done) HIP_TRY(ctx, hipEventSynchronize(S.done));
  if (ctx->dev_samples.valid && ctx->dev_samples.slot == slot) ctx->dev_samples.overwritten = true; /* curvis_ctx_samples: see fetch_device_samples */
  S.valid = false;
  /* jobs: one per distinct radial coordinate of the cameras (bit pattern) */
  S.job_of_frame.assign(n_frames, 0u);
  S.l_job.clear();
  S.l_frame.resize(n_frames);
  {
    std::map<uint64_t, unsigned> seen;
    for (uint32_t f = 0; f < n_frames; ++f) {
      uint64_t key;
      std::memcpy(&key, &cams[f].pos[1], sizeof key);
      auto it = seen.find(key);
      if (it == seen.end()) {
        it = seen.emplace(key, (unsigned)S.l_job.size()).first;
        S.l_job.push_back(cams[f].pos[1]);
      }
      S.job_of_frame[f] = it->second;
      S.l_frame[f] = cams[f].pos[1];
    }
  }
  const unsigned n_jobs = (unsigned)S.l_job.size();
  const size_t T = (size_t)n_jobs * cvk::kSamplerCap, SS = (size_t)n_jobs * cvk::kSpecSlots;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  /* staged from the host in one copy ... */
  const size_t o_jf = carve(sizeof(unsigned) * n_frames), o_to = carve(sizeof(unsigned) * n_frames), o_go = carve(sizeof(unsigned) * n_frames),
               o_l = carve(sizeof(double) * n_jobs);
  const size_t staged = off;
  /* ... written by the sampler kernel */
  S.o_tab_off = o_to;
  S.o_grid_off = o_go;
  S.o_grid = carve(sizeof(unsigned) * n_jobs * (cvk::kInterpGrid + 1u));
  S.o_tab_n = carve(sizeof(unsigned) * n_frames);
  S.o_res = carve(sizeof(cvk::SamplerResult) * n_jobs);
  for (size_t &o : S.o_tab) o = carve(sizeof(double) * T);
  /* the jobs' evaluation caches (cv_sampler_dev.h SpecTable): 256 KB each */
  const size_t o_sk = carve(sizeof(unsigned long long) * SS), o_se = carve(sizeof(double) * SS), o_ss = carve(sizeof(double) * SS),
               o_st = carve(sizeof(unsigned) * SS), o_su = carve(sizeof(int) * SS);
  int rc = ensure_device(ctx, S.d, S.d_cap, off);
  if (rc) return rc;
  S.res_bytes = sizeof(cvk::SamplerResult) * n_jobs;
  S.h_res_off = staged;
  const size_t pinned = staged + ((S.res_bytes + 255) & ~(size_t)255);
  if (S.h_cap < pinned) {
    if (S.h) HIP_TRY(ctx, hipHostFree(S.h));
    S.h = nullptr;
    S.h_cap = 0;
    HIP_TRY(ctx, hipHostMalloc((void **)&S.h, pinned + pinned / 2));
    S.h_cap = pinned + pinned / 2;
  }
  if (!S.done) HIP_TRY(ctx, hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
  if (!S.t0) {
    HIP_TRY(ctx, hipEventCreate(&S.t0));
    HIP_TRY(ctx, hipEventCreate(&S.t1));
  }
  std::memcpy(S.h + o_jf, S.job_of_frame.data(), sizeof(unsigned) * n_frames);
  {
    auto *to = reinterpret_cast<unsigned *>(S.h + o_to);
    auto *go = reinterpret_cast<unsigned *>(S.h + o_go);
    for (uint32_t f = 0; f < n_frames; ++f) {
      to[f] = S.job_of_frame[f] * cvk::kSamplerCap;
      go[f] = S.job_of_frame[f] * (cvk::kInterpGrid + 1u);
    }
  }
  std::memcpy(S.h + o_l, S.l_job.data(), sizeof(double) * n_jobs);
  HIP_TRY(ctx, hipMemcpyAsync(S.d, S.h, staged, hipMemcpyHostToDevice, stream));
  HIP_TRY(ctx, hipMemsetAsync(S.d + o_sk, 0xFF, sizeof(unsigned long long) * SS, stream)); /* every key = kSpecEmpty */
  SamplerParams SP;
  SP.metric = MP;
  SP.l_cam = (const double *)(S.d + o_l);
  SP.n_jobs = n_jobs;
  SP.n_frames = n_frames;
  SP.job_of_frame = (const unsigned *)(S.d + o_jf);
  SP.tab_n = (unsigned *)(S.d + S.o_tab_n);
  SP.n0 = alpha_nums;
  SP.max_iterations = max_iterations_sampling;
  SP.max_iter = max_iter;
  SP.a_min = -0.1 * CV_PI; /* src/systems.rs:437-438 */
  SP.a_max = 1.1 * CV_PI;
  SP.thr1 = thr1;
  SP.thr2 = thr2;
  SP.max_radius = max_radius;
  SP.delta = delta;
  SP.fast_ok = cvk::metric_fast_ok(metric->kind, MP, max_radius) ? 1 : 0;
  SP.sx = (double *)(S.d + S.o_tab[0]);
  SP.se = (double *)(S.d + S.o_tab[1]);
  SP.ss = (double *)(S.d + S.o_tab[2]);
  SP.m_e = (double *)(S.d + S.o_tab[3]);
  SP.c_e = (double *)(S.d + S.o_tab[4]);
  SP.m_s = (double *)(S.d + S.o_tab[5]);
  SP.c_s = (double *)(S.d + S.o_tab[6]);
  SP.grid = (unsigned *)(S.d + S.o_grid);
  SP.res = (cvk::SamplerResult *)(S.d + S.o_res);
  SP.spec_key = (unsigned long long *)(S.d + o_sk);
  SP.spec_e = (double *)(S.d + o_se);
  SP.spec_s = (double *)(S.d + o_ss);
  SP.spec_steps = (unsigned *)(S.d + o_st);
  SP.spec_status = (int *)(S.d + o_su);
  SP.speculate = ctx->sampling_speculation != 0 ? 1 : 0; /* option "sampling_speculation" = 0 switches it off here too */
  const bool fast = ctx->fast_math != 0;
  HIP_TRY(ctx, hipEventRecord(S.t0, stream));
  switch (metric->kind) {
    case CURVIS_METRIC_ELLIS: rc = launch_sampler_kind<cvk::METRIC_ELLIS>(ctx, stream, fast, SP); break;
    case CURVIS_METRIC_INTERSTELLAR: rc = launch_sampler_kind<cvk::METRIC_INTERSTELLAR>(ctx, stream, fast, SP); break;
    default: rc = launch_sampler_kind<cvk::METRIC_FLAT>(ctx, stream, fast, SP); break;
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(S.t1, stream));
  HIP_TRY(ctx, hipMemcpyAsync(S.h + S.h_res_off, S.d + S.o_res, S.res_bytes, hipMemcpyDeviceToHost, stream));
  HIP_TRY(ctx, hipEventRecord(S.done, stream));
  S.metric = *metric;
  S.n_frames = n_frames;
  S.max_iter = max_iter;
  S.alpha_nums = alpha_nums;
  S.max_iterations_sampling = max_iterations_sampling;
  S.params[0] = max_radius, S.params[1] = delta, S.params[2] = thr1, S.params[3] = thr2;
  S.fast = fast ? 1 : 0;
  S.speculate = SP.speculate;
  S.seq = ++ctx->samp_seq;
  S.valid = true;
  return CURVIS_OK;
}

/* curvis_ctx_prefetch_efficient: the sampler of a FUTURE curvis_render_efficient_batch call, launched now on a stream of its own.
 * The sampler's cost is latency (a handful of Euler chains on a few compute units), the per-pixel kernel's and the PNG front end's
 * is throughput, and between them a render call leaves the GPU to the host (stream download, hand-over): the next call's sampler
 * fits into all of that.  The call with the same metric, settings and camera radii then waits for the event instead of sampling. */
int prefetch_efficient_impl(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cams, uint32_t n_frames, uint32_t max_iter,
                            double max_radius, double delta, uint32_t alpha_nums, uint32_t max_iterations_sampling, double thr1, double thr2) {
  if (!ctx) return CURVIS_E_INVALID;
  if (!metric || !cams || n_frames == 0) return fail(ctx, CURVIS_E_INVALID, "null metric/camera or zero frames");
  int rc = curvis_metric_validate(metric);
  if (rc != CURVIS_OK) return fail(ctx, rc, "invalid metric parameters (src/metrics.rs:409-456)");
  if (alpha_nums < 3 || alpha_nums > cvk::kSamplerCap || alpha_nums > cvk::kSamplerPendCap) return CURVIS_OK; /* not a case for the device sampler */
  if (!(ctx->device_sampler > 0 || (ctx->device_sampler < 0 && n_frames >= (uint32_t)ctx->device_sampler_min_frames)))
    return CURVIS_OK; /* the render call will take the host-paced sampler: nothing to run ahead */
  for (uint32_t f = 0; f < n_frames; ++f)
    if (std::fabs(cams[f].pos[1]) > max_radius) return CURVIS_OK; /* the render call will report it */
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->sampler_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->sampler_stream, hipStreamNonBlocking));
  const cvk::MetricParams MP = make_metric(*metric);
  const unsigned slot = ctx->samp_next;
  rc = sampler_submit(ctx, slot, ctx->sampler_stream, metric, MP, cams, n_frames, max_iter, max_radius, delta, alpha_nums, max_iterations_sampling,
                      thr1, thr2);
  if (rc) return rc;
  ctx->samp[slot].prefetched = true;
  ctx->samp_next = slot ^ 1u;
  ctx->prefetches++;
  return CURVIS_OK;
}

int render_efficient_device(curvis_ctx *ctx, const curvis_metric *metric, const cvk::MetricParams &MP, const curvis_camera *cams,
                            uint32_t n_frames, const std::vector<cvk::EfficientFrame> &eframes, uint32_t max_iter, double max_radius,
                            double delta, uint32_t alpha_nums, uint32_t max_iterations_sampling, double thr1, double thr2,
                            uint8_t *rgb_out, curvis_stats *stats, std::chrono::steady_clock::time_point t_begin) {
  const uint32_t W = cams[0].res_x, H = cams[0].res_y;
  const size_t npix = (size_t)W * H;
  if (npix > 0xFFFFFFFFull || n_frames > 65535u) return fail(ctx, CURVIS_E_INVALID, "frame or batch too large");
  /* the tables: prefetched by curvis_ctx_prefetch_efficient (either slot may hold them), or sampled now on this call's stream */
  const int fast_i = ctx->fast_math != 0 ? 1 : 0, spec_i = ctx->sampling_speculation != 0 ? 1 : 0;
  int slot = -1;
  for (unsigned k = 0; k < 2u; ++k) /* both may match (every batch of an orbit has the same radii): the one submitted FIRST is the finished one */
    if (ctx->samp[k].prefetched && sampler_key_equal(ctx->samp[k], *metric, cams, n_frames, max_iter, max_radius, delta, alpha_nums,
                                                     max_iterations_sampling, thr1, thr2, fast_i, spec_i) &&
        (slot < 0 || ctx->samp[k].seq < ctx->samp[slot].seq))
      slot = (int)k;
  const bool prefetched = slot >= 0;
  int rc;
  if (!prefetched) {
    slot = (int)ctx->samp_next;
    rc = sampler_submit(ctx, (unsigned)slot, ctx->stream, metric, MP, cams, n_frames, max_iter, max_radius, delta, alpha_nums,
                        max_iterations_sampling, thr1, thr2);
    if (rc) return rc;
    ctx->samp[slot].prefetched = false;
    ctx->samp_next = (unsigned)slot ^ 1u;
  } else {
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->samp[slot].done, 0));
    ctx->prefetch_hits++;
  }
  curvis_ctx::SamplerSlot &S = ctx->samp[slot];
  S.prefetched = false; /* consumed (the tables stay readable until the slot is submitted to again) */
  const unsigned n_jobs = (unsigned)S.l_job.size();
  /* this call's own staging: cameras and per-frame constants */
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_cams = carve(sizeof(cvk::CameraParams) * n_frames), o_fr = carve(sizeof(cvk::EfficientFrame) * n_frames);
  rc = ensure_device(ctx, ctx->d_eff, ctx->eff_cap, off);
  if (rc) return rc;
  if (ctx->h_eff_cap < off) {
    if (ctx->h_eff) HIP_TRY(ctx, hipHostFree(ctx->h_eff));
    ctx->h_eff = nullptr;
    ctx->h_eff_cap = 0;
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_eff, off + off / 2));
    ctx->h_eff_cap = off + off / 2;
  }
  unsigned char *stage = ctx->h_eff;
  {
    auto *cp = reinterpret_cast<cvk::CameraParams *>(stage + o_cams);
    for (uint32_t f = 0; f < n_frames; ++f) cp[f] = make_camera(cams[f]);
    std::memcpy(stage + o_fr, eframes.data(), sizeof(cvk::EfficientFrame) * n_frames);
  }
  const size_t fb_bytes = npix * 3 * n_frames;
  rc = fb_begin_write(ctx, fb_bytes);
  if (rc) return rc;
  ctx->fb_bytes = fb_bytes;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eff, stage, off, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  FrameCounters FC;
  rc = prepare_counters(ctx, n_frames, FC, 64u);
  if (rc) return rc;
  const size_t cnt_words = counter_words(n_frames, FC.slots);
  EfficientPixelParams Q;
  for (int k = 0; k < 2; ++k) {
    Q.sky[k].texels = (const unsigned *)ctx->d_sky[k];
    Q.sky[k].w = ctx->sky_w[k];
    Q.sky[k].h = ctx->sky_h[k];
    for (int i = 0; i < 9; ++i) Q.sky[k].inv_rot[i] = ctx->sky_inv_rot[k][i];
  }
  Q.cams = (const cvk::CameraParams *)(ctx->d_eff + o_cams);
  Q.frames = (const cvk::EfficientFrame *)(ctx->d_eff + o_fr);
  Q.tab_off = (const unsigned *)(S.d + S.o_tab_off);
  Q.tab_n = (const unsigned *)(S.d + S.o_tab_n);
  Q.grid_off = (const unsigned *)(S.d + S.o_grid_off);
  Q.grid = (const unsigned *)(S.d + S.o_grid);
  Q.sx = (const double *)(S.d + S.o_tab[0]);
  Q.m_e = (const double *)(S.d + S.o_tab[3]);
  Q.c_e = (const double *)(S.d + S.o_tab[4]);
  Q.m_s = (const double *)(S.d + S.o_tab[5]);
  Q.c_s = (const double *)(S.d + S.o_tab[6]);
  Q.n_frames = n_frames;
  Q.W = W;
  Q.H = H;
  Q.fb = ctx->d_fb;
  Q.counters = FC;
  rc = ensure_pixel_recips(ctx, (double)W, (double)H, Q.recips);
  if (rc) return rc;
  Q.w_magic = W > 1u ? ~0ull / W + 1ull : 0ull; /* floor((2^64 - 1) / W) + 1 = floor(2^64 / W) + 1 unless W divides 2^64, where it is 2^64 / W: exact too */
  hipLaunchKernelGGL(efficient_pixel_kernel, dim3((unsigned)((npix + 255) / 256), n_frames), dim3(256), 0, ctx->stream, Q);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev2, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(unsigned long long) * cnt_words, hipMemcpyDeviceToHost, ctx->stream));
  const auto t_launched = std::chrono::steady_clock::now();
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); /* (behind the sampler's event: its results are in S.h) */
  if (getenv("CURVIS_DEBUG_TIMING")) {
    float k_ms = 0.f;
    (void)hipEventElapsedTime(&k_ms, ctx->ev1, ctx->ev2);
    const auto t_now = std::chrono::steady_clock::now();
    fprintf(stderr, "[curvis] efficient call, %u frames (ms): host before the sync %.3f, in the sync %.3f (staging copy -> end of the per-pixel kernel: %.3f)\n",
            n_frames, std::chrono::duration<double, std::milli>(t_launched - t_begin).count(),
            std::chrono::duration<double, std::milli>(t_now - t_launched).count(), k_ms);
  }
  auto *h_res = reinterpret_cast<const cvk::SamplerResult *>(S.h + S.h_res_off);
  bool overflow = false, panic = false;
  for (unsigned j = 0; j < n_jobs; ++j) {
    overflow = overflow || h_res[j].status == cvk::SAMPLER_OVERFLOW;
    panic = panic || h_res[j].status == cvk::SAMPLER_PANIC;
  }
  if (getenv("CURVIS_DEBUG_TIMING"))
    for (unsigned j = 0; j < n_jobs; ++j)
      fprintf(stderr, "[curvis] device sampler job %u: l = %.17g -> %u samples, %u rounds, %llu calls, %llu steps, warned %d, status %d; "
              "%u Euler chains, %u points integrated%s\n", j, S.l_job[j], h_res[j].n, h_res[j].rounds, (unsigned long long)h_res[j].calls,
              (unsigned long long)h_res[j].steps, h_res[j].warned, h_res[j].status, h_res[j].eval_phases, h_res[j].evaluated,
              prefetched ? " (prefetched)" : "");
  if (overflow) return kSamplerFallback;
  float sample_ms = 0.f, ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&sample_ms, S.t0, S.t1)); /* the sampler kernel, wherever and whenever it ran */
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev1, ctx->ev2));
  ctx->last_sampling_launches = 1;
  ctx->last_sampling_prefetched = prefetched ? 1 : 0;
  uint64_t total_steps = 0, evaluated = 0;
  ctx->last_samples.assign(n_frames, {});
  ctx->last_sampling_info.assign(n_frames, curvis_sampling_info{});
  for (uint32_t f = 0; f < n_frames; ++f) {
    const cvk::SamplerResult &r = h_res[S.job_of_frame[f]];
    curvis_sampling_info &si = ctx->last_sampling_info[f];
    si.n_samples = r.n;
    si.rounds = r.rounds;
    si.calls = r.calls; /* what the reference's sampler of THIS frame calls and steps, whether or not frames shared the work */
    si.steps = r.steps;
    si.warned_max_iterations = r.warned;
    total_steps += r.steps;
  }
  uint32_t chains = 0;
  for (unsigned j = 0; j < n_jobs; ++j) {
    evaluated += h_res[j].evaluated;
    chains = std::max(chains, h_res[j].eval_phases);
  }
  ctx->last_sampling_evaluated = evaluated;
  ctx->last_sampling_chains = chains; /* Euler chains the slowest job waited for: what the launch's latency is made of */
  ctx->dev_samples.valid = !panic;
  ctx->dev_samples.overwritten = false;
  ctx->dev_samples.slot = (unsigned)slot;
  if (panic)
    return fail(ctx, CURVIS_E_SAMPLING,
                "sampler panic: fewer than 3 finite samples (src/sampling.rs:155-157) or undefined tangent rotation "
                "(src/algebra.rs:95-97)");
  if (rgb_out) {
    rc = fb_download(ctx, rgb_out, fb_bytes);
    if (rc) return rc;
  }
  uint64_t tot[FC_N] = {0};
  ctx->last_frame_stats.assign(n_frames, curvis_stats{});
  for (uint32_t f = 0; f < n_frames; ++f) {
    uint64_t fc[FC_N];
    sum_frame_counters(ctx->h_counters, FC.slots, f, fc);
    for (int k = 0; k < FC_N; ++k) tot[k] += fc[k];
    curvis_stats &fs = ctx->last_frame_stats[f];
    fs.rays = (uint64_t)npix;
    fs.steps = ctx->last_sampling_info[f].steps;
    fs.n_pos = fc[FC_POS];
    fs.n_neg = fc[FC_NEG];
    fs.n_none = fc[FC_NONE];
    fs.n_oob = fc[FC_OOB];
    fs.integrate_ms = sample_ms / n_frames;
    fs.shade_ms = ms / n_frames;
    fs.kernel_ms = fs.integrate_ms + fs.shade_ms;
    fs.total_ms = fs.kernel_ms;
  }
  if (stats) {
    stats->rays = (uint64_t)npix * n_frames;
    stats->steps = total_steps;
    stats->n_pos = tot[FC_POS];
    stats->n_neg = tot[FC_NEG];
    stats->n_none = tot[FC_NONE];
    stats->n_oob = tot[FC_OOB];
    stats->integrate_ms = sample_ms;
    stats->shade_ms = ms;
    stats->kernel_ms = sample_ms + ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return CURVIS_OK;
}

/* the sample table of frame `frame` of the last render_efficient call that used the device-resident sampler: fetched from the
 * context's scratch on demand (curvis_ctx_samples), once per frame asked for */
int fetch_device_samples(curvis_ctx *ctx, uint32_t frame) {
  if (!ctx->dev_samples.valid) return CURVIS_OK;
  const curvis_ctx::SamplerSlot &S = ctx->samp[ctx->dev_samples.slot];
  if (frame >= S.job_of_frame.size() || frame >= ctx->last_samples.size()) return CURVIS_OK;
  if (!ctx->last_samples[frame].empty() || ctx->last_sampling_info[frame].n_samples == 0) return CURVIS_OK;
  if (ctx->dev_samples.overwritten)
    return fail(ctx, CURVIS_E_INVALID, "the sample tables of that render call are gone: a later curvis_ctx_prefetch_efficient has taken their slot "
                                       "(ask for them before the second prefetch after the call)");
  const size_t n = ctx->last_sampling_info[frame].n_samples, o = (size_t)S.job_of_frame[frame] * cvk::kSamplerCap * sizeof(double);
  std::vector<double> a(n), e(n), s(n);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpy(a.data(), S.d + S.o_tab[0] + o, n * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(e.data(), S.d + S.o_tab[1] + o, n * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(s.data(), S.d + S.o_tab[2] + o, n * sizeof(double), hipMemcpyDeviceToHost));
  auto &pts = ctx->last_samples[frame];
  pts.resize(n);
  for (size_t i = 0; i < n; ++i) pts[i] = cvs::BiPoint{a[i], e[i], s[i]};
  return CURVIS_OK;
}

int render_efficient_impl(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cams, uint32_t n_frames,
                          uint32_t max_iter, double max_radius, double delta, uint32_t alpha_nums,
                          uint32_t max_iterations_sampling, double thr1, double thr2, uint8_t *rgb_out,
                          curvis_stats *stats) {
  if (!ctx) return CURVIS_E_INVALID;
  if (!metric || !cams || n_frames == 0) return fail(ctx, CURVIS_E_INVALID, "null metric/camera or zero frames");
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = curvis_metric_validate(metric);
  if (rc != CURVIS_OK) return fail(ctx, rc, "invalid metric parameters (src/metrics.rs:409-456)");
  const uint32_t W = cams[0].res_x, H = cams[0].res_y;
  if (W == 0 || H == 0) return fail(ctx, CURVIS_E_INVALID, "resolution must be greater than 0 (src/cameras.rs:98)");
  if (alpha_nums < 3) return fail(ctx, CURVIS_E_SAMPLING, "alpha_nums < 3: the sampler panics (src/sampling.rs:155-157)");
  for (uint32_t f = 0; f < n_frames; ++f) {
    if (cams[f].res_x != W || cams[f].res_y != H)
      return fail(ctx, CURVIS_E_INVALID, "all cameras of a batch must share one resolution");
    if (std::fabs(cams[f].pos[1]) > max_radius)
      return fail(ctx, CURVIS_E_CAMERA_OUTSIDE,
                  "Photon already beyond the maximum radius. Cannot evaluate escape. (src/systems.rs:122-124)");
  }
  if (!ctx->d_sky[0] || !ctx->d_sky[1]) return fail(ctx, CURVIS_E_NO_SKY, "both background images must be set");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const cvk::MetricParams MP = make_metric(*metric);

  /* step 1 (host): camera direction on the background space and the tangent->background rotation */
  std::vector<cvk::EfficientFrame> eframes(n_frames);
  for (uint32_t f = 0; f < n_frames; ++f) {
    if (!cvk::efficient_frame_pose(cams[f].pos[2], cams[f].pos[3], eframes[f])) /* platform libm: cv_frame_host.h */
      return fail(ctx, CURVIS_E_PARALLEL, "v1 and v2 must not be parallel (src/algebra.rs:95-97, camera on the x axis)");
  }

  ctx->dev_samples.valid = false;
  /* step 3 on the device (sampler_kernel: no host in the refinement loop) for calls of `device_sampler_min_frames` frames and
   * more -- its latency is rounds x one Euler chain, about three times the speculating host-paced sampler's below, and what it
   * saves is host time and launches per frame, so single images and small batches stay on the host-paced path (cross-over measured
   * between 32 and 64 frames per call); option "device_sampler": 1 always, 0 never, -1 (default) by that threshold */
  const bool want_device = ctx->device_sampler > 0 || (ctx->device_sampler < 0 && n_frames >= (uint32_t)ctx->device_sampler_min_frames);
  if (want_device && alpha_nums <= cvk::kSamplerCap && alpha_nums <= cvk::kSamplerPendCap) {
    rc = render_efficient_device(ctx, metric, MP, cams, n_frames, eframes, max_iter, max_radius, delta, alpha_nums, max_iterations_sampling,
                                 thr1, thr2, rgb_out, stats, t_begin);
    ctx->last_sampler_path = rc == kSamplerFallback ? 2 : 1;
    if (rc != kSamplerFallback) return rc;
    ctx->dev_samples.valid = false; /* a table outgrew the kernel's arrays: the host-paced sampler takes the call */
  } else {
    ctx->last_sampler_path = 0;
  }
  /* step 3 on the host: one sampler per frame, advanced in lock step; every round is ONE kernel launch */
  std::vector<cvs::Sampler> smp(n_frames);
  for (uint32_t f = 0; f < n_frames; ++f) {
    smp[f].a_min = -0.1 * CV_PI; /* src/systems.rs:437-438 */
    smp[f].a_max = 1.1 * CV_PI;
    smp[f].n0 = alpha_nums;
    smp[f].max_iterations = max_iterations_sampling;
    smp[f].thr1 = thr1;
    smp[f].thr2 = thr2;
  }
  /* Evaluation cache + speculation.  Every point the sampler will ever ask for is the midpoint of two
   * samples that are adjacent at that time, i.e. a node of the dyadic tree below an interval of the current
   * table, computed by the same (lo + hi) / 2.0.  So whenever some requested alpha is not cached yet, the
   * launch also evaluates the whole subtree of depth `spec` below the interval it comes from (and, on the
   * first launch, below every interval of the uniform grid): the GPU is idle anyway -- a round is a single
   * wave's 2000-step dependency chain -- and the following rounds are then served from the cache without
   * a launch.  The sampler consumes exactly the values the sequential algorithm would compute; calls and
   * steps are counted at consumption, so the bookkeeping equals the reference's. */
  /* open-addressing table keyed by the bit pattern of alpha; state 0 = empty, 1 = queued for the next launch,
   * 2 = evaluated (a node-based std::unordered_map cost more host time per batch than the kernels) */
  struct Cached {
    uint64_t key;
    double e, s;
    uint32_t steps;
    int status;
    uint32_t state;
  };
  struct EvalCache {
    std::vector<Cached> slots;
    size_t used = 0;
    explicit EvalCache(size_t capacity = 4096) : slots(capacity, Cached{0, 0.0, 0.0, 0, 0, 0}) {}
    static size_t hash(uint64_t k) { return (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20); }
    Cached *find(uint64_t k) { /* the slot holding k, or the empty slot where it would go */
      const size_t mask = slots.size() - 1;
      size_t i = hash(k) & mask;
      while (slots[i].state != 0 && slots[i].key != k) i = (i + 1) & mask;
      return &slots[i];
    }
    Cached *claim(uint64_t k) { /* find, inserting an empty (state 0) entry for a new key */
      if (2 * (used + 1) > slots.size()) {
        std::vector<Cached> old;
        old.swap(slots);
        slots.assign(old.size() * 2, Cached{0, 0.0, 0.0, 0, 0, 0});
        for (const Cached &c : old)
          if (c.state != 0) *find(c.key) = c;
      }
      Cached *c = find(k);
      if (c->state == 0) c->key = k;
      return c;
    }
  };
  auto key_of = [](double a) {
    uint64_t u;
    std::memcpy(&u, &a, sizeof u);
    return u;
  };
  /* automatic depths: about 30-50 k points per launch (tools/gpu_eff_two_launch.py, tools/gpu_eff_batch_spec.py) */
  const int spec = ctx->sampling_speculation < 0 ? (n_frames <= 2 ? 10 : n_frames <= 5 ? 6 : 4)
                                                 : (ctx->sampling_speculation > 11 ? 11 : ctx->sampling_speculation);
  /* depth of the subtrees evaluated below the intervals of the initial uniform grid (first launch) */
  const int first_cap = ctx->sampling_speculation_first < 0 ? (n_frames <= 2 ? 8 : n_frames <= 5 ? 4 : 3)
                                                            : (ctx->sampling_speculation_first > 11 ? 11 : ctx->sampling_speculation_first);
  /* sized for the first launch (grid x subtree) plus as much again, so that the table is not rebuilt four times on
   * the way up from a small default (a quarter of the host time of a single image) */
  size_t cache_cap = 4096;
  {
    const size_t first = (size_t)alpha_nums << (spec > 0 ? (spec > first_cap ? first_cap : spec) : 0);
    while (cache_cap < 4 * first && cache_cap < ((size_t)1 << 22)) cache_cap *= 2;
  }
  std::vector<EvalCache> cache;
  cache.reserve(n_frames);
  for (uint32_t f = 0; f < n_frames; ++f) cache.emplace_back(cache_cap);
  std::vector<char> planned(n_frames, 0);
  double sample_ms = 0.0;
  uint64_t evaluated = 0;
  uint32_t launches = 0;
  std::vector<double> b_alpha, b_l, r_angle, r_space, ce, cs;
  std::vector<uint32_t> r_steps, cst;
  std::vector<int> r_status;
  std::vector<uint32_t> b_frame;
  bool panic = false;
  const bool dbg_timing = getenv("CURVIS_DEBUG_TIMING") != nullptr;
  double t_adv = 0.0, t_build = 0.0, t_eval = 0.0, t_ins = 0.0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  for (;;) {
    const auto tp0 = now();
    /* advance every sampler as far as the cache allows */
    bool any_waiting = false;
    for (uint32_t f = 0; f < n_frames; ++f) {
      for (;;) {
        if (!planned[f]) {
          if (!smp[f].plan()) break; /* finished */
          planned[f] = 1;
        }
        bool all_cached = true;
        for (double a : smp[f].pending)
          if (cache[f].find(key_of(a))->state != 2) {
            all_cached = false;
            break;
          }
        if (!all_cached) {
          any_waiting = true;
          break;
        }
        const size_t n = smp[f].pending.size();
        ce.resize(n);
        cs.resize(n);
        cst.resize(n);
        for (size_t k = 0; k < n; ++k) {
          const Cached &c = *cache[f].find(key_of(smp[f].pending[k]));
          ce[k] = c.e;
          cs[k] = c.s;
          cst[k] = c.steps;
          if (c.status == cvk::ESC_PANIC) panic = true;
        }
        smp[f].consume(ce.data(), cs.data(), cst.data());
        planned[f] = 0;
      }
    }
    const auto tp1 = now();
    t_adv += secs(tp0, tp1);
    if (!any_waiting) break;
    /* one launch: the missing points of every waiting frame plus their speculative subtrees */
    b_alpha.clear();
    b_l.clear();
    b_frame.clear();
    for (uint32_t f = 0; f < n_frames; ++f) {
      if (!planned[f]) continue;
      auto want = [&](double a) {
        Cached *c = cache[f].claim(key_of(a));
        if (c->state != 0) return; /* evaluated, or already queued for this launch */
        c->state = 1;
        cache[f].used++;
        b_alpha.push_back(a);
        b_l.push_back(cams[f].pos[1]);
        b_frame.push_back(f);
      };
      struct Node {
        double lo, hi;
        int depth;
      };
      std::vector<Node> stack;
      const cvs::Sampler &S = smp[f];
      for (size_t k = 0; k < S.pending.size(); ++k) {
        want(S.pending[k]);
        if (spec <= 0) continue;
        if (S.pend_lo[k] == S.pend_lo[k]) {
          stack.push_back(Node{S.pend_lo[k], S.pend_hi[k], spec});
        } else if (k + 1 < S.pending.size()) { /* uniform grid: subtree below [x_k, x_{k+1}] */
          stack.push_back(Node{S.pending[k], S.pending[k + 1], spec > first_cap ? first_cap : spec});
        }
        while (!stack.empty()) {
          const Node nd = stack.back();
          stack.pop_back();
          const double mid = (nd.lo + nd.hi) / 2.0;
          if (!(mid > nd.lo && mid < nd.hi)) continue; /* interval exhausted in double precision */
          want(mid);
          if (nd.depth > 1) {
            stack.push_back(Node{nd.lo, mid, nd.depth - 1});
            stack.push_back(Node{mid, nd.hi, nd.depth - 1});
          }
        }
      }
    }
    const auto tp2 = now();
    t_build += secs(tp1, tp2);
    rc = eval_escape_batch(ctx, metric, MP, b_alpha, b_l, max_iter, max_radius, delta, r_angle, r_space, r_steps,
                           r_status, &sample_ms);
    if (rc) return rc;
    const auto tp3 = now();
    t_eval += secs(tp2, tp3);
    ++launches;
    evaluated += b_alpha.size();
    for (size_t k = 0; k < b_alpha.size(); ++k) {
      Cached *c = cache[b_frame[k]].find(key_of(b_alpha[k]));
      c->e = r_angle[k];
      c->s = r_space[k];
      c->steps = r_steps[k];
      c->status = r_status[k];
      c->state = 2;
    }
    t_ins += secs(tp3, now());
  }
  if (dbg_timing)
    fprintf(stderr, "[curvis] sampling host phases (ms): advance %.3f, build %.3f, evaluate (copies+kernel+sync) %.3f of which kernels %.3f, cache insert %.3f; launches %u, points %llu\n",
            t_adv, t_build, t_eval, sample_ms, t_ins, launches, (unsigned long long)evaluated);
  ctx->last_sampling_launches = launches;
  ctx->last_sampling_evaluated = evaluated;
  ctx->last_samples.assign(n_frames, {});
  ctx->last_sampling_info.assign(n_frames, curvis_sampling_info{});
  uint64_t total_steps = 0;
  for (uint32_t f = 0; f < n_frames; ++f) {
    if (smp[f].panicked) panic = true;
    ctx->last_samples[f] = smp[f].pts;
    curvis_sampling_info &si = ctx->last_sampling_info[f];
    si.n_samples = (uint32_t)smp[f].pts.size();
    si.rounds = smp[f].rounds;
    si.calls = smp[f].calls;
    si.steps = smp[f].steps;
    si.warned_max_iterations = smp[f].warned ? 1 : 0;
    total_steps += smp[f].steps;
  }
  if (panic)
    return fail(ctx, CURVIS_E_SAMPLING,
                "sampler panic: fewer than 3 finite samples (src/sampling.rs:155-157) or undefined tangent rotation "
                "(src/algebra.rs:95-97)");

  /* step 4 tables (interp 1.0.3) */
  std::vector<double> sx, m_e, c_e, m_s, c_s, x, ye, ys, m, c;
  std::vector<unsigned> tab_off(n_frames), tab_n(n_frames), grid_off(n_frames), grid; /* grid: cv_efficient.h interp_index_grid */
  for (uint32_t f = 0; f < n_frames; ++f) {
    const auto &pts = smp[f].pts;
    x.clear();
    ye.clear();
    ys.clear();
    for (const auto &b : pts) {
      x.push_back(b.a);
      ye.push_back(b.e);
      ys.push_back(b.s);
    }
    tab_off[f] = (unsigned)sx.size();
    tab_n[f] = (unsigned)pts.size();
    const size_t slots = std::max<size_t>(pts.size(), 1);
    cvs::interp_tables(x, ye, m, c);
    m.resize(slots, 0.0);
    c.resize(slots, 0.0);
    m_e.insert(m_e.end(), m.begin(), m.end());
    c_e.insert(c_e.end(), c.begin(), c.end());
    cvs::interp_tables(x, ys, m, c);
    m.resize(slots, 0.0);
    c.resize(slots, 0.0);
    m_s.insert(m_s.end(), m.begin(), m.end());
    c_s.insert(c_s.end(), c.begin(), c.end());
    grid_off[f] = (unsigned)grid.size();
    grid.resize(grid.size() + cvk::kInterpGrid + 1u, 0u);
    for (unsigned i = 0; i <= (unsigned)pts.size(); ++i) cvk::interp_grid_fill(x.data(), (unsigned)pts.size(), i, grid.data() + grid_off[f]);
    x.resize(slots, 0.0);
    sx.insert(sx.end(), x.begin(), x.end());
  }

  /* device buffers for K3 */
  const size_t npix = (size_t)W * H;
  if (npix > 0xFFFFFFFFull || n_frames > 65535u) return fail(ctx, CURVIS_E_INVALID, "frame or batch too large");
  const size_t fb_bytes = npix * 3 * n_frames;
  rc = fb_begin_write(ctx, fb_bytes);
  if (rc) return rc;
  ctx->fb_bytes = fb_bytes;
  const size_t T = sx.size();
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 15) & ~(size_t)15;
    return o;
  };
  const size_t o_cams = carve(sizeof(cvk::CameraParams) * n_frames), o_fr = carve(sizeof(cvk::EfficientFrame) * n_frames),
               o_to = carve(sizeof(unsigned) * n_frames), o_tn = carve(sizeof(unsigned) * n_frames),
               o_go = carve(sizeof(unsigned) * n_frames), o_gr = carve(sizeof(unsigned) * grid.size()),
               o_sx = carve(sizeof(double) * T), o_me = carve(sizeof(double) * T), o_ce = carve(sizeof(double) * T),
               o_ms = carve(sizeof(double) * T), o_cs = carve(sizeof(double) * T);
  rc = ensure_device(ctx, ctx->d_eff, ctx->eff_cap, off);
  if (rc) return rc;
  std::vector<unsigned char> stage(off);
  std::vector<cvk::CameraParams> cp(n_frames);
  for (uint32_t f = 0; f < n_frames; ++f) cp[f] = make_camera(cams[f]);
  std::memcpy(stage.data() + o_cams, cp.data(), sizeof(cvk::CameraParams) * n_frames);
  std::memcpy(stage.data() + o_fr, eframes.data(), sizeof(cvk::EfficientFrame) * n_frames);
  std::memcpy(stage.data() + o_to, tab_off.data(), sizeof(unsigned) * n_frames);
  std::memcpy(stage.data() + o_tn, tab_n.data(), sizeof(unsigned) * n_frames);
  std::memcpy(stage.data() + o_go, grid_off.data(), sizeof(unsigned) * n_frames);
  std::memcpy(stage.data() + o_gr, grid.data(), sizeof(unsigned) * grid.size());
  std::memcpy(stage.data() + o_sx, sx.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_me, m_e.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_ce, c_e.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_ms, m_s.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_cs, c_s.data(), sizeof(double) * T);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eff, stage.data(), off, hipMemcpyHostToDevice, ctx->stream));
  FrameCounters FC;
  rc = prepare_counters(ctx, n_frames, FC, 64u); /* one workgroup in 256 pixels adds to them: spread over 64 lines per frame */
  if (rc) return rc;
  const size_t cnt_words = counter_words(n_frames, FC.slots);
  EfficientPixelParams Q;
  for (int k = 0; k < 2; ++k) {
    Q.sky[k].texels = (const unsigned *)ctx->d_sky[k];
    Q.sky[k].w = ctx->sky_w[k];
    Q.sky[k].h = ctx->sky_h[k];
    for (int i = 0; i < 9; ++i) Q.sky[k].inv_rot[i] = ctx->sky_inv_rot[k][i];
  }
  Q.cams = (const cvk::CameraParams *)(ctx->d_eff + o_cams);
  Q.frames = (const cvk::EfficientFrame *)(ctx->d_eff + o_fr);
  Q.tab_off = (const unsigned *)(ctx->d_eff + o_to);
  Q.tab_n = (const unsigned *)(ctx->d_eff + o_tn);
  Q.grid_off = (const unsigned *)(ctx->d_eff + o_go);
  Q.grid = (const unsigned *)(ctx->d_eff + o_gr);
  Q.sx = (const double *)(ctx->d_eff + o_sx);
  Q.m_e = (const double *)(ctx->d_eff + o_me);
  Q.c_e = (const double *)(ctx->d_eff + o_ce);
  Q.m_s = (const double *)(ctx->d_eff + o_ms);
  Q.c_s = (const double *)(ctx->d_eff + o_cs);
  Q.n_frames = n_frames;
  Q.W = W;
  Q.H = H;
  Q.fb = ctx->d_fb;
  Q.counters = FC;
  rc = ensure_pixel_recips(ctx, (double)W, (double)H, Q.recips);
  if (rc) return rc;
  Q.w_magic = W > 1u ? ~0ull / W + 1ull : 0ull; /* floor((2^64 - 1) / W) + 1 = floor(2^64 / W) + 1 unless W divides 2^64, where it is 2^64 / W: exact too */
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(efficient_pixel_kernel, dim3((unsigned)((npix + 255) / 256), n_frames), dim3(256), 0, ctx->stream, Q);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(unsigned long long) * cnt_words,
                              hipMemcpyDeviceToHost, ctx->stream));
  if (rgb_out) {
    rc = fb_download(ctx, rgb_out, fb_bytes); /* leaves the stream idle; option "async_download": the frames follow */
    if (rc) return rc;
  } else {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  float ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  uint64_t tot[FC_N] = {0};
  ctx->last_frame_stats.assign(n_frames, curvis_stats{});
  for (uint32_t f = 0; f < n_frames; ++f) {
    uint64_t fc[FC_N];
    sum_frame_counters(ctx->h_counters, FC.slots, f, fc);
    for (int k = 0; k < FC_N; ++k) tot[k] += fc[k];
    curvis_stats &fs = ctx->last_frame_stats[f];
    fs.rays = (uint64_t)npix; /* pixels; the integrator calls of the frame's sampler are in curvis_ctx_sampling_info */
    fs.steps = smp[f].steps;
    fs.n_pos = fc[FC_POS];
    fs.n_neg = fc[FC_NEG];
    fs.n_none = fc[FC_NONE];
    fs.n_oob = fc[FC_OOB];
    /* the samplers of a batch share their launches: times are the batch's, shared out evenly */
    fs.integrate_ms = sample_ms / n_frames;
    fs.shade_ms = ms / n_frames;
    fs.kernel_ms = fs.integrate_ms + fs.shade_ms;
    fs.total_ms = fs.kernel_ms;
  }
  if (stats) {
    stats->rays = (uint64_t)npix * n_frames;
    stats->steps = total_steps;
    stats->n_pos = tot[FC_POS];
    stats->n_neg = tot[FC_NEG];
    stats->n_none = tot[FC_NONE];
    stats->n_oob = tot[FC_OOB];
    stats->integrate_ms = sample_ms;
    stats->shade_ms = ms;
    stats->kernel_ms = sample_ms + ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return CURVIS_OK;
}

template <int KIND>
int launch_direct_kind(curvis_ctx *ctx, bool fast, const DirectParams &P) {
  const unsigned blocks = (unsigned)((P.total_rays + 255ull) / 256ull);
  if (fast)
    hipLaunchKernelGGL((direct_kernel<KIND, true>), dim3(blocks), dim3(256), 0, ctx->stream, P);
  else
    hipLaunchKernelGGL((direct_kernel<KIND, false>), dim3(blocks), dim3(256), 0, ctx->stream, P);
  HIP_TRY(ctx, hipGetLastError());
  return CURVIS_OK;
}

int render_direct_impl(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cam, uint32_t max_iter,
                       double max_radius, double delta, uint8_t *rgb_out, curvis_stats *stats) {
  if (!ctx) return CURVIS_E_INVALID;
  if (!metric || !cam) return fail(ctx, CURVIS_E_INVALID, "null metric/camera");
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = curvis_metric_validate(metric);
  if (rc != CURVIS_OK) return fail(ctx, rc, "invalid metric parameters (src/metrics.rs:409-456)");
  const uint32_t W = cam->res_x, H = cam->res_y;
  if (W == 0 || H == 0) return fail(ctx, CURVIS_E_INVALID, "resolution must be greater than 0 (src/cameras.rs:98)");
  if (std::fabs(cam->pos[1]) > max_radius)
    return fail(ctx, CURVIS_E_CAMERA_OUTSIDE, "Photon already beyond the maximum radius. Cannot evaluate escape. (src/systems.rs:122-124)");
  if (!ctx->d_sky[0] || !ctx->d_sky[1]) return fail(ctx, CURVIS_E_NO_SKY, "both background images must be set");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DirectParams P;
  P.metric = make_metric(*metric);
  P.cam = make_camera(*cam);
  if (!cvk::efficient_frame_pose(cam->pos[2], cam->pos[3], P.frame)) /* src/systems.rs:393-397, :411 */
    return fail(ctx, CURVIS_E_PARALLEL, "v1 and v2 must not be parallel (src/algebra.rs:95-97, camera on the x axis)");
  for (int k = 0; k < 2; ++k) {
    P.sky[k].texels = (const unsigned *)ctx->d_sky[k];
    P.sky[k].w = ctx->sky_w[k];
    P.sky[k].h = ctx->sky_h[k];
    for (int i = 0; i < 9; ++i) P.sky[k].inv_rot[i] = ctx->sky_inv_rot[k][i];
  }
  P.W = W;
  P.H = H;
  P.tiles_x = (W + 7) / 8;
  P.tiles_y = (H + 7) / 8;
  P.total_rays = (unsigned long long)P.tiles_x * P.tiles_y * 64ull;
  if (P.total_rays / 64ull > 0xFFFFFFFFull) return fail(ctx, CURVIS_E_INVALID, "frame too large");
  P.max_iter = max_iter;
  P.max_radius = max_radius;
  P.delta = delta;
  P.fast_ok = cvk::metric_fast_ok(metric->kind, P.metric, max_radius) ? 1 : 0;
  const size_t npix = (size_t)W * H, fb_bytes = npix * 3;
  rc = fb_begin_write(ctx, fb_bytes);
  if (rc) return rc;
  ctx->fb_bytes = fb_bytes;
  P.fb = ctx->d_fb;
  FrameCounters FC;
  rc = prepare_counters(ctx, 1, FC);
  if (rc) return rc;
  P.counters = FC;
  const size_t cnt_words = counter_words(1, FC.slots);
  const bool fast = ctx->fast_math != 0;
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  switch (metric->kind) {
    case CURVIS_METRIC_ELLIS:
      rc = launch_direct_kind<cvk::METRIC_ELLIS>(ctx, fast, P);
      break;
    case CURVIS_METRIC_INTERSTELLAR:
      rc = launch_direct_kind<cvk::METRIC_INTERSTELLAR>(ctx, fast, P);
      break;
    default:
      rc = launch_direct_kind<cvk::METRIC_FLAT>(ctx, fast, P);
      break;
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(unsigned long long) * cnt_words, hipMemcpyDeviceToHost, ctx->stream));
  if (rgb_out) {
    rc = fb_download(ctx, rgb_out, fb_bytes); /* leaves the stream idle; option "async_download": the frames follow */
    if (rc) return rc;
  } else {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  float ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  uint64_t fc[FC_N];
  sum_frame_counters(ctx->h_counters, FC.slots, 0, fc);
  curvis_stats st;
  std::memset(&st, 0, sizeof st);
  st.rays = fc[FC_RAYS];
  st.steps = fc[FC_STEPS];
  st.n_pos = fc[FC_POS];
  st.n_neg = fc[FC_NEG];
  st.n_none = fc[FC_NONE];
  st.n_oob = fc[FC_OOB];
  st.kernel_ms = st.integrate_ms = ms;
  st.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  ctx->last_frame_stats.assign(1, st);
  ctx->last_integrate_ms = ms;
  ctx->last_shade_ms = 0.0;
  ctx->last_relay_launches = 0;
  if (stats) *stats = st;
  return CURVIS_OK;
}

}  // namespace
