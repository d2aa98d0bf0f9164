/* render_host.h -- host side of the per-pixel path: struct curvis_ctx, launch selection (static / relay / persistent),
 * render_impl with per-frame statistics and the relay seat belt.
 * Part of the ONE translation unit curvis_hip.hip (included there, nowhere else). */
#pragma once

namespace {

/* ------------------------------------------------------------------------------------------ host */

thread_local std::string g_create_error;

}  // namespace

struct curvis_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipDeviceProp_t prop{};
  std::string err;
  /* skies */
  void *d_sky[2] = {nullptr, nullptr};
  bool sky_owned[2] = {false, false};
  unsigned sky_w[2] = {0, 0}, sky_h[2] = {0, 0};
  double sky_inv_rot[2][9];
  /* frame resources */
  unsigned char *d_fb = nullptr;
  size_t fb_cap = 0, fb_bytes = 0;
  /* overlapped download (option "async_download", fb_begin_write / fb_download below): a copy stream of its own, the second
   * frame buffer the next render call writes while the copy engine still reads the first, and the one download in flight */
  int async_download = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_fb = nullptr, ev_dl = nullptr; /* frames complete on `stream` / download complete on `copy_stream` */
  unsigned char *d_fb_alt = nullptr;
  size_t fb_alt_cap = 0;
  bool dl_pending = false;
  const unsigned char *dl_src = nullptr;       /* the device buffer the pending download reads */
  uint64_t downloads_overlapped = 0;           /* downloads queued behind the caller's back so far (option, read-only) */
  /* option "async_streams": curvis_ctx_deflate_frames returns while its streams are still on their way to the caller's buffer (copy
   * stream); they are there after curvis_ctx_download_wait, and before the next deflate call touches the scratch they are read from */
  int async_streams = 0;
  bool streams_pending = false;
  hipEvent_t ev_streams = nullptr;
  curvis_ray_debug *d_dbg = nullptr;
  size_t dbg_cap = 0;
  unsigned char *d_store = nullptr; /* RayStore arrays, carved from one allocation */
  unsigned char *d_rq = nullptr;    /* RelayQueue + ticket ring of the relay kernel */
  unsigned char *d_verify = nullptr; /* copy of the relay kernel's frame while the static kernel re-renders it (seat belt) */
  size_t verify_cap = 0;
  unsigned char *d_png = nullptr;    /* scratch of the device PNG front end (kernels_png.h): histograms, codes, offsets, streams */
  size_t png_cap = 0;
  double last_png_ms = 0.0;          /* HIP-event time of the last curvis_ctx_deflate_frames */
  size_t last_png_stream_bytes = 0;  /* bytes the streams of the last curvis_ctx_deflate_frames take (also when it failed for want of room) */
  int relay_segment = 0;            /* steps between two hand-over points; 0 = automatic */
  int relay_max_hops = 0;           /* hand-overs per tile at most; 0 = no limit */
  int relay_max_parks = 0;          /* hand-overs per launch at most; 0 = no limit */
  int relay_max_frames = 8;         /* largest launch (frames) the relay kernel is used for: the end-game it repairs is
                                       ~5 % of a one-frame launch and 1-2 % of a launch of three to six frames;
                                       beyond that its staging area (56 B per ray) buys nothing */
  int relay_disabled = 0;           /* set when a relay launch reported waves that gave up waiting: the context falls back
                                       to the static kernel for good (the relay kernel leans on the dispatcher starting
                                       workgroups in blockIdx order, which HIP does not promise) */
  int relay_verify = 0;             /* debug option: every relay render is repeated with the static kernel and the two
                                       frames and statistics compared (CURVIS_E_HIP on a difference) */
  int relay_test_fault = 0;         /* test hook: pretend the next relay launch reported a wave that gave up */
  int relay_test_corrupt = 0;       /* test hook: the next relay launch perturbs the first tile it hands over */
  int relay_auto_verify = 1;        /* seat belt (default on): the FIRST relay launch of every launch shape (W, H, frames, metric,
                                       step flavour) of this context is repeated by the static kernel and compared; on a
                                       difference the context drops to the static kernel for good (relay_mismatches counts) */
  uint32_t relay_mismatches = 0;
  /* launch shapes of the relay kernel -> relay launches of that shape so far; the first and then every
   * relay_recheck_every-th is repeated by the static kernel and compared */
  std::map<std::array<uint32_t, 9>, uint64_t> relay_verified;
  int relay_recheck_every = 1024;
  uint64_t relay_checks = 0;        /* launches checked so far */
  uint32_t relay_fallbacks = 0;     /* renders that fell back from the relay to the static kernel */
  long long relay_min_blocks = -1;  /* smallest grid (fresh workgroups) the relay kernel is used for; -1 = automatic
                                       (4 per CU: with fewer workgroups than that nearly the whole grid is resident at
                                       once, there is no dispatch phase, and the static kernel is as good) */
  uint32_t last_relay_launches = 0;
  uint64_t last_relay_parks = 0, last_relay_waiters = 0;
  unsigned relay_resident_blocks[3][2] = {{0, 0}, {0, 0}, {0, 0}}; /* cached occupancy query per kernel instantiation */
  int relay_resident_threads = 0;                                  /* ... valid for this workgroup size */
  int block_threads = 0; /* workgroup size of the static / relay kernels: 64, 128 or 256; 0 = automatic */
  size_t store_cap = 0;
  hipEvent_t ev2 = nullptr;
  /* efficient mode scratch (device) */
  unsigned char *d_eff = nullptr;
  size_t eff_cap = 0;
  unsigned char *h_eff = nullptr; /* pinned staging mirror of d_eff for the sampling launches */
  size_t h_eff_cap = 0;
  /* sample tables of the last efficient render, per frame (for tests / statistics) */
  std::vector<std::vector<cvs::BiPoint>> last_samples;
  std::vector<curvis_sampling_info> last_sampling_info;
  cvk::CameraParams *d_cams = nullptr;
  size_t cams_cap = 0;
  cvk::CameraParams *h_cams = nullptr; /* pinned */
  size_t h_cams_cap = 0;
  unsigned long long *d_counters = nullptr; /* FrameCounters block, sized for the largest launch so far */
  size_t counters_cap = 0;
  unsigned long long *h_counters = nullptr; /* pinned mirror (+ 8 words for the relay queue header) */
  size_t h_counters_cap = 0;
  /* statistics of the last render, per frame (curvis_ctx_frame_stats) */
  std::vector<curvis_stats> last_frame_stats;
  /* options */
  int variant = -1;         /* -1 automatic (default): relay kernel for launches of up to relay_max_frames frames and at least
                               relay_min_blocks workgroups, static kernel otherwise;
                               1 static one-ray-per-thread, 2 relay (subject to relay_min_blocks), 0 persistent lane-refill */
  int refill_threshold = 16;
  int blocks_per_cu = 0;    /* 0 = occupancy query */
  int fast_math = 1;        /* 1 shared-reciprocal step (ray_step_fast), 0 compiler IEEE div/sqrt */
  int fuse_shade = 1;       /* static kernel shades in its epilogue (no ray store, no shade launch) */
  int sampling_speculation = -1; /* efficient renderer: depth of the speculative subtree evaluated below every
                                    refined interval (0 = one launch per refinement round, no speculation;
                                    -1 = automatic: 10 for one or two frames, 6 for three to five, 4 for larger batches;
                                    at most 11) */
  int sampling_speculation_first = -1; /* the same for the first launch (below the uniform grid); -1 = automatic: 8 / 4 / 3 */
  int device_sampler = -1;           /* efficient renderer: 1 = sampler_kernel (device-resident refinement loop), 0 = host-paced sampler with
                                        speculation, -1 = automatic: the device for calls of device_sampler_min_frames frames and more */
  int device_sampler_min_frames = 48; /* measured cross-over against the host-paced sampler: between 32 and 64 frames per call (profiles/round6_eff_device_sampler.txt) */
  int last_sampler_path = 0;         /* of the last render_efficient call: 0 host-paced, 1 device, 2 device -> fell back to the host (overflow) */
  /* device-resident sampler: two slots (device buffer + page-locked mirror each) that take turns -- a call samples into one on its own
   * stream, or finds one filled ahead of time by curvis_ctx_prefetch_efficient on `sampler_stream`; the tables of the last render
   * stay readable in their slot (curvis_ctx_samples) until that slot is submitted to again, i.e. for one more submission */
  struct SamplerSlot {
    bool valid = false, prefetched = false;
    unsigned char *d = nullptr, *h = nullptr;
    size_t d_cap = 0, h_cap = 0, res_bytes = 0, h_res_off = 0;
    size_t o_tab_off = 0, o_tab_n = 0, o_grid_off = 0, o_grid = 0, o_res = 0, o_tab[7] = {0, 0, 0, 0, 0, 0, 0};
    hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr;
    std::vector<unsigned> job_of_frame;
    std::vector<double> l_job, l_frame;
    /* what the tables depend on */
    curvis_metric metric{};
    uint32_t n_frames = 0, max_iter = 0, alpha_nums = 0, max_iterations_sampling = 0;
    double params[4] = {0, 0, 0, 0};
    int fast = 0, speculate = 0;
    uint64_t seq = 0; /* order of submission */
  } samp[2];
  unsigned samp_next = 0;
  uint64_t samp_seq = 0;
  hipStream_t sampler_stream = nullptr;
  uint64_t prefetches = 0, prefetch_hits = 0;
  int last_sampling_prefetched = 0;
  struct DevSamples {                /* which slot holds the tables of the last device-sampled call (curvis_ctx_samples fetches on demand) */
    bool valid = false, overwritten = false;
    unsigned slot = 0;
  } dev_samples;
  struct PixRecips {                 /* efficient pixel kernel: reciprocals of the call's constant denominators, formed on the device once
                                        per resolution (efficient_host.h ensure_pixel_recips) */
    bool valid = false;
    double res_x = 0.0, res_y = 0.0;
    cvk::PixelRecips y{};
  } pix_recips;
  uint32_t last_sampling_chains = 0; /* device sampler: Euler chains (rounds that had to integrate) of the slowest job of the last call */
  uint32_t last_sampling_launches = 0;
  uint64_t last_sampling_evaluated = 0;
  size_t max_store_bytes = (size_t)8 << 30; /* frames of a batch are rendered in chunks below this */
  double last_integrate_ms = 0.0, last_shade_ms = 0.0;
};

namespace {

int fail(curvis_ctx *ctx, int code, const std::string &msg) {
  if (ctx)
    ctx->err = msg;
  else
    g_create_error = msg;
  return code;
}

#define HIP_TRY(ctx, call)                                                                         \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(ctx, CURVIS_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
  } while (0)

template <typename T>
int ensure_device(curvis_ctx *ctx, T *&ptr, size_t &cap, size_t need) {
  if (need <= cap) return CURVIS_OK;
  if (ptr) HIP_TRY(ctx, hipFree(ptr));
  ptr = nullptr;
  cap = 0;
  HIP_TRY(ctx, hipMalloc((void **)&ptr, need * sizeof(T)));
  cap = need;
  return CURVIS_OK;
}

/* ---- overlapped download of the frames (option "async_download" = 1) -------------------------------------------------
 * The reference's render_image returns an owned host image (src/systems.rs:314-329), so a host that calls one render per
 * frame pays the PCIe copy after every kernel: +0.25 ms on a 10.2 ms 1080p frame (bench.py: value_with_download, -2.4 %).
 * With the option set a render call given `rgb_out` returns when its kernels are done and the copy is QUEUED on the
 * context's copy stream; the next call renders into the OTHER frame buffer while the copy engine drains the first, and,
 * before it queues its own download, waits for the previous one (long finished: it ran under this call's kernels).  So the
 * contract is a pipeline one frame deep: `rgb_out` of call k is complete when call k + 1 on the same context returns, or
 * after curvis_ctx_download_wait.  ctx->d_fb is always the buffer of the LAST render (what curvis_ctx_deflate_frames,
 * curvis_ctx_download and the seat belt read); only a call that is about to WRITE frames steps aside. */
int download_wait(curvis_ctx *ctx) {
  if (ctx->streams_pending) { /* option "async_streams": the zlib streams of the last deflate call */
    ctx->streams_pending = false;
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_streams));
  }
  if (!ctx->dl_pending) return CURVIS_OK;
  ctx->dl_pending = false;
  ctx->dl_src = nullptr;
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_dl));
  return CURVIS_OK;
}
int ensure_copy_stream(curvis_ctx *ctx) {
  if (!ctx->copy_stream) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fb, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_dl, hipEventDisableTiming));
  }
  if (!ctx->ev_streams) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_streams, hipEventDisableTiming));
  return CURVIS_OK;
}
/* call before anything writes `bytes` of frames into ctx->d_fb */
int fb_begin_write(curvis_ctx *ctx, size_t bytes) {
  if (ctx->dl_pending && ctx->dl_src == ctx->d_fb) { /* the copy engine is still reading it: write the other one */
    std::swap(ctx->d_fb, ctx->d_fb_alt);
    std::swap(ctx->fb_cap, ctx->fb_alt_cap);
  }
  return ensure_device(ctx, ctx->d_fb, ctx->fb_cap, bytes);
}
/* frames [0, bytes) of ctx->d_fb -> rgb_out, after everything queued on ctx->stream so far.  Synchronous unless the
 * option is set; either way ctx->stream is idle on return. */
int fb_download(curvis_ctx *ctx, unsigned char *rgb_out, size_t bytes) {
  if (!ctx->async_download) {
    HIP_TRY(ctx, hipMemcpyAsync(rgb_out, ctx->d_fb, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CURVIS_OK;
  }
  {
    const int rcs = ensure_copy_stream(ctx);
    if (rcs) return rcs;
  }
  const int rc = download_wait(ctx); /* the previous call's: it had this call's kernels to hide under */
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fb, ctx->stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_fb, 0));
  HIP_TRY(ctx, hipMemcpyAsync(rgb_out, ctx->d_fb, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
  /* from here on a copy into the caller's buffer is in flight: it is tracked BEFORE anything else can fail, and a failure of
   * one of the remaining calls drains the copy stream before it is reported -- no error return with rgb_out still being written */
  ctx->dl_pending = true;
  ctx->dl_src = ctx->d_fb;
  ctx->downloads_overlapped++;
  hipError_t e = hipEventRecord(ctx->ev_dl, ctx->copy_stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); /* kernels, counters, debug dump: done (the frames are still on their way) */
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    ctx->dl_pending = false;
    ctx->dl_src = nullptr;
    return fail(ctx, CURVIS_E_HIP, std::string("asynchronous frame download: ") + hipGetErrorString(e));
  }
  return CURVIS_OK;
}

/* counter block for a launch of n_frames frames: device block + pinned mirror, zeroed on the stream */
int prepare_counters(curvis_ctx *ctx, unsigned n_frames, FrameCounters &C, unsigned slots = 0u) {
  C.slots = slots ? slots : counter_slots_for(n_frames); /* a power of two */
  const size_t words = counter_words(n_frames, C.slots);
  int rc = ensure_device(ctx, ctx->d_counters, ctx->counters_cap, words);
  if (rc) return rc;
  if (ctx->h_counters_cap < words + 8) {
    if (ctx->h_counters) HIP_TRY(ctx, hipHostFree(ctx->h_counters));
    ctx->h_counters = nullptr;
    ctx->h_counters_cap = 0;
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_counters, sizeof(unsigned long long) * (words + 8)));
    ctx->h_counters_cap = words + 8;
  }
  C.base = ctx->d_counters;
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * words, ctx->stream));
  return CURVIS_OK;
}
/* sum the replicas of frame f of the mirrored block into out[FC_N] */
void sum_frame_counters(const unsigned long long *h, unsigned slots, unsigned f, uint64_t out[FC_N]) {
  for (int k = 0; k < FC_N; ++k) out[k] = 0;
  for (unsigned r = 0; r < slots; ++r) {
    const unsigned long long *line = h + (size_t)CNT_STRIDE * (1u + (size_t)f * slots + r);
    for (int k = 0; k < FC_N; ++k) out[k] += line[k];
  }
}

cvk::MetricParams make_metric(const curvis_metric &m) {
  cvk::MetricParams M;
  M.rho = m.rho;
  M.rho2 = m.rho * m.rho;
  M.m = m.m;
  M.a = m.a;
  M.pim = CV_PI * m.m;
  M.inv_pim = 1.0 / M.pim;
  M.two_o_pi = 2.0 / CV_PI;
  M.T = cv_sc_table(); /* host tables; kernels substitute their own copies (LDS or __constant__) */
  M.LT = cv_log_table();
  M.AT = cv_atan_table();
  return M;
}

/* one Euler step on the host, all eight components: the body of trajectory_kernel's loop */
template <int KIND>
void host_euler_step(const cvk::MetricParams &MP, double x[4], double p[4], double delta) {
  cvk::Ray q;
  q.l = x[1];
  q.th = x[2];
  q.ph = x[3];
  q.p1 = p[1];
  q.p2 = p[2];
  q.p3 = p[3];
  q.p3sq = q.p3 * q.p3;
  cvk::ray_step<KIND, true>(MP, q, delta);
  x[0] = x[0] + (p[0] * (1.0 / -1.0)) * delta; /* dx0 = p0 * g00.powi(-1), as in trajectory_kernel */
  x[1] = q.l;
  x[2] = q.th;
  x[3] = q.ph;
  p[0] = p[0] + 0.0 * delta;
  p[1] = q.p1;
  p[2] = q.p2;
  p[3] = p[3] + 0.0 * delta;
}

cvk::CameraParams make_camera(const curvis_camera &c) {
  cvk::CameraParams C;
  for (int i = 0; i < 4; ++i) C.pos[i] = c.pos[i];
  for (int i = 0; i < 9; ++i) C.rot[i] = c.rot[i];
  C.focal = c.focal;
  C.sensor_w = c.sensor_w;
  C.sensor_h = c.sensor_h;
  C.res_x = (double)c.res_x;
  C.res_y = (double)c.res_y;
  return C;
}

/* workgroup size of the static and relay kernels ("block_threads"; total_rays is a multiple of 64) */
unsigned integrate_block_threads(const curvis_ctx *ctx, int kind) {
  (void)kind;
  const int bt = ctx->block_threads;
  return (bt == 64 || bt == 128 || bt == 256) ? (unsigned)bt : 256u;
}

/* grid = fresh workgroups + relay workgroups; see geodesic_relay */
template <int KIND, bool FAST>
int launch_relay(curvis_ctx *ctx, const IntegrateParams &P, bool relay_only) {
  const size_t bytes = sizeof(RelayQueue) + sizeof(unsigned) * kRelayRing;
  if (!ctx->d_rq) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_rq, bytes));
  RelayArgs A;
  A.q = (RelayQueue *)ctx->d_rq;
  A.n_tiles = P.total_rays / 64ull;
  const unsigned bt = integrate_block_threads(ctx, KIND);
  const unsigned long long fresh_blocks = relay_only ? 0ull : (P.total_rays + bt - 1ull) / bt;
  A.fresh_blocks = (unsigned)fresh_blocks;
  /* segment = 0.6 R / delta steps: an ordinary ray (about R / delta steps from a camera near the throat, +-10 %)
   * then crosses ONE hand-over point and ends well inside its second segment.  With 0.5 R / delta the second
   * boundary falls inside the spread of ray lengths and a third of the tiles is handed over a second time for their
   * last few dozen steps (1080p: 10.8 ms against 10.6 with 0.4 or 0.6; tools/gpu_seg_sweep.py); segments below
   * ~0.3 R / delta cost more in boundary checks and workgroup launches than the finer balance returns. */
  {
    const double half = 0.6 * P.max_radius / P.delta;
    unsigned seg = (half >= 256.0 && half <= 65536.0) ? (unsigned)half : 1024u;
    A.seg = ctx->relay_segment > 0 ? (unsigned)ctx->relay_segment : seg;
  }
  A.max_hops = (unsigned)std::max(0, ctx->relay_max_hops);
  A.max_parks = (unsigned)std::max(0, ctx->relay_max_parks);
  A.corrupt_ticket = ctx->relay_test_corrupt ? 1u : 0u;
  ctx->relay_test_corrupt = 0;
  if (ctx->relay_resident_threads != (int)bt) {
    for (auto &row : ctx->relay_resident_blocks) row[0] = row[1] = 0;
    ctx->relay_resident_threads = (int)bt;
  }
  unsigned &cached = ctx->relay_resident_blocks[KIND][FAST ? 1 : 0];
  if (cached == 0) {
    int per_cu = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, geodesic_relay<KIND, FAST>, (int)bt, 0));
    if (per_cu <= 0) per_cu = 1;
    cached = (unsigned)per_cu * (unsigned)ctx->prop.multiProcessorCount;
  }
  const unsigned long long resident_blocks = cached;
  if (!relay_only) HIP_TRY(ctx, hipMemsetAsync(ctx->d_rq, 0, bytes, ctx->stream));
  /* every tile in flight when the fresh workgroups run out (at most the resident waves) is passed on once per
   * segment of its remaining steps: (max_iter / seg) <= 16 hand-overs each, usually ~2; surplus relay
   * workgroups leave at once */
  unsigned long long relay_blocks = resident_blocks * 24ull;
  if (relay_blocks > fresh_blocks * 2ull + resident_blocks) relay_blocks = fresh_blocks * 2ull + resident_blocks;
  hipLaunchKernelGGL((geodesic_relay<KIND, FAST>), dim3((unsigned)(fresh_blocks + relay_blocks)), dim3(bt), 0, ctx->stream,
                     P, A);
  HIP_TRY(ctx, hipGetLastError());
  return CURVIS_OK;
}

template <int KIND, bool PHI, bool FAST>
int launch_integrate(curvis_ctx *ctx, const IntegrateParams &P, bool fused, int relay) {
  if (relay && fused) return launch_relay<KIND, FAST>(ctx, P, relay == 2);
  if (ctx->variant != 0) {
    const unsigned bt = integrate_block_threads(ctx, KIND);
    const unsigned long long blocks = (P.total_rays + bt - 1ull) / bt;
    if (fused)
      hipLaunchKernelGGL((geodesic_static<KIND, false, FAST, true>), dim3((unsigned)blocks), dim3(bt), 0, ctx->stream, P);
    else
      hipLaunchKernelGGL((geodesic_static<KIND, PHI, FAST, false>), dim3((unsigned)blocks), dim3(bt), 0, ctx->stream, P);
  } else {
    int per_cu = ctx->blocks_per_cu;
    if (per_cu <= 0) {
      HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, geodesic_persistent<KIND, PHI, FAST>, 256, 0));
      if (per_cu <= 0) per_cu = 1;
    }
    unsigned long long blocks = (unsigned long long)per_cu * (unsigned long long)ctx->prop.multiProcessorCount;
    const unsigned long long max_useful = (P.total_rays + 255ull) / 256ull;
    if (blocks > max_useful) blocks = max_useful;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((geodesic_persistent<KIND, PHI, FAST>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, P);
  }
  HIP_TRY(ctx, hipGetLastError());
  return CURVIS_OK;
}

template <int KIND>
int launch_integrate_kind(curvis_ctx *ctx, bool phi, bool fast, bool fused, int relay, const IntegrateParams &P) {
  if (phi)
    return fast ? launch_integrate<KIND, true, true>(ctx, P, false, 0) : launch_integrate<KIND, true, false>(ctx, P, false, 0);
  return fast ? launch_integrate<KIND, false, true>(ctx, P, fused, relay) : launch_integrate<KIND, false, false>(ctx, P, fused, relay);
}
int launch_integrate_any(curvis_ctx *ctx, int kind, bool phi, bool fast, bool fused, int relay, const IntegrateParams &P) {
  switch (kind) {
    case CURVIS_METRIC_ELLIS: return launch_integrate_kind<cvk::METRIC_ELLIS>(ctx, phi, fast, fused, relay, P);
    case CURVIS_METRIC_INTERSTELLAR: return launch_integrate_kind<cvk::METRIC_INTERSTELLAR>(ctx, phi, fast, fused, relay, P);
    default: return launch_integrate_kind<cvk::METRIC_FLAT>(ctx, phi, fast, fused, relay, P);
  }
}

template <int KIND>
int launch_shade_kind(curvis_ctx *ctx, bool debug, const ShadeParams &P) {
  const unsigned long long blocks = (P.n_pixels + 255ull) / 256ull;
  if (debug)
    hipLaunchKernelGGL((shade_kernel<KIND, true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, P);
  else
    hipLaunchKernelGGL((shade_kernel<KIND, false>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, P);
  HIP_TRY(ctx, hipGetLastError());
  return CURVIS_OK;
}

RayStore carve_store(unsigned char *base, size_t npix) {
  RayStore S;
  double *d = (double *)base;
  S.l = d;
  S.th = d + npix;
  S.ph = d + 2 * npix;
  S.p1 = d + 3 * npix;
  S.p2 = d + 4 * npix;
  S.p3 = d + 5 * npix;
  S.steps = (unsigned *)(d + 6 * npix);
  S.code = (int *)(S.steps + npix);
  return S;
}
constexpr size_t kStoreBytesPerPixel = 6 * sizeof(double) + sizeof(unsigned) + sizeof(int);

int render_impl(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cams, uint32_t n_frames,
                uint32_t max_iterations, double max_radius, double delta, uint8_t *rgb_out,
                curvis_ray_debug *dbg_out, curvis_stats *stats, uint32_t row_begin = 0, uint32_t row_count = 0) {
  if (!ctx) return CURVIS_E_INVALID;
  if (!metric || !cams || n_frames == 0) return fail(ctx, CURVIS_E_INVALID, "null metric/camera or zero frames");
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = curvis_metric_validate(metric);
  if (rc != CURVIS_OK) return fail(ctx, rc, "invalid metric parameters (src/metrics.rs:409-456)");
  const uint32_t W = cams[0].res_x, H_full = cams[0].res_y;
  if (W == 0 || H_full == 0) return fail(ctx, CURVIS_E_INVALID, "resolution must be greater than 0 (src/cameras.rs:98)");
  /* row band (curvis_render_brute_rows): the launch covers image rows [row_begin, row_begin + row_count); the
   * cameras keep the full resolution, which is what pixel -> direction uses */
  const bool band = row_count != 0;
  if (band && ((uint64_t)row_begin + row_count > H_full || n_frames != 1 || dbg_out))
    return fail(ctx, CURVIS_E_INVALID, "row band outside the frame (or used with a batch / the debug dump)");
  const uint32_t H = band ? row_count : H_full;
  for (uint32_t f = 0; f < n_frames; ++f) {
    if (cams[f].res_x != W || cams[f].res_y != H_full)
      return fail(ctx, CURVIS_E_INVALID, "all cameras of a batch must share one resolution");
    if (std::fabs(cams[f].pos[1]) > max_radius)
      return fail(ctx, CURVIS_E_CAMERA_OUTSIDE,
                  "Photon already beyond the maximum radius. Cannot evaluate escape. (src/systems.rs:122-124)");
  }
  if (!ctx->d_sky[0] || !ctx->d_sky[1]) return fail(ctx, CURVIS_E_NO_SKY, "both background images must be set");
  HIP_TRY(ctx, hipSetDevice(ctx->device));

  const size_t npix = (size_t)W * H;
  const size_t fb_bytes = npix * 3 * n_frames;
  rc = fb_begin_write(ctx, fb_bytes);
  if (rc) return rc;
  ctx->fb_bytes = fb_bytes;
  if (dbg_out) {
    rc = ensure_device(ctx, ctx->d_dbg, ctx->dbg_cap, npix * n_frames);
    if (rc) return rc;
  }
  /* fused shading: static kernel, no debug dump (option "fuse_shade", default on) -- no ray store at all.
   * Otherwise frames are rendered in chunks whose ray store stays below max_store_bytes. */
  const bool fused = ctx->variant != 0 && ctx->fuse_shade != 0 && dbg_out == nullptr;
  /* relay kernel ("variant" = 2, and the automatic choice for big enough single images): end-game hand-over of
   * tiles; only launches of a few frames have a tail worth its staging area (56 B per ray) -- larger batches
   * use the static kernel, and so do frames too small to have a dispatch phase (measured against the static
   * kernel: 640x360 +2 %, 720x405 -9 %, 800x450 -9 %, 960x540 -15 %, 1280x720 -6 %, 1920x1080 -3..-5 %,
   * 2560x1440 -1 %; tools/gpu_relay_sizes.py, tools/gpu_relay_threshold.py) */
  const unsigned long long relay_fresh_blocks = ((unsigned long long)((W + 7) / 8) * ((H + 7) / 8) * n_frames + 3ull) / 4ull;
  const unsigned long long relay_min = ctx->relay_min_blocks >= 0 ? (unsigned long long)ctx->relay_min_blocks
                                                                   : 4ull * (unsigned long long)ctx->prop.multiProcessorCount;
  const size_t relay_staging = (size_t)((W + 7) / 8) * ((H + 7) / 8) * 64u * n_frames * kStoreBytesPerPixel;
  const bool relay = (ctx->variant == 2 || ctx->variant < 0) && !ctx->relay_disabled && fused && n_frames <= (uint32_t)ctx->relay_max_frames &&
                     relay_fresh_blocks >= relay_min && relay_staging <= ctx->max_store_bytes;
  uint32_t chunk = n_frames;
  if (relay) {
    const size_t rays = (size_t)((W + 7) / 8) * ((H + 7) / 8) * 64u * n_frames;
    rc = ensure_device(ctx, ctx->d_store, ctx->store_cap, rays * kStoreBytesPerPixel);
    if (rc) return rc;
  }
  if (!fused) {
    chunk = (uint32_t)std::max<size_t>(1, ctx->max_store_bytes / (npix * kStoreBytesPerPixel));
    if (chunk > n_frames) chunk = n_frames;
    rc = ensure_device(ctx, ctx->d_store, ctx->store_cap, (size_t)chunk * npix * kStoreBytesPerPixel);
    if (rc) return rc;
  }
  rc = ensure_device(ctx, ctx->d_cams, ctx->cams_cap, (size_t)n_frames);
  if (rc) return rc;
  if (ctx->h_cams_cap < n_frames) {
    if (ctx->h_cams) HIP_TRY(ctx, hipHostFree(ctx->h_cams));
    ctx->h_cams = nullptr;
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_cams, sizeof(cvk::CameraParams) * n_frames));
    ctx->h_cams_cap = n_frames;
  }
  for (uint32_t f = 0; f < n_frames; ++f) ctx->h_cams[f] = make_camera(cams[f]);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_cams, ctx->h_cams, sizeof(cvk::CameraParams) * n_frames, hipMemcpyHostToDevice,
                              ctx->stream));

  const cvk::MetricParams MP = make_metric(*metric);
  const bool phi = dbg_out != nullptr; /* phi is only read by the debug dump on this path */
  const bool fast = ctx->fast_math != 0;
  uint64_t tot[FC_N] = {0};
  double integrate_ms = 0.0, shade_ms = 0.0;
  ctx->last_frame_stats.assign(n_frames, curvis_stats{});

  for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk) {
    const uint32_t nf = std::min(chunk, n_frames - f0);
    FrameCounters FC;
    rc = prepare_counters(ctx, nf, FC);
    if (rc) return rc;
    const size_t cnt_words = counter_words(nf, FC.slots);
    IntegrateParams P;
    P.metric = MP;
    P.cams = ctx->d_cams + f0;
    P.n_frames = nf;
    P.W = W;
    P.H = H;
    P.row0 = band ? row_begin : 0u;
    P.tiles_x = (W + 7) / 8;
    P.tiles_y = (H + 7) / 8;
    const unsigned long long rpf = (unsigned long long)P.tiles_x * P.tiles_y * 64ull;
    if (rpf > 0xFFFFFFFFull || rpf * nf / 64ull > 0xFFFFFFFFull) return fail(ctx, CURVIS_E_INVALID, "frame or batch too large");
    P.rays_per_frame = (unsigned)rpf;
    P.total_rays = rpf * nf;
    P.max_iter = max_iterations;
    P.max_radius = max_radius;
    P.delta = delta;
    P.store = relay ? carve_store(ctx->d_store, (size_t)P.total_rays)
                    : fused ? RayStore{} : carve_store(ctx->d_store, (size_t)nf * npix);
    P.counters = FC;
    for (int k = 0; k < 2; ++k) {
      P.sky[k].texels = (const unsigned *)ctx->d_sky[k];
      P.sky[k].w = ctx->sky_w[k];
      P.sky[k].h = ctx->sky_h[k];
      for (int i = 0; i < 9; ++i) P.sky[k].inv_rot[i] = ctx->sky_inv_rot[k][i];
    }
    P.fb = ctx->d_fb + (size_t)f0 * npix * 3;
    P.refill_threshold = ctx->refill_threshold < 1 ? 1 : (ctx->refill_threshold > 64 ? 64 : ctx->refill_threshold);
    P.fast_ok = cvk::metric_fast_ok(metric->kind, MP, max_radius) ? 1 : 0;
    P.trace = nullptr;
    const char *trace_file = getenv("CURVIS_TRACE_FILE");
    const size_t trace_words = (size_t)(P.total_rays / 64ull) * 4u;
    size_t trace_alloc_words = trace_words;
    if (relay) trace_alloc_words = (size_t)(P.total_rays / 64ull) * 3u * 4u + 65536u * 16u; /* every wave of the grid */
    if (trace_file && *trace_file && (ctx->variant != 0 || relay)) {
      HIP_TRY(ctx, hipMalloc((void **)&P.trace, trace_alloc_words * sizeof(unsigned long long)));
      HIP_TRY(ctx, hipMemsetAsync(P.trace, 0, trace_alloc_words * sizeof(unsigned long long), ctx->stream));
    }

    ShadeParams Q;
    Q.metric = MP;
    for (int k = 0; k < 2; ++k) {
      Q.sky[k].texels = (const unsigned *)ctx->d_sky[k];
      Q.sky[k].w = ctx->sky_w[k];
      Q.sky[k].h = ctx->sky_h[k];
      for (int i = 0; i < 9; ++i) Q.sky[k].inv_rot[i] = ctx->sky_inv_rot[k][i];
    }
    Q.store = P.store;
    Q.n_pixels = (unsigned long long)nf * npix;
    Q.fb = ctx->d_fb + (size_t)f0 * npix * 3;
    Q.dbg = dbg_out ? ctx->d_dbg + (size_t)f0 * npix : nullptr;
    Q.npix = npix;
    Q.counters = FC;

    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    rc = launch_integrate_any(ctx, metric->kind, phi, fast, fused, relay ? 1 : 0, P);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (P.trace) { /* diagnostics only: dump the per-wave records of this launch (binary u64 x 4 per wave) */
      std::vector<unsigned long long> tr(trace_alloc_words);
      HIP_TRY(ctx, hipMemcpyAsync(tr.data(), P.trace, trace_alloc_words * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      HIP_TRY(ctx, hipFree(P.trace));
      if (FILE *fp = fopen(trace_file, "wb")) {
        fwrite(tr.data(), sizeof(unsigned long long), tr.size(), fp);
        fclose(fp);
      }
    }
    if (!fused) {
      switch (metric->kind) {
        case CURVIS_METRIC_ELLIS:
          rc = launch_shade_kind<cvk::METRIC_ELLIS>(ctx, dbg_out != nullptr, Q);
          break;
        case CURVIS_METRIC_INTERSTELLAR:
          rc = launch_shade_kind<cvk::METRIC_INTERSTELLAR>(ctx, dbg_out != nullptr, Q);
          break;
        default:
          rc = launch_shade_kind<cvk::METRIC_FLAT>(ctx, dbg_out != nullptr, Q);
          break;
      }
      if (rc) return rc;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev2, ctx->stream));
    ctx->last_relay_launches = relay ? 1 : 0;
    for (;;) {
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(unsigned long long) * cnt_words,
                                  hipMemcpyDeviceToHost, ctx->stream));
      if (relay) /* queue header rides along with the counters: finished / error */
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters + cnt_words, ctx->d_rq, sizeof(unsigned long long) * 8,
                                    hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (!relay) break;
      /* normally the one launch finished every tile; more relay workgroups only if the grid ran out of them
       * with tiles still parked */
      const RelayQueue *hq = (const RelayQueue *)(ctx->h_counters + cnt_words);
      const unsigned long long n_tiles = P.total_rays / 64ull;
      if (hq->error != 0 || ctx->relay_test_fault) {
        /* waves gave up waiting for a tile (a logic error, or a dispatcher that did not start the workgroups in
         * order): not a hang and not a wrong frame -- the frame is rendered again by the static kernel, which has no
         * inter-workgroup dependency, and this context stops using the relay kernel */
        ctx->relay_test_fault = 0;
        ctx->relay_disabled = 1;
        ctx->relay_fallbacks++;
        fprintf(stderr, "[curvis] relay kernel: %llu waves gave up waiting (%llu tiles unfinished); falling back to the static kernel for this context\n",
                (unsigned long long)hq->error, (unsigned long long)(n_tiles - hq->finished));
        return render_impl(ctx, metric, cams, n_frames, max_iterations, max_radius, delta, rgb_out, dbg_out, stats, row_begin,
                           row_count);
      }
      ctx->last_relay_parks = hq->tail;
      ctx->last_relay_waiters = hq->head;
      if (hq->finished >= n_tiles) break;
      if (ctx->last_relay_launches++ > 64)
        return fail(ctx, CURVIS_E_HIP, "relay kernel: tiles still unfinished after 64 relay launches");
      rc = launch_integrate_any(ctx, metric->kind, phi, fast, fused, 2, P);
      if (rc) return rc;
      HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
      HIP_TRY(ctx, hipEventRecord(ctx->ev2, ctx->stream));
    }
    float ms_i = 0.f, ms_s = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms_i, ctx->ev0, ctx->ev1));
    integrate_ms += ms_i;
    HIP_TRY(ctx, hipEventElapsedTime(&ms_s, ctx->ev1, ctx->ev2));
    shade_ms += ms_s;
    for (uint32_t f = 0; f < nf; ++f) {
      uint64_t fc[FC_N];
      sum_frame_counters(ctx->h_counters, FC.slots, f, fc);
      for (int k = 0; k < FC_N; ++k) tot[k] += fc[k];
      curvis_stats &fs = ctx->last_frame_stats[f0 + f];
      fs.rays = fc[FC_RAYS];
      fs.steps = fc[FC_STEPS];
      fs.n_pos = fc[FC_POS];
      fs.n_neg = fc[FC_NEG];
      fs.n_none = fc[FC_NONE];
      fs.n_oob = fc[FC_OOB];
      /* the frames of a launch run interleaved on the GPU: times are the launch's, shared out by executed steps */
      fs.integrate_ms = ms_i;
      fs.shade_ms = ms_s;
    }
    { /* time share of each frame of this launch, in proportion to its Euler steps */
      uint64_t launch_steps = 0;
      for (uint32_t f = 0; f < nf; ++f) launch_steps += ctx->last_frame_stats[f0 + f].steps;
      for (uint32_t f = 0; f < nf; ++f) {
        curvis_stats &fs = ctx->last_frame_stats[f0 + f];
        const double share = launch_steps ? (double)fs.steps / (double)launch_steps : 1.0 / nf;
        fs.integrate_ms *= share;
        fs.shade_ms *= share;
        fs.kernel_ms = fs.integrate_ms + fs.shade_ms;
        fs.total_ms = fs.kernel_ms;
      }
    }
  }
  ctx->last_integrate_ms = integrate_ms;
  ctx->last_shade_ms = shade_ms;
  /* The relay kernel's hand-over rests on gfx950 facts (write-through sc0 sc1 stores, s_waitcnt vmcnt(0) before
   * the ticket store) rather than on the HIP memory model, so it wears a seat belt: the first relay launch of every
   * launch shape is repeated by the static kernel -- no inter-workgroup traffic at all -- and frames and counters are
   * compared.  Option "relay_verify" = 1 checks EVERY launch and makes a difference an error (debugging); the automatic
   * check (option "relay_auto_verify", default 1) costs one static launch per shape and context and, on a difference,
   * reports it on stderr, counts it ("relay_mismatches"), switches the context to the static kernel and returns the
   * static kernel's frame. */
  /* everything that shapes the hand-over pattern: frame size and count, metric and step flavour, the band, the step cap, the
   * segment length and hop limit in force */
  const std::array<uint32_t, 9> shape = {W, H, n_frames, (uint32_t)metric->kind, (uint32_t)(fast ? 1 : 0), row_begin, row_count, max_iterations,
                                         (uint32_t)ctx->relay_segment * 256u + (uint32_t)std::max(0, ctx->relay_max_hops)};
  bool auto_check = false;
  if (relay && !ctx->relay_verify && ctx->relay_auto_verify) {
    const uint64_t seen = ctx->relay_verified[shape]++; /* relay launches of this shape before this one */
    auto_check = seen == 0 || (ctx->relay_recheck_every > 0 && seen % (uint64_t)ctx->relay_recheck_every == 0);
  }
  if (relay && (ctx->relay_verify || auto_check)) {
    /* the relay frame is kept in a second device buffer and compared there: no host copies (two pageable D2H copies of a
     * batch cost more than the static re-render and left the NEXT render call 20 ms slower) */
    const size_t padded = (fb_bytes + 7) & ~(size_t)7;
    rc = ensure_device(ctx, ctx->d_verify, ctx->verify_cap, padded + 8);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_verify + (padded - 8), 0, 16, ctx->stream)); /* tail padding + the counter */
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_verify, ctx->d_fb, fb_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (ctx->fb_cap < padded) { /* room for the zeroed tail the word-wise compare reads (the frame is re-rendered below anyway) */
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      rc = ensure_device(ctx, ctx->d_fb, ctx->fb_cap, padded);
      if (rc) return rc;
    }
    const std::vector<curvis_stats> fs = ctx->last_frame_stats;
    const uint32_t launches = ctx->last_relay_launches;
    const uint64_t parks = ctx->last_relay_parks, waiters = ctx->last_relay_waiters;
    const double keep_i = ctx->last_integrate_ms, keep_s = ctx->last_shade_ms;
    const int saved = ctx->variant;
    ctx->variant = 1;
    rc = render_impl(ctx, metric, cams, n_frames, max_iterations, max_radius, delta, nullptr, nullptr, nullptr, row_begin, row_count);
    ctx->variant = saved;
    if (rc) return rc;
    /* d_fb holds the static kernel's frame now */
    if (padded != fb_bytes) HIP_TRY(ctx, hipMemsetAsync(ctx->d_fb + fb_bytes, 0, padded - fb_bytes, ctx->stream));
    unsigned long long *d_cnt = (unsigned long long *)(ctx->d_verify + padded);
    const size_t n_words = padded / 8;
    hipLaunchKernelGGL(compare_kernel, dim3((unsigned)std::min<size_t>((n_words + 255) / 256, 4096)), dim3(256), 0, ctx->stream,
                       (const unsigned long long *)ctx->d_verify, (const unsigned long long *)ctx->d_fb, n_words, d_cnt);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long n_diff_words = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n_diff_words, d_cnt, sizeof n_diff_words, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    bool same = n_diff_words == 0 && fs.size() == ctx->last_frame_stats.size();
    for (size_t f = 0; same && f < fs.size(); ++f) {
      const curvis_stats &x = fs[f], &y = ctx->last_frame_stats[f];
      same = x.rays == y.rays && x.steps == y.steps && x.n_pos == y.n_pos && x.n_neg == y.n_neg && x.n_none == y.n_none && x.n_oob == y.n_oob;
    }
    if (!same) {
      ctx->relay_mismatches++;
      if (ctx->relay_verify) return fail(ctx, CURVIS_E_HIP, "relay_verify: the relay kernel and the static kernel disagree on this launch");
      fprintf(stderr, "[curvis] relay kernel: a checked launch of shape %ux%u x %u frame(s) differs from the static kernel (%llu of %zu 8-byte words%s); "
                      "this context uses the static kernel from now on\n", W, H, n_frames, n_diff_words, n_words, n_diff_words ? "" : ", counters only");
      ctx->relay_disabled = 1;
      ctx->relay_fallbacks++;
      return render_impl(ctx, metric, cams, n_frames, max_iterations, max_radius, delta, rgb_out, dbg_out, stats, row_begin, row_count);
    }
    ctx->relay_checks++;
    /* the launch that counts is the relay one: its frame is what d_fb holds again (same bytes), and so are its statistics */
    ctx->last_frame_stats = fs;
    ctx->last_relay_launches = launches;
    ctx->last_relay_parks = parks;
    ctx->last_relay_waiters = waiters;
    ctx->last_integrate_ms = keep_i;
    ctx->last_shade_ms = keep_s;
  }
  if (dbg_out)
    HIP_TRY(ctx, hipMemcpyAsync(dbg_out, ctx->d_dbg, sizeof(curvis_ray_debug) * npix * n_frames,
                                hipMemcpyDeviceToHost, ctx->stream));
  if (rgb_out) {
    rc = fb_download(ctx, rgb_out, fb_bytes);
    if (rc) return rc;
  } else {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (dbg_out) {
    /* dead lanes of the integrator, replayed on the host: t_{k+1} = t_k + (p_t * g^tt) * delta with
     * p_t = 1, g^tt = -1 (src/metrics.rs:237, :295); p_t = p_t + 0*delta stays 1. */
    std::vector<double> t_of_steps;
    for (uint32_t f = 0; f < n_frames; ++f) {
      curvis_ray_debug *d = dbg_out + (size_t)f * npix;
      uint32_t most = 0; /* the table only needs to reach the largest step count of the frame, not the cap */
      for (size_t i = 0; i < npix; ++i) most = std::max(most, d[i].steps);
      t_of_steps.resize((size_t)most + 1);
      double t = cams[f].pos[0];
      t_of_steps[0] = t;
      for (uint32_t k = 1; k <= most; ++k) {
        t = t + (1.0 * -1.0) * delta;
        t_of_steps[k] = t;
      }
      for (size_t i = 0; i < npix; ++i) d[i].x[0] = t_of_steps[d[i].steps];
    }
  }
  if (stats) {
    stats->rays = tot[FC_RAYS];
    stats->steps = tot[FC_STEPS];
    stats->n_pos = tot[FC_POS];
    stats->n_neg = tot[FC_NEG];
    stats->n_none = tot[FC_NONE];
    stats->n_oob = tot[FC_OOB];
    stats->kernel_ms = integrate_ms + shade_ms;
    stats->integrate_ms = integrate_ms;
    stats->shade_ms = shade_ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return CURVIS_OK;
}

}  // namespace
