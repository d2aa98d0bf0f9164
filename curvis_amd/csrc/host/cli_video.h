/* cli_video.h -- part of the `curvis` binary (host/curvis_cli.cpp includes the parts in order; one translation unit):
 * `curvis video` (src/main.rs:135-168, src/rendering.rs:258-327): device worker threads, shared retry queue, sky distribution, per-frame statistics. */
#ifndef CURVIS_CLI_VIDEO_H
#define CURVIS_CLI_VIDEO_H

namespace {

int video_main(const Args &a_in) {
  Args a = a_in; /* --batch 0 (automatic) is resolved below, once the frame size is known */
  std::printf("Video rendering\n");
  Common c;
  VideoSettings vs;
  std::string err;
  if (!a.video_toml.empty()) {
    if (!path_exists(a.video_toml)) die("Error with video settings: File \"" + a.video_toml + "\" not found.");
    if (!from_toml(a.video_toml, vs, err)) die("Error with video settings: " + err);
  }
  PhaseClock clk;
  load_common(a, c, "video");
  clk.mark("settings + background images");
  const bool batch_auto = a.batch < 1;
  if (a.batch < 1) {
    /* frames per launch.  The per-pixel modes keep the GPU busy with 8 frames per launch.  --mode efficient is launch- and
     * host-paced (lone waves between host-side sampler rounds): longer batches amortise both -- 1 568 / 2 202 / 2 757 frames/s at
     * 8 / 16 / 32 frames per launch with one context, the same trend with two to four (profiles/round5_eff_contexts_sweep.txt) --
     * as far as the page-locked batch buffers stay modest (three per worker of batch x frame bytes: 256 MB each at most) */
    const size_t fbytes0 = (size_t)c.cam.resolution_x * c.cam.resolution_y * 3;
    a.batch = a.mode == "efficient" ? (int)std::max<size_t>(4, std::min<size_t>(32, ((size_t)256 << 20) / std::max<size_t>(1, fbytes0))) : 8;
  }
  vs.filepath_to_camera_path = resolve_path(vs.filepath_to_camera_path); /* normalize() */
  if (vs.video_name.empty()) die("Error in rendering video: Video name cannot be an empty string.");
  if (!has_extension(vs.filepath_to_camera_path, "csv"))
    die("Error in rendering video: The camera path \"" + vs.filepath_to_camera_path + "\" is not a csv file.");
  if (!path_exists(vs.filepath_to_camera_path))
    die("Error in rendering video: The camera path \"" + vs.filepath_to_camera_path + "\" does not exist.");
  CameraPath path;
  if (!load_path(vs.filepath_to_camera_path, path, err)) die("Error in rendering video: " + err, 101);
  /* times_of_frames (src/rendering.rs:224-238) */
  std::vector<double> times;
  {
    const double min_time = path.pos[0], max_time = path.pos[4 * (path.n - 1)], dt = 1.0 / vs.frame_rate;
    for (double t = min_time; t < max_time; t += dt) times.push_back(t);
  }
  if (!path_exists(c.out) && ::mkdir(c.out.c_str(), 0777) != 0)
    die("Error in rendering video: Could not create video output folder \"" + c.out + "\"");
  const std::string tmp = c.out + "/tmp";
  if (a.resume) { /* opt-in: keep the frames a previous (interrupted) run has written */
    if (!path_exists(tmp) && ::mkdir(tmp.c_str(), 0777) != 0)
      die("Error in rendering video: Could not create tmp output folder \"" + tmp + "\"");
  } else {
    if (path_exists(tmp) && rm_rf(tmp) != 0)
      die("Error in rendering video: Could not remove pre-existing tmp folder \"" + tmp + "\"");
    if (::mkdir(tmp.c_str(), 0777) != 0) die("Error in rendering video: Could not create tmp output folder \"" + tmp + "\"");
  }
  std::printf("Rendering %zu frames...\n", times.size());

  /* cameras of all frames; the reference panics when it reaches the broken last segment, after having
   * written the frames before it: frames up to the first failing one are rendered, then exit 101. */
  std::vector<curvis_camera> cams;
  std::string panic_msg;
  for (size_t k = 0; k < times.size(); ++k) {
    double pos[4], fwd[3], up[3];
    const int prc = path_camera(path, times[k], pos, fwd, up);
    if (prc != 0) {
      panic_msg = prc == 2 ? "index out of bounds in the camera-path interpolation (src/interpolation.rs:76-90)"
                           : "Interpolation time outside the camera path";
      break;
    }
    curvis_camera cam;
    const int rc = curvis_camera_init(&cam, pos, fwd, up, c.cam.focal_length, c.cam.diagonal, c.cam.resolution_x, c.cam.resolution_y);
    if (rc != CURVIS_OK) {
      panic_msg = "Forward and up vectors must not be parallel";
      break;
    }
    cams.push_back(cam);
  }
  const size_t n_frames = cams.size();
  const size_t fbytes = (size_t)c.cam.resolution_x * c.cam.resolution_y * 3;
  std::mutex io_mu;
  std::atomic<int> failed{0};
  FILE *stats_f = a.stats.empty() ? nullptr : std::fopen(a.stats.c_str(), "w");
  WriterPool writers(a.writers);
  /* per-stage host profile (--stats): what the writer threads spent encoding and writing, what each device worker
   * spent inside the render call (GPU kernels + D2H of the batch), waiting for work and handing frames over */
  pngio::EncodeTimes enc_total;
  pngio::EncodeTimes bench_total; /* --encode-bench: the extra encodes, kept apart */
  struct DeviceSummary {
    std::string pci_bus_id;
    int sclk_mhz = -1, power_w = -1;
    size_t frames = 0, batches = 0;
    double render_s = 0, kernel_ms = 0, submit_s = 0, wait_s = 0, busy_s = 0, pool_wait_s = 0;
    double png_ms = 0; /* device PNG front end: HIP-event time of its kernels */
    size_t png_frames = 0, png_fallback_frames = 0, png_regrown = 0; /* png_regrown: batches whose streams needed a larger buffer */
    long long prefetches = 0, prefetch_hits = 0; /* efficient mode: samplers started ahead of their render call / found ready by it */
    double sky_s = 0, sky_bcast_s = 0; /* skies into this device's HBM: all of it / the curvis_ctx_bcast_skies call alone */
    unsigned long long steps = 0;
  };
  /* workers = devices x contexts-per-device; worker r drives device r / contexts with a context of its own (`--mode efficient`
   * spends half of a frame's render call on the host -- the adaptive sampler between its launches --, so a second context
   * on the same GPU fills the gaps) */
  if (a.contexts_auto && a.contexts > 1) {
    /* every extra context costs a start-up of its own (stream, buffers, first launches: ~30-50 ms) and pays only over enough
     * frames: 240 / 960 / 2 400 / 6 000 1080p frames on one GPU took 0.63 / 0.91 / 1.36 / 2.46 s with one context, 0.64 / 0.77 /
     * 1.11 / 1.76 s with two, 0.71 / 0.85 / 1.04 / 1.50 s with four (profiles/round5_cli_startup.txt) */
    const size_t per_device = (n_frames + (size_t)a.devices - 1) / (size_t)a.devices;
    /* round 6 (device-resident sampler, 128 frames per call, the next call's sampler prefetched under this call's kernels): one
     * context no longer waits for the host, and a second one is all that still pays -- 1 / 2 / 3 / 4 contexts read 8 605 / 11 632 /
     * 10 509 / 11 343 frames/s on the orbit and 7 852 / 9 979 / - / 9 027 on the fly-through at 29 970 frames
     * (profiles/round6_eff_contexts_sweep.txt) */
    const int cap = per_device < 600 ? 1 : 2;
    a.contexts = std::min(a.contexts, cap);
  }
  if (batch_auto && a.mode == "efficient" && (n_frames + (size_t)a.devices - 1) / (size_t)a.devices >= 2000) {
    /* long videos: 128 frames per call.  From 48 frames on the library samples on the device (sampler_kernel: one launch per call,
     * no host in the refinement loop), whose latency -- a handful of Euler chains, 5-12 ms -- is paid once per call however many
     * frames share it, and hidden altogether when the worker prefetches the next call's sampler (below): 29 970 frames at 64 / 128 /
     * 256 per call read 5 809 / 7 852 / 7 906 frames/s on the fly-through and - / 8 605 / 8 644 on the orbit with one context
     * (round 5, host-paced sampler at 64 per call: 3 280).  The frames of a call sit together in HBM (128 x 6.2 MB at 1080p; 1 GB at
     * most) next to the PNG front end's scratch (~10 MB per frame); the page-locked batch buffers are sized for streams, not pixels */
    a.batch = (int)std::max<size_t>(4, std::min<size_t>(128, ((size_t)1 << 30) / std::max<size_t>(1, fbytes)));
  }
  const int n_workers = a.devices * a.contexts;
  auto device_of = [&](int rank) { return a.device + rank / a.contexts; };
  std::vector<DeviceSummary> dev_sum((size_t)n_workers);
  clk.mark("camera path, folders");
  std::vector<std::unique_ptr<PinnedPool>> pools((size_t)n_workers); /* destroyed after writers.finish() below */
  const double t_video0 = pngio::now_s();
  /* sky distribution: rank 0 uploads the two textures once; with --sky-broadcast rccl (default for
   * --devices > 1) the other GPUs receive them with ncclBroadcast over xGMI (curvis_ctx_bcast_skies),
   * otherwise every GPU uploads from host memory. */
  std::vector<ncclComm_t> comms;
  const bool share_device = std::getenv("CURVIS_TEST_SHARE_DEVICE") != nullptr; /* test hook: every worker on GPU a.device */
  bool use_rccl = a.sky_broadcast == "rccl" && !share_device && (a.devices > 1 || std::getenv("CURVIS_FORCE_RCCL"));
  if (use_rccl) {
    std::vector<int> devs;
    for (int r = 0; r < a.devices; ++r) devs.push_back(a.device + r);
    comms.resize(a.devices);
    /* one node, one process: RCCL's bootstrap needs no network.  Left to itself it picks the first "real" interface, and on
     * hosts where that one is slow or unroutable communicator set-up was seen to take 6 s (lo: 2.7 s) up to ~80 s */
    ::setenv("NCCL_SOCKET_IFNAME", "lo", 0 /* a value the user has set stays */);
    ::setenv("NCCL_DEBUG", "WARN", 0); /* RCCL's own warnings: a first contact between two devices that fails says why ... */
    ::setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0); /* ... on stderr, its version banner included: stdout stays the reference's */
    const ncclResult_t nrc = ncclCommInitAll(comms.data(), a.devices, devs.data());
    if (nrc != ncclSuccess) {
      const char *last = ncclGetLastError(nullptr);
      const std::string why = std::string("stage ncclCommInitAll over devices ") + std::to_string(a.device) + ".." +
                              std::to_string(a.device + a.devices - 1) + ": " + ncclGetErrorString(nrc) +
                              (last && *last ? std::string("; last RCCL error: ") + last : std::string());
      /* asked for explicitly: a broken xGMI broadcast must not hide behind a silent fallback */
      if (a.sky_broadcast_explicit) die("Error in rendering video: --sky-broadcast rccl: " + why);
      std::fprintf(stderr, "warning: sky broadcast: %s; uploading the skies to every device instead\n", why.c_str());
      comms.clear();
      use_rccl = false;
    }
  }
  /* Work list: frame k belongs to device k mod N (src/rendering.rs:291-316 has no cross-frame state), in batches of
   * --batch frames per launch.  With --resume the frames already on disk are dropped first.  A batch whose render
   * call fails goes to a shared retry queue and is taken by a DIFFERENT device (by the same one when there is only
   * one); after max(2, N) failed attempts the run fails. */
  struct Batch {
    std::vector<size_t> frames;
    int attempts = 0, last_device = -1, last_worker = -1; /* where the last failed attempt ran: GPU and worker thread */
  };
  std::vector<std::deque<Batch>> own((size_t)n_workers);
  size_t n_skipped = 0, n_batches = 0;
  for (int r = 0; r < n_workers; ++r) {
    Batch cur;
    for (size_t k = (size_t)r; k < n_frames; k += (size_t)n_workers) {
      if (a.resume) {
        struct stat sb;
        const std::string file = tmp + "/frame_" + std::to_string(k) + ".png";
        if (::stat(file.c_str(), &sb) == 0 && sb.st_size > 0) {
          ++n_skipped;
          continue;
        }
      }
      cur.frames.push_back(k);
      if (cur.frames.size() == (size_t)a.batch) {
        own[(size_t)r].push_back(cur);
        cur.frames.clear();
      }
    }
    if (!cur.frames.empty()) own[(size_t)r].push_back(cur);
    n_batches += own[(size_t)r].size();
  }
  if (a.resume) std::printf("Resuming: %zu of %zu frames already present in \"%s\"\n", n_skipped, n_frames, tmp.c_str());
  /* PNG front end on the device: with the fast writer (the default) in the modes whose frames of a batch sit together in the
   * context's framebuffer; --encode-bench measures the HOST encoder and therefore keeps it */
  const bool gpu_png = a.gpu_png == "on" ? (a.mode != "direct")
                       : a.gpu_png == "auto" ? (a.png_level < 0 && a.mode != "direct" && a.encode_bench == 0) : false;
  std::mutex q_mu;
  std::condition_variable q_cv;
  std::deque<Batch> retry;
  size_t batches_done = 0;
  const int max_attempts = std::max(2, n_workers);
  /* fault injection for the tests: "rank:n" makes the n-th render call of that worker fail once */
  int fail_rank = -1, fail_call = -1;
  if (const char *fi = std::getenv("CURVIS_TEST_FAIL_BATCH")) std::sscanf(fi, "%d:%d", &fail_rank, &fail_call);
  auto worker = [&](int rank) {
    const double t_w_start = pngio::now_s();
    auto wmark = [&](const char *what) { /* CURVIS_DEBUG_TIMING: where a worker's start-up goes */
      if (clk.on) {
        std::lock_guard<std::mutex> gi(io_mu);
        std::fprintf(stderr, "[curvis timing]   worker %d: %-24s at %8.1f ms of the worker\n", rank, what, (pngio::now_s() - t_w_start) * 1e3);
      }
    };
    curvis_ctx *ctx = make_ctx_bare(share_device ? a.device : device_of(rank), "video");
    wmark("context created");
    DeviceSummary &ds = dev_sum[(size_t)rank];
    {
      char id[64] = {0};
      (void)curvis_ctx_device_status(ctx, id, sizeof id, nullptr, nullptr);
      ds.pci_bus_id = id;
    }
    const double t_worker0 = pngio::now_s();
    if (use_rccl && rank % a.contexts == 0) { /* one context per device takes part in the broadcast; its siblings upload */
      if (rank == 0) upload_skies(ctx, c, "video");
      const double t_b0 = pngio::now_s();
      check(curvis_ctx_bcast_skies(ctx, comms[rank / a.contexts], 0), ctx, "video");
      ds.sky_bcast_s = pngio::now_s() - t_b0;
      /* every GPU checks what arrived over xGMI against the decoded files (head, middle and tail of both textures):
       * a broken broadcast must stop the run, not colour its frames */
      const pngio::Image *sk[2] = {&c.sky1, &c.sky2};
      for (int w = 0; w < 2; ++w) {
        const size_t total = sk[w]->rgba.size(), piece = std::min<size_t>(total, (size_t)1 << 16);
        std::vector<uint8_t> got(piece);
        for (size_t off : {(size_t)0, (total - piece) / 2, total - piece}) {
          check(curvis_ctx_read_sky(ctx, w, off, piece, got.data()), ctx, "video");
          if (std::getenv("CURVIS_TEST_CORRUPT_BCAST")) got[piece / 2] ^= 0x10;  /* test hook: pretend a flipped bit */
          if (std::memcmp(got.data(), sk[w]->rgba.data() + off, piece) != 0)
            die("Error in rendering video: background " + std::to_string(w + 1) + " arrived corrupted on device " +
                std::to_string(device_of(rank)) + " after the RCCL broadcast");
        }
      }
    } else {
      upload_skies(ctx, c, "video");
    }
    ds.sky_s = pngio::now_s() - t_worker0;
    wmark("skies in HBM");
    std::vector<curvis_camera> bc;
    std::vector<uint8_t> rgb_pageable; /* only if page-locked memory could not be had */
    /* one being filled, up to two with the writers.  The pool belongs to video_main's scope: writer jobs hold its
     * buffers (and its mutex, through the deleter) after this worker has returned */
    /* With the device PNG front end a batch buffer receives zlib streams, not pixels: an eighth of the raw size holds what
     * frames rendered from smooth or star-field backgrounds compress to several times over (0.05-0.7 MB per 6.2 MB frame).  A
     * batch that needs more makes the pool hand out larger buffers from then on (below); only frames that do not compress at
     * all take the host encoder, through pageable memory. */
    const size_t pool_bytes = gpu_png ? std::max<size_t>((size_t)8 << 20, (size_t)a.batch * fbytes / 8) : (size_t)a.batch * fbytes;
    size_t pool_first = std::min(pool_bytes, (size_t)a.batch * fbytes);
    if (const char *tb = std::getenv("CURVIS_TEST_STREAM_POOL_BYTES")) /* test hook: start with buffers this small, so that the growth path runs */
      if (gpu_png) pool_first = std::max<size_t>(64, (size_t)std::atoll(tb));
    pools[(size_t)rank].reset(new PinnedPool(pool_first, 3));
    PinnedPool &pool = *pools[(size_t)rank];
    if (pool.buffers() < 2) {
      std::lock_guard<std::mutex> gi(io_mu);
      std::fprintf(stderr, "warning: device %d: no page-locked host memory for the frame buffers, using pageable memory\n", device_of(rank));
    }
    wmark("page-locked buffers");
    /* The writer queue holds 64 jobs; a worker hands over a whole batch at once (128 frames in efficient mode) and waited ~1.2 ms per
     * batch for room, GPU idle.  With ONE context per GPU the bound goes up to two batches (the jobs of pooled batches own no memory:
     * cli_writers.h): 13 300 -> 14 400 frames/s.  With two contexts it stays: measured, the stall is what keeps the two workers out of
     * phase -- 15 600 frames/s with it, 13 400-13 700 without, 14 600 with the render calls taking explicit turns instead
     * (profiles/round6_png_codes_kernel.txt).  CURVIS_WRITER_QUEUE=<n> sets the bound for experiments. */
    if (pool.buffers() >= 2) {
      const char *qe = std::getenv("CURVIS_WRITER_QUEUE");
      if (qe)
        writers.raise_bound((size_t)std::max(1, std::atoi(qe)));
      else if (a.contexts == 1)
        writers.raise_bound((size_t)2 * (size_t)n_workers * (size_t)std::max(1, a.batch));
    }
    /* the zlib streams of a batch travel to its (page-locked) buffer while the NEXT batch renders (option "async_streams"): the
     * writer jobs of that batch are held back here and handed to the pool once the copy engine is done -- right after the next
     * render call has returned, or before this thread leaves.  (Held back, not submitted to wait on a flag: the pool's queue is
     * bounded, and a queue full of waiting jobs would block the very thread that has to raise the flag.) */
    const bool async_streams = gpu_png && pool.buffers() >= 2 && !std::getenv("CURVIS_NO_ASYNC_STREAMS") &&
                               curvis_ctx_set_option(ctx, "async_streams", 1) == CURVIS_OK;
    std::vector<std::function<void()>> deferred;
    auto settle = [&] {
      if (deferred.empty()) return;
      if (curvis_ctx_download_wait(ctx) != CURVIS_OK) {
        std::lock_guard<std::mutex> gi(io_mu);
        std::fprintf(stderr, "Error in rendering video: device %d: the frames' streams did not arrive: %s\n", device_of(rank), curvis_last_error(ctx));
        failed = 1;
      }
      for (auto &job : deferred) writers.submit(std::move(job));
      deferred.clear();
    };
    int calls = 0;
    size_t prefetched_first = (size_t)-1; /* first frame of the batch whose sampler is already in flight (curvis_ctx_prefetch_efficient) */
    std::vector<curvis_camera> pc;
    for (;;) {
      Batch b;
      pc.clear();
      {
        std::unique_lock<std::mutex> g(q_mu);
        for (;;) {
          if (failed || batches_done == n_batches) {
            g.unlock();
            q_cv.notify_all(); /* nobody may sleep on while the others leave */
            ds.busy_s = pngio::now_s() - t_worker0;
            settle();
            wmark("last batch done");
            {
              int64_t v = 0;
              if (curvis_ctx_get_option(ctx, "prefetches", &v) == CURVIS_OK) ds.prefetches = v;
              if (curvis_ctx_get_option(ctx, "prefetch_hits", &v) == CURVIS_OK) ds.prefetch_hits = v;
            }
            curvis_ctx_destroy(ctx);
            wmark("context destroyed");
            return;
          }
          auto it = retry.begin();
          /* a failed batch goes to another GPU; with several contexts per GPU the sibling context of the SAME GPU is not
           * "another device" (a faulty GPU used up two attempts that way).  One GPU only: another context of it. */
          while (it != retry.end() && n_workers > 1 &&
                 (a.devices > 1 ? it->last_device == device_of(rank) : it->last_worker == rank))
            ++it;
          if (it != retry.end()) {
            b = *it;
            retry.erase(it);
            break;
          }
          if (!own[(size_t)rank].empty()) {
            b = own[(size_t)rank].front();
            own[(size_t)rank].pop_front();
            break;
          }
          const double tw = pngio::now_s();
          q_cv.wait_for(g, std::chrono::milliseconds(200)); /* re-checks `failed`: a writer thread sets it without this lock */
          ds.wait_s += pngio::now_s() - tw;
        }
      }
      const size_t nb = b.frames.size();
      bc.clear();
      for (size_t j = 0; j < nb; ++j) bc.push_back(cams[b.frames[j]]);
      /* efficient mode: the sampler of the batch this worker takes NEXT starts now, on a stream of its own -- it runs under this
       * batch's per-pixel kernel, PNG front end, stream download and hand-over, and its render call finds the tables ready */
      if (a.mode == "efficient") {
        {
          std::lock_guard<std::mutex> g(q_mu);
          if (!own[(size_t)rank].empty() && own[(size_t)rank].front().frames[0] != prefetched_first) {
            const Batch &nx = own[(size_t)rank].front();
            for (size_t j = 0; j < nx.frames.size(); ++j) pc.push_back(cams[nx.frames[j]]);
            prefetched_first = nx.frames[0];
          }
        }
        /* the worker's very first batch goes first: batches whose cameras share their radii (an orbit) have equal keys, a render call
         * takes the OLDEST matching prefetch, and only a pipeline primed this way keeps it one batch ahead of the one in flight */
        if (calls == 0 && !pc.empty()) prefetch_frames(ctx, a, c, bc.data(), (uint32_t)nb, c.sim.sampling_convergence_threshold_1);
        /* (Measured and not taken: starting a one-job sampler AFTER this batch's render call, where it fills the gap the PNG front end's
         * host side leaves on the GPU -- no kernel in flight 25 % -> 11 % of the span with one context, but the next render call then
         * waits for its tail: 10 536 -> 8 999 frames/s on the orbit, 9 240 -> 6 602 on the fly-through.) */
        if (!pc.empty()) prefetch_frames(ctx, a, c, pc.data(), (uint32_t)pc.size(), c.sim.sampling_convergence_threshold_1);
      }
      std::shared_ptr<uint8_t> batch_buf;
      uint8_t *rgb_ptr = nullptr;
      if (pool.buffers() >= 2) batch_buf = pool.take(&ds.pool_wait_s);
      if (batch_buf) {
        rgb_ptr = batch_buf.get();
      } else {
        rgb_pageable.resize(nb * fbytes);
        rgb_ptr = rgb_pageable.data();
      }
      curvis_stats st;
      /* src/rendering.rs:305-306: threshold_1 is passed for both thresholds */
      const double t_r0 = pngio::now_s();
      /* with the device PNG front end the pixels stay in HBM (rgb_out = NULL) and the batch buffer receives the frames' zlib
       * streams instead; should they not fit (frames that do not compress: > 1 byte per byte) the raw frames are fetched after
       * all and the host encoder takes them */
      std::vector<size_t> zoff;
      std::vector<uint32_t> zcrc;
      bool streams = false, have_crc = false; /* have_crc: the device also computed the PNG chunks' CRC-32 (two-pass front end) */
      int rc = render_frames(ctx, a, c, bc.data(), (uint32_t)nb, c.sim.sampling_convergence_threshold_1, gpu_png ? nullptr : rgb_ptr, &st);
      settle(); /* the previous batch's streams had this call to arrive under */
      if (rc == CURVIS_OK && gpu_png) {
        zoff.resize(nb + 1);
        double pms = 0.0;
        /* test hook: pretend the streams do not fit (frames that do not compress), so that the fall-back below runs */
        const size_t zcap = std::getenv("CURVIS_TEST_SMALL_PNG_BUFFER") ? (size_t)64 : batch_buf ? pool.bytes_each() : nb * fbytes;
        zcrc.assign(nb, 0u);
        int crc_ok = 0;
        int zrc = curvis_ctx_deflate_frames_crc(ctx, c.cam.resolution_x, c.cam.resolution_y, (uint32_t)nb, rgb_ptr, zcap, zoff.data(), &pms,
                                                zcrc.data(), &crc_ok);
        if (zrc != CURVIS_OK && batch_buf && !std::getenv("CURVIS_TEST_SMALL_PNG_BUFFER")) {
          /* the streams need more than the stream-sized buffer holds (backgrounds that compress badly): the pool hands out
           * larger buffers from now on -- half as much again as this batch needs, the raw size at most -- and the streams are
           * made once more (their kernels take ~0.02 ms per 1080p frame) */
          int64_t need = 0;
          (void)curvis_ctx_get_option(ctx, "last_png_stream_bytes", &need);
          if (need > 0 && (size_t)need > pool.bytes_each() && (size_t)need <= (size_t)a.batch * fbytes) {
            pool.resize(std::min((size_t)a.batch * fbytes, (size_t)need + (size_t)need / 2));
            batch_buf.reset(); /* back to the pool, which lets go of it */
            batch_buf = pool.take(&ds.pool_wait_s);
            if (batch_buf) {
              rgb_ptr = batch_buf.get();
              zrc = curvis_ctx_deflate_frames_crc(ctx, c.cam.resolution_x, c.cam.resolution_y, (uint32_t)nb, rgb_ptr, pool.bytes_each(), zoff.data(),
                                                  &pms, zcrc.data(), &crc_ok);
              ds.png_regrown += 1;
            } else {
              rgb_pageable.resize(nb * fbytes);
              rgb_ptr = rgb_pageable.data();
            }
          }
        }
        have_crc = crc_ok != 0;
        if (zrc == CURVIS_OK) {
          streams = true;
          ds.png_ms += pms;
          ds.png_frames += nb;
        } else {
          if (batch_buf && pool.bytes_each() < nb * fbytes) { /* the stream-sized buffer cannot take the pixels: pageable memory, copied per frame below */
            batch_buf.reset();
            rgb_pageable.resize(nb * fbytes);
            rgb_ptr = rgb_pageable.data();
          }
          rc = curvis_ctx_download(ctx, rgb_ptr, nb * fbytes);
          ds.png_fallback_frames += nb;
        }
      }
      const double batch_call_ms = (pngio::now_s() - t_r0) * 1e3;
      ds.render_s += batch_call_ms * 1e-3;
      const bool injected = rank == fail_rank && calls == fail_call;
      if (injected) rc = CURVIS_E_HIP;
      ++calls;
      if (rc != CURVIS_OK) {
        std::lock_guard<std::mutex> g(q_mu);
        {
          std::lock_guard<std::mutex> gi(io_mu);
          std::fprintf(stderr, "warning: device %d: rendering frames %zu.. failed: %s (code %d), attempt %d of %d%s\n",
                       device_of(rank), b.frames[0], injected ? "injected test fault" : curvis_last_error(ctx), rc,
                       b.attempts + 1, max_attempts, b.attempts + 1 < max_attempts ? "; re-queued" : "");
        }
        b.attempts++;
        b.last_device = device_of(rank);
        b.last_worker = rank;
        if (b.attempts >= max_attempts) {
          std::lock_guard<std::mutex> gi(io_mu);
          std::fprintf(stderr, "Error in rendering video: frames %zu.. could not be rendered on any device\n", b.frames[0]);
          failed = 1;
        } else {
          retry.push_back(b);
        }
        q_cv.notify_all();
        continue;
      }
      /* hand the frames of this batch to the writer pool (each job owns a copy of its frame and ITS statistics:
       * the kernels keep one set of counters per frame of a launch) */
      ds.frames += nb;
      ds.batches += 1;
      ds.kernel_ms += st.kernel_ms;
      ds.steps += st.steps;
      if (ds.batches % 8 == 1) { /* clock and power while the device is under load */
        int sclk = -1, pw = -1;
        (void)curvis_ctx_device_status(ctx, nullptr, 0, &sclk, &pw);
        if (sclk > 0) ds.sclk_mhz = sclk;
        if (pw > 0) ds.power_w = pw;
      }
      const double t_s0 = pngio::now_s();
      const bool hold_back = streams && async_streams && batch_buf; /* the streams are still on their way: see `deferred` above */
      if (streams && async_streams && !batch_buf) (void)curvis_ctx_download_wait(ctx); /* pageable fall-back buffer: the jobs own copies made right here */
      for (size_t j = 0; j < nb; ++j) {
        const size_t k = b.frames[j];
        /* the writer job keeps the batch buffer alive and reads its frame in place; with pageable memory it owns a copy */
        std::shared_ptr<std::vector<uint8_t>> copy;
        const size_t f_off = streams ? zoff[j] : j * fbytes, f_len = streams ? zoff[j + 1] - zoff[j] : fbytes;
        if (!batch_buf) copy = std::make_shared<std::vector<uint8_t>>(rgb_ptr + f_off, rgb_ptr + f_off + f_len);
        const uint8_t *frame = batch_buf ? batch_buf.get() + f_off : copy->data();
        curvis_stats fs;
        std::memset(&fs, 0, sizeof fs);
        if (a.mode == "direct" && j < g_direct_frame_stats.size())
          fs = g_direct_frame_stats[j];
        else
          (void)curvis_ctx_frame_stats(ctx, (uint32_t)j, &fs);
        const double batch_ms = st.kernel_ms;
        const uint32_t frame_crc = streams && have_crc ? zcrc[j] : 0u;
        const bool frame_has_crc = streams && have_crc;
        auto job = [&, k, frame, f_len, streams, batch_buf, copy, fs, nb, rank, batch_ms, batch_call_ms, frame_crc, frame_has_crc] {
          const std::string file = tmp + "/frame_" + std::to_string(k) + ".png";
          const std::string part = file + ".part"; /* written under another name, then renamed: --resume never sees half a file */
          std::string e;
          pngio::EncodeTimes tm, tb;
          bool ok = streams ? pngio::save_zlib_stream_rgb8(part, frame, f_len, c.cam.resolution_x, c.cam.resolution_y, e, &tm, frame_has_crc ? &frame_crc : nullptr)
                            : pngio::save_rgb8(part, frame, c.cam.resolution_x, c.cam.resolution_y, e, a.png_level, &tm);
          if (ok && std::rename(part.c_str(), file.c_str()) != 0) {
            ok = false;
            e = std::strerror(errno);
          }
          for (int rep = 0; ok && rep < a.encode_bench; ++rep) { /* diagnostics: the host's encode capacity with one GPU feeding it */
            std::string e2;
            if (streams)
              (void)pngio::save_zlib_stream_rgb8(part + ".bench", frame, f_len, c.cam.resolution_x, c.cam.resolution_y, e2, &tb, frame_has_crc ? &frame_crc : nullptr);
            else
              (void)pngio::save_rgb8(part + ".bench", frame, c.cam.resolution_x, c.cam.resolution_y, e2, a.png_level, &tb);
          }
          if (a.encode_bench) std::remove((part + ".bench").c_str());
          std::lock_guard<std::mutex> g(io_mu);
          if (!ok) {
            std::fprintf(stderr, "Error in rendering video: Could not save image frame \"%s\" due to error: %s\n", file.c_str(), e.c_str());
            failed = 1;
            q_cv.notify_all(); /* device workers waiting for work must see it */
            return;
          }
          for (auto pr : {std::make_pair(&enc_total, &tm), std::make_pair(&bench_total, &tb)}) {
            pr.first->filter += pr.second->filter;
            pr.first->deflate += pr.second->deflate;
            pr.first->checksum += pr.second->checksum;
            pr.first->write += pr.second->write;
            pr.first->raw_bytes += pr.second->raw_bytes;
            pr.first->file_bytes += pr.second->file_bytes;
            pr.first->frames += pr.second->frames;
          }
          std::printf("Rendering frame %zu/%zu...\n", k + 1, times.size());
          if (stats_f)
            std::fprintf(stats_f, "{\"frame\": %zu, \"time\": %.17g, \"device\": %d, \"mode\": \"%s\", \"rays\": %llu, \"steps\": %llu, \"n_pos\": %llu, \"n_neg\": %llu, \"n_none\": %llu, \"n_oob\": %llu, \"kernel_ms\": %.4f, \"mray_steps_per_s\": %.1f, \"batch_frames\": %zu, \"batch_kernel_ms\": %.4f, \"batch_call_ms\": %.4f}\n",
                         k, times[k], device_of(rank), a.mode.c_str(), (unsigned long long)fs.rays, (unsigned long long)fs.steps,
                         (unsigned long long)fs.n_pos, (unsigned long long)fs.n_neg, (unsigned long long)fs.n_none,
                         (unsigned long long)fs.n_oob, fs.kernel_ms, fs.kernel_ms > 0.0 ? (double)fs.steps / fs.kernel_ms / 1e3 : 0.0, nb,
                         batch_ms, batch_call_ms);
        };
        if (hold_back)
          deferred.emplace_back(std::move(job));
        else
          writers.submit(std::move(job));
      }
      ds.submit_s += pngio::now_s() - t_s0; /* frame copies + time blocked on a full writer queue */
      {
        std::lock_guard<std::mutex> g(q_mu);
        ++batches_done;
      }
      q_cv.notify_all();
    }
  };
  std::vector<std::thread> th;
  for (int r = 0; r < n_workers; ++r) th.emplace_back(worker, r);
  for (auto &t : th) t.join();
  for (ncclComm_t cm : comms) ncclCommDestroy(cm);
  const double t_workers_done = pngio::now_s();
  clk.mark("device workers (contexts, skies, frames)");
  writers.finish();
  clk.mark("writer drain");
  pools.clear(); /* every writer job is done: the page-locked buffers can go */
  const double t_video1 = pngio::now_s();
  clk.mark("page-locked buffers released");
  if (stats_f) std::fclose(stats_f);
  if (!a.stats.empty()) { /* <stats>.summary.json + a table: who rendered what at which clock, where the host's time went */
    const double wall = t_video1 - t_video0;
    size_t total_frames = 0;
    for (const DeviceSummary &d : dev_sum) total_frames += d.frames;
    std::string js = "{\"frames\": " + std::to_string(total_frames) + ", \"wall_s\": " + std::to_string(wall) +
                     ", \"frames_per_s\": " + std::to_string(wall > 0 ? total_frames / wall : 0.0) +
                     ", \"writers\": " + std::to_string(a.writers) + ", \"png_level\": " + std::to_string(a.png_level) +
                     ", \"gpu_png\": " + (gpu_png ? "true" : "false") +
                     ", \"writer_drain_s\": " + std::to_string(t_video1 - t_workers_done);
    { /* how the two textures reached the devices: the slowest device's time; for RCCL the broadcast call alone as well
       * (root: upload first, then header + 2 x ncclBroadcast; the first collective of a communicator carries its set-up) */
      double sky_max = 0, bcast_max = 0;
      for (const DeviceSummary &d : dev_sum) {
        sky_max = std::max(sky_max, d.sky_s);
        bcast_max = std::max(bcast_max, d.sky_bcast_s);
      }
      const double sky_bytes = (double)c.sky1.rgba.size() + (double)c.sky2.rgba.size();
      char buf[384];
      std::snprintf(buf, sizeof buf,
                    ", \"sky_distribution\": {\"via\": \"%s\", \"bytes\": %.0f, \"seconds\": %.4f, \"broadcast_call_s\": %.4f, "
                    "\"sky_broadcast_gbps\": %.2f}",
                    use_rccl ? "rccl: ncclCommInitAll + curvis_ctx_bcast_skies" : "upload to every device", sky_bytes, sky_max, bcast_max,
                    bcast_max > 0 ? sky_bytes / bcast_max / 1e9 : 0.0);
      js += buf;
    }
    if (a.devices > 1 && !share_device) { /* how the GPUs are connected: what sky_broadcast_gbps has to be read against */
      js += ", \"links\": [";
      bool first = true;
      for (int i = 0; i < a.devices; ++i)
        for (int k = i + 1; k < a.devices; ++k) {
          int lt = -1, hops = -1, peer = -1, perf = -1, atom = -1;
          (void)curvis_device_link(a.device + i, a.device + k, &lt, &hops, &peer, &perf, &atom);
          const char *name = lt == 4 ? "xGMI" : lt == 2 ? "PCIe" : lt == 0 ? "same device" : "unknown";
          char buf[256];
          std::snprintf(buf, sizeof buf, "%s{\"a\": %d, \"b\": %d, \"link\": \"%s\", \"link_type\": %d, \"hops\": %d, \"peer_access\": %d, "
                        "\"performance_rank\": %d}", first ? "" : ", ", a.device + i, a.device + k, name, lt, hops, peer, perf);
          js += buf;
          if (i == 0) std::printf("link device %d <-> %d: %s, %d hop(s), peer access %d\n", a.device + i, a.device + k, name, hops, peer);
          first = false;
        }
      js += "]";
    }
    js += ", \"devices\": [";
    std::printf("device  pci_bus_id     frames  kernel ms/frame  render-call ms/frame  fps    sclk MHz  power W  wait s  hand-over s\n");
    for (size_t r = 0; r < dev_sum.size(); ++r) {
      const DeviceSummary &d = dev_sum[r];
      const double kf = d.frames ? d.kernel_ms / d.frames : 0.0, rf = d.frames ? d.render_s * 1e3 / d.frames : 0.0;
      std::printf("%-7zu %-14s %-7zu %-16.3f %-21.3f %-6.1f %-9d %-8d %-7.2f %.2f\n", (size_t)device_of((int)r), d.pci_bus_id.c_str(), d.frames, kf, rf,
                  d.busy_s > 0 ? d.frames / d.busy_s : 0.0, d.sclk_mhz, d.power_w, d.wait_s, d.submit_s);
      char buf[1024];
      std::snprintf(buf, sizeof buf,
                    "%s{\"device\": %zu, \"pci_bus_id\": \"%s\", \"frames\": %zu, \"batches\": %zu, \"kernel_ms_per_frame\": %.4f, "
                    "\"render_call_ms_per_frame\": %.4f, \"frames_per_s\": %.2f, \"mray_steps_per_s\": %.1f, \"sclk_mhz\": %d, \"power_w\": %d, "
                    "\"wait_s\": %.3f, \"hand_over_s\": %.3f, \"buffer_wait_s\": %.3f, \"busy_s\": %.3f, \"gpu_png_frames\": %zu, "
                    "\"gpu_png_kernel_ms_per_frame\": %.4f, \"gpu_png_fallback_frames\": %zu, \"gpu_png_buffer_regrown\": %zu, "
                    "\"sampler_prefetches\": %lld, \"sampler_prefetch_hits\": %lld}",
                    r ? ", " : "", (size_t)device_of((int)r), d.pci_bus_id.c_str(), d.frames, d.batches, kf, rf, d.busy_s > 0 ? d.frames / d.busy_s : 0.0,
                    d.kernel_ms > 0 ? (double)d.steps / d.kernel_ms / 1e3 : 0.0, d.sclk_mhz, d.power_w, d.wait_s, d.submit_s, d.pool_wait_s, d.busy_s,
                    d.png_frames, d.png_frames ? d.png_ms / d.png_frames : 0.0, d.png_fallback_frames, d.png_regrown, d.prefetches, d.prefetch_hits);
      js += buf;
    }
    js += "]";
    for (auto pr : {std::make_pair("encode", &enc_total), std::make_pair("encode_bench", &bench_total)}) {
      const pngio::EncodeTimes &t = *pr.second;
      if (!t.frames) continue;
      const double per = 1e3 / (double)t.frames, cpu = t.filter + t.deflate + t.checksum + t.write;
      char buf[640];
      std::snprintf(buf, sizeof buf,
                    ", \"%s\": {\"frames\": %zu, \"filter_ms\": %.3f, \"deflate_ms\": %.3f, \"checksum_ms\": %.3f, \"write_ms\": %.3f, "
                    "\"thread_ms_per_frame\": %.3f, \"raw_mb_per_frame\": %.3f, \"file_mb_per_frame\": %.3f, \"mb_per_s_per_thread\": %.1f, "
                    "\"frames_per_s_per_thread\": %.1f}",
                    pr.first, t.frames, t.filter * per, t.deflate * per, t.checksum * per, t.write * per, cpu * per, t.raw_bytes / 1e6 / t.frames,
                    t.file_bytes / 1e6 / t.frames, cpu > 0 ? t.raw_bytes / 1e6 / cpu : 0.0, cpu > 0 ? t.frames / cpu : 0.0);
      js += buf;
      std::printf("%s: %zu frames, per frame and writer thread: filter %.2f + deflate %.2f + checksums %.2f + file write %.2f = %.2f ms "
                  "(%.0f MB/s, %.1f frames/s per thread), %.2f -> %.2f MB\n",
                  pr.first, t.frames, t.filter * per, t.deflate * per, t.checksum * per, t.write * per, cpu * per, cpu > 0 ? t.raw_bytes / 1e6 / cpu : 0.0,
                  cpu > 0 ? t.frames / cpu : 0.0, t.raw_bytes / 1e6 / t.frames, t.file_bytes / 1e6 / t.frames);
    }
    js += "}\n";
    std::printf("video: %zu frames in %.2f s wall = %.1f frames/s (%d writer threads, png level %d; writers still busy %.2f s after the last render)\n",
                total_frames, wall, wall > 0 ? total_frames / wall : 0.0, a.writers, a.png_level, t_video1 - t_workers_done);
    if (FILE *sf = std::fopen((a.stats + ".summary.json").c_str(), "w")) {
      std::fputs(js.c_str(), sf);
      std::fclose(sf);
    }
  }
  if (failed) return 1;
  if (!panic_msg.empty()) {
    std::fprintf(stderr, "thread 'main' panicked: %s (frame %zu of %zu)\n", panic_msg.c_str(), n_frames, times.size());
    return 101;
  }
  return 0;
}

}  // namespace

#endif /* CURVIS_CLI_VIDEO_H */
