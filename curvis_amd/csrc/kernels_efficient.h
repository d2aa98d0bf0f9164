/* kernels_efficient.h -- device side of render_image_efficient (E1-E3), direct mode, trajectories and the math self-test.
 * Part of the ONE translation unit curvis_hip.hip (included there, nowhere else). */
#pragma once

namespace {

/* ------------------------------------------------------------------------------------------------
 * Efficient renderer (render_image_efficient, src/systems.rs:333-527): the CLI's variant. */

struct EscapeAngleParams {
  cvk::MetricParams metric;
  const double *alpha; /* n */
  const double *l_cam; /* n: radial coordinate of the camera the sample belongs to */
  double *angle;       /* n: escape angle, NaN when not escaped */
  double *space;       /* n: +1 / -1, NaN when not escaped */
  unsigned *steps;     /* n */
  int *status;         /* n: escape code, or ESC_PANIC */
  unsigned n;
  unsigned max_iter;
  double max_radius, delta;
  int fast_ok;
};

/* compute_escape_angle (src/systems.rs:203-261) for ONE alpha on one lane: photon at (0, l, pi/2, 0) with tangent direction
 * (cos a, 0, sin a), Euler loop WITH phi, world direction, angle.  Every lane of the wave that is active here must have entered
 * together (the step counter is wave-uniform). */
template <int KIND, bool FAST>
__device__ __forceinline__ void escape_angle_lane(const cvk::MetricParams &M, double alpha, double l_cam, unsigned max_iter, double max_radius,
                                                  double delta, int fast_ok, double &angle, double &space, unsigned &steps_out, int &status) {
  double sa, ca;
  cv_sincos(alpha, &sa, &ca);
  const double pos[4] = {0.0, l_cam, CV_PI / 2.0, 0.0};
  cvk::Ray q;
  cvk::ray_init_dir<KIND>(M, pos, ca, 0.0, sa, q);
  const bool lane_ok = FAST && fast_ok && cvk::ray_fast_ok(q);
  /* same loop shape as geodesic_static: wave-uniform step counter (all lanes start together), one escape compare,
   * the per-lane step count captured under a scalar branch in the iterations in which some lane escapes */
  unsigned steps = max_iter;
  int code = cvk::CODE_NONE;
  if (max_iter != 0) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned k = 0;
    for (;;) {
      ++k;
      one_step<KIND, true, FAST, true>(M, delta, q, lane_ok); /* equatorial photons: see ray_step_fast */
      const bool esc = ray_escaped(q.l, max_radius);
      const unsigned long long em = __builtin_amdgcn_ballot_w64(esc);
      if (em) {
        unsigned kv;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kv) : "s"(k));
        if ((em >> lane) & 1ull) steps = kv;
      }
      if (esc) break;
      if (k >= max_iter) break;
    }
    if (ray_escaped(q.l, max_radius)) code = escape_code(q.l);
  } else {
    steps = 0;
  }
  const double nan = __builtin_nan("");
  angle = nan;
  space = nan;
  status = code;
  if (code != cvk::CODE_NONE) {
    if (cvk::escape_angle_of<KIND>(M, q, angle)) {
      space = (code == cvk::CODE_POS) ? 1.0 : -1.0;
    } else {
      angle = nan;
      status = cvk::ESC_PANIC;
    }
  }
  steps_out = steps;
}

/* K2: compute_escape_angle for a batch of alphas (the host-paced sampler's launches: efficient_host.h eval_escape_batch) */
template <int KIND, bool FAST>
__global__ __launch_bounds__(64) void escape_angle_kernel(const EscapeAngleParams P) {
  __shared__ MathTablesLds<KIND> s_tab;
  cvk::MetricParams M = P.metric;
  load_math_tables<KIND>(s_tab, M);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  double angle, space;
  unsigned steps;
  int status;
  escape_angle_lane<KIND, FAST>(M, P.alpha[i], P.l_cam[i], P.max_iter, P.max_radius, P.delta, P.fast_ok, angle, space, steps, status);
  P.angle[i] = angle;
  P.space[i] = space;
  P.steps[i] = steps;
  P.status[i] = status;
}

/* K2': the reference's adaptive sampler (doubly_sample_function, src/sampling.rs:46-195, called at src/systems.rs:437-486) as the
 * job of ONE WORKGROUP, start to finish, with no host in the loop.  Table, plan and counters live in LDS (cv_sampler_dev.h
 * SamplerState, 82 KB); lane 0 plans a round (the walk over the triples is sequential by definition: `i += 1` or `i += 2`
 * depending on the triple just seen), every lane integrates one of the round's new alphas -- a 2000-step dependency chain per
 * round is what a job costs, however many lanes take part --, lane 0 assembles, and round again.  When the table has settled, all
 * lanes write it out together with interp 1.0.3's slopes and intercepts over it (the per-pixel kernel's inputs: m = dy / dx,
 * c = y - x m, IEEE division and unfused multiply-subtract like the host's cvs::interp_tables) and the sample count of every
 * frame that uses this job.  Frames whose cameras share the radial coordinate l share a job: compute_escape_angle(l, alpha) sees
 * nothing else of the camera (src/systems.rs:473-485), so their tables are the same table.
 * One job per workgroup, one workgroup per CU (LDS): a launch of up to 256 jobs runs them all at once; what bounds a launch is
 * the latency of the longest job (rounds x ~2000 steps x ~600 cycles), not the number of jobs. */
struct SamplerParams {
  cvk::MetricParams metric;
  const double *l_cam;            /* n_jobs */
  unsigned n_jobs, n_frames;
  const unsigned *job_of_frame;   /* n_frames */
  unsigned *tab_n;                /* n_frames: samples in the table of the frame's job (read by efficient_pixel_kernel) */
  unsigned n0, max_iterations, max_iter;
  double a_min, a_max, thr1, thr2, max_radius, delta;
  int fast_ok;
  double *sx, *se, *ss, *m_e, *c_e, *m_s, *c_s; /* n_jobs x kSamplerCap each */
  unsigned *grid;                 /* n_jobs x (kInterpGrid + 1): bucket grid over sx (cv_efficient.h interp_index_grid) */
  cvk::SamplerResult *res;        /* n_jobs */
  /* evaluation cache of every job (cv_sampler_dev.h SpecTable), n_jobs x kSpecSlots each; keys preset to kSpecEmpty by the host */
  unsigned long long *spec_key;
  double *spec_e, *spec_s;
  unsigned *spec_steps;
  int *spec_status;
  int speculate;                  /* 0: every round integrates exactly its pending points (no subtree, table still used) */
};
constexpr unsigned kSamplerThreads = 512; /* two waves per SIMD: what a lone chain leaves idle anyway (a step is ~85 dependent FP64
                                             instructions of ~8 cycles latency, issued in 4) */

template <int KIND, bool FAST>
__global__ __launch_bounds__(kSamplerThreads) void sampler_kernel(const SamplerParams P) {
  __shared__ MathTablesLds<KIND> s_tab;
  __shared__ cvk::SamplerState S;
  __shared__ int s_panic;
  cvk::MetricParams M = P.metric;
  if (threadIdx.x == 0u) {
    cvk::sampler_reset(S);
    s_panic = 0;
  }
  load_math_tables<KIND>(s_tab, M); /* ends with a barrier */
  const unsigned job = blockIdx.x;
  const double l_cam = P.l_cam[job];
  cvk::SpecTable T;
  {
    const size_t so = (size_t)job * cvk::kSpecSlots;
    T.key = P.spec_key + so;
    T.e = P.spec_e + so;
    T.s = P.spec_s + so;
    T.steps = P.spec_steps + so;
    T.status = P.spec_status + so;
  }
  for (;;) {
    /* plan the round: cv_sampler_dev.h -- only the walk over the triples is one lane's work */
    if (threadIdx.x == 0u) {
      S.go = cvk::sampler_plan_begin(S, P.n0, P.max_iterations);
      S.n_miss = S.n_eval = 0u;
    }
    __syncthreads();
    const int mode = S.go; /* uniform: read between this barrier and the next, written only after the one that follows */
    if (mode == cvk::PLAN_STOP) break;
    if (mode == cvk::PLAN_GRID) {
      for (unsigned i = threadIdx.x; i < P.n0; i += kSamplerThreads) cvk::sampler_grid_point(S, i, P.n0, P.a_min, P.a_max);
    } else {
      for (unsigned i = threadIdx.x; i + 2u < S.n; i += kSamplerThreads) cvk::sampler_flag(S, i, P.thr1, P.thr2);
      __syncthreads();
      if (threadIdx.x == 0u) S.go = cvk::sampler_walk(S) ? cvk::PLAN_REFINE : cvk::PLAN_STOP;
      __syncthreads();
      if (S.go == cvk::PLAN_STOP) break; /* out of room: uniform */
      for (unsigned v = threadIdx.x; v < S.n_vis; v += kSamplerThreads) cvk::sampler_place(S, v);
    }
    __syncthreads();
    const unsigned np = S.n_pend;
    /* phase 1: what the table already holds goes straight into the new table; the rest is noted */
    for (unsigned t = threadIdx.x; t < np; t += kSamplerThreads) {
      int panic = 0;
      if (!cvk::sampler_take(S, T, t, panic)) S.miss[atomicAdd(&S.n_miss, 1u)] = (unsigned short)t;
      if (panic) s_panic = 1;
    }
    __syncthreads();
    const unsigned nm = S.n_miss;
    if (nm) { /* uniform */
      /* phase 2: queue the missing points (they always fit), then the subtrees below their intervals on the lanes one chain
       * would leave idle */
      for (unsigned m = threadIdx.x; m < nm; m += kSamplerThreads) cvk::sampler_want(S, T, S.pend_a[S.miss[m]], true);
      __syncthreads();
      if (P.speculate && !S.overflow) {
        const unsigned depth = S.started ? cvk::sampler_spec_depth(nm, kSamplerThreads) : 0u;
        const unsigned grid_depth = (!S.started && kSamplerThreads > nm) ? cvk::sampler_spec_depth(nm - 1u, kSamplerThreads - nm) : 0u;
        for (unsigned m = threadIdx.x; m < nm; m += kSamplerThreads) cvk::sampler_speculate(S, T, m, depth, grid_depth);
      }
      __syncthreads();
      /* phase 3: integrate */
      const unsigned ne = S.n_eval < cvk::kSpecEvalCap ? S.n_eval : cvk::kSpecEvalCap;
      for (unsigned base = 0; base < ne; base += kSamplerThreads) {
        const unsigned k = base + threadIdx.x;
        if (k < ne) {
          double angle, space;
          unsigned steps;
          int status;
          escape_angle_lane<KIND, FAST>(M, S.eval_a[k], l_cam, P.max_iter, P.max_radius, P.delta, P.fast_ok, angle, space, steps, status);
          const unsigned slot = S.eval_slot[k];
          CV_SPEC_ST(&T.e[slot], angle);
          CV_SPEC_ST(&T.s[slot], space);
          CV_SPEC_ST(&T.steps[slot], steps);
          CV_SPEC_ST(&T.status[slot], status);
        }
      }
      if (threadIdx.x == 0u) {
        S.eval_phases += (ne + kSamplerThreads - 1u) / kSamplerThreads;
        S.evaluated += ne;
      }
      __syncthreads();
      /* phase 4: the missing points are in the table now */
      for (unsigned m = threadIdx.x; m < nm; m += kSamplerThreads) {
        int panic = 0;
        if (!cvk::sampler_take(S, T, S.miss[m], panic)) S.overflow = 1; /* cannot happen unless phase 2 ran out of room */
        if (panic) s_panic = 1;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0u) {
      if (S.overflow) S.finished = 1;
      cvk::sampler_consume(S, P.max_iterations);
    }
    __syncthreads();
  }
  /* the finished table and the interpolation tables over it; slot max(n, 1) - 1 is padding (zeros), as on the host */
  const unsigned n = S.n, cur = S.cur;
  const size_t o = (size_t)job * cvk::kSamplerCap;
  const double *A = S.a[cur], *E = S.e[cur], *Z = S.s[cur];
  const unsigned slots = n ? n : 1u;
  for (unsigned i = threadIdx.x; i < slots; i += kSamplerThreads) {
    const bool have = i < n;
    P.sx[o + i] = have ? A[i] : 0.0;
    P.se[o + i] = have ? E[i] : 0.0;
    P.ss[o + i] = have ? Z[i] : 0.0;
    double me = 0.0, ce = 0.0, ms = 0.0, cs = 0.0;
    if (n == 1u) { /* interp 1.0.3 with one point: the "intercept" is y[0] */
      ce = E[0];
      cs = Z[0];
    } else if (i + 1u < n) {
      const double dx = A[i + 1u] - A[i];
      me = (E[i + 1u] - E[i]) / dx;
      ce = E[i] - A[i] * me;
      ms = (Z[i + 1u] - Z[i]) / dx;
      cs = Z[i] - A[i] * ms;
    }
    P.m_e[o + i] = me;
    P.c_e[o + i] = ce;
    P.m_s[o + i] = ms;
    P.c_s[o + i] = cs;
  }
  for (unsigned i = threadIdx.x; i <= n; i += kSamplerThreads) cvk::interp_grid_fill(A, n, i, P.grid + (size_t)job * (cvk::kInterpGrid + 1u));
  for (unsigned f = threadIdx.x; f < P.n_frames; f += kSamplerThreads)
    if (P.job_of_frame[f] == job) P.tab_n[f] = n;
  if (threadIdx.x == 0u) {
    cvk::SamplerResult r;
    r.n = n;
    r.rounds = S.rounds;
    r.calls = S.calls;
    r.steps = S.steps;
    r.warned = S.warned;
    r.status = S.overflow ? cvk::SAMPLER_OVERFLOW : (S.panicked || s_panic) ? cvk::SAMPLER_PANIC : cvk::SAMPLER_OK;
    r.eval_phases = S.eval_phases;
    r.evaluated = S.evaluated;
    P.res[job] = r;
  }
}

struct EfficientPixelParams {
  cvk::SkyParams sky[2];
  const cvk::CameraParams *cams;      /* n_frames */
  const cvk::EfficientFrame *frames;  /* n_frames */
  const unsigned *tab_off;            /* n_frames: offset of the frame's tables in sx/m/c */
  const unsigned *tab_n;              /* n_frames: number of samples */
  const unsigned *grid_off;           /* n_frames: offset of the frame's bucket grid in `grid` */
  const unsigned *grid;               /* kInterpGrid + 1 entries per table */
  const double *sx, *m_e, *c_e, *m_s, *c_s;
  unsigned n_frames, W, H;
  unsigned char *fb;
  FrameCounters counters;
  cvk::PixelRecips recips;            /* cv_device.h: reciprocals of the call's constant denominators (ensure_pixel_recips) */
  unsigned long long w_magic;         /* floor(2^64 / W) + 1: pixel index / W = the high 64 bits of index x w_magic, exactly, for every
                                         index < 2^32 (index e / (W 2^64) < 1 / W with e = w_magic W - 2^64 <= W); 0 for W = 1 */
};

/* y = recip_chain(d) for a handful of constants: the first half of the device's own f64 division, run ON the device so that the
 * pixel kernel's shared quotients are the compiler's quotients operation for operation (cv_device.h) */
__global__ void recip_chain_kernel(const double *d, double *y, unsigned n) {
  if (threadIdx.x < n) y[threadIdx.x] = cvk::recip_chain(d[threadIdx.x]);
}

/* K3: steps 2, 4, 5 of render_image_efficient + sky lookup, one thread per pixel.
 *
 * Grid: (groups of 256 pixels, frames) -- the frame index is uniform, the camera and the frame's constants come in through
 * scalar loads.  Statistics (escaped to +l / -l, black, texel index clamped) are reduced per WORKGROUP -- ballots and
 * population counts per wave, LDS across the four waves -- and then added to the frame's counters by one lane; the counters are
 * spread over 64 cache lines per frame.  Round 5 found the first version, in which every WAVE added its counts straight to one of
 * 8 lines per frame, spending 42 % of its time in those atomics (a million waves per 32-frame launch: without the statistics the
 * same kernel ran 0.057 instead of 0.099 ms per 1080p frame) -- and this kernel is more than half of the GPU time of
 * `curvis video` in the reference's default mode.  (Walking several groups per workgroup to reduce even less often was worse: the
 * compiler hoists the frame's constants out of the loop into 155 VGPRs -- 3 waves per SIMD instead of 8 --, and as a
 * non-inlined call the body spills.)  The RGB8 bytes of a wave's 64 pixels are transposed through LDS and stored as 48 dwords. */
__global__ __launch_bounds__(256) void efficient_pixel_kernel(const EfficientPixelParams P) {
  __shared__ __attribute__((aligned(16))) unsigned char s_rgb[256 * 3];
  __shared__ unsigned s_cnt[5];
  const unsigned f = blockIdx.y;
  const unsigned npix = P.W * P.H; /* < 2^32: checked on the host */
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  const bool valid = pix < npix;
  if (threadIdx.x < 5u) s_cnt[threadIdx.x] = 0u;
  bool pos = false, neg = false, none = false, oob = false;
  unsigned texel = 0xFF000000u;
  if (valid) {
    const unsigned py = P.w_magic ? (unsigned)__umul64hi((unsigned long long)pix, P.w_magic) : pix / P.W, px = pix - py * P.W;
    const unsigned off = P.tab_off[f], n = P.tab_n[f];
    double fin[3], space;
    cvk::efficient_pixel<true>(P.cams[f], P.frames[f], px, py, P.sx + off, P.m_e + off, P.c_e + off, P.m_s + off, P.c_s + off, n, fin, space,
                               &P.recips, P.grid + P.grid_off[f]);
    /* match escape_space { 1.0 => ..., -1.0 => ..., _ => black }.  One sky after the other, each under its own branch: the lanes of
     * a wave nearly always look at the same sky, and then its rotation, size and texel pointer are scalar operands of that one
     * pass -- selecting them per lane cost 22 v_cndmask and the VGPRs to hold the result */
    pos = (space == 1.0);
    neg = (space == -1.0);
    none = !(pos || neg);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 0 ? pos : neg) {
        const cvk::SkyParams &S = P.sky[k];
        unsigned tx, ty;
        cvk::sky_indices<true>(S, fin[0], fin[1], fin[2], tx, ty, P.recips.y_pi, P.recips.y_two_pi);
        if (tx >= S.w || ty >= S.h) oob = true;
        if (tx >= S.w) tx = S.w - 1;
        if (ty >= S.h) ty = S.h - 1;
        texel = S.texels[(size_t)ty * S.w + tx];
      }
    }
  }
  unsigned char *fb = P.fb + (size_t)f * npix * 3u;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
  /* dword stores need the frame to start on a 4-byte boundary (every frame of a batch does unless W*H*3 is not a multiple of 4) */
  if (vm == ~0ull && ((((size_t)f * npix * 3u) & 3u) == 0u)) {
    unsigned char *sw = s_rgb + wave * 192u;
    sw[3u * lane + 0u] = (unsigned char)(texel & 0xFF);
    sw[3u * lane + 1u] = (unsigned char)((texel >> 8) & 0xFF);
    sw[3u * lane + 2u] = (unsigned char)((texel >> 16) & 0xFF);
    __builtin_amdgcn_wave_barrier(); /* LDS operations of one wave complete in order; the buffer is this wave's alone */
    if (lane < 48u) reinterpret_cast<unsigned *>(fb + (size_t)(pix - lane) * 3u)[lane] = reinterpret_cast<const unsigned *>(sw)[lane];
  } else if (valid) {
    unsigned char *dst = fb + (size_t)pix * 3u;
    dst[0] = (unsigned char)(texel & 0xFF);
    dst[1] = (unsigned char)((texel >> 8) & 0xFF);
    dst[2] = (unsigned char)((texel >> 16) & 0xFF);
  }
  /* statistics: ballots per wave, LDS across the waves, one lane of the workgroup for the frame's counters */
  const unsigned n_pos = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(pos)), n_neg = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(neg));
  const unsigned n_none = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(none)), n_oob = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(oob));
  __syncthreads(); /* s_cnt cleared */
  if (lane == 0u && vm) {
    atomicAdd(&s_cnt[0], (unsigned)__popcll(vm));
    if (n_pos) atomicAdd(&s_cnt[1], n_pos);
    if (n_neg) atomicAdd(&s_cnt[2], n_neg);
    if (n_none) atomicAdd(&s_cnt[3], n_none);
    if (n_oob) atomicAdd(&s_cnt[4], n_oob);
  }
  __syncthreads();
  if (threadIdx.x == 0u && s_cnt[0]) {
    unsigned long long *c = frame_counter_line(P.counters, f);
    atomicAdd(&c[FC_RAYS], (unsigned long long)s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&c[FC_POS], (unsigned long long)s_cnt[1]);
    if (s_cnt[2]) atomicAdd(&c[FC_NEG], (unsigned long long)s_cnt[2]);
    if (s_cnt[3]) atomicAdd(&c[FC_NONE], (unsigned long long)s_cnt[3]);
    if (s_cnt[4]) atomicAdd(&c[FC_OOB], (unsigned long long)s_cnt[4]);
  }
}

/* "direct" mode (NOT in the reference; SURVEY 8f N1 names it as a quality option): what render_image_efficient
 * approximates by sampling + interpolation, computed exactly -- compute_escape_angle(l_cam, alpha) for the alpha of
 * EVERY pixel (src/systems.rs:203-261 on the result of :405-433), then step 5 (:498-523) with that escape angle and
 * space.  One thread per pixel, 8x8 tiles per wave (neighbouring alphas: coherent step counts); every photon lives in
 * the equatorial plane, so the loop is the sampling kernel's (phi integrated, equatorial step form). */
struct DirectParams {
  cvk::MetricParams metric;
  cvk::SkyParams sky[2];
  cvk::CameraParams cam;
  cvk::EfficientFrame frame;
  unsigned W, H, tiles_x, tiles_y;
  unsigned long long total_rays; /* tiles_x * tiles_y * 64 */
  unsigned max_iter;
  double max_radius, delta;
  int fast_ok;
  unsigned char *fb;
  FrameCounters counters;
};

template <int KIND, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KIND == cvk::METRIC_INTERSTELLAR ? 4 : 6)))
void direct_kernel(const DirectParams P) {
  __shared__ MathTablesLds<KIND> s_tab;
  cvk::MetricParams M = P.metric;
  load_math_tables<KIND>(s_tab, M);
  const unsigned long long id = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned tile = (unsigned)(id >> 6), k6 = (unsigned)id & 63u;
  const unsigned tyi = tile / P.tiles_x, txi = tile - tyi * P.tiles_x;
  const unsigned px = txi * 8u + (k6 & 7u), py = tyi * 8u + (k6 >> 3);
  const bool valid = id < P.total_rays && px < P.W && py < P.H;
  unsigned steps = 0, pos = 0, neg = 0, none = 0, oob = 0;
  if (valid) {
    double alpha, axis[3];
    cvk::efficient_pixel_geometry(P.cam, P.frame, px, py, alpha, axis);
    double sa, ca;
    cv_sincos(alpha, &sa, &ca);
    const double p4[4] = {0.0, P.cam.pos[1], CV_PI / 2.0, 0.0};
    cvk::Ray q;
    cvk::ray_init_dir<KIND>(M, p4, ca, 0.0, sa, q);
    const bool lane_ok = FAST && P.fast_ok && cvk::ray_fast_ok(q);
    int code = cvk::CODE_NONE;
    /* a per-lane loop (lanes outside the frame are idle from the start, so the counter is not wave-uniform) */
    for (unsigned k = 0; k < P.max_iter; ++k) {
      one_step<KIND, true, FAST, true>(M, P.delta, q, lane_ok);
      ++steps;
      if (ray_escaped(q.l, P.max_radius)) {
        code = escape_code(q.l);
        break;
      }
    }
    unsigned texel = 0xFF000000u; /* NotEscaped / undefined tangent rotation: black */
    double angle;
    if (code != cvk::CODE_NONE && cvk::escape_angle_of<KIND>(M, q, angle)) {
      cvk::efficient_pixel_geometry(P.cam, P.frame, px, py, alpha, axis); /* again: not kept live across the loop */
      double fin[3];
      cvk::efficient_final_direction(P.frame, axis, angle, fin);
      const cvk::SkyParams &S = P.sky[code == cvk::CODE_POS ? 0 : 1];
      unsigned tx, ty;
      cvk::sky_indices(S, fin[0], fin[1], fin[2], tx, ty);
      if (tx >= S.w || ty >= S.h) oob = 1;
      if (tx >= S.w) tx = S.w - 1;
      if (ty >= S.h) ty = S.h - 1;
      texel = S.texels[(size_t)ty * S.w + tx];
      pos = (code == cvk::CODE_POS);
      neg = (code == cvk::CODE_NEG);
    } else {
      none = 1;
    }
    unsigned char *dst = P.fb + ((size_t)py * P.W + px) * 3;
    dst[0] = (unsigned char)(texel & 0xFF);
    dst[1] = (unsigned char)((texel >> 8) & 0xFF);
    dst[2] = (unsigned char)((texel >> 16) & 0xFF);
  }
  flush_frame_counts(P.counters, 0u, valid, steps, 1u, pos, neg, none, oob);
}

/* compute_photon_trajectory (src/systems.rs:77-92): the state BEFORE each of `iterations` Euler steps,
 * all eight components (t and p_t included: x_t += (p_t * -1) * delta, p_t += 0 * delta), one thread per
 * photon.  Momentum is covariant on entry (what new_photon produces). */
struct TrajectoryParams {
  cvk::MetricParams metric;
  const double *x0, *p0; /* n*4 each */
  double *out;           /* n * iterations * 8: [photon][iteration][x0..x3, p0..p3] */
  unsigned n, iterations;
  double delta;
};

template <int KIND>
__global__ __launch_bounds__(64) void trajectory_kernel(const TrajectoryParams P) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  cvk::MetricParams M = P.metric;
  M.T = cv_sc_table();
  M.LT = cv_log_table();
  M.AT = cv_atan_table();
  double t = P.x0[4 * i], pt = P.p0[4 * i];
  cvk::Ray q;
  q.l = P.x0[4 * i + 1];
  q.th = P.x0[4 * i + 2];
  q.ph = P.x0[4 * i + 3];
  q.p1 = P.p0[4 * i + 1];
  q.p2 = P.p0[4 * i + 2];
  q.p3 = P.p0[4 * i + 3];
  q.p3sq = q.p3 * q.p3;
  double p3 = q.p3;
  double *o = P.out + (size_t)i * P.iterations * 8;
  for (unsigned k = 0; k < P.iterations; ++k) {
    o[0] = t;
    o[1] = q.l;
    o[2] = q.th;
    o[3] = q.ph;
    o[4] = pt;
    o[5] = q.p1;
    o[6] = q.p2;
    o[7] = p3;
    o += 8;
    cvk::ray_step<KIND, true>(M, q, P.delta);
    t = t + (pt * (1.0 / -1.0)) * P.delta; /* dx0 = p0 * g00.powi(-1) */
    pt = pt + 0.0 * P.delta;
    p3 = p3 + 0.0 * P.delta;
  }
}

__global__ void selftest_math_kernel(int op, const double *a, const double *b, double *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0;
  double r;
  switch (op) {
    case 0:
      r = cv_sin(x);
      break;
    case 1:
      r = cv_cos(x);
      break;
    case 2:
      r = cv_atan(x);
      break;
    case 3:
      r = cv_acos(x);
      break;
    case 4:
      r = cv_log(x);
      break;
    case 5:
      r = cv_atan2(x, y);
      break;
    case 6:
      r = x / y;
      break;
    case 7:
      r = CV_SQRT(x);
      break;
    case 9:
      r = __builtin_amdgcn_rcp(x); /* raw v_rcp_f64 seed */
      break;
    case 10:
      r = __builtin_amdgcn_rsq(x); /* raw v_rsq_f64 seed */
      break;
    default:
      r = CV_FMA(x, y, x);
      break;
  }
  out[i] = r;
}

/* Directed cases for the primitives of the fast step (tests/test_gpu_fast_step.py): three inputs per element.
 * op 0 div_with_recip(a, b, c)   1 root of sqrt_and_rsqrt(a)   2 its y ~ 1/sqrt(a)
 *    3 the square root's last residual step for given (x, g, y) = (a, b, c): fma(fma(-g, g, x), 0.5 y, g)
 *    4 recip_refined(a)   5 cv_div_nr(a, b)   6 recip_newton(a, b)
 *    7 / 8 / 9 component 0 / 1 / 2 of unit3<true>(v = (a, b, c), |v|)   10 div_index<true>(a, b, recip_chain(b))   11 div_angle<true>(a, b, recip_chain(b))   12 sqrt_plain<true>(a)   13 (double)rust_as_u32(a) */
__global__ void selftest_math3_kernel(int op, const double *a, const double *b, const double *c, double *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0, z = c ? c[i] : 0.0;
  double r, t;
  switch (op) {
    case 0:
      r = cvk::div_with_recip(x, y, z);
      break;
    case 1:
      cvk::sqrt_and_rsqrt(x, r, t);
      break;
    case 2:
      cvk::sqrt_and_rsqrt(x, t, r);
      break;
    case 3:
      r = CV_FMA(CV_FMA(-y, y, x), 0.5 * z, y);
      break;
    case 4:
      r = cvk::recip_refined(x);
      break;
    case 5:
      r = cv_div_nr(x, y);
      break;
    case 6:
      r = cvk::recip_newton(x, y);
      break;
    case 7: /* 7, 8, 9: component 0, 1, 2 of unit3<true>((a, b, c), |(a, b, c)|) */
    case 8:
    case 9: {
      const double v[3] = {x, y, z};
      double u[3];
      cvk::unit3<true>(v, cvk::norm3(v), u);
      r = u[op - 7];
      break;
    }
    case 10: /* an index by a constant */
      r = cvk::div_index<true>(x, y, cvk::recip_chain(y));
      break;
    case 11: /* an angle by a constant */
      r = cvk::div_angle<true>(x, y, cvk::recip_chain(y));
      break;
    case 12: /* the square root without the expansion's range wrappers */
      r = cvk::sqrt_plain<true>(x);
      break;
    default: /* 13: Rust's `as u32` */
      r = (double)cvk::rust_as_u32(x);
      break;
  }
  out[i] = r;
}

/* One fast Euler step per input state with every quotient of the step recorded (cvk::NoProbe's counterpart): for
 * each state 6 x {numerator, denominator, the shared reciprocal the step used, the step's quotient, the IEEE
 * quotient, the remainder n - d RN(n y), 1 - d y} (NaN rows: quotient not formed -- k = 0 outside Ellis, or the state took the strict step), then the new
 * state (l, theta, phi, p_l, p_theta) of the fast step and of the strict step, and a flag (fast path taken). */
struct StepRecorder {
  double *o;
  __device__ void rec(int k, double n, double d, double y) const {
    o[7 * k + 0] = n;
    o[7 * k + 1] = d;
    o[7 * k + 2] = y;
    o[7 * k + 3] = cvk::div_with_recip(n, d, y);
    o[7 * k + 4] = n / d;
    o[7 * k + 5] = CV_FMA(-d, n * y, n); /* the remainder div_with_recip sees (exact) */
    o[7 * k + 6] = CV_FMA(-d, y, 1.0);   /* 1 - d y: how far the shared reciprocal is from 1/d (exact) */
  }
};


template <int KIND>
__global__ void selftest_fast_step_kernel(cvk::MetricParams M, double delta, double max_radius, const double *st, size_t n,
                                          double *out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  M.T = cv_sc_table();
  M.LT = cv_log_table();
  M.AT = cv_atan_table();
  cvk::Ray q;
  q.l = st[5 * i + 0];
  q.th = st[5 * i + 1];
  q.ph = 0.0;
  q.p1 = st[5 * i + 2];
  q.p2 = st[5 * i + 3];
  q.p3 = st[5 * i + 4];
  q.p3sq = q.p3 * q.p3;
  cvk::Ray q0 = q;
  double *o = out + i * CURVIS_FAST_STEP_RECORD;
  for (int k = 0; k < 42; ++k) o[k] = __builtin_nan("");
  const bool lane_ok = cvk::metric_fast_ok(KIND, M, max_radius) && cvk::ray_fast_ok(q);
  StepRecorder pr{o};
  cvk::ray_step_fast<KIND, true, false, false, StepRecorder>(M, q, delta, lane_ok, pr);
  cvk::ray_step<KIND, true>(M, q0, delta);
  o[42] = q.l, o[43] = q.th, o[44] = q.ph, o[45] = q.p1, o[46] = q.p2;
  o[47] = q0.l, o[48] = q0.th, o[49] = q0.ph, o[50] = q0.p1, o[51] = q0.p2;
  o[52] = (o[8] == o[8]) ? 1.0 : 0.0; /* quotient 1 (1/r^2) is formed by every fast step */
}

}  // namespace
