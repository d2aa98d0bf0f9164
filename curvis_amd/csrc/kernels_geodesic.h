/* kernels_geodesic.h -- the per-pixel path on the device: per-frame counters, launch parameters,
 * geodesic_static / geodesic_persistent / geodesic_relay, the relay seat belt's compare kernel, shade_kernel.
 * Part of the ONE translation unit curvis_hip.hip (included there, nowhere else). */
#pragma once

namespace {


/* Statistics counters, PER FRAME (src/rendering.rs:291-316 renders frame by frame; BASELINE configs[4] asks for
 * per-frame early-termination statistics, and a batch of frames is ONE launch here).  Layout of the counter block,
 * in 128-byte lines of CNT_STRIDE words: line 0 holds the persistent kernel's queue head (CNT_NEXT) and nothing
 * else; then `slots` replica lines per frame, each {FC_STEPS, FC_RAYS, FC_POS, FC_NEG, FC_NONE, FC_OOB}.  A wave
 * adds its sums to the replica (blockIdx.x mod slots) of its frame: tens of thousands of waves adding to ONE
 * address serialise in a single L2 channel (it made the 0.06 ms per-pixel kernel of the efficient renderer take
 * 0.40 ms), so a frame's counters are spread over 64 lines in launches of a few frames and over 8 in larger
 * batches.  The host sums the replicas of a frame, and the frames for the totals of the call. */
enum { CNT_NEXT = 0 };
enum { FC_STEPS = 0, FC_RAYS, FC_POS, FC_NEG, FC_NONE, FC_OOB, FC_N };
enum { CNT_STRIDE = 16 };
struct FrameCounters {
  unsigned long long *base; /* device: CNT_STRIDE * (1 + n_frames * slots) words */
  unsigned slots;           /* replica lines per frame, a power of two */
};
__host__ __device__ inline unsigned counter_slots_for(unsigned n_frames) { return n_frames >= 8u ? 8u : 64u; }
__host__ __device__ inline size_t counter_words(unsigned n_frames, unsigned slots) {
  return (size_t)CNT_STRIDE * (1u + (size_t)n_frames * slots);
}
__device__ __forceinline__ unsigned long long *frame_counter_line(const FrameCounters &C, unsigned frame) {
  return C.base + (size_t)CNT_STRIDE * (1u + (size_t)frame * C.slots + (blockIdx.x & (C.slots - 1u)));
}
/* frame of a wave's 8x8 tile, as a scalar: computed in the epilogue from the wave-uniform tile number so that no
 * per-lane frame index stays live across the Euler loop (it cost the Interstellar relay kernel its fifth wave) */
__device__ __forceinline__ unsigned frame_of_tile(unsigned long long tile, unsigned rays_per_frame) {
  const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tile); /* tiles < 2^32 (checked on the host) */
  return t / (rays_per_frame >> 6);
}
/* Add a wave's contribution to the per-frame counters.  `frame` is per lane; lanes with !valid contribute
 * nothing.  When every valid lane of the wave belongs to one frame (always true for the 8x8-tile kernels, and for
 * all but the waves straddling a frame boundary in the per-pixel kernels) the wave reduces first and one lane
 * issues the atomics; otherwise each valid lane adds its own. */
__device__ __forceinline__ void flush_frame_counts(const FrameCounters &C, unsigned frame, bool valid,
                                                   unsigned long long steps, unsigned rays, unsigned pos, unsigned neg,
                                                   unsigned none, unsigned oob) {
  const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
  if (!vm) return;
  const unsigned f0 = (unsigned)__builtin_amdgcn_readlane((int)frame, (int)__builtin_ctzll(vm));
  if (!valid) steps = 0ull, rays = pos = neg = none = oob = 0u;
  if (__builtin_amdgcn_ballot_w64(valid && frame != f0) == 0ull) {
    for (int off = 32; off > 0; off >>= 1) {
      steps += __shfl_xor(steps, off);
      rays += __shfl_xor(rays, off);
      pos += __shfl_xor(pos, off);
      neg += __shfl_xor(neg, off);
      none += __shfl_xor(none, off);
      oob += __shfl_xor(oob, off);
    }
    if ((threadIdx.x & 63u) == 0u) {
      unsigned long long *c = frame_counter_line(C, f0);
      if (steps) atomicAdd(&c[FC_STEPS], steps);
      if (rays) atomicAdd(&c[FC_RAYS], (unsigned long long)rays);
      if (pos) atomicAdd(&c[FC_POS], (unsigned long long)pos);
      if (neg) atomicAdd(&c[FC_NEG], (unsigned long long)neg);
      if (none) atomicAdd(&c[FC_NONE], (unsigned long long)none);
      if (oob) atomicAdd(&c[FC_OOB], (unsigned long long)oob);
    }
  } else if (valid) {
    unsigned long long *c = frame_counter_line(C, frame);
    if (steps) atomicAdd(&c[FC_STEPS], steps);
    if (rays) atomicAdd(&c[FC_RAYS], (unsigned long long)rays);
    if (pos) atomicAdd(&c[FC_POS], (unsigned long long)pos);
    if (neg) atomicAdd(&c[FC_NEG], (unsigned long long)neg);
    if (none) atomicAdd(&c[FC_NONE], (unsigned long long)none);
    if (oob) atomicAdd(&c[FC_OOB], (unsigned long long)oob);
  }
}

/* the same for kernels whose lanes contribute FLAGS (one pixel each, no step counts): four ballots and population counts per
 * wave instead of 42 cross-lane shuffles -- the per-pixel kernel of the efficient renderer runs ~1 000 instructions per pixel,
 * the generic reduction was ~5 % of them */
__device__ __forceinline__ void flush_frame_flags(const FrameCounters &C, unsigned frame, bool valid, bool pos, bool neg, bool none,
                                                  bool oob) {
  const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
  if (!vm) return;
  const unsigned f0 = (unsigned)__builtin_amdgcn_readlane((int)frame, (int)__builtin_ctzll(vm));
  if (__builtin_amdgcn_ballot_w64(valid && frame != f0) == 0ull) {
    const unsigned n_pos = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid && pos)), n_neg = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid && neg));
    const unsigned n_none = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid && none)), n_oob = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid && oob));
    if ((threadIdx.x & 63u) == 0u) {
      unsigned long long *c = frame_counter_line(C, f0);
      atomicAdd(&c[FC_RAYS], (unsigned long long)__popcll(vm));
      if (n_pos) atomicAdd(&c[FC_POS], (unsigned long long)n_pos);
      if (n_neg) atomicAdd(&c[FC_NEG], (unsigned long long)n_neg);
      if (n_none) atomicAdd(&c[FC_NONE], (unsigned long long)n_none);
      if (n_oob) atomicAdd(&c[FC_OOB], (unsigned long long)n_oob);
    }
  } else if (valid) { /* a wave that straddles two frames: every lane for itself */
    unsigned long long *c = frame_counter_line(C, frame);
    atomicAdd(&c[FC_RAYS], 1ull);
    if (pos) atomicAdd(&c[FC_POS], 1ull);
    if (neg) atomicAdd(&c[FC_NEG], 1ull);
    if (none) atomicAdd(&c[FC_NONE], 1ull);
    if (oob) atomicAdd(&c[FC_OOB], 1ull);
  }
}

/* Final ray states, structure-of-arrays in HBM, indexed by pixel id = frame*W*H + py*W + px.
 * Written by the integration kernel, read once by the shading kernel (48-56 B per ray against
 * ~2000 Euler steps of arithmetic: the staging costs ~0.2% of a frame). */
struct RayStore {
  double *l, *th, *ph, *p1, *p2, *p3;
  unsigned *steps;
  int *code;
};

struct IntegrateParams {
  cvk::MetricParams metric;
  const cvk::CameraParams *cams; /* device, n_frames entries */
  unsigned n_frames, W, H, tiles_x, tiles_y; /* H = rows rendered by this launch (a band of the frame or all of it) */
  unsigned row0;                 /* first image row of the band: pixel (px, py) of the launch is image row row0 + py */
  unsigned rays_per_frame;       /* tiles_x*tiles_y*64 (padded to whole 8x8 tiles) */
  unsigned long long total_rays; /* n_frames * rays_per_frame */
  unsigned max_iter;
  double max_radius, delta;
  RayStore store;
  FrameCounters counters;
  int refill_threshold;
  int fast_ok; /* host-side part of the fast-step guard */
  /* fused shading (static kernel, non-debug): the epilogue looks the sky up and writes RGB8 itself */
  cvk::SkyParams sky[2];
  unsigned char *fb;
  /* diagnostics (CURVIS_TRACE_FILE): per wave of the static kernel {start, end (wall_clock64 ticks), HW_ID,
   * XCC_ID}; null in normal operation */
  unsigned long long *trace;
};

struct ShadeParams {
  cvk::MetricParams metric;
  cvk::SkyParams sky[2];
  RayStore store;
  unsigned long long n_pixels; /* n_frames*W*H */
  unsigned char *fb;           /* RGB8 */
  curvis_ray_debug *dbg;       /* or null */
  unsigned long long npix;     /* pixels per frame: frame of pixel o = o / npix */
  FrameCounters counters;
};

/* Per-workgroup LDS copy of the sin/cos table (4 KiB; 8 KiB in its 256-row form): the Euler loop evaluates
 * sincos once per step per lane with a data-dependent index; two ds_read_b128 from LDS instead of divergent
 * __constant__ loads. */
template <int KIND>
struct alignas(16) MathTablesLds {
  static constexpr unsigned LOG_ROWS = (KIND == cvk::METRIC_INTERSTELLAR) ? (unsigned)CV_LOG_TABLE_N : 2u;
  static constexpr unsigned ATAN_ROWS = (KIND == cvk::METRIC_INTERSTELLAR) ? (unsigned)CV_ATAN_TABLE_N : 1u;
  /* Order and alignment are chosen for the address arithmetic of the lookups: the 24-byte log rows sit at offset
   * 0, so ds_read2_b64 (whose offset field is short) and ds_read_b64 share one address register; the 32- and
   * 64-byte rows of the other two tables are read with ds_read_b128, whose offset field reaches any LDS
   * address, so their base offsets cost no instruction either. */
  /* 256-row form of the sin/cos table (cv_sincos_tw: no index mask) in every kernel.  Footprint of the Interstellar
   * kernels: 12 KiB log (512 rows x 24 B) + 8 KiB sin/cos + 8.06 KiB atan (the 129 reciprocal-branch rows x 64 B; the
   * direct-branch rows are read from __constant__ memory by the few steps next to the throat) = 28.1 KiB per
   * workgroup: five workgroups (= five waves per SIMD, what amdgpu_waves_per_eu(5) asks for) fit the CU's 160 KiB,
   * six would not.  The static_assert below keeps a table change from silently costing that occupancy. */
  static constexpr bool WIDE_SC = true;
  double lg[LOG_ROWS][3]; /* only the Interstellar metric evaluates a logarithm and an arc tangent per step */
  double sc[WIDE_SC ? 256 : 128][4];
  double at[ATAN_ROWS][8];
};

static_assert(sizeof(MathTablesLds<cvk::METRIC_INTERSTELLAR>) * 5 <= 160 * 1024,
              "five workgroups of the Interstellar kernels (5 waves per SIMD) must fit the CU's 160 KiB of LDS");
static_assert(sizeof(MathTablesLds<cvk::METRIC_ELLIS>) * 8 <= 160 * 1024, "the Ellis / flat kernels run at up to 8 workgroups per CU");

/* copy the elementary-function tables of cv_math.h into LDS and point the metric at them */
template <int KIND>
__device__ __forceinline__ void load_math_tables(MathTablesLds<KIND> &L, cvk::MetricParams &M) {
  const double *src = &cv_sc_table_dev[0][0];
  double *dst = &L.sc[0][0];
  for (unsigned i = threadIdx.x; i < (MathTablesLds<KIND>::WIDE_SC ? 1024u : 512u); i += blockDim.x) dst[i] = src[i & 511u];
  M.T = L.sc;
  if (KIND == cvk::METRIC_INTERSTELLAR) {
    const double *lsrc = &cv_log_table_dev[0][0];
    double *ldst = &L.lg[0][0];
    for (unsigned i = threadIdx.x; i < 3u * CV_LOG_TABLE_N; i += blockDim.x) ldst[i] = lsrc[i];
    M.LT = L.lg;
    const double *asrc = &cv_atan_table_dev[0][0];
    double *adst = &L.at[0][0];
    for (unsigned i = threadIdx.x; i < 8u * CV_ATAN_TABLE_N; i += blockDim.x) adst[i] = asrc[i];
    M.AT = L.at;
  } else {
    M.LT = cv_log_table();
    M.AT = cv_atan_table();
  }
  __syncthreads();
}

/* ray id -> (frame, pixel).  Rays are numbered by 8x8 pixel tiles so the 64 rays a wave draws
 * together are spatial neighbours (similar step counts, neighbouring texels). */
__device__ __forceinline__ bool decode_ray(const IntegrateParams &P, unsigned long long id, unsigned &frame,
                                           unsigned &px, unsigned &py) {
  frame = (unsigned)(id / P.rays_per_frame);
  const unsigned rem = (unsigned)(id - (unsigned long long)frame * P.rays_per_frame);
  const unsigned tile = rem >> 6, k = rem & 63u;
  const unsigned tyi = tile / P.tiles_x, txi = tile - tyi * P.tiles_x;
  px = txi * 8u + (k & 7u);
  py = tyi * 8u + (k >> 3);
  return px < P.W && py < P.H;
}

template <bool PHI>
__device__ __forceinline__ void store_ray(const RayStore &S, size_t o, const cvk::Ray &q, unsigned steps, int code) {
  S.l[o] = q.l;
  S.th[o] = q.th;
  if (PHI) S.ph[o] = q.ph;
  S.p1[o] = q.p1;
  S.p2[o] = q.p2;
  S.p3[o] = q.p3;
  S.steps[o] = steps;
  S.code[o] = code;
}

/* Escape test of src/systems.rs:129-134: `l > R` -> PositiveSpace, else `l < -R` -> NegativeSpace.
 * One compare per step: |l| > R is true exactly when one of the two is (false for NaN, like both);
 * which one is decided once, after the loop. */
__device__ __forceinline__ bool ray_escaped(double l, double R) { return __builtin_fabs(l) > R; }
__device__ __forceinline__ int escape_code(double l) { return l > 0.0 ? cvk::CODE_POS : cvk::CODE_NEG; }

template <int KIND, bool PHI, bool FAST, bool EQ = false>
__device__ __forceinline__ void one_step(const cvk::MetricParams &M, double delta, cvk::Ray &q, bool lane_ok) {
  if (FAST)
    cvk::ray_step_fast<KIND, PHI, MathTablesLds<KIND>::WIDE_SC, EQ>(M, q, delta, lane_ok);
  else
    cvk::ray_step<KIND, PHI, MathTablesLds<KIND>::WIDE_SC>(M, q, delta);
}

/* final photon -> tangent direction -> nearest sky texel (rows R9-R10 of SURVEY.md 8a) */
template <int KIND>
__device__ __forceinline__ unsigned shade_ray(const cvk::MetricParams &M, const cvk::SkyParams *sky, const cvk::Ray &q,
                                              int code, unsigned &tx, unsigned &ty, unsigned &oob) {
  unsigned texel = 0xFF000000u; /* Rgba([0,0,0,255]) */
  tx = ty = 0;
  if (code != cvk::CODE_NONE) {
    double d0, d1, d2;
    cvk::ray_direction<KIND>(M, q, d0, d1, d2);
    const cvk::SkyParams &S = sky[code == cvk::CODE_POS ? 0 : 1];
    cvk::sky_indices(S, d0, d1, d2, tx, ty);
    unsigned cx = tx, cy = ty;
    if (cx >= S.w || cy >= S.h) oob = 1; /* reference: image::get_pixel panics; defined here: clamp + count */
    if (cx >= S.w) cx = S.w - 1;
    if (cy >= S.h) cy = S.h - 1;
    texel = S.texels[(size_t)cy * S.w + cx];
  }
  return texel;
}

/* K1, persistent form: lanes draw rays from a global queue with one wave-aggregated atomic whenever
 * `refill_threshold` lanes are free; terminated rays are stored together at that point. */
template <int KIND, bool PHI, bool FAST>
__global__ __launch_bounds__(256) void geodesic_persistent(const IntegrateParams P) {
  __shared__ MathTablesLds<KIND> s_tab;
  cvk::MetricParams M = P.metric;
  load_math_tables<KIND>(s_tab, M);
  const unsigned lane = threadIdx.x & 63u;
  cvk::Ray q;
  q.l = q.th = q.ph = q.p1 = q.p2 = q.p3 = q.p3sq = 0.0;
  size_t slot = 0;
  unsigned steps = 0;
  int code = cvk::CODE_NONE;
  bool active = false; /* lane is integrating */
  bool done = false;   /* lane holds a terminated ray that has not been stored yet */
  bool dry = false;    /* queue exhausted (wave-uniform) */
  bool lane_ok = false;

  for (;;) {
    if (done) { /* staged path: shade_kernel reads the store and keeps the per-frame statistics */
      store_ray<PHI>(P.store, slot, q, steps, code);
      done = false;
    }
    if (!dry) {
      const bool need = !active;
      const unsigned long long mask = __ballot(need);
      if (mask) {
        const unsigned n = (unsigned)__popcll(mask);
        const int leader = __ffsll((long long)mask) - 1;
        unsigned long long base = 0;
        if ((int)lane == leader) base = atomicAdd(&P.counters.base[CNT_NEXT], (unsigned long long)n);
        base = __shfl(base, leader);
        const unsigned rank =
            __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        const unsigned long long mine = base + rank;
        if (need && mine < P.total_rays) {
          unsigned frame, px, py;
          if (decode_ray(P, mine, frame, px, py)) {
            cvk::ray_init<KIND>(M, P.cams[frame], px, py + P.row0, q);
            slot = (size_t)frame * P.W * P.H + (size_t)py * P.W + px;
            steps = 0;
            lane_ok = FAST && P.fast_ok && cvk::ray_fast_ok(q);
            if (P.max_iter == 0) {
              code = cvk::CODE_NONE;
              done = true;
            } else {
              active = true;
            }
          }
        }
        if (base + n >= P.total_rays) dry = true;
      }
    }
    if (!__ballot(active)) {
      if (__ballot(done)) continue; /* max_iter == 0 corner */
      if (dry) break;
      continue; /* every drawn id was tile padding: draw again */
    }
    const int thr = dry ? 64 : P.refill_threshold;
    for (;;) { /* integrate until `thr` lanes are free */
      if (active) {
        one_step<KIND, PHI, FAST>(M, P.delta, q, lane_ok);
        ++steps;
        const bool esc = ray_escaped(q.l, P.max_radius);
        if (esc | (steps >= P.max_iter)) { /* loop bound of src/systems.rs:126 */
          code = esc ? escape_code(q.l) : cvk::CODE_NONE;
          active = false;
          done = true;
        }
      }
      if (__popcll(__ballot(!active)) >= thr) break;
    }
  }
}

/* K1, static form: one ray per thread, hardware block scheduling does the load balancing.
 * FUSED: the epilogue shades the pixel itself (direction, sky lookup, RGB8 store) instead of staging the
 * final state in HBM for shade_kernel -- the epilogue needs fewer registers than the loop, so the fusion is
 * free in occupancy and removes ~200 MB of HBM traffic and one launch per frame. */
template <int KIND, bool PHI, bool FAST, bool FUSED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KIND == cvk::METRIC_INTERSTELLAR ? 5 : 7)))
void geodesic_static(const IntegrateParams P) {
  __shared__ MathTablesLds<KIND> s_tab;
  cvk::MetricParams M = P.metric;
  load_math_tables<KIND>(s_tab, M);
  const unsigned long long id = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long t_start = P.trace ? wall_clock64() : 0ull;
  unsigned frame, px, py;
  cvk::Ray q;
  q.l = q.th = q.ph = q.p1 = q.p2 = q.p3 = q.p3sq = 0.0;
  bool valid = false, active = false, lane_ok_w = false;
  unsigned steps = 0;
  int code = cvk::CODE_NONE;
  if (id < P.total_rays && decode_ray(P, id, frame, px, py)) {
    cvk::ray_init<KIND>(M, P.cams[frame], px, py + P.row0, q);
    lane_ok_w = FAST && P.fast_ok && cvk::ray_fast_ok(q);
    valid = true;
    active = P.max_iter != 0;
  }
  /* All lanes of a wave start together, so the step counter is wave-uniform (an SGPR).  The loop is a plain
   * divergent loop: a lane leaves it (drops out of EXEC) when it escapes; the back-edge is "EXEC still
   * non-empty", so activity costs no VALU instruction.  The counter is recorded per lane only in the
   * iterations in which some lane escapes (a scalar branch on the ballot; the per-lane test goes through the
   * ballot mask so that the compiler keeps the block inside the loop instead of sinking it behind the exit,
   * which would cost a counter copy to a VGPR in every iteration).  Lanes still inside when the counter
   * reaches max_iterations are NotEscaped (code stays CODE_NONE). */
  if (active) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned k = 0;
    steps = P.max_iter;
    for (;;) {
      ++k;
      one_step<KIND, PHI, FAST>(M, P.delta, q, lane_ok_w);
      const bool esc = ray_escaped(q.l, P.max_radius);
      const unsigned long long em = __builtin_amdgcn_ballot_w64(esc);
      if (em) { /* rare: at most 64 times per wave.  The volatile asm keeps this a real (scalar) branch. */
        unsigned kv;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kv) : "s"(k));
        if ((em >> lane) & 1ull) steps = kv;
      }
      if (esc) break;
      if (k >= P.max_iter) break;
    }
    if (ray_escaped(q.l, P.max_radius)) code = escape_code(q.l); /* the state is final: same test as in the loop */
  }
  unsigned pos = 0, neg = 0, none = 0, oob = 0;
  /* The pixel position is decoded AGAIN here, from the laundered block index, instead of being kept in registers
   * across the Euler loop (frame, px, py or a 64-bit slot: 3-4 VGPRs the loop is better off without). */
  unsigned bid = blockIdx.x;
  asm volatile("" : "+s"(bid));
  const unsigned long long id2 = (unsigned long long)bid * blockDim.x + threadIdx.x;
  valid = id2 < P.total_rays && decode_ray(P, id2, frame, px, py);
  const size_t slot = valid ? (size_t)frame * P.W * P.H + (size_t)py * P.W + px : 0;
  if (valid) {
    if (FUSED) {
      unsigned tx, ty;
      const unsigned texel = shade_ray<KIND>(M, P.sky, q, code, tx, ty, oob);
      unsigned char *dst = P.fb + slot * 3;
      dst[0] = (unsigned char)(texel & 0xFF);
      dst[1] = (unsigned char)((texel >> 8) & 0xFF);
      dst[2] = (unsigned char)((texel >> 16) & 0xFF);
      pos = (code == cvk::CODE_POS);
      neg = (code == cvk::CODE_NEG);
      none = (code == cvk::CODE_NONE);
    } else {
      store_ray<PHI>(P.store, slot, q, steps, code);
    }
  }
  /* statistics of the wave's tile go to the counters of ITS frame (a tile never straddles frames); on the staged
   * path shade_kernel keeps them */
  if (FUSED) flush_frame_counts(P.counters, frame_of_tile(id2 >> 6, P.rays_per_frame), valid, steps, 1u, pos, neg, none, oob);
  if (P.trace && (threadIdx.x & 63u) == 0) {
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    unsigned long long *rec = P.trace + 4ull * (id >> 6);
    rec[0] = t_start;
    rec[1] = wall_clock64();
    rec[2] = hw_id;
    rec[3] = xcc_id;
  }
}

/* ------------------------------------------------------------------------------------------------
 * K1, relay form ("variant" = 2, launches of up to "relay_max_frames" = 8 frames): the static kernel plus a hand-over of
 * unfinished tiles in the END-GAME of the launch, with the hardware workgroup dispatcher as load balancer.
 *
 * A single-frame launch ends with ~2 ms in which no fresh workgroup is left and every SIMD finishes the 5-6
 * waves it happens to hold; the SIMDs finish between 10.8 and 12.0 ms (wave trace): ~1 ms of a
 * 12 ms frame is imbalance that cannot be repaired because a wave, once placed, stays where it is.
 *
 * Here the grid is [fresh workgroups | relay workgroups].  A fresh wave runs the static kernel's loop in
 * segments of `seg` steps.  Once every fresh workgroup has started (a counter), a wave that reaches a segment
 * boundary with unfinished rays PARKS its tile -- state of the 64 rays to HBM (56 B per ray), tile id into a
 * ticket ring -- and exits.  Its slot goes to the next relay workgroup, which the dispatcher places on
 * whichever CU has room: a relay wave takes the oldest parked tile, integrates one more segment, and parks
 * it again or shades it.  Tiles therefore drift, `seg` steps at a time, from CUs with a backlog to CUs that
 * ran dry.  There is no persistent loop (inside one it costs 86-155 VGPRs instead of 61): every
 * wave does one piece of work and exits.  Results are bit-identical: same per-ray arithmetic, state
 * round-trips through HBM as doubles.
 *
 * Ring protocol: `tail` / `head` hand out tickets; a parker writes tile+1 into slot ticket%CAP (release), a
 * relay wave waits for its slot to become non-zero (acquire) and clears it.  At most one tile per resident
 * wave is parked, CAP = 32768.  `remaining` counts unfinished tiles; waiting relay waves leave when it
 * reaches zero (their tickets are then never filled).  If the relay workgroups of a launch run out while
 * tiles are still parked, the host launches more (relay-only grid) until every tile is finished; a wave that has
 * waited ~20 s sets `error` and leaves, so a logic error shows up as CURVIS_E_HIP, not as a hang (a relay wave
 * legitimately waits at most for the rest of the launch: the kernel is meant for launches of milliseconds). */
struct RelayQueue { /* all zero before the first launch of a frame (one hipMemsetAsync) */
  unsigned long long started;  /* fresh workgroups that have begun */
  unsigned long long head, tail;
  unsigned long long finished; /* tiles shaded so far */
  unsigned long long error;
  unsigned long long pad[3];
  unsigned ring[1];            /* kRelayRing entries follow */
};
constexpr unsigned kRelayRing = 32768; /* >= resident waves (256 CUs x 32) with margin */
struct RelayArgs {
  RelayQueue *q;
  unsigned long long n_tiles;
  unsigned fresh_blocks; /* workgroups [0, fresh_blocks) start tiles, the rest relay parked ones */
  unsigned seg;          /* steps per segment */
  unsigned max_hops;     /* a tile is handed over at most this many times (0 = no limit) */
  unsigned max_parks;    /* hand-overs per launch at most (0 = no limit) */
  unsigned corrupt_ticket; /* test hook (option "relay_test_corrupt"): non-zero = every relay wave of this launch perturbs the
                              state it reloads, so that the first-launch check below has something to find; 0 = off */
};

/* Hand-over traffic of the relay kernel goes around the caches: system-scope relaxed atomics compile to
 * write-through stores / cache-bypassing loads (sc0 sc1), so publishing a tile needs only "my stores have
 * been acknowledged" (a workgroup-scope release = s_waitcnt) instead of an agent-scope release fence, which on
 * this multi-XCD part writes back the whole L2 (buffer_wbl2) -- measured ~80 us per hand-over. */
template <typename T>
__device__ __forceinline__ void st_sys(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <typename T>
__device__ __forceinline__ T ld_sys(const T *p) { return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

/* register budget: the Interstellar instantiation must stay at 5 waves per SIMD (<= 96 VGPRs; its LDS tables allow
 * no more anyway): left alone the allocator takes 97 and drops to four (+6 % time) */
template <int KIND, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KIND == cvk::METRIC_INTERSTELLAR ? 5 : 7)))
void geodesic_relay(const IntegrateParams P, const RelayArgs A) {
  __shared__ MathTablesLds<KIND> s_tab;
  RelayQueue *const Q = A.q;
  const unsigned long long t_start = P.trace ? wall_clock64() : 0ull;
  unsigned long long t_work = 0ull; /* diagnostics: when the wave had its tile */
  const bool fresh = blockIdx.x < A.fresh_blocks;
  /* a relay workgroup that starts when every tile is finished leaves at once, before the table load: the grid
   * holds many more relay workgroups than are usually needed.  The decision is made ONCE per workgroup (thread 0
   * reads the counter, LDS + barrier hand it to the other waves): waves reading it on their own could disagree,
   * and a workgroup of which only some waves reach load_math_tables' barrier must not exist. */
  if (!fresh) {
    __shared__ int s_leave;
    if (threadIdx.x == 0)
      s_leave = __hip_atomic_load(&Q->finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.n_tiles ? 1 : 0;
    __syncthreads();
    if (s_leave) return;
  }
  cvk::MetricParams M = P.metric;
  load_math_tables<KIND>(s_tab, M);
  const unsigned lane = threadIdx.x & 63u;
  unsigned long long tile;
  bool corrupt = false;
  if (fresh) {
    if (threadIdx.x == 0) atomicAdd(&Q->started, 1ull);
    tile = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  } else {
    /* relay wave: take the oldest parked tile, or leave when every tile is finished */
    unsigned long long tk = 0;
    if (lane == 0) tk = atomicAdd(&Q->head, 1ull);
    unsigned *slot_p = Q->ring + (__builtin_amdgcn_readfirstlane((unsigned)tk) & (kRelayRing - 1u));
    unsigned v = 0, spins = 0;
    for (;;) {
      v = ld_sys(slot_p);
      if (v) break;
      /* the own ring slot is polled every ~1 us, the shared `remaining` word only every 8th time */
      if ((spins & 7u) == 0u && __hip_atomic_load(&Q->finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.n_tiles) break;
      if (++spins > 20000000u) { /* ~20 s of waiting: longer than any launch this kernel is chosen for */
        if (lane == 0) atomicAdd(&Q->error, 1ull);
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
    if (!v) return;
    if (lane == 0) st_sys(slot_p, 0u);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); /* order the state loads below after the ticket load */
    tile = (unsigned long long)(v - 1u);
    corrupt = A.corrupt_ticket != 0u; /* test hook: every tile this launch hands over arrives perturbed */
  }
  if (P.trace) t_work = wall_clock64();
  const unsigned long long id = tile * 64ull + lane;
  unsigned frame, px, py;
  cvk::Ray q;
  q.l = q.th = q.ph = q.p1 = q.p2 = q.p3 = q.p3sq = 0.0;
  const bool valid = id < P.total_rays && decode_ray(P, id, frame, px, py);
  bool active = false;
  unsigned steps = 0, k0 = 0, hops = 0; /* hops: how often this tile has been handed over so far */
  int code = cvk::CODE_NONE;
  if (fresh) {
    if (valid) {
      cvk::ray_init<KIND>(M, P.cams[frame], px, py + P.row0, q);
      active = P.max_iter != 0;
      steps = P.max_iter;
    }
  } else {
    /* 40 B per parked ray: l, theta, p_l, p_theta and one packed {steps, code} word.  p_phi is a constant of the motion
     * (dp_phi = 0, src/metrics.rs:262-268; the loop never writes it): ray_init recomputes it -- and p_phi^2 -- from the
     * pixel, bit for bit what the parking wave held. */
    if (valid) cvk::ray_init<KIND>(M, P.cams[frame], px, py + P.row0, q);
    q.l = ld_sys(&P.store.l[id]);
    q.th = ld_sys(&P.store.th[id]);
    q.p1 = ld_sys(&P.store.p1[id]);
    q.p2 = ld_sys(&P.store.p2[id]);
    if (corrupt) { /* what a hand-over that lost stores would look like: the rays of the tile land elsewhere */
      q.th = q.th + 0.25;
      q.p1 = -q.p1;
    }
    const unsigned long long sc = ld_sys((const unsigned long long *)&P.store.ph[id]); /* the phi slot: unused outside the debug dump */
    steps = (unsigned)(sc >> 3);
    hops = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sc >> 40)) & 0xffu; /* the same in every lane of the tile */
    const int c = (int)(sc & 7ull);
    active = (c & 4) != 0;
    code = (c & 3) == 1 ? cvk::CODE_POS : (c & 3) == 2 ? cvk::CODE_NEG : cvk::CODE_NONE;
    /* the step counter is wave-uniform: every active lane parked it in `steps` (a parked tile has one) */
    const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
    k0 = __builtin_amdgcn_readlane(steps, am ? (int)__builtin_ctzll(am) : 0);
    if (active) steps = P.max_iter;
  }
  const bool lane_ok_w = FAST && P.fast_ok && cvk::ray_fast_ok(q);
  const unsigned SEG = A.seg;
  bool parked = false;
  for (;;) {
    const unsigned seg_end = (P.max_iter - k0 > SEG) ? k0 + SEG : P.max_iter;
    if (active) {
      unsigned k = k0;
      for (;;) {
        ++k;
        one_step<KIND, false, FAST>(M, P.delta, q, lane_ok_w);
        const bool esc = ray_escaped(q.l, P.max_radius);
        const unsigned long long em = __builtin_amdgcn_ballot_w64(esc);
        if (em) { /* rare; the volatile asm keeps this a real (scalar) branch */
          unsigned kv;
          asm volatile("v_mov_b32 %0, %1" : "=v"(kv) : "s"(k));
          if ((em >> lane) & 1ull) steps = kv;
        }
        if (esc) break;
        if (k >= seg_end) break;
      }
      if (ray_escaped(q.l, P.max_radius)) {
        code = escape_code(q.l);
        active = false;
      }
    }
    k0 = seg_end;
    if (!__builtin_amdgcn_ballot_w64(active) || k0 >= P.max_iter) break; /* tile finished */
    /* keep the tile while fresh workgroups are still being started, and afterwards unless a relay wave is
     * waiting for a ticket right now (head > tail): a hand-over then costs the tile ~1 us, whereas a tile parked
     * with nobody waiting would sit idle until the dispatcher has placed another relay workgroup */
    if (fresh && __hip_atomic_load(&Q->started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)A.fresh_blocks)
      continue;
    {
      const unsigned long long tail_now = __hip_atomic_load(&Q->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(&Q->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= tail_now) continue;
      if (A.max_parks != 0u && tail_now >= A.max_parks) continue; /* hand-over budget of the launch spent */
    }
    /* a tile that has been handed over `max_hops` times stays where it is: every hop costs 2 x 64 x 40 B of HBM traffic */
    if (!fresh && A.max_hops != 0u && hops >= A.max_hops) continue;
    parked = true;
    break;
  }
  unsigned pos = 0, neg = 0, none = 0, oob = 0;
  if (parked) {
    st_sys(&P.store.l[id], q.l);
    st_sys(&P.store.th[id], q.th);
    st_sys(&P.store.p1[id], q.p1);
    st_sys(&P.store.p2[id], q.p2);
    st_sys((unsigned long long *)&P.store.ph[id],
           ((unsigned long long)(hops < 255u ? hops + 1u : 255u) << 40) | ((unsigned long long)(active ? k0 : steps) << 3) |
               (unsigned long long)((code == cvk::CODE_POS ? 1 : code == cvk::CODE_NEG ? 2 : 0) | (active ? 4 : 0)));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* every lane's write-through stores acknowledged */
    if (lane == 0) {
      const unsigned long long tk = atomicAdd(&Q->tail, 1ull);
      st_sys(Q->ring + ((unsigned)tk & (kRelayRing - 1u)), (unsigned)tile + 1u);
    }
  } else {
    /* pixel position decoded again from the laundered tile number rather than kept live across the loop */
    unsigned tile_s = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tile);
    asm volatile("" : "+s"(tile_s));
    unsigned frame2, px2, py2;
    const unsigned long long id2 = (unsigned long long)tile_s * 64ull + lane;
    if (id2 < P.total_rays && decode_ray(P, id2, frame2, px2, py2)) {
      unsigned tx, ty;
      const unsigned texel = shade_ray<KIND>(M, P.sky, q, code, tx, ty, oob);
      unsigned char *dst = P.fb + ((size_t)frame2 * P.W * P.H + (size_t)py2 * P.W + px2) * 3;
      dst[0] = (unsigned char)(texel & 0xFF);
      dst[1] = (unsigned char)((texel >> 8) & 0xFF);
      dst[2] = (unsigned char)((texel >> 16) & 0xFF);
      pos = (code == cvk::CODE_POS);
      neg = (code == cvk::CODE_NEG);
      none = (code == cvk::CODE_NONE);
    }
    if (lane == 0 && tile < A.n_tiles) atomicAdd(&Q->finished, 1ull);
  }
  /* a tile is counted once, by the wave that finishes it, in the counters of its frame */
  flush_frame_counts(P.counters, frame_of_tile(tile, P.rays_per_frame), valid && !parked, steps, 1u, pos, neg, none, oob);
  if (P.trace && lane == 0) { /* CURVIS_TRACE_FILE: {start, end, HW_ID, XCC_ID | flags, got-tile time, tile} per wave */
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    unsigned long long *rec = P.trace + 4ull * ((unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    rec[0] = t_start;
    rec[1] = wall_clock64();
    rec[2] = (unsigned long long)hw_id | ((unsigned long long)(xcc_id & 0xf) << 32) | ((unsigned long long)(fresh ? 1 : 0) << 40) |
             ((unsigned long long)(parked ? 1 : 0) << 41) | ((unsigned long long)(k0 & 0xffff) << 44);
    rec[3] = t_work;
  }
}

/* seat belt of the relay kernel: number of differing 8-byte words of two framebuffers (one atomic per thread that saw one:
 * none at all in the expected case) */
__global__ __launch_bounds__(256) void compare_kernel(const unsigned long long *a, const unsigned long long *b, size_t n_words,
                                                      unsigned long long *n_diff) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned mine = 0;
  for (; i < n_words; i += stride) mine += a[i] != b[i];
  if (mine) atomicAdd(n_diff, (unsigned long long)mine);
}

/* K2: final photon -> tangent direction -> nearest sky texel -> RGB8 (rows R9-R10 of SURVEY.md 8a).
 * One thread per pixel, coalesced reads of the ray store, 3-byte stores of consecutive pixels. */
template <int KIND, bool DEBUG>
__global__ __launch_bounds__(256) void shade_kernel(const ShadeParams P) {
  const unsigned long long o = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned pos = 0, neg = 0, none = 0, oob = 0, n_steps = 0;
  const bool valid = o < P.n_pixels;
  if (valid) {
    cvk::Ray q;
    q.l = P.store.l[o];
    q.th = P.store.th[o];
    q.ph = DEBUG ? P.store.ph[o] : 0.0;
    q.p1 = P.store.p1[o];
    q.p2 = P.store.p2[o];
    q.p3 = P.store.p3[o];
    q.p3sq = 0.0;
    const int code = P.store.code[o];
    const unsigned steps = P.store.steps[o];
    n_steps = steps;
    unsigned tx, ty;
    cvk::MetricParams M = P.metric;
    M.T = cv_sc_table();
    M.LT = cv_log_table();
    M.AT = cv_atan_table();
    const unsigned texel = shade_ray<KIND>(M, P.sky, q, code, tx, ty, oob);
    unsigned char *dst = P.fb + o * 3;
    dst[0] = (unsigned char)(texel & 0xFF);
    dst[1] = (unsigned char)((texel >> 8) & 0xFF);
    dst[2] = (unsigned char)((texel >> 16) & 0xFF);
    pos = (code == cvk::CODE_POS);
    neg = (code == cvk::CODE_NEG);
    none = (code == cvk::CODE_NONE);
    if (DEBUG) {
      curvis_ray_debug *d = P.dbg + o;
      d->x[0] = 0.0; /* t and p_t: dead lanes of the integrator, filled in by the host */
      d->x[1] = q.l;
      d->x[2] = q.th;
      d->x[3] = q.ph;
      d->p[0] = 1.0;
      d->p[1] = q.p1;
      d->p[2] = q.p2;
      d->p[3] = steps ? q.p3 + 0.0 : q.p3; /* p3 + 0.0*delta of the reference (-0 -> +0) */
      d->steps = steps;
      d->code = code;
      d->tx = tx;
      d->ty = ty;
    }
  }
  flush_frame_counts(P.counters, (unsigned)(o / P.npix), valid, n_steps, 1u, pos, neg, none, oob);
}

}  // namespace
