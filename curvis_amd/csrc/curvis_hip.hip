/* curvis_hip.hip -- gfx950 kernels and C ABI of libcurvis_hip.so (see include/curvis_hip.h).
 *
 * Kernels (per-ray arithmetic lives in cv_device.h / cv_efficient.h / cv_math.h)
 *   geodesic_static<KIND,PHI,FAST,FUSED>   the default hot kernel of RelativisticSystem::render_image
 *       (src/systems.rs:307-330): one ray per lane -- pixel -> photon -> forward-Euler loop to escape or
 *       cap with a wave-uniform step counter -> (FUSED) tangent direction, nearest sky texel, RGB8 store in
 *       the epilogue, i.e. rows R1-R10 of SURVEY.md 8a in ONE launch per batch of frames and no
 *       intermediate HBM traffic.  Hardware block scheduling balances the grid.
 *   geodesic_persistent<KIND,PHI,FAST>     persistent waves: when `refill_threshold` lanes of a wave have
 *       terminated they are stored together and the free lanes are refilled from a global ray queue with
 *       ONE wave-aggregated atomic (ballot + popcount + mbcnt rank), so lanes never idle behind a slow
 *       neighbour.  Final states are staged in the ray store and shaded by shade_kernel.  Selectable
 *       ("variant" = 0); measured 3-8 % slower than the static kernel on every workload tried.
 *   shade_kernel<KIND,DEBUG>               staged shading (persistent kernel, debug dump of every ray).
 *   escape_angle_kernel<KIND,FAST>, efficient_pixel_kernel   render_image_efficient (src/systems.rs:333-527).
 *   selftest_math_kernel                   cv_math.h / IEEE div / sqrt / hardware seeds for the tests.
 *   FAST = shared-reciprocal Euler step (cv_device.h ray_step_fast), !FAST = compiler IEEE div/sqrt;
 *   PHI = integrate phi as well (debug dump, escape angles).
 *
 * Ray order: rays are numbered by 8x8 pixel tiles (tile-major, then row-major inside the tile) so the 64
 * rays of a wave are spatial neighbours: similar step counts, neighbouring sky texels, and 3-byte stores
 * that cover whole 24-byte row segments.
 *
 * No MFMA, no LDS: the loop is an issue-bound chain of FP64 VALU ops (5 divisions, sqrt, sincos per step)
 * on five registers of state; HBM traffic is 3 B out + 4 B in per ~2000 steps (DESIGN.md sections 5-6).
 */
#include <hip/hip_runtime.h>
#include <dirent.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/curvis_hip.h"
#include "cv_device.h"
#include "cv_efficient.h"
#include "cv_host.h"
#include "cv_sampler.h"
#include "host/jpeg_io.h" /* PNG + JPEG decoders shared with the curvis binary */

#pragma clang fp contract(off)

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
  unsigned steps = 0, k0 = 0;
  int code = cvk::CODE_NONE;
  if (fresh) {
    if (valid) {
      cvk::ray_init<KIND>(M, P.cams[frame], px, py + P.row0, q);
      active = P.max_iter != 0;
      steps = P.max_iter;
    }
  } else {
    q.l = ld_sys(&P.store.l[id]);
    q.th = ld_sys(&P.store.th[id]);
    q.p1 = ld_sys(&P.store.p1[id]);
    q.p2 = ld_sys(&P.store.p2[id]);
    q.p3 = ld_sys(&P.store.p3[id]);
    q.p3sq = q.p3 * q.p3;
    if (corrupt) { /* what a hand-over that lost stores would look like: the rays of the tile land elsewhere */
      q.th = q.th + 0.25;
      q.p1 = -q.p1;
    }
    steps = ld_sys(&P.store.steps[id]);
    const int c = ld_sys(&P.store.code[id]);
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
    if (__hip_atomic_load(&Q->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <=
        __hip_atomic_load(&Q->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      continue;
    parked = true;
    break;
  }
  unsigned pos = 0, neg = 0, none = 0, oob = 0;
  if (parked) {
    st_sys(&P.store.l[id], q.l);
    st_sys(&P.store.th[id], q.th);
    st_sys(&P.store.p1[id], q.p1);
    st_sys(&P.store.p2[id], q.p2);
    st_sys(&P.store.p3[id], q.p3);
    st_sys(&P.store.steps[id], active ? k0 : steps);
    st_sys(&P.store.code[id], (code == cvk::CODE_POS ? 1 : code == cvk::CODE_NEG ? 2 : 0) | (active ? 4 : 0));
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

/* seat belt of the relay kernel: number of differing 8-byte words of two framebuffers (one atomic per wave that saw one) */
__global__ __launch_bounds__(256) void compare_kernel(const unsigned long long *a, const unsigned long long *b, size_t n_words,
                                                      unsigned long long *n_diff) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned mine = 0;
  for (; i < n_words; i += stride) mine += a[i] != b[i];
  const unsigned long long m = __builtin_amdgcn_ballot_w64(mine != 0);
  if (mine) atomicAdd(n_diff, (unsigned long long)mine);
  (void)m;
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

/* K2: compute_escape_angle (src/systems.rs:203-261) for a batch of alphas: photon at (0, l, pi/2, 0)
 * with tangent direction (cos a, 0, sin a), Euler loop WITH phi, world direction, angle. */
template <int KIND, bool FAST>
__global__ __launch_bounds__(64) void escape_angle_kernel(const EscapeAngleParams P) {
  __shared__ MathTablesLds<KIND> s_tab;
  cvk::MetricParams M = P.metric;
  load_math_tables<KIND>(s_tab, M);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const double alpha = P.alpha[i];
  double sa, ca;
  cv_sincos(alpha, &sa, &ca);
  const double pos[4] = {0.0, P.l_cam[i], CV_PI / 2.0, 0.0};
  cvk::Ray q;
  cvk::ray_init_dir<KIND>(M, pos, ca, 0.0, sa, q);
  const bool lane_ok = FAST && P.fast_ok && cvk::ray_fast_ok(q);
  /* same loop shape as geodesic_static: wave-uniform step counter (all lanes start together), one escape compare,
   * the per-lane step count captured under a scalar branch in the iterations in which some lane escapes */
  unsigned steps = P.max_iter;
  int code = cvk::CODE_NONE;
  if (P.max_iter != 0) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned k = 0;
    for (;;) {
      ++k;
      one_step<KIND, true, FAST, true>(M, P.delta, q, lane_ok); /* equatorial photons: see ray_step_fast */
      const bool esc = ray_escaped(q.l, P.max_radius);
      const unsigned long long em = __builtin_amdgcn_ballot_w64(esc);
      if (em) {
        unsigned kv;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kv) : "s"(k));
        if ((em >> lane) & 1ull) steps = kv;
      }
      if (esc) break;
      if (k >= P.max_iter) break;
    }
    if (ray_escaped(q.l, P.max_radius)) code = escape_code(q.l);
  } else {
    steps = 0;
  }
  const double nan = __builtin_nan("");
  double angle = nan, space = nan;
  int status = code;
  if (code != cvk::CODE_NONE) {
    if (cvk::escape_angle_of<KIND>(M, q, angle)) {
      space = (code == cvk::CODE_POS) ? 1.0 : -1.0;
    } else {
      angle = nan;
      status = cvk::ESC_PANIC;
    }
  }
  P.angle[i] = angle;
  P.space[i] = space;
  P.steps[i] = steps;
  P.status[i] = status;
}

struct EfficientPixelParams {
  cvk::SkyParams sky[2];
  const cvk::CameraParams *cams;      /* n_frames */
  const cvk::EfficientFrame *frames;  /* n_frames */
  const unsigned *tab_off;            /* n_frames: offset of the frame's tables in sx/m/c */
  const unsigned *tab_n;              /* n_frames: number of samples */
  const double *sx, *m_e, *c_e, *m_s, *c_s;
  unsigned n_frames, W, H;
  unsigned char *fb;
  FrameCounters counters;
};

/* K3: steps 2, 4, 5 of render_image_efficient + sky lookup, one thread per pixel. */
__global__ __launch_bounds__(256) void efficient_pixel_kernel(const EfficientPixelParams P) {
  const unsigned long long o = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long npix = (unsigned long long)P.W * P.H;
  unsigned pos = 0, neg = 0, none = 0, oob = 0;
  const bool valid = o < npix * P.n_frames;
  const unsigned f = valid ? (unsigned)(o / npix) : 0u;
  if (valid) {
    const unsigned pix = (unsigned)(o - (unsigned long long)f * npix);
    const unsigned py = pix / P.W, px = pix - py * P.W;
    const unsigned off = P.tab_off[f], n = P.tab_n[f];
    double fin[3], space;
    cvk::efficient_pixel(P.cams[f], P.frames[f], px, py, P.sx + off, P.m_e + off, P.c_e + off, P.m_s + off,
                         P.c_s + off, n, fin, space);
    unsigned texel = 0xFF000000u;
    if (space == 1.0 || space == -1.0) { /* match escape_space { 1.0 => ..., -1.0 => ..., _ => black } */
      const cvk::SkyParams &S = P.sky[space == 1.0 ? 0 : 1];
      unsigned tx, ty;
      cvk::sky_indices(S, fin[0], fin[1], fin[2], tx, ty);
      if (tx >= S.w || ty >= S.h) oob = 1;
      if (tx >= S.w) tx = S.w - 1;
      if (ty >= S.h) ty = S.h - 1;
      texel = S.texels[(size_t)ty * S.w + tx];
      pos = (space == 1.0);
      neg = (space == -1.0);
    } else {
      none = 1;
    }
    unsigned char *dst = P.fb + o * 3;
    dst[0] = (unsigned char)(texel & 0xFF);
    dst[1] = (unsigned char)((texel >> 8) & 0xFF);
    dst[2] = (unsigned char)((texel >> 16) & 0xFF);
  }
  flush_frame_counts(P.counters, f, valid, 0ull, 1u, pos, neg, none, oob);
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
  curvis_ray_debug *d_dbg = nullptr;
  size_t dbg_cap = 0;
  unsigned char *d_store = nullptr; /* RayStore arrays, carved from one allocation */
  unsigned char *d_rq = nullptr;    /* RelayQueue + ticket ring of the relay kernel */
  unsigned char *d_verify = nullptr; /* copy of the relay kernel's frame while the static kernel re-renders it (seat belt) */
  size_t verify_cap = 0;
  int relay_segment = 0;            /* steps between two hand-over points; 0 = automatic */
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
  std::set<std::array<uint32_t, 5>> relay_verified; /* shapes whose first relay launch has been checked */
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

/* counter block for a launch of n_frames frames: device block + pinned mirror, zeroed on the stream */
int prepare_counters(curvis_ctx *ctx, unsigned n_frames, FrameCounters &C) {
  C.slots = counter_slots_for(n_frames);
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
  rc = ensure_device(ctx, ctx->d_fb, ctx->fb_cap, fb_bytes);
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
  const std::array<uint32_t, 5> shape = {W, H, n_frames, (uint32_t)metric->kind, (uint32_t)(fast ? 1 : 0)};
  const bool auto_check = relay && !ctx->relay_verify && ctx->relay_auto_verify && !ctx->relay_verified.count(shape);
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
      fprintf(stderr, "[curvis] relay kernel: first launch of shape %ux%u x %u frame(s) differs from the static kernel (%llu of %zu 8-byte words%s); "
                      "this context uses the static kernel from now on\n", W, H, n_frames, n_diff_words, n_words, n_diff_words ? "" : ", counters only");
      ctx->relay_disabled = 1;
      ctx->relay_fallbacks++;
      return render_impl(ctx, metric, cams, n_frames, max_iterations, max_radius, delta, rgb_out, dbg_out, stats, row_begin, row_count);
    }
    ctx->relay_verified.insert(shape);
    /* the launch that counts is the relay one: its frame is what d_fb holds again (same bytes), and so are its statistics */
    ctx->last_frame_stats = fs;
    ctx->last_relay_launches = launches;
    ctx->last_relay_parks = parks;
    ctx->last_relay_waiters = waiters;
    ctx->last_integrate_ms = keep_i;
    ctx->last_shade_ms = keep_s;
  }
  if (rgb_out) HIP_TRY(ctx, hipMemcpyAsync(rgb_out, ctx->d_fb, fb_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (dbg_out)
    HIP_TRY(ctx, hipMemcpyAsync(dbg_out, ctx->d_dbg, sizeof(curvis_ray_debug) * npix * n_frames,
                                hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
    cvk::vector3_from_theta_phi(cams[f].pos[2], cams[f].pos[3], eframes[f].cam_bg);
    const double ex[3] = {1.0, 0.0, 0.0};
    if (!cvk::rotation_from_two_vectors(ex, eframes[f].cam_bg, eframes[f].rot_bg))
      return fail(ctx, CURVIS_E_PARALLEL, "v1 and v2 must not be parallel (src/algebra.rs:95-97, camera on the x axis)");
  }

  /* step 3: one sampler per frame, advanced in lock step; every round is ONE kernel launch */
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
  std::vector<unsigned> tab_off(n_frames), tab_n(n_frames);
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
    x.resize(slots, 0.0);
    sx.insert(sx.end(), x.begin(), x.end());
  }

  /* device buffers for K3 */
  const size_t npix = (size_t)W * H;
  const size_t fb_bytes = npix * 3 * n_frames;
  rc = ensure_device(ctx, ctx->d_fb, ctx->fb_cap, fb_bytes);
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
  std::memcpy(stage.data() + o_sx, sx.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_me, m_e.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_ce, c_e.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_ms, m_s.data(), sizeof(double) * T);
  std::memcpy(stage.data() + o_cs, c_s.data(), sizeof(double) * T);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eff, stage.data(), off, hipMemcpyHostToDevice, ctx->stream));
  FrameCounters FC;
  rc = prepare_counters(ctx, n_frames, FC);
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
  const unsigned long long blocks = ((unsigned long long)npix * n_frames + 255ull) / 256ull;
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(efficient_pixel_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, Q);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(unsigned long long) * cnt_words,
                              hipMemcpyDeviceToHost, ctx->stream));
  if (rgb_out) HIP_TRY(ctx, hipMemcpyAsync(rgb_out, ctx->d_fb, fb_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
  cvk::vector3_from_theta_phi(cam->pos[2], cam->pos[3], P.frame.cam_bg); /* src/systems.rs:393-397 */
  const double ex[3] = {1.0, 0.0, 0.0};
  if (!cvk::rotation_from_two_vectors(ex, P.frame.cam_bg, P.frame.rot_bg))
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
  rc = ensure_device(ctx, ctx->d_fb, ctx->fb_cap, fb_bytes);
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
  if (rgb_out) HIP_TRY(ctx, hipMemcpyAsync(rgb_out, ctx->d_fb, fb_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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

/* ------------------------------------------------------------------------------------------ ABI */
extern "C" {

const char *curvis_version(void) { return "curvis_amd 0.1 (gfx950, abi 1)"; }

const char *curvis_last_error(const curvis_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int curvis_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int curvis_ctx_create(int device, curvis_ctx **out) {
  if (!out) return fail(nullptr, CURVIS_E_INVALID, "out is null");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, CURVIS_E_NO_DEVICE,
                "no HIP device visible: libcurvis_hip has no CPU fallback (hipGetDeviceCount: " +
                    std::string(hipGetErrorString(e)) + ")");
  if (device < 0 || device >= n) return fail(nullptr, CURVIS_E_NO_DEVICE, "device index out of range");
  curvis_ctx *ctx = new curvis_ctx();
  ctx->device = device;
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 9; ++i) ctx->sky_inv_rot[s][i] = (i % 4 == 0) ? 1.0 : 0.0;
  auto bail = [&](const std::string &m) {
    g_create_error = m;
    curvis_ctx_destroy(ctx);
    return CURVIS_E_HIP;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(std::string("hipSetDevice: ") + hipGetErrorString(e));
  if ((e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess)
    return bail(std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is ") + ctx->prop.gcnArchName + ", this library carries gfx950 code only";
    curvis_ctx_destroy(ctx);
    return CURVIS_E_NO_DEVICE;
  }
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return bail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
      (e = hipEventCreate(&ctx->ev2)) != hipSuccess)
    return bail(std::string("hipEventCreate: ") + hipGetErrorString(e));
  *out = ctx;
  return CURVIS_OK;
}

void curvis_ctx_destroy(curvis_ctx *ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int s = 0; s < 2; ++s)
    if (ctx->d_sky[s] && ctx->sky_owned[s]) (void)hipFree(ctx->d_sky[s]);
  if (ctx->d_fb) (void)hipFree(ctx->d_fb);
  if (ctx->d_dbg) (void)hipFree(ctx->d_dbg);
  if (ctx->d_store) (void)hipFree(ctx->d_store);
  if (ctx->d_rq) (void)hipFree(ctx->d_rq);
  if (ctx->d_verify) (void)hipFree(ctx->d_verify);
  if (ctx->d_eff) (void)hipFree(ctx->d_eff);
  if (ctx->h_eff) (void)hipHostFree(ctx->h_eff);
  if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
  if (ctx->d_cams) (void)hipFree(ctx->d_cams);
  if (ctx->h_cams) (void)hipHostFree(ctx->h_cams);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int curvis_ctx_device_info(const curvis_ctx *ctx, char *name, size_t name_cap, int *compute_units, int *clock_mhz) {
  if (!ctx) return CURVIS_E_INVALID;
  if (name && name_cap) {
    std::snprintf(name, name_cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  }
  if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
  if (clock_mhz) *clock_mhz = ctx->prop.clockRate / 1000;
  return CURVIS_OK;
}

/* first integer of a small sysfs file matched by `glob`-less path pieces; -1 when unreadable */
static long read_sysfs_long(const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "r");
  if (!f) return -1;
  long v = -1;
  if (std::fscanf(f, "%ld", &v) != 1) v = -1;
  std::fclose(f);
  return v;
}

int curvis_ctx_device_status(const curvis_ctx *ctx, char *pci_bus_id, size_t cap, int *sclk_mhz, int *power_w) {
  if (!ctx) return CURVIS_E_INVALID;
  char id[64] = {0};
  if (hipDeviceGetPCIBusId(id, (int)sizeof id, ctx->device) != hipSuccess) id[0] = 0;
  for (char *p = id; *p; ++p) *p = (char)std::tolower((unsigned char)*p); /* sysfs spells the address in lower case */
  if (pci_bus_id && cap) std::snprintf(pci_bus_id, cap, "%s", id);
  const std::string dev = std::string("/sys/bus/pci/devices/") + id;
  if (sclk_mhz) { /* pp_dpm_sclk: one line per level, "1: 2100Mhz *" marks the current one */
    *sclk_mhz = -1;
    if (FILE *f = std::fopen((dev + "/pp_dpm_sclk").c_str(), "r")) {
      char line[128];
      while (std::fgets(line, sizeof line, f)) {
        int level = 0, mhz = 0;
        if (std::strchr(line, '*') && std::sscanf(line, "%d: %dMhz", &level, &mhz) == 2) *sclk_mhz = mhz;
      }
      std::fclose(f);
    }
  }
  if (power_w) { /* hwmon/hwmonN/power1_average (or power1_input), microwatts */
    *power_w = -1;
    if (DIR *dir = opendir((dev + "/hwmon").c_str())) {
      while (struct dirent *e = readdir(dir)) {
        if (std::strncmp(e->d_name, "hwmon", 5) != 0) continue;
        const std::string h = dev + "/hwmon/" + e->d_name;
        long uw = read_sysfs_long(h + "/power1_average");
        if (uw < 0) uw = read_sysfs_long(h + "/power1_input");
        if (uw >= 0) {
          *power_w = (int)(uw / 1000000);
          break;
        }
      }
      closedir(dir);
    }
  }
  return CURVIS_OK;
}

static int set_sky_common(curvis_ctx *ctx, int which, uint32_t w, uint32_t h) {
  if (!ctx) return CURVIS_E_INVALID;
  if (which < 0 || which > 1) return fail(ctx, CURVIS_E_INVALID, "which must be 0 (+l) or 1 (-l)");
  if (w == 0 || h == 0) return fail(ctx, CURVIS_E_INVALID, "empty sky image");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->d_sky[which] && ctx->sky_owned[which]) HIP_TRY(ctx, hipFree(ctx->d_sky[which]));
  ctx->d_sky[which] = nullptr;
  ctx->sky_owned[which] = false;
  ctx->sky_w[which] = w;
  ctx->sky_h[which] = h;
  return CURVIS_OK;
}

int curvis_ctx_set_sky(curvis_ctx *ctx, int which, const uint8_t *rgba, uint32_t w, uint32_t h) {
  int rc = set_sky_common(ctx, which, w, h);
  if (rc) return rc;
  const size_t bytes = (size_t)w * h * 4;
  HIP_TRY(ctx, hipMalloc(&ctx->d_sky[which], bytes));
  ctx->sky_owned[which] = true;
  if (rgba) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sky[which], rgba, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return CURVIS_OK;
}

int curvis_ctx_set_sky_device(curvis_ctx *ctx, int which, const void *dev_rgba, uint32_t w, uint32_t h, int copy) {
  if (!dev_rgba) return fail(ctx, CURVIS_E_INVALID, "null device pointer");
  int rc = set_sky_common(ctx, which, w, h);
  if (rc) return rc;
  const size_t bytes = (size_t)w * h * 4;
  if (copy) {
    HIP_TRY(ctx, hipMalloc(&ctx->d_sky[which], bytes));
    ctx->sky_owned[which] = true;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sky[which], dev_rgba, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    ctx->d_sky[which] = const_cast<void *>(dev_rgba);
    ctx->sky_owned[which] = false;
  }
  return CURVIS_OK;
}

int curvis_ctx_set_sky_orientation(curvis_ctx *ctx, int which, const double forward[3], const double up[3]) {
  if (!ctx || !forward || !up || which < 0 || which > 1) return fail(ctx, CURVIS_E_INVALID, "bad argument");
  cvh::Orientation o;
  if (!cvh::orientation_new(cvh::Vec3{forward[0], forward[1], forward[2]}, cvh::Vec3{up[0], up[1], up[2]}, o))
    return fail(ctx, CURVIS_E_PARALLEL, "Forward and up vectors must not be parallel (src/algebra.rs:19-21)");
  for (int i = 0; i < 9; ++i) ctx->sky_inv_rot[which][i] = o.inverse_rotation.m[i];
  return CURVIS_OK;
}

int curvis_ctx_bcast_skies(curvis_ctx *ctx, void *nccl_comm, int root) {
  if (!ctx || !nccl_comm) return fail(ctx, CURVIS_E_INVALID, "null context or communicator");
  ncclComm_t comm = (ncclComm_t)nccl_comm;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rank = -1;
  if (ncclCommUserRank(comm, &rank) != ncclSuccess) return fail(ctx, CURVIS_E_RCCL, "ncclCommUserRank failed");
  /* header first: {root_ok, w0, h0, w1, h1}, then the two textures.  root_ok travels with the shapes so that a
   * root without skies makes EVERY rank return CURVIS_E_NO_SKY together -- a root that returned before the
   * collective would leave its peers waiting inside ncclBroadcast for ever. */
  uint32_t *d_hdr = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&d_hdr, 5 * sizeof(uint32_t)));
  uint32_t hdr[5] = {0u, ctx->sky_w[0], ctx->sky_h[0], ctx->sky_w[1], ctx->sky_h[1]};
  if (rank == root) {
    hdr[0] = (ctx->d_sky[0] && ctx->d_sky[1]) ? 1u : 0u;
    HIP_TRY(ctx, hipMemcpyAsync(d_hdr, hdr, sizeof hdr, hipMemcpyHostToDevice, ctx->stream));
  }
  if (ncclBroadcast(d_hdr, d_hdr, 5, ncclUint32, root, comm, ctx->stream) != ncclSuccess) {
    (void)hipFree(d_hdr);
    return fail(ctx, CURVIS_E_RCCL, "ncclBroadcast(header) failed");
  }
  HIP_TRY(ctx, hipMemcpyAsync(hdr, d_hdr, sizeof hdr, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(d_hdr);
  if (hdr[0] != 1u)
    return fail(ctx, CURVIS_E_NO_SKY, rank == root ? "root rank must hold both skies before the broadcast"
                                                   : "the root rank of the sky broadcast holds no skies");
  const uint32_t *shape = hdr + 1;
  for (int s = 0; s < 2; ++s) {
    const uint32_t w = shape[2 * s], h = shape[2 * s + 1];
    if (rank != root) {
      int rc = curvis_ctx_set_sky(ctx, s, nullptr, w, h);
      if (rc) return rc;
    }
    if (ncclBroadcast(ctx->d_sky[s], ctx->d_sky[s], (size_t)w * h * 4, ncclUint8, root, comm, ctx->stream) !=
        ncclSuccess)
      return fail(ctx, CURVIS_E_RCCL, "ncclBroadcast(sky) failed");
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CURVIS_OK;
}

static_assert(sizeof(ncclUniqueId) == CURVIS_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes in RCCL");

int curvis_rccl_unique_id(uint8_t id[CURVIS_RCCL_ID_BYTES]) {
  if (!id) return fail(nullptr, CURVIS_E_INVALID, "null id");
  ncclUniqueId u;
  const ncclResult_t rc = ncclGetUniqueId(&u);
  if (rc != ncclSuccess) return fail(nullptr, CURVIS_E_RCCL, std::string("ncclGetUniqueId: ") + ncclGetErrorString(rc));
  std::memcpy(id, &u, sizeof u);
  return CURVIS_OK;
}

int curvis_ctx_rccl_comm_init(curvis_ctx *ctx, const uint8_t id[CURVIS_RCCL_ID_BYTES], int n_ranks, int rank,
                              void **comm_out) {
  if (!ctx || !id || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return fail(ctx, CURVIS_E_INVALID, "bad argument");
  *comm_out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = ncclCommInitRank(&comm, n_ranks, u, rank);
  if (rc != ncclSuccess) return fail(ctx, CURVIS_E_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(rc));
  *comm_out = (void *)comm;
  return CURVIS_OK;
}

int curvis_rccl_comm_destroy(void *nccl_comm) {
  if (!nccl_comm) return CURVIS_OK;
  return ncclCommDestroy((ncclComm_t)nccl_comm) == ncclSuccess ? CURVIS_OK : CURVIS_E_RCCL;
}

int curvis_ctx_read_sky(curvis_ctx *ctx, int which, size_t offset, size_t bytes, uint8_t *out) {
  if (!ctx || !out || which < 0 || which > 1) return fail(ctx, CURVIS_E_INVALID, "bad argument");
  if (!ctx->d_sky[which]) return fail(ctx, CURVIS_E_NO_SKY, "sky not set");
  const size_t total = (size_t)ctx->sky_w[which] * ctx->sky_h[which] * 4;
  if (offset > total || bytes > total - offset) return fail(ctx, CURVIS_E_INVALID, "range outside the texture");
  if (bytes == 0) return CURVIS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(out, (const uint8_t *)ctx->d_sky[which] + offset, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CURVIS_OK;
}

int curvis_orientation_init(const double forward[3], const double up[3], double rot[9], double inv_rot[9],
                            double up_out[3]) {
  if (!forward || !up) return CURVIS_E_INVALID;
  cvh::Orientation o;
  if (!cvh::orientation_new(cvh::Vec3{forward[0], forward[1], forward[2]}, cvh::Vec3{up[0], up[1], up[2]}, o))
    return CURVIS_E_PARALLEL;
  if (rot) std::memcpy(rot, o.rotation.m, sizeof o.rotation.m);
  if (inv_rot) std::memcpy(inv_rot, o.inverse_rotation.m, sizeof o.inverse_rotation.m);
  if (up_out) {
    up_out[0] = o.up.x;
    up_out[1] = o.up.y;
    up_out[2] = o.up.z;
  }
  return CURVIS_OK;
}

int curvis_camera_init(curvis_camera *out, const double pos[4], const double forward[3], const double up[3],
                       double focal_length, double sensor_diagonal, uint32_t res_x, uint32_t res_y) {
  if (!out || !pos || !forward || !up) return CURVIS_E_INVALID;
  if (!(focal_length > 0.0)) return CURVIS_E_INVALID;    /* src/cameras.rs:92 */
  if (!(sensor_diagonal > 0.0)) return CURVIS_E_INVALID; /* :95 */
  if (res_x == 0 || res_y == 0) return CURVIS_E_INVALID; /* :98 */
  int rc = curvis_orientation_init(forward, up, out->rot, nullptr, nullptr);
  if (rc) return rc;
  for (int i = 0; i < 4; ++i) out->pos[i] = pos[i];
  const double aspect = (double)res_x / (double)res_y; /* :107-110 */
  const double aspect2 = aspect * aspect;
  out->sensor_h = std::sqrt(sensor_diagonal * sensor_diagonal / (aspect2 + 1.0));
  out->sensor_w = aspect * out->sensor_h;
  out->focal = focal_length;
  out->res_x = res_x;
  out->res_y = res_y;
  return CURVIS_OK;
}

int curvis_metric_validate(const curvis_metric *m) {
  if (!m) return CURVIS_E_INVALID;
  switch (m->kind) {
    case CURVIS_METRIC_ELLIS:
      return (m->rho <= 0.0 || m->rho != m->rho) ? CURVIS_E_METRIC : CURVIS_OK;
    case CURVIS_METRIC_INTERSTELLAR:
      if (m->m <= 0.0 || m->a <= 0.0 || m->rho <= 0.0) return CURVIS_E_METRIC;
      if (m->m != m->m || m->a != m->a || m->rho != m->rho) return CURVIS_E_METRIC;
      return CURVIS_OK;
    case CURVIS_METRIC_FLAT:
      return CURVIS_OK;
    default:
      return CURVIS_E_METRIC;
  }
}

int curvis_metric_functions(const curvis_metric *m, double l, double *r, double *r_squared, double *r_derivative) {
  if (!m) return CURVIS_E_INVALID;
  if (curvis_metric_validate(m) != CURVIS_OK) return CURVIS_E_METRIC;
  const cvk::MetricParams MP = make_metric(*m);
  double rr, r2, rd;
  switch (m->kind) {
    case CURVIS_METRIC_ELLIS: cvk::metric_eval<cvk::METRIC_ELLIS>(MP, l, rr, r2, rd); break;
    case CURVIS_METRIC_INTERSTELLAR: cvk::metric_eval<cvk::METRIC_INTERSTELLAR>(MP, l, rr, r2, rd); break;
    default: cvk::metric_eval<cvk::METRIC_FLAT>(MP, l, rr, r2, rd); break;
  }
  if (r) *r = rr;
  if (r_squared) *r_squared = r2;
  if (r_derivative) *r_derivative = rd;
  return CURVIS_OK;
}

int curvis_metric_tensor(const curvis_metric *m, const double position[4], double g_cov[4], double g_contr[4]) {
  if (!m || !position) return CURVIS_E_INVALID;
  double r2;
  const int rc = curvis_metric_functions(m, position[1], nullptr, &r2, nullptr);
  if (rc != CURVIS_OK) return rc;
  const double s = cv_sin(position[2]);
  const double g[4] = {-1.0, 1.0, r2, r2 * (s * s)};
  for (int i = 0; i < 4; ++i) {
    if (g_cov) g_cov[i] = g[i];
    if (g_contr) g_contr[i] = 1.0 / g[i];
  }
  return CURVIS_OK;
}

int curvis_camera_outward_vector(const curvis_camera *camera, uint32_t px, uint32_t py, double camera_space[3],
                                 double world_space[3]) {
  if (!camera || camera->res_x == 0 || camera->res_y == 0) return CURVIS_E_INVALID;
  /* the first half of cvk::ray_init, expression for expression */
  const double h = 0.5 - ((double)py / (double)camera->res_y);
  const double w = ((double)px / (double)camera->res_x) - 0.5;
  double vx = camera->focal * 1.0;
  double vy = -camera->sensor_w * w;
  double vz = camera->sensor_h * h;
  const double n = std::sqrt(vx * vx + vy * vy + vz * vz);
  vx = vx / n;
  vy = vy / n;
  vz = vz / n;
  if (camera_space) {
    camera_space[0] = vx;
    camera_space[1] = vy;
    camera_space[2] = vz;
  }
  if (world_space) cvk::mat3_vec(camera->rot, vx, vy, vz, world_space[0], world_space[1], world_space[2]);
  return CURVIS_OK;
}

int curvis_vector_to_direction(const curvis_metric *metric, const double position[4], const double p_cov[4],
                               double direction[3]) {
  if (!metric || !position || !p_cov || !direction) return CURVIS_E_INVALID;
  if (curvis_metric_validate(metric) != CURVIS_OK) return CURVIS_E_METRIC;
  const cvk::MetricParams MP = make_metric(*metric);
  cvk::Ray q;
  q.l = position[1];
  q.th = position[2];
  q.ph = position[3];
  q.p1 = p_cov[1];
  q.p2 = p_cov[2];
  q.p3 = p_cov[3];
  q.p3sq = q.p3 * q.p3;
  switch (metric->kind) {
    case CURVIS_METRIC_ELLIS: cvk::ray_direction<cvk::METRIC_ELLIS>(MP, q, direction[0], direction[1], direction[2]); break;
    case CURVIS_METRIC_INTERSTELLAR:
      cvk::ray_direction<cvk::METRIC_INTERSTELLAR>(MP, q, direction[0], direction[1], direction[2]);
      break;
    default: cvk::ray_direction<cvk::METRIC_FLAT>(MP, q, direction[0], direction[1], direction[2]); break;
  }
  return CURVIS_OK;
}

int curvis_update_relativistic_object(const curvis_metric *metric, double x[4], double p_cov[4], double delta) {
  if (!metric || !x || !p_cov) return CURVIS_E_INVALID;
  if (curvis_metric_validate(metric) != CURVIS_OK) return CURVIS_E_METRIC;
  const cvk::MetricParams MP = make_metric(*metric);
  switch (metric->kind) {
    case CURVIS_METRIC_ELLIS: host_euler_step<cvk::METRIC_ELLIS>(MP, x, p_cov, delta); break;
    case CURVIS_METRIC_INTERSTELLAR: host_euler_step<cvk::METRIC_INTERSTELLAR>(MP, x, p_cov, delta); break;
    default: host_euler_step<cvk::METRIC_FLAT>(MP, x, p_cov, delta); break;
  }
  return CURVIS_OK;
}

int curvis_sky_texel_index(uint32_t w, uint32_t h, const double inv_rot[9], const double v[3], uint32_t *x, uint32_t *y) {
  if (!v || !x || !y || w == 0 || h == 0) return CURVIS_E_INVALID;
  cvk::SkyParams S;
  S.texels = nullptr;
  S.w = w;
  S.h = h;
  for (int i = 0; i < 9; ++i) S.inv_rot[i] = inv_rot ? inv_rot[i] : ((i % 4 == 0) ? 1.0 : 0.0);
  unsigned tx = 0, ty = 0;
  cvk::sky_indices(S, v[0], v[1], v[2], tx, ty);
  *x = tx;
  *y = ty;
  return (tx >= w || ty >= h) ? CURVIS_E_INVALID : CURVIS_OK;
}

int curvis_render_brute(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                        uint32_t max_iterations, double max_radius, double delta, uint8_t *rgb_out,
                        curvis_stats *stats) {
  return render_impl(ctx, metric, camera, 1, max_iterations, max_radius, delta, rgb_out, nullptr, stats);
}

int curvis_render_brute_rows(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                             uint32_t row_begin, uint32_t row_count, uint32_t max_iterations, double max_radius,
                             double delta, uint8_t *rgb_out, curvis_stats *stats) {
  if (row_count == 0) return fail(ctx, CURVIS_E_INVALID, "row_count must be greater than 0");
  return render_impl(ctx, metric, camera, 1, max_iterations, max_radius, delta, rgb_out, nullptr, stats, row_begin,
                     row_count);
}

int curvis_render_brute_debug(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                              uint32_t max_iterations, double max_radius, double delta, uint8_t *rgb_out,
                              curvis_ray_debug *dbg_out, curvis_stats *stats) {
  if (!dbg_out) return fail(ctx, CURVIS_E_INVALID, "dbg_out is null");
  return render_impl(ctx, metric, camera, 1, max_iterations, max_radius, delta, rgb_out, dbg_out, stats);
}

int curvis_render_brute_batch(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cameras,
                              uint32_t n_frames, uint32_t max_iterations, double max_radius, double delta,
                              uint8_t *rgb_out, curvis_stats *stats) {
  return render_impl(ctx, metric, cameras, n_frames, max_iterations, max_radius, delta, rgb_out, nullptr, stats);
}

int curvis_render_efficient(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                            uint32_t max_iterations_propagation, double max_radius, double delta, uint32_t alpha_nums,
                            uint32_t max_iterations_sampling, double sampling_convergence_threshold_1,
                            double sampling_convergence_threshold_2, uint8_t *rgb_out, curvis_stats *stats) {
  return render_efficient_impl(ctx, metric, camera, 1, max_iterations_propagation, max_radius, delta, alpha_nums,
                               max_iterations_sampling, sampling_convergence_threshold_1,
                               sampling_convergence_threshold_2, rgb_out, stats);
}

int curvis_render_efficient_batch(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cameras,
                                  uint32_t n_frames, uint32_t max_iterations_propagation, double max_radius,
                                  double delta, uint32_t alpha_nums, uint32_t max_iterations_sampling,
                                  double sampling_convergence_threshold_1, double sampling_convergence_threshold_2,
                                  uint8_t *rgb_out, curvis_stats *stats) {
  return render_efficient_impl(ctx, metric, cameras, n_frames, max_iterations_propagation, max_radius, delta,
                               alpha_nums, max_iterations_sampling, sampling_convergence_threshold_1,
                               sampling_convergence_threshold_2, rgb_out, stats);
}

int curvis_render_direct(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera, uint32_t max_iterations,
                         double max_radius, double delta, uint8_t *rgb_out, curvis_stats *stats) {
  return render_direct_impl(ctx, metric, camera, max_iterations, max_radius, delta, rgb_out, stats);
}

int curvis_ctx_sampling_info(const curvis_ctx *ctx, uint32_t frame, curvis_sampling_info *info) {
  if (!ctx || !info || frame >= ctx->last_sampling_info.size()) return CURVIS_E_INVALID;
  *info = ctx->last_sampling_info[frame];
  return CURVIS_OK;
}

int curvis_ctx_frame_stats(const curvis_ctx *ctx, uint32_t frame, curvis_stats *stats) {
  if (!ctx || !stats || frame >= ctx->last_frame_stats.size()) return CURVIS_E_INVALID;
  *stats = ctx->last_frame_stats[frame];
  return CURVIS_OK;
}

int curvis_ctx_samples(const curvis_ctx *ctx, uint32_t frame, double *alpha, double *escape_angle,
                       double *escape_space, size_t cap) {
  if (!ctx || frame >= ctx->last_samples.size()) return CURVIS_E_INVALID;
  const auto &pts = ctx->last_samples[frame];
  if (cap < pts.size()) return CURVIS_E_INVALID;
  for (size_t i = 0; i < pts.size(); ++i) {
    if (alpha) alpha[i] = pts[i].a;
    if (escape_angle) escape_angle[i] = pts[i].e;
    if (escape_space) escape_space[i] = pts[i].s;
  }
  return CURVIS_OK;
}

int curvis_new_photon(const curvis_metric *metric, const double position[4], const double direction[3], double x[4],
                      double p_cov[4]) {
  if (!metric || !position || !direction || !x || !p_cov) return CURVIS_E_INVALID;
  if (curvis_metric_validate(metric) != CURVIS_OK) return CURVIS_E_METRIC;
  const cvk::MetricParams MP = make_metric(*metric);
  cvk::Ray q;
  switch (metric->kind) {
    case CURVIS_METRIC_ELLIS:
      cvk::ray_init_dir<cvk::METRIC_ELLIS>(MP, position, direction[0], direction[1], direction[2], q);
      break;
    case CURVIS_METRIC_INTERSTELLAR:
      cvk::ray_init_dir<cvk::METRIC_INTERSTELLAR>(MP, position, direction[0], direction[1], direction[2], q);
      break;
    default:
      cvk::ray_init_dir<cvk::METRIC_FLAT>(MP, position, direction[0], direction[1], direction[2], q);
      break;
  }
  for (int i = 0; i < 4; ++i) x[i] = position[i];
  p_cov[0] = 1.0;
  p_cov[1] = q.p1;
  p_cov[2] = q.p2;
  p_cov[3] = q.p3;
  return CURVIS_OK;
}

int curvis_photon_trajectories(curvis_ctx *ctx, const curvis_metric *metric, uint32_t n_photons, const double *x0,
                               const double *p0_cov, uint32_t iterations, double delta, double *out) {
  if (!ctx) return CURVIS_E_INVALID;
  if (!metric || !x0 || !p0_cov || !out) return fail(ctx, CURVIS_E_INVALID, "null argument");
  if (curvis_metric_validate(metric) != CURVIS_OK) return fail(ctx, CURVIS_E_METRIC, "invalid metric parameters");
  if (n_photons == 0 || iterations == 0) return CURVIS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t in_bytes = (size_t)n_photons * 4 * sizeof(double);
  const size_t out_bytes = (size_t)n_photons * iterations * 8 * sizeof(double);
  int rc = ensure_device(ctx, ctx->d_eff, ctx->eff_cap, 2 * in_bytes + out_bytes);
  if (rc) return rc;
  double *d_x = (double *)ctx->d_eff, *d_p = d_x + (size_t)n_photons * 4, *d_out = d_p + (size_t)n_photons * 4;
  HIP_TRY(ctx, hipMemcpyAsync(d_x, x0, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_p, p0_cov, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  TrajectoryParams P;
  P.metric = make_metric(*metric);
  P.x0 = d_x;
  P.p0 = d_p;
  P.out = d_out;
  P.n = n_photons;
  P.iterations = iterations;
  P.delta = delta;
  const unsigned blocks = (n_photons + 63u) / 64u;
  switch (metric->kind) {
    case CURVIS_METRIC_ELLIS:
      hipLaunchKernelGGL((trajectory_kernel<cvk::METRIC_ELLIS>), dim3(blocks), dim3(64), 0, ctx->stream, P);
      break;
    case CURVIS_METRIC_INTERSTELLAR:
      hipLaunchKernelGGL((trajectory_kernel<cvk::METRIC_INTERSTELLAR>), dim3(blocks), dim3(64), 0, ctx->stream, P);
      break;
    default:
      hipLaunchKernelGGL((trajectory_kernel<cvk::METRIC_FLAT>), dim3(blocks), dim3(64), 0, ctx->stream, P);
      break;
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CURVIS_OK;
}

int curvis_compute_escape_angles(curvis_ctx *ctx, const curvis_metric *metric, double l, const double *alphas,
                                 uint32_t n, double delta, uint32_t max_iterations, double max_radius,
                                 double *angle, int32_t *space, uint32_t *steps) {
  if (!ctx) return CURVIS_E_INVALID;
  if (!metric || !alphas || !angle || !space) return fail(ctx, CURVIS_E_INVALID, "null argument");
  if (curvis_metric_validate(metric) != CURVIS_OK) return fail(ctx, CURVIS_E_METRIC, "invalid metric parameters");
  if (std::fabs(l) > max_radius)
    return fail(ctx, CURVIS_E_CAMERA_OUTSIDE,
                "Photon already beyond the maximum radius. Cannot evaluate escape. (src/systems.rs:122-124)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const cvk::MetricParams MP = make_metric(*metric);
  std::vector<double> a(alphas, alphas + n), ls(n, l), ang, spc;
  std::vector<uint32_t> st;
  std::vector<int> status;
  int rc = eval_escape_batch(ctx, metric, MP, a, ls, max_iterations, max_radius, delta, ang, spc, st, status, nullptr);
  if (rc) return rc;
  bool panic = false;
  for (uint32_t i = 0; i < n; ++i) {
    angle[i] = ang[i];
    space[i] = status[i] == cvk::ESC_PANIC ? 0 : status[i];
    if (steps) steps[i] = st[i];
    if (status[i] == cvk::ESC_PANIC) panic = true;
  }
  if (panic) return fail(ctx, CURVIS_E_PARALLEL, "v1 and v2 must not be parallel (src/algebra.rs:95-97) for at least one sample");
  return CURVIS_OK;
}

int curvis_image_load(const char *path, uint8_t **rgba_out, uint32_t *w, uint32_t *h) {
  if (!path || !rgba_out || !w || !h) return fail(nullptr, CURVIS_E_INVALID, "null argument");
  *rgba_out = nullptr;
  pngio::Image img;
  std::string err;
  if (!jpegio::load_image(path, img, err)) return fail(nullptr, CURVIS_E_IO, std::string(path) + ": " + err);
  uint8_t *buf = (uint8_t *)std::malloc(img.rgba.size() ? img.rgba.size() : 1);
  if (!buf) return fail(nullptr, CURVIS_E_IO, "out of memory");
  std::memcpy(buf, img.rgba.data(), img.rgba.size());
  *rgba_out = buf;
  *w = img.w;
  *h = img.h;
  return CURVIS_OK;
}

void curvis_image_free(uint8_t *rgba) { std::free(rgba); }

int curvis_image_save_rgb8(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h) {
  if (!path || !rgb || w == 0 || h == 0) return fail(nullptr, CURVIS_E_INVALID, "null argument or empty image");
  std::string err;
  if (!pngio::save_rgb8(path, rgb, w, h, err)) return fail(nullptr, CURVIS_E_IO, err);
  return CURVIS_OK;
}

int curvis_image_save_rgb8_level(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h, int level) {
  if (!path || !rgb || w == 0 || h == 0) return fail(nullptr, CURVIS_E_INVALID, "null argument or empty image");
  if (level < -1 || level > 9) return fail(nullptr, CURVIS_E_INVALID, "level must be -1 (fast writer) or 0..9 (zlib)");
  std::string err;
  if (!pngio::save_rgb8(path, rgb, w, h, err, level)) return fail(nullptr, CURVIS_E_IO, err);
  return CURVIS_OK;
}

int curvis_host_alloc(size_t bytes, void **out) {
  if (!out || bytes == 0) return fail(nullptr, CURVIS_E_INVALID, "curvis_host_alloc: null pointer or zero bytes");
  *out = nullptr;
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable); /* usable from every device's context */
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(nullptr, CURVIS_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  }
  return CURVIS_OK;
}
void curvis_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int curvis_ctx_framebuffer(curvis_ctx *ctx, void **dev_ptr, size_t *bytes) {
  if (!ctx) return CURVIS_E_INVALID;
  if (dev_ptr) *dev_ptr = ctx->d_fb;
  if (bytes) *bytes = ctx->fb_bytes;
  return CURVIS_OK;
}

int curvis_ctx_download(curvis_ctx *ctx, uint8_t *rgb_out, size_t bytes) {
  if (!ctx || !rgb_out) return CURVIS_E_INVALID;
  if (bytes > ctx->fb_bytes) return fail(ctx, CURVIS_E_INVALID, "download larger than the last frame");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(rgb_out, ctx->d_fb, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CURVIS_OK;
}

int curvis_ctx_synchronize(curvis_ctx *ctx) {
  if (!ctx) return CURVIS_E_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CURVIS_OK;
}

int curvis_ctx_set_option(curvis_ctx *ctx, const char *key, int64_t value) {
  if (!ctx || !key) return CURVIS_E_INVALID;
  const std::string k(key);
  if (k == "variant")
    ctx->variant = (int)value;
  else if (k == "refill_threshold")
    ctx->refill_threshold = (int)value;
  else if (k == "blocks_per_cu")
    ctx->blocks_per_cu = (int)value;
  else if (k == "block_threads")
    ctx->block_threads = (int)value;
  else if (k == "relay_segment")
    ctx->relay_segment = (int)value;
  else if (k == "relay_max_frames")
    ctx->relay_max_frames = (int)value;
  else if (k == "relay_min_blocks")
    ctx->relay_min_blocks = (long long)value;
  else if (k == "relay_verify")
    ctx->relay_verify = (int)value;
  else if (k == "relay_auto_verify") {
    ctx->relay_auto_verify = (int)value;
    ctx->relay_verified.clear(); /* switching it (back) on checks every shape afresh */
  } else if (k == "relay_test_corrupt")
    ctx->relay_test_corrupt = (int)value;
  else if (k == "relay_disabled")
    ctx->relay_disabled = (int)value;
  else if (k == "relay_test_fault")
    ctx->relay_test_fault = (int)value;

  else if (k == "fast_math")
    ctx->fast_math = (int)value;
  else if (k == "fuse_shade")
    ctx->fuse_shade = (int)value;
  else if (k == "sampling_speculation")
    ctx->sampling_speculation = (int)value;
  else if (k == "sampling_speculation_first")
    ctx->sampling_speculation_first = (int)value;
  else if (k == "max_store_bytes")
    ctx->max_store_bytes = (size_t)value;
  else
    return fail(ctx, CURVIS_E_INVALID, "unknown option " + k);
  return CURVIS_OK;
}

int curvis_ctx_get_option(const curvis_ctx *ctx, const char *key, int64_t *value) {
  if (!ctx || !key || !value) return CURVIS_E_INVALID;
  const std::string k(key);
  if (k == "variant")
    *value = ctx->variant;
  else if (k == "refill_threshold")
    *value = ctx->refill_threshold;
  else if (k == "blocks_per_cu")
    *value = ctx->blocks_per_cu;
  else if (k == "block_threads")
    *value = ctx->block_threads;
  else if (k == "relay_segment")
    *value = ctx->relay_segment;
  else if (k == "relay_max_frames")
    *value = ctx->relay_max_frames;
  else if (k == "relay_min_blocks")
    *value = ctx->relay_min_blocks;
  else if (k == "relay_verify")
    *value = ctx->relay_verify;
  else if (k == "relay_disabled")
    *value = ctx->relay_disabled;
  else if (k == "relay_auto_verify")
    *value = ctx->relay_auto_verify;
  else if (k == "relay_mismatches")
    *value = ctx->relay_mismatches;
  else if (k == "relay_verified_shapes")
    *value = (int64_t)ctx->relay_verified.size();
  else if (k == "relay_fallbacks")
    *value = ctx->relay_fallbacks;
  else if (k == "last_frames")
    *value = (int64_t)ctx->last_frame_stats.size();
  else if (k == "last_relay_launches")
    *value = ctx->last_relay_launches;
  else if (k == "last_relay_parks")
    *value = (int64_t)ctx->last_relay_parks;
  else if (k == "last_relay_waiters")
    *value = (int64_t)ctx->last_relay_waiters;

  else if (k == "fast_math")
    *value = ctx->fast_math;
  else if (k == "fuse_shade")
    *value = ctx->fuse_shade;
  else if (k == "sampling_speculation")
    *value = ctx->sampling_speculation;
  else if (k == "sampling_speculation_first")
    *value = ctx->sampling_speculation_first;
  else if (k == "last_sampling_launches")
    *value = ctx->last_sampling_launches;
  else if (k == "last_sampling_evaluated")
    *value = (int64_t)ctx->last_sampling_evaluated;
  else if (k == "max_store_bytes")
    *value = (int64_t)ctx->max_store_bytes;
  else
    return CURVIS_E_INVALID;
  return CURVIS_OK;
}

int curvis_selftest_math(curvis_ctx *ctx, int op, const double *a, const double *b, double *out, size_t n) {
  if (!ctx || !a || !out) return CURVIS_E_INVALID;
  if (n == 0) return CURVIS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&da, n * sizeof(double)));
  HIP_TRY(ctx, hipMalloc((void **)&dout, n * sizeof(double)));
  HIP_TRY(ctx, hipMemcpy(da, a, n * sizeof(double), hipMemcpyHostToDevice));
  if (b) {
    HIP_TRY(ctx, hipMalloc((void **)&db, n * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(db, b, n * sizeof(double), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, da, db,
                     dout, n);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(out, dout, n * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(da);
  (void)hipFree(dout);
  if (db) (void)hipFree(db);
  return CURVIS_OK;
}

} /* extern "C" */
