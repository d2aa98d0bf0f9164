/* curvis_hip.hip -- the ONE translation unit of libcurvis_hip.so (C ABI: include/curvis_hip.h), gfx950 only.
 *
 * Pieces (each included here and nowhere else; `make asm` before/after the split: the same instructions):
 *   kernels_geodesic.h   device side of RelativisticSystem::render_image (src/systems.rs:307-330, rows R1-R10 of SURVEY.md 8a)
 *   kernels_efficient.h  device side of render_image_efficient (src/systems.rs:333-527), direct mode, trajectories, math self-test
 *   render_host.h        struct curvis_ctx, kernel selection, render_impl, per-frame statistics, the relay seat belt
 *   efficient_host.h     the adaptive sampler's driver (src/sampling.rs) over batched escape-angle launches, per-pixel launch
 *   kernels_png.h, png_host.h   PNG front end on the device: the frames in HBM -> one zlib stream per frame (src/rendering.rs:110, :311)
 *   (this file)          the extern "C" entry points
 *   per-ray arithmetic: cv_device.h / cv_efficient.h / cv_sampler.h / cv_math.h (shared with the host twin of the tests)
 *
 * Kernels
 *   geodesic_relay<KIND,FAST>              DEFAULT for launches of <= 8 frames that fill the chip (>= 4 workgroups per CU):
 *       the static kernel's loop in segments; in the end-game of a launch unfinished 8x8 tiles are parked in HBM (40 B per
 *       ray, write-through) and picked up by relay workgroups the dispatcher places wherever slots are free.
 *   geodesic_static<KIND,PHI,FAST,FUSED>   default for larger batches, small frames and the debug dump, and the relay
 *       kernel's checker: one ray per lane -- pixel -> photon -> forward-Euler loop to escape or cap with a wave-uniform
 *       step counter -> (FUSED) tangent direction, nearest sky texel, RGB8 store in the epilogue, i.e. R1-R10 in ONE
 *       launch per batch of frames and no intermediate HBM traffic.  Hardware block scheduling balances the grid.
 *   geodesic_persistent<KIND,PHI,FAST>     persistent waves: when `refill_threshold` lanes of a wave have terminated they
 *       are stored together and the free lanes are refilled from a global ray queue with ONE wave-aggregated atomic
 *       (ballot + popcount + mbcnt rank).  Final states are staged in the ray store and shaded by shade_kernel.
 *       Selectable ("variant" = 0); measured 3-8 % slower than the static kernel on every workload tried.
 *   shade_kernel<KIND,DEBUG>               staged shading (persistent kernel, debug dump of every ray).
 *   escape_angle_kernel<KIND,FAST>, efficient_pixel_kernel, direct_kernel, trajectory_kernel   efficient mode and extras.
 *   selftest_math_kernel                   cv_math.h / IEEE div / sqrt / hardware seeds for the tests.
 *   FAST = shared-reciprocal Euler step (cv_device.h ray_step_fast), !FAST = compiler IEEE div/sqrt;
 *   PHI = integrate phi as well (debug dump, escape angles).
 *
 * Ray order: rays are numbered by 8x8 pixel tiles (tile-major, then row-major inside the tile) so the 64 rays of a wave
 * are spatial neighbours: similar step counts, neighbouring sky texels, 3-byte stores that cover whole 24-byte row segments.
 *
 * No MFMA (nothing here is a contraction).  Ray state lives in registers (5 doubles per ray); LDS holds the read-only
 * function tables of cv_math.h (8 KiB Ellis, 28 KiB Interstellar per workgroup).  The loop is an issue-bound chain of FP64
 * VALU ops (5 divisions, sqrt, sincos per step); HBM traffic is 3 B out + 4 B in per ~2000 steps.
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
#include <map>
#include <set>
#include <string>
#include <atomic>
#include <system_error>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/curvis_hip.h"
#include "cv_device.h"
#include "cv_efficient.h"
#include "cv_frame_host.h"
#include "cv_host.h"
#include "cv_sampler.h"
#include "cv_sampler_dev.h"
#include "host/jpeg_io.h" /* PNG + JPEG decoders shared with the curvis binary */

#pragma clang fp contract(off)

#include "kernels_geodesic.h"
#include "kernels_efficient.h"
#include "render_host.h"
#include "efficient_host.h"
#include "png_codes.h"
#include "kernels_png.h"
#include "png_host.h"

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
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream); /* a download still in flight reads d_fb / d_fb_alt */
  if (ctx->sampler_stream) (void)hipStreamSynchronize(ctx->sampler_stream); /* a prefetched sampler still writes its slot */
  for (auto &S : ctx->samp) {
    if (S.d) (void)hipFree(S.d);
    if (S.h) (void)hipHostFree(S.h);
    if (S.done) (void)hipEventDestroy(S.done);
    if (S.t0) (void)hipEventDestroy(S.t0);
    if (S.t1) (void)hipEventDestroy(S.t1);
  }
  if (ctx->sampler_stream) (void)hipStreamDestroy(ctx->sampler_stream);
  for (int s = 0; s < 2; ++s)
    if (ctx->d_sky[s] && ctx->sky_owned[s]) (void)hipFree(ctx->d_sky[s]);
  if (ctx->d_fb) (void)hipFree(ctx->d_fb);
  if (ctx->d_fb_alt) (void)hipFree(ctx->d_fb_alt);
  if (ctx->ev_fb) (void)hipEventDestroy(ctx->ev_fb);
  if (ctx->ev_dl) (void)hipEventDestroy(ctx->ev_dl);
  if (ctx->ev_streams) (void)hipEventDestroy(ctx->ev_streams);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->d_dbg) (void)hipFree(ctx->d_dbg);
  if (ctx->d_store) (void)hipFree(ctx->d_store);
  if (ctx->d_rq) (void)hipFree(ctx->d_rq);
  if (ctx->d_verify) (void)hipFree(ctx->d_verify);
  if (ctx->d_png) (void)hipFree(ctx->d_png);
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
    /* some hosts' driver stack reports no marketing name (hipDeviceProp_t::name empty: no amdgpu.ids entry): take the
     * board's product name from sysfs then, and say so rather than print nothing */
    std::string nm = ctx->prop.name;
    if (nm.find_first_not_of(' ') == std::string::npos) {
      char id[64] = {0};
      if (hipDeviceGetPCIBusId(id, (int)sizeof id, ctx->device) == hipSuccess) {
        for (char *p = id; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
        if (FILE *f = std::fopen((std::string("/sys/bus/pci/devices/") + id + "/product_name").c_str(), "r")) {
          char line[128] = {0};
          if (std::fgets(line, sizeof line, f)) nm = line;
          std::fclose(f);
          while (!nm.empty() && (nm.back() == '\n' || nm.back() == ' ')) nm.pop_back();
        }
      }
      if (nm.find_first_not_of(' ') == std::string::npos) nm = "AMD GPU, name not reported by the driver";
    }
    std::snprintf(name, name_cap, "%s (%s)", nm.c_str(), ctx->prop.gcnArchName);
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

/* what RCCL itself has to say about a failure: the result's text, the communicator's last error, an asynchronous error */
static std::string rccl_detail(ncclComm_t comm, ncclResult_t rc) {
  std::string m = ncclGetErrorString(rc);
  const char *last = ncclGetLastError(comm);
  if (last && *last) m += std::string("; last RCCL error: ") + last;
  ncclResult_t async = ncclSuccess;
  if (comm && ncclCommGetAsyncError(comm, &async) == ncclSuccess && async != ncclSuccess)
    m += std::string("; asynchronous: ") + ncclGetErrorString(async);
  return m;
}
/* stream synchronisation of one stage of the broadcast, with the stage's name in the error (first contact between two
 * devices must say WHERE it broke: a collective's enqueue succeeds, its failure shows at the synchronisation) */
static int bcast_stage_sync(curvis_ctx *ctx, ncclComm_t comm, const char *stage) {
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  ncclResult_t async = ncclSuccess;
  (void)ncclCommGetAsyncError(comm, &async);
  if (e != hipSuccess || async != ncclSuccess)
    return fail(ctx, e != hipSuccess ? CURVIS_E_HIP : CURVIS_E_RCCL,
                std::string("sky broadcast, stage ") + stage + ": " + (e != hipSuccess ? hipGetErrorString(e) : "stream synchronised") +
                    "; RCCL: " + rccl_detail(comm, async));
  return CURVIS_OK;
}

int curvis_ctx_bcast_skies(curvis_ctx *ctx, void *nccl_comm, int root) {
  if (!ctx || !nccl_comm) return fail(ctx, CURVIS_E_INVALID, "null context or communicator");
  ncclComm_t comm = (ncclComm_t)nccl_comm;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rank = -1;
  ncclResult_t nrc = ncclCommUserRank(comm, &rank);
  if (nrc != ncclSuccess) return fail(ctx, CURVIS_E_RCCL, "sky broadcast, stage communicator: ncclCommUserRank: " + rccl_detail(comm, nrc));
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
  nrc = ncclBroadcast(d_hdr, d_hdr, 5, ncclUint32, root, comm, ctx->stream);
  if (nrc != ncclSuccess) {
    (void)hipFree(d_hdr);
    return fail(ctx, CURVIS_E_RCCL, "sky broadcast, stage header_broadcast: ncclBroadcast: " + rccl_detail(comm, nrc));
  }
  int rc = bcast_stage_sync(ctx, comm, "header_broadcast");
  if (rc == CURVIS_OK) {
    const hipError_t e = hipMemcpy(hdr, d_hdr, sizeof hdr, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(ctx, CURVIS_E_HIP, std::string("sky broadcast, stage header_broadcast: reading the header back: ") + hipGetErrorString(e));
  }
  (void)hipFree(d_hdr);
  if (rc != CURVIS_OK) return rc;
  if (hdr[0] != 1u)
    return fail(ctx, CURVIS_E_NO_SKY, rank == root ? "root rank must hold both skies before the broadcast"
                                                   : "the root rank of the sky broadcast holds no skies");
  const uint32_t *shape = hdr + 1;
  for (int s = 0; s < 2; ++s) {
    const uint32_t w = shape[2 * s], h = shape[2 * s + 1];
    const char *stage = s == 0 ? "texture_broadcast(+l sky)" : "texture_broadcast(-l sky)";
    if (rank != root) {
      rc = curvis_ctx_set_sky(ctx, s, nullptr, w, h);
      if (rc) return fail(ctx, rc, std::string("sky broadcast, stage ") + stage + ": allocating the receiving texture: " + ctx->err);
    }
    nrc = ncclBroadcast(ctx->d_sky[s], ctx->d_sky[s], (size_t)w * h * 4, ncclUint8, root, comm, ctx->stream);
    if (nrc != ncclSuccess)
      return fail(ctx, CURVIS_E_RCCL, std::string("sky broadcast, stage ") + stage + ": ncclBroadcast: " + rccl_detail(comm, nrc));
    rc = bcast_stage_sync(ctx, comm, stage); /* one synchronisation per texture: the error names the texture */
    if (rc != CURVIS_OK) return rc;
  }
  return CURVIS_OK;
}

/* How two devices of this node are connected (hipExtGetLinkTypeAndHopCount / hipDeviceGetP2PAttribute): read next to the
 * first measured sky_broadcast_gbps -- xGMI is point-to-point, 7 links x ~153 GB/s per MI355X; a pair that only has PCIe
 * between it explains a broadcast an order of magnitude slower.  link_type: HSA_AMD_LINK_INFO_TYPE_* (2 PCIe, 4 xGMI),
 * 0 with hops 0 for a == b, -1 unknown.  Every output pointer may be NULL. */
int curvis_device_link(int device_a, int device_b, int *link_type, int *hops, int *peer_access, int *performance_rank,
                       int *native_atomics) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_a < 0 || device_b < 0 || device_a >= n || device_b >= n)
    return fail(nullptr, CURVIS_E_INVALID, "curvis_device_link: no such device");
  int lt = -1, hp = -1, pa = -1, pr = -1, na = -1;
  if (device_a == device_b) {
    lt = 0, hp = 0, pa = 1;
  } else {
    uint32_t t = 0, h = 0;
    if (hipExtGetLinkTypeAndHopCount(device_a, device_b, &t, &h) == hipSuccess) lt = (int)t, hp = (int)h;
    int v = 0;
    if (hipDeviceCanAccessPeer(&v, device_a, device_b) == hipSuccess) pa = v;
    if (hipDeviceGetP2PAttribute(&v, hipDevP2PAttrPerformanceRank, device_a, device_b) == hipSuccess) pr = v;
    if (hipDeviceGetP2PAttribute(&v, hipDevP2PAttrNativeAtomicSupported, device_a, device_b) == hipSuccess) na = v;
    (void)hipGetLastError();
  }
  if (link_type) *link_type = lt;
  if (hops) *hops = hp;
  if (peer_access) *peer_access = pa;
  if (performance_rank) *performance_rank = pr;
  if (native_atomics) *native_atomics = na;
  return CURVIS_OK;
}

static_assert(sizeof(ncclUniqueId) == CURVIS_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes in RCCL");

/* the ranks of these communicators sit on ONE node (frames of a video shard over the GPUs of a node): RCCL's bootstrap goes
 * over the loopback interface unless the user has chosen one -- on hosts whose first interface is slow or unroutable the
 * default choice was seen to cost 6 s to ~80 s of communicator set-up */
static void rccl_single_node_defaults() { ::setenv("NCCL_SOCKET_IFNAME", "lo", 0); }

int curvis_rccl_unique_id(uint8_t id[CURVIS_RCCL_ID_BYTES]) {
  if (!id) return fail(nullptr, CURVIS_E_INVALID, "null id");
  rccl_single_node_defaults();
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
  rccl_single_node_defaults();
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = ncclCommInitRank(&comm, n_ranks, u, rank);
  if (rc != ncclSuccess)
    return fail(ctx, CURVIS_E_RCCL, "ncclCommInitRank (rank " + std::to_string(rank) + " of " + std::to_string(n_ranks) + ", device " +
                                        std::to_string(ctx->device) + "): " + rccl_detail(nullptr, rc));
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

int curvis_ctx_prefetch_efficient(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cameras, uint32_t n_frames,
                                  uint32_t max_iterations_propagation, double max_radius, double delta, uint32_t alpha_nums,
                                  uint32_t max_iterations_sampling, double sampling_convergence_threshold_1,
                                  double sampling_convergence_threshold_2) {
  return prefetch_efficient_impl(ctx, metric, cameras, n_frames, max_iterations_propagation, max_radius, delta, alpha_nums,
                                 max_iterations_sampling, sampling_convergence_threshold_1, sampling_convergence_threshold_2);
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
  /* device-resident sampler: the tables stayed in HBM; this frame's is fetched now (the context is the caller's to mutate: a
   * context is not thread-safe, and the const in the signature promises nothing about caches) */
  const int frc = fetch_device_samples(const_cast<curvis_ctx *>(ctx), frame);
  if (frc != CURVIS_OK) return frc;
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

int curvis_ctx_deflate_frames(curvis_ctx *ctx, uint32_t res_x, uint32_t res_y, uint32_t n_frames, uint8_t *zlib_out, size_t out_cap,
                              size_t *offsets, double *kernel_ms) {
  return deflate_frames_impl(ctx, res_x, res_y, n_frames, zlib_out, out_cap, offsets, kernel_ms);
}

int curvis_ctx_deflate_frames_crc(curvis_ctx *ctx, uint32_t res_x, uint32_t res_y, uint32_t n_frames, uint8_t *zlib_out, size_t out_cap,
                                  size_t *offsets, double *kernel_ms, uint32_t *idat_crc, int *crc_valid) {
  if (!idat_crc || !crc_valid) return fail(ctx, CURVIS_E_INVALID, "null idat_crc / crc_valid");
  return deflate_frames_impl(ctx, res_x, res_y, n_frames, zlib_out, out_cap, offsets, kernel_ms, idat_crc, crc_valid);
}

int curvis_image_save_zlib_rgb8_crc(const char *path, const uint8_t *zlib_stream, size_t len, uint32_t w, uint32_t h, uint32_t idat_crc) {
  if (!path || !zlib_stream || len < 6 || w == 0 || h == 0) return fail(nullptr, CURVIS_E_INVALID, "null argument or empty stream");
  std::string err;
  if (!pngio::save_zlib_stream_rgb8(path, zlib_stream, len, w, h, err, nullptr, &idat_crc)) return fail(nullptr, CURVIS_E_IO, err);
  return CURVIS_OK;
}

int curvis_image_save_zlib_rgb8(const char *path, const uint8_t *zlib_stream, size_t len, uint32_t w, uint32_t h) {
  if (!path || !zlib_stream || len < 6 || w == 0 || h == 0) return fail(nullptr, CURVIS_E_INVALID, "null argument or empty stream");
  std::string err;
  if (!pngio::save_zlib_stream_rgb8(path, zlib_stream, len, w, h, err)) return fail(nullptr, CURVIS_E_IO, err);
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

/* the other direction: RGB8 frames from host memory into the context's framebuffer (it grows as needed), so that frames
 * made elsewhere -- a host that composites, a test with chosen contents -- can go through curvis_ctx_deflate_frames */
int curvis_ctx_upload(curvis_ctx *ctx, const uint8_t *rgb, size_t bytes) {
  if (!ctx || !rgb || bytes == 0) return fail(ctx, CURVIS_E_INVALID, "bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = fb_begin_write(ctx, bytes);
  if (rc) return rc;
  ctx->fb_bytes = bytes;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_fb, rgb, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CURVIS_OK;
}

int curvis_ctx_synchronize(curvis_ctx *ctx) {
  if (!ctx) return CURVIS_E_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CURVIS_OK;
}

/* option "async_download": the frames of the last render call that was given `rgb_out` are in host memory on return */
int curvis_ctx_download_wait(curvis_ctx *ctx) {
  if (!ctx) return CURVIS_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return download_wait(ctx);
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
  else if (k == "relay_max_hops")
    ctx->relay_max_hops = (int)value;
  else if (k == "relay_max_parks")
    ctx->relay_max_parks = (int)value;
  else if (k == "relay_recheck_every")
    ctx->relay_recheck_every = (int)value;
  else if (k == "async_download") { /* overlapped download of the frames, see fb_download (render_host.h) */
    if (!value) {
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      const int rc = download_wait(ctx);
      if (rc) return rc;
    }
    ctx->async_download = value ? 1 : 0;
  } else if (k == "async_streams") { /* curvis_ctx_deflate_frames returns with its streams still on their way: png_host.h */
    if (!value) {
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      const int rc = download_wait(ctx);
      if (rc) return rc;
    }
    ctx->async_streams = value ? 1 : 0;
  } else if (k == "relay_max_frames")
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
  else if (k == "device_sampler")
    ctx->device_sampler = (int)value;
  else if (k == "device_sampler_min_frames")
    ctx->device_sampler_min_frames = value < 1 ? 1 : (int)value;
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
  else if (k == "relay_max_hops")
    *value = ctx->relay_max_hops;
  else if (k == "relay_max_parks")
    *value = ctx->relay_max_parks;
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
  else if (k == "relay_checks")
    *value = (int64_t)ctx->relay_checks;
  else if (k == "last_png_stream_bytes")
    *value = (int64_t)ctx->last_png_stream_bytes;
  else if (k == "relay_recheck_every")
    *value = ctx->relay_recheck_every;
  else if (k == "async_download")
    *value = ctx->async_download;
  else if (k == "async_streams")
    *value = ctx->async_streams;
  else if (k == "streams_pending")
    *value = ctx->streams_pending ? 1 : 0;
  else if (k == "downloads_overlapped")
    *value = (int64_t)ctx->downloads_overlapped;
  else if (k == "download_pending")
    *value = ctx->dl_pending ? 1 : 0;
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
  else if (k == "device_sampler")
    *value = ctx->device_sampler;
  else if (k == "device_sampler_min_frames")
    *value = ctx->device_sampler_min_frames;
  else if (k == "last_sampler_path")
    *value = ctx->last_sampler_path;
  else if (k == "last_sampling_chains")
    *value = ctx->last_sampling_chains;
  else if (k == "last_sampling_prefetched")
    *value = ctx->last_sampling_prefetched;
  else if (k == "prefetches")
    *value = (int64_t)ctx->prefetches;
  else if (k == "prefetch_hits")
    *value = (int64_t)ctx->prefetch_hits;
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

int curvis_selftest_math3(curvis_ctx *ctx, int op, const double *a, const double *b, const double *c, double *out, size_t n) {
  if (!ctx || !a || !out) return CURVIS_E_INVALID;
  if (n == 0) return CURVIS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const double *in[3] = {a, b, c};
  double *dev[4] = {nullptr, nullptr, nullptr, nullptr};
  int rc = CURVIS_OK;
  for (int k = 0; k < 4 && rc == CURVIS_OK; ++k) {
    if (k < 3 && !in[k]) continue;
    if (hipMalloc((void **)&dev[k], n * sizeof(double)) != hipSuccess ||
        (k < 3 && hipMemcpy(dev[k], in[k], n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess))
      rc = fail(ctx, CURVIS_E_HIP, "curvis_selftest_math3: device buffer");
  }
  if (rc == CURVIS_OK) {
    hipLaunchKernelGGL(selftest_math3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, dev[0], dev[1],
                       dev[2], dev[3], n);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(out, dev[3], n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(ctx, CURVIS_E_HIP, "curvis_selftest_math3: launch");
  }
  for (double *d : dev)
    if (d) (void)hipFree(d);
  return rc;
}

int curvis_selftest_fast_step(curvis_ctx *ctx, const curvis_metric *metric, double delta, double max_radius, const double *states,
                              size_t n, double *out) {
  if (!ctx || !metric || !states || !out) return CURVIS_E_INVALID;
  if (curvis_metric_validate(metric) != CURVIS_OK) return fail(ctx, CURVIS_E_INVALID, "curvis_selftest_fast_step: invalid metric");
  if (n == 0) return CURVIS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const cvk::MetricParams MP = make_metric(*metric);
  double *din = nullptr, *dout = nullptr;
  int rc = CURVIS_OK;
  if (hipMalloc((void **)&din, n * 5 * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&dout, n * CURVIS_FAST_STEP_RECORD * sizeof(double)) != hipSuccess ||
      hipMemcpy(din, states, n * 5 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(ctx, CURVIS_E_HIP, "curvis_selftest_fast_step: device buffer");
  if (rc == CURVIS_OK) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (metric->kind) {
      case CURVIS_METRIC_ELLIS:
        hipLaunchKernelGGL((selftest_fast_step_kernel<cvk::METRIC_ELLIS>), grid, block, 0, ctx->stream, MP, delta, max_radius, din, n, dout);
        break;
      case CURVIS_METRIC_INTERSTELLAR:
        hipLaunchKernelGGL((selftest_fast_step_kernel<cvk::METRIC_INTERSTELLAR>), grid, block, 0, ctx->stream, MP, delta, max_radius, din, n, dout);
        break;
      default:
        hipLaunchKernelGGL((selftest_fast_step_kernel<cvk::METRIC_FLAT>), grid, block, 0, ctx->stream, MP, delta, max_radius, din, n, dout);
        break;
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(out, dout, n * CURVIS_FAST_STEP_RECORD * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(ctx, CURVIS_E_HIP, "curvis_selftest_fast_step: launch");
  }
  if (din) (void)hipFree(din);
  if (dout) (void)hipFree(dout);
  return rc;
}

} /* extern "C" */
