/* curvis_hip.h -- C ABI of libcurvis_hip.so: the MI355X (gfx950) replacement for the
 * per-pixel geodesic hot path of fragarriss/CurVis.
 *
 * The reference has no FFI/plugin seam; the seam this library replaces is the method pair its
 * orchestration calls (src/rendering.rs:97 and :299):
 *
 *   RelativisticSystem<M>::render_image(&self, max_iterations: u32, max_radius: f64, delta: f64)
 *        -> image::DynamicImage                                          src/systems.rs:307-330
 *   RelativisticSystem<M>::render_image_efficient(&self, max_iterations_propagation, max_radius, delta,
 *        alpha_nums, max_iterations_sampling, thr1, thr2) -> DynamicImage  src/systems.rs:333-527
 *
 * with self = { metric: M, background_positive, background_negative: SphericalImage, camera: Camera }
 * (src/systems.rs:68-73).  A context (`curvis_ctx`) plays the role of `self`: it owns one GPU, the
 * two sky textures resident in HBM, and the device framebuffer.  A Rust host binds these with a
 * plain `extern "C"` block (INTEGRATION.md shows the stub); the C++ host in curvis_amd/csrc/host and
 * the Python mirror in curvis_amd/ bind the same symbols.
 *
 * Conventions: every function returns CURVIS_OK (0) or a negative CURVIS_E_* code;
 * curvis_last_error() gives the message.  The reference's panics map to error codes.  A context is
 * not thread-safe; distinct contexts (one per GPU) are independent.  No CPU fallback exists: without
 * a gfx950 device curvis_ctx_create fails with CURVIS_E_NO_DEVICE.
 */
#ifndef CURVIS_HIP_H
#define CURVIS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CURVIS_ABI_VERSION 1

enum {
  CURVIS_OK = 0,
  CURVIS_E_INVALID = -1,        /* bad argument (null pointer, zero resolution, ...) */
  CURVIS_E_NO_DEVICE = -2,      /* no usable gfx950 GPU: the product path has no CPU fallback */
  CURVIS_E_HIP = -3,            /* a HIP runtime call failed */
  CURVIS_E_CAMERA_OUTSIDE = -4, /* |l_camera| > max_radius: panic at src/systems.rs:122-124 */
  CURVIS_E_NO_SKY = -5,         /* a background image has not been set */
  CURVIS_E_PARALLEL = -6,       /* forward/up parallel: panic at src/algebra.rs:19-21 */
  CURVIS_E_METRIC = -7,         /* invalid metric parameters: panics at src/metrics.rs:409-456 */
  CURVIS_E_RCCL = -8,           /* RCCL call failed */
  CURVIS_E_SAMPLING = -9,       /* sampler panic (< 3 finite points): src/sampling.rs:155-157 */
  CURVIS_E_IO = -10
};

/* enum Metric (src/metrics.rs:575-578) + FlatSphericalMetric (src/metrics.rs:492-505) */
enum { CURVIS_METRIC_ELLIS = 0, CURVIS_METRIC_INTERSTELLAR = 1, CURVIS_METRIC_FLAT = 2 };

/* EllisMetric { rho } (src/metrics.rs:399-401), InterstellarMetric { m, a, rho } (:431-435). */
typedef struct curvis_metric {
  int32_t kind;
  int32_t _pad;
  double rho, m, a;
} curvis_metric;

/* Camera (src/cameras.rs:16-34) reduced to what the kernels read: position (t,l,theta,phi),
 * camera_to_world_rotation_matrix (row-major), focal_length, sensor_width/height, resolution. */
typedef struct curvis_camera {
  double pos[4];
  double rot[9];
  double focal, sensor_w, sensor_h;
  uint32_t res_x, res_y;
} curvis_camera;

/* PhotonEscape (src/systems.rs:39-44) */
enum { CURVIS_NOT_ESCAPED = 0, CURVIS_POSITIVE_SPACE = 1, CURVIS_NEGATIVE_SPACE = -1 };

/* per-ray final state, for parity tests: RelativisticObject (src/vectors.rs:135-139) after
 * escape_photon, number of Euler steps executed, escape code, raw texel indices
 * (`as u32` results of src/images.rs:118-119, before clamping). */
typedef struct curvis_ray_debug {
  double x[4];
  double p[4];
  uint32_t steps;
  int32_t code;
  uint32_t tx, ty;
} curvis_ray_debug;

typedef struct curvis_stats {
  uint64_t rays;      /* rays traced */
  uint64_t steps;     /* Euler steps executed (sum over rays) */
  uint64_t n_pos;     /* escaped to +l */
  uint64_t n_neg;     /* escaped to -l */
  uint64_t n_none;    /* hit the iteration cap */
  uint64_t n_oob;     /* texel index == W or == H (reference would panic); clamped */
  double kernel_ms;   /* HIP-event time of the kernels on the context's stream (integrate + shade) */
  double total_ms;    /* wall time of the call including H2D/D2H */
  double integrate_ms; /* HIP-event time of the geodesic integration kernel(s) alone */
  double shade_ms;     /* HIP-event time of the shading (direction + sky lookup) kernel(s) */
} curvis_stats;

typedef struct curvis_ctx curvis_ctx;

const char *curvis_version(void);
/* message of the last error on this context (or of the last failed curvis_ctx_create if ctx == NULL) */
const char *curvis_last_error(const curvis_ctx *ctx);
int curvis_device_count(void);

int curvis_ctx_create(int device, curvis_ctx **out);
void curvis_ctx_destroy(curvis_ctx *ctx);
/* name of the device + number of CUs, for bench reports */
int curvis_ctx_device_info(const curvis_ctx *ctx, char *name, size_t name_cap, int *compute_units, int *clock_mhz);
/* which physical GPU this context sits on and what it is doing right now, for the per-device tables of multi-GPU runs
 * (bench.py `per_rank`, `curvis video --stats`): PCI address "dddd:bb:dd.f" (hipDeviceGetPCIBusId), the current
 * shader clock in MHz and the board power in W as the amdgpu driver reports them in sysfs (pp_dpm_sclk,
 * hwmon power1_average / power1_input); a value that cannot be read comes back as -1, never as an error.  No
 * reference counterpart (the reference is single-threaded CPU code). */
int curvis_ctx_device_status(const curvis_ctx *ctx, char *pci_bus_id, size_t cap, int *sclk_mhz, int *power_w);

/* SphericalImage (src/images.rs:51-56): which = 0 -> background_positive (+l), 1 -> background_negative.
 * `rgba` is the decoded image as Rgba8 (what DynamicImage::get_pixel returns, src/images.rs:107-111),
 * row-major, host memory; it is copied to HBM and kept for all frames. */
int curvis_ctx_set_sky(curvis_ctx *ctx, int which, const uint8_t *rgba, uint32_t w, uint32_t h);
/* same, from a device pointer on this context's GPU (e.g. a buffer filled by an RCCL broadcast);
 * copy != 0 copies it, copy == 0 borrows it (caller keeps it alive). */
int curvis_ctx_set_sky_device(curvis_ctx *ctx, int which, const void *dev_rgba, uint32_t w, uint32_t h, int copy);
/* SphericalImage::set_forward_up (src/images.rs:102-104); default forward = x, up = z. */
int curvis_ctx_set_sky_orientation(curvis_ctx *ctx, int which, const double forward[3], const double up[3]);
/* Broadcast both sky textures from rank `root` over an existing RCCL communicator (ncclComm_t):
 * the shapes (4 x u32) first, then two ncclBroadcast calls of w*h*4 bytes each over xGMI.
 * Non-root ranks need no prior set_sky: their textures are allocated from the broadcast shapes. */
int curvis_ctx_bcast_skies(curvis_ctx *ctx, void *nccl_comm, int root);

/* RCCL plumbing for a host that has no RCCL binding of its own (one process per GPU -- a Rust host, `bench.py`):
 * rank 0 draws an id (ncclGetUniqueId) and ships its CURVIS_RCCL_ID_BYTES bytes to the other processes by whatever
 * out-of-band channel it has; every process then joins with its own context (ncclCommInitRank on the context's
 * device) and passes the returned communicator to curvis_ctx_bcast_skies.  The communicator is the caller's:
 * curvis_rccl_comm_destroy it after the broadcast.  A one-process / N-thread host (`curvis video --devices N`) needs
 * none of this: it calls ncclCommInitAll itself.  Single node by design (frames shard over the GPUs of ONE node): unless
 * NCCL_SOCKET_IFNAME is set these two calls set it to "lo", so RCCL's bootstrap does not wait on an unroutable interface.
 * No reference counterpart (single-threaded CPU code). */
#define CURVIS_RCCL_ID_BYTES 128
int curvis_rccl_unique_id(uint8_t id[CURVIS_RCCL_ID_BYTES]);
int curvis_ctx_rccl_comm_init(curvis_ctx *ctx, const uint8_t id[CURVIS_RCCL_ID_BYTES], int n_ranks, int rank,
                              void **comm_out);
int curvis_rccl_comm_destroy(void *nccl_comm);

/* Read back `bytes` bytes at byte offset `offset` of sky texture `which` from HBM (e.g. to verify on every rank that
 * a broadcast texture equals the root's file). */
int curvis_ctx_read_sky(curvis_ctx *ctx, int which, size_t offset, size_t bytes, uint8_t *out);

/* how two devices of this node are connected (hipExtGetLinkTypeAndHopCount, hipDeviceCanAccessPeer, hipDeviceGetP2PAttribute):
 * link_type = HSA_AMD_LINK_INFO_TYPE_* (2 PCIe, 4 xGMI; 0 with hops 0 for a == b; -1 unknown).  Read beside the measured
 * sky-broadcast rate: xGMI is point-to-point, 7 links x ~153 GB/s per MI355X.  Output pointers may be NULL.  The errors of
 * curvis_ctx_bcast_skies / curvis_ctx_rccl_comm_init name the stage that failed ("sky broadcast, stage header_broadcast: ...")
 * and carry RCCL's own last error. */
int curvis_device_link(int device_a, int device_b, int *link_type, int *hops, int *peer_access, int *performance_rank,
                       int *native_atomics);

/* Camera::new (src/cameras.rs:79-122) incl. Orientation::new (src/algebra.rs:16-38). */
int curvis_camera_init(curvis_camera *out, const double pos[4], const double forward[3], const double up[3],
                       double focal_length, double sensor_diagonal, uint32_t res_x, uint32_t res_y);
/* Orientation::new: rotation matrix, its inverse and the orthogonalised up (any may be NULL). */
int curvis_orientation_init(const double forward[3], const double up[3], double rot[9], double inv_rot[9],
                            double up_out[3]);
/* EllisMetric::new / InterstellarMetric::new parameter checks (src/metrics.rs:407-459). */
int curvis_metric_validate(const curvis_metric *m);
/* The three required methods of trait DiagonalSphericalMetric (src/metrics.rs:40-48: r, r_squared, r_derivative;
 * Ellis :417-421, Interstellar :467-485, flat :501-505) at radial coordinate l, evaluated on the host with the same
 * arithmetic (cv_math.h) the kernels use -- what every ray of a render is integrated with.  Any output may be NULL. */
int curvis_metric_functions(const curvis_metric *m, double l, double *r, double *r_squared, double *r_derivative);
/* The diagonal of the metric tensor at `position` = (t, l, theta, phi): covariant g_ii = (-1, 1, r^2(l),
 * r^2(l) sin^2(theta)) (src/metrics.rs:49-68; sin().powi(2) is s * s) and contravariant g^ii = g_ii.powi(-1) = 1 / g_ii
 * (:84-93) -- what to_covariant / to_contravariant (:163-219) multiply a vector's components by.  Host-side, same
 * arithmetic as the kernels; either output may be NULL. */
int curvis_metric_tensor(const curvis_metric *m, const double position[4], double g_cov[4], double g_contr[4]);

/* RelativisticSystem::render_image (src/systems.rs:307-330): one ray per pixel, forward-Euler
 * integration until |l| > max_radius or max_iterations steps, nearest-texel sky lookup.
 * rgb_out: host buffer res_y*res_x*3 (row-major, RGB8) or NULL to leave the frame in HBM
 * (see curvis_ctx_framebuffer / curvis_ctx_download). */
int curvis_render_brute(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                        uint32_t max_iterations, double max_radius, double delta, uint8_t *rgb_out,
                        curvis_stats *stats);
/* Host-side accessors of the per-ray functions either end of the Euler loop, in the kernels' arithmetic (the same
 * host/device source): what the kernels do for every pixel, callable for one.
 * Camera::outward_vector_on_camera_space (src/cameras.rs:150-164; unit vector, x forward / y left / z up) and
 * outward_vector_on_world_space_from_x_y (:169-172, the former rotated by camera_to_world); either output may be NULL. */
int curvis_camera_outward_vector(const curvis_camera *camera, uint32_t px, uint32_t py, double camera_space[3],
                                 double world_space[3]);
/* DiagonalSphericalMetric::relativistic_vector_to_direction (src/metrics.rs:339-349 with to_contravariant :190-203):
 * covariant momentum at `position` -> tangent-space direction (not normalised; z uses frame_field_22, as the
 * reference does). */
int curvis_vector_to_direction(const curvis_metric *metric, const double position[4], const double p_cov[4],
                               double direction[3]);
/* DiagonalSphericalMetric::update_relativistic_object (src/metrics.rs:283-297) for a covariant momentum: ONE forward-Euler
 * step of (x, p_cov) in place, on the host, with the IEEE form of the step the kernels fall back to (the fast step
 * returns the same bits) -- the body of curvis_photon_trajectories' loop, all eight components. */
int curvis_update_relativistic_object(const curvis_metric *metric, double x[4], double p_cov[4], double delta);
/* SphericalImage::get_pixel_from_vector3's texel (src/images.rs:115-142, 171-174; src/algebra.rs:106-134) for an image of
 * w x h texels whose inverse orientation is inv_rot (NULL = the default forward x / up z): raw `as u32` indices.
 * Returns CURVIS_OK, or CURVIS_E_INVALID with the indices still set when x == w or y == h (the reference's
 * get_pixel panics there; the kernels clamp and count such rays, curvis_stats.n_oob). */
int curvis_sky_texel_index(uint32_t w, uint32_t h, const double inv_rot[9], const double v[3], uint32_t *x, uint32_t *y);

/* A band of image rows [row_begin, row_begin + row_count) of the same frame: rays are independent
 * (src/systems.rs:316-326), so a single image can be split across GPUs by rows and assembled on the host
 * (SURVEY 8e).  rgb_out: row_count*res_x*3 or NULL; stats cover the band. */
int curvis_render_brute_rows(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                             uint32_t row_begin, uint32_t row_count, uint32_t max_iterations, double max_radius,
                             double delta, uint8_t *rgb_out, curvis_stats *stats);
/* same as curvis_render_brute, plus the final state of every ray (dbg_out: res_y*res_x entries, row-major). */
int curvis_render_brute_debug(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                              uint32_t max_iterations, double max_radius, double delta, uint8_t *rgb_out,
                              curvis_ray_debug *dbg_out, curvis_stats *stats);
/* n_frames cameras of identical resolution rendered by ONE launch (video shards):
 * rgb_out is n_frames*res_y*res_x*3 or NULL. */
int curvis_render_brute_batch(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cameras,
                              uint32_t n_frames, uint32_t max_iterations, double max_radius, double delta,
                              uint8_t *rgb_out, curvis_stats *stats);

/* RelativisticSystem::render_image_efficient (src/systems.rs:333-527) -- what `curvis image` and
 * `curvis video` call (src/rendering.rs:97-106, :299-307): adaptive 1-D sampling of the escape angle over
 * alpha in [-0.1 pi, 1.1 pi] on the equatorial plane (src/sampling.rs), linear interpolation per pixel
 * (interp 1.0.3), axis-angle rotation of the camera direction, nearest-texel lookup.  Argument names and
 * order are the reference's.  Reproduces the reference's behaviour including the shrinking sample domain
 * (src/sampling.rs:161), extrapolation beyond the last sample and black +/- transitions. */
int curvis_render_efficient(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera,
                            uint32_t max_iterations_propagation, double max_radius, double delta, uint32_t alpha_nums,
                            uint32_t max_iterations_sampling, double sampling_convergence_threshold_1,
                            double sampling_convergence_threshold_2, uint8_t *rgb_out, curvis_stats *stats);
/* n_frames cameras: the per-frame samplers advance in lock step, one kernel launch per refinement round
 * for the whole batch, one per-pixel launch for all frames. */
int curvis_render_efficient_batch(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cameras,
                                  uint32_t n_frames, uint32_t max_iterations_propagation, double max_radius,
                                  double delta, uint32_t alpha_nums, uint32_t max_iterations_sampling,
                                  double sampling_convergence_threshold_1, double sampling_convergence_threshold_2,
                                  uint8_t *rgb_out, curvis_stats *stats);
/* The sampler of a FUTURE curvis_render_efficient_batch call, launched NOW on a stream of its own (device-resident sampler:
 * option "device_sampler").  Returns at once.  The render call with the same metric, settings and camera radii (the l of every
 * frame, in order) then finds its sample tables ready -- it waits for the sampler's event on its own stream instead of sampling --
 * so that a caller rendering batch after batch hides the sampler's latency (a handful of ~2000-step Euler chains on a few
 * compute units) under the previous batch's per-pixel kernel, PNG front end and host work:
 *     prefetch(batch 0); for k: { prefetch(batch k + 1); render(batch k); deflate / download(batch k); }
 * A prefetch nobody consumes costs its kernel and is overwritten by the second prefetch after it (two slots).  Settings the
 * device sampler does not take (alpha_nums beyond its arrays, a camera beyond max_radius) make this a no-op: the render call
 * deals with them.  Identical results with and without; read-only options "prefetches", "prefetch_hits",
 * "last_sampling_prefetched".  Reference seam: none -- the reference samples inside render_image_efficient
 * (src/systems.rs:437-486); this only moves WHEN the same sampling runs. */
int curvis_ctx_prefetch_efficient(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *cameras, uint32_t n_frames,
                                  uint32_t max_iterations_propagation, double max_radius, double delta, uint32_t alpha_nums,
                                  uint32_t max_iterations_sampling, double sampling_convergence_threshold_1,
                                  double sampling_convergence_threshold_2);

/* "direct" mode -- NOT a function of the reference (SURVEY.md 8f N1 names it as an option): the image that
 * render_image_efficient approximates by adaptive sampling + linear interpolation, computed without either:
 * compute_escape_angle (src/systems.rs:203-261) is evaluated for the alpha of EVERY pixel (:405-433) and step 5
 * (:498-523) applied to its result.  A not-escaped photon, or one whose tangent rotation is undefined, gives a black
 * pixel (counted in n_none).  Parity: bit-exact against this repository's oracle (cvo_render_image_direct); against the
 * reference only through what it approximates (its efficient image differs where the interpolation does). */
int curvis_render_direct(curvis_ctx *ctx, const curvis_metric *metric, const curvis_camera *camera, uint32_t max_iterations,
                         double max_radius, double delta, uint8_t *rgb_out, curvis_stats *stats);
/* sampler bookkeeping of the last efficient render (per frame): final table size, refinement rounds,
 * integrator calls, Euler steps, and whether the "maximum number of iterations" warning fired. */
typedef struct curvis_sampling_info {
  uint32_t n_samples, rounds;
  uint64_t calls, steps;
  int32_t warned_max_iterations;
  int32_t _pad;
} curvis_sampling_info;
int curvis_ctx_sampling_info(const curvis_ctx *ctx, uint32_t frame, curvis_sampling_info *info);
/* Statistics of frame `frame` of the last render call (brute, rows, batch, efficient, efficient batch): the
 * per-frame loop of VideoRenderingSystem::render (src/rendering.rs:291-316) is ONE launch per batch here, so the
 * kernels keep one set of counters per frame.  rays / steps / n_pos / n_neg / n_none / n_oob are exact for that
 * frame (for the efficient renderer: rays = pixels, steps = Euler steps of the frame's sampler, n_* = pixels by
 * escape space); the *_ms fields are the frame's share of the launch time (by executed steps), not a separate
 * measurement.  "last_frames" (curvis_ctx_get_option) = number of frames available. */
int curvis_ctx_frame_stats(const curvis_ctx *ctx, uint32_t frame, curvis_stats *stats);
/* the (alpha, escape angle, escape space) table of a frame of the last efficient render; cap >= n_samples */
int curvis_ctx_samples(const curvis_ctx *ctx, uint32_t frame, double *alpha, double *escape_angle,
                       double *escape_space, size_t cap);

/* compute_escape_angles_range / compute_escape_angle (src/systems.rs:203-281, re-exported by src/lib.rs:37):
 * photons at (0, l, pi/2, 0) with tangent direction (cos a, 0, sin a); angle[i] = escape angle in [0, 2 pi)
 * (NaN when not escaped), space[i] = +1 / -1 / 0 (EscapeAngle::{PositiveSpace, NegativeSpace, NotEscaped}). */
int curvis_compute_escape_angles(curvis_ctx *ctx, const curvis_metric *metric, double l, const double *alphas,
                                 uint32_t n, double delta, uint32_t max_iterations, double max_radius,
                                 double *angle, int32_t *space, uint32_t *steps /* nullable */);
/* DiagonalSphericalMetric::new_photon (src/metrics.rs:301-334), host-side: position (t,l,theta,phi) and a
 * tangent-space direction -> contravariant position x and covariant momentum p. */
int curvis_new_photon(const curvis_metric *metric, const double position[4], const double direction[3], double x[4],
                      double p_cov[4]);
/* compute_photon_trajectory (src/systems.rs:77-92, re-exported by src/lib.rs:37) for n photons:
 * out[photon][iteration][0..7] = (x_t, x_l, x_theta, x_phi, p_t, p_l, p_theta, p_phi) BEFORE that iteration's
 * Euler step; momentum covariant. */
int curvis_photon_trajectories(curvis_ctx *ctx, const curvis_metric *metric, uint32_t n_photons, const double *x0,
                               const double *p0_cov, uint32_t iterations, double delta, double *out);

/* images::load_image / save_image (src/images.rs:7-20; image::open / DynamicImage::save of image 0.25.2) for the
 * two file formats this library decodes itself: PNG (every colour type and bit depth, converted to Rgba8 the way
 * DynamicImage::get_pixel does: grey -> (v, v, v, 255), 16 bit -> (v + 128) / 257, palette and tRNS expanded) and
 * JPEG (8-bit Huffman baseline / progressive; see csrc/host/jpeg_io.h on why JPEG is outside the pixel-parity
 * claims); the format is taken from the file's signature.  *rgba_out is w*h*4 bytes owned by the library until
 * curvis_image_free.  Errors: CURVIS_E_IO, message through curvis_last_error(NULL).  Host-only, no GPU needed. */
int curvis_image_load(const char *path, uint8_t **rgba_out, uint32_t *w, uint32_t *h);
void curvis_image_free(uint8_t *rgba);
/* DynamicImage::ImageRgb8(..).save(path) as PNG (8-bit RGB, non-interlaced) */
int curvis_image_save_rgb8(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h);
/* same with the encoder chosen: level -1 = the library's fast PNG writer (filter Up + one dynamic-Huffman block whose
 * only matches are zero runs: what `curvis video` writes its frames with; the reference's image crate also saves
 * with its fast setting), 0..9 = zlib at that level.  Same decoded pixels whatever the level. */
int curvis_image_save_rgb8_level(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h, int level);

/* PNG front end ON THE DEVICE, for hosts that save every frame (src/rendering.rs:110, :311): the `n_frames` RGB8 frames of
 * res_x x res_y pixels that the last render call left in the context's framebuffer (call it with rgb_out = NULL: the pixels
 * then never cross PCIe) are filtered (type 2, Up), Huffman-coded (one dynamic block per frame, distance-1 matches for zero
 * runs) and check-summed (Adler-32) by HIP kernels; what comes back is one finished zlib stream per frame, back to back in
 * `zlib_out` (host memory, preferably from curvis_host_alloc), frame f at [offsets[f], offsets[f + 1]) -- `offsets` has
 * n_frames + 1 entries.  curvis_image_save_zlib_rgb8 wraps such a stream into a PNG file (signature, IHDR, IDAT + CRC-32,
 * IEND: ~0.1 ms of a host thread per MB of stream instead of 4-7 ms per 1080p frame for filtering and coding on the
 * host).  CURVIS_E_INVALID with "output buffer too small" when out_cap does not suffice (worst case: 1.5 x the raw
 * frames + 200 bytes each; typical frames compress 10-100 x).  kernel_ms (may be NULL): HIP-event time of the launches. */
int curvis_ctx_deflate_frames(curvis_ctx *ctx, uint32_t res_x, uint32_t res_y, uint32_t n_frames, uint8_t *zlib_out, size_t out_cap,
                              size_t *offsets, double *kernel_ms);
int curvis_image_save_zlib_rgb8(const char *path, const uint8_t *zlib_stream, size_t len, uint32_t w, uint32_t h);
/* the same with the PNG chunk's CRC-32 from the device: idat_crc[f] = CRC-32 of "IDAT" + frame f's stream (Adler-32 trailer
 * included), *crc_valid = 1 on success (always, since round 6: one device path for every frame width; the flag stays in the
 * signature -- until round 5 frames whose rows were no multiple of 64 bytes left it 0 and idat_crc untouched).  curvis_image_save_zlib_rgb8_crc then writes the file without reading the stream again: what is left
 * to a writer thread is the write itself. */
int curvis_ctx_deflate_frames_crc(curvis_ctx *ctx, uint32_t res_x, uint32_t res_y, uint32_t n_frames, uint8_t *zlib_out, size_t out_cap,
                                  size_t *offsets, double *kernel_ms, uint32_t *idat_crc, int *crc_valid);
int curvis_image_save_zlib_rgb8_crc(const char *path, const uint8_t *zlib_stream, size_t len, uint32_t w, uint32_t h, uint32_t idat_crc);

/* Page-locked host memory for the `rgb_out` buffers of the render calls: into such a buffer the device-to-host copy of
 * a frame is ONE DMA transfer (~25 GB/s over PCIe 5), into ordinary pageable memory the runtime stages it through
 * bounce buffers (~2 GB/s measured: 12 ms of a 54 ms 4K frame).  `curvis video` keeps a small pool of these and hands
 * them to its PNG writer threads without another copy.  Free with curvis_host_free; no reference counterpart. */
int curvis_host_alloc(size_t bytes, void **out);
void curvis_host_free(void *p);

/* device framebuffer of the last render (RGB8, frames back to back) */
int curvis_ctx_framebuffer(curvis_ctx *ctx, void **dev_ptr, size_t *bytes);
int curvis_ctx_download(curvis_ctx *ctx, uint8_t *rgb_out, size_t bytes);
/* RGB8 frames from host memory into the context's framebuffer (frames back to back; it grows as needed): what
 * curvis_ctx_deflate_frames then compresses */
int curvis_ctx_upload(curvis_ctx *ctx, const uint8_t *rgb, size_t bytes);
int curvis_ctx_synchronize(curvis_ctx *ctx);
/* Overlapped stream download (option "async_streams" = 1, default 0): curvis_ctx_deflate_frames[_crc] returns while the frames'
 * zlib streams are still travelling to `zlib_out` on the context's copy stream.  Offsets, the Adler-32 trailers (written by the
 * host behind every stream: they need the sums, not the bytes) and the chunk CRCs are final on return; the stream bytes are
 * there after curvis_ctx_download_wait -- call it before anything reads `zlib_out`, e.g. after the NEXT render call has
 * returned, which is what the copy hides under (`curvis video`: one context 8 600 -> 10 600 frames/s).  The next deflate call
 * waits by itself before it reuses the scratch the streams are read from; so does switching the option off and
 * curvis_ctx_destroy.  `zlib_out` must stay valid until then (page-locked memory from curvis_host_alloc for a real overlap).
 * Read-only option "streams_pending". */

/* Overlapped download (option "async_download" = 1, default 0).  The reference's render_image returns an owned host
 * image (src/systems.rs:314-329), so a host that renders frame after frame pays the PCIe copy behind every kernel
 * (+0.25 ms on a 10.2 ms 1080p frame).  With the option set, a render call given `rgb_out` (brute, rows, batch, efficient,
 * direct) returns as soon as its kernels have finished and its statistics are valid, with the copy into `rgb_out` QUEUED
 * on a copy stream of the context; the next call renders into a second frame buffer while the copy engine drains the
 * first.  A pipeline one frame deep: `rgb_out` of call k is complete when call k + 1 with an `rgb_out` on the same
 * context returns, or when curvis_ctx_download_wait returns -- not before.  `rgb_out` should come from
 * curvis_host_alloc (into pageable memory the runtime's copy is not asynchronous; still correct).  Everything that reads
 * "the frames of the last render" (curvis_ctx_deflate_frames, curvis_ctx_download, curvis_ctx_framebuffer) keeps seeing them;
 * the device pointer curvis_ctx_framebuffer returns alternates between two buffers from call to call.  Setting the
 * option back to 0 waits for the download in flight, and so does curvis_ctx_destroy.  get_option: "downloads_overlapped"
 * (so far), "download_pending" (0 / 1). */
int curvis_ctx_download_wait(curvis_ctx *ctx);

/* tuning knobs (not part of the reference surface): "variant" (-1 = automatic, the default: the static
 * one-ray-per-thread kernel, and for launches of up to "relay_max_frames" (default 8) frames with at least "relay_min_blocks" workgroups
 * -- default 4 per CU, i.e. from about 700x400 -- the relay kernel; 0 = persistent lane-refill kernel;
 * 1 = static kernel always; 2 = relay kernel = the static kernel with end-game hand-over of unfinished tiles
 * between waves, still subject to "relay_min_blocks"; "relay_segment" = steps between hand-over points,
 * 0 = automatic; all variants give identical results), "block_threads" (workgroup size of the static / relay
 * kernels: 64, 128 or 256; 0 = automatic = 256), "refill_threshold",
 * "blocks_per_cu", "fast_math"
 * (1 = shared-reciprocal Euler step, 0 = compiler IEEE division/sqrt; identical results), "fuse_shade"
 * (1 = the static kernel shades in its epilogue, 0 = final states staged in HBM + separate shade kernel),
 * "max_store_bytes" (ray-store budget that bounds the frames per launch of a batch),
 * "sampling_speculation" (efficient renderer: depth of the speculative dyadic subtree evaluated below every
 * refined interval; 0 = one launch per refinement round; default -1 = automatic, 10 for one or two frames, 6 for three to five and 4
 * for larger batches) and "sampling_speculation_first" (the same below the intervals of the initial uniform grid,
 * i.e. for the first launch; default -1 = automatic, 8 / 4 / 3; depths up to 11), "device_sampler" (efficient renderer: 1 = the
 * reference's whole adaptive sampler -- rounds, speculation and all -- runs on the device, ONE launch per call and a workgroup
 * per distinct camera radius, no host in the refinement loop; 0 = the host-paced sampler, several launches per call; default -1 =
 * automatic: the device from "device_sampler_min_frames" (default 48) frames per call on; identical sample tables and pixels either
 * way; a table that outgrows the kernel's fixed arrays -- 1536 samples -- sends the call to the host-paced sampler;
 * "sampling_speculation" = 0 switches speculation off on the device too); read-only after an efficient render:
 * "last_sampling_launches", "last_sampling_evaluated", "last_sampler_path" (0 host-paced, 1 device, 2 device fell back to the
 * host), "last_sampling_chains" (device: Euler chains the slowest job waited for); after any render: "last_frames"; after a relay render: "last_relay_launches",
 * "last_relay_parks".  Relay safety net: if a relay launch reports waves that gave up waiting (the kernel leans
 * on in-order workgroup dispatch, which HIP does not promise), the frame is rendered again by the static kernel and
 * "relay_disabled" becomes 1 for the context ("relay_fallbacks" counts such renders); "relay_verify" = 1 (debug)
 * repeats every relay render with the static kernel and fails with CURVIS_E_HIP if frames or counters differ. */
int curvis_ctx_set_option(curvis_ctx *ctx, const char *key, int64_t value);
int curvis_ctx_get_option(const curvis_ctx *ctx, const char *key, int64_t *value);

/* self-test hooks used by tests/ (device vs host bit-equality of cv_math.h and of IEEE div/sqrt):
 * op: 0 sin, 1 cos, 2 atan, 3 acos, 4 log, 5 atan2(a,b), 6 a/b, 7 sqrt(a), 8 fma(a,b,a),
 * 9 raw v_rcp_f64(a), 10 raw v_rsq_f64(a) (hardware seeds, for accuracy measurements) */
int curvis_selftest_math(curvis_ctx *ctx, int op, const double *a, const double *b, double *out, size_t n);

/* the primitives of the shared-reciprocal Euler step, element-wise on three inputs (b, c may be NULL where unused):
 * op 0 div_with_recip(n = a, d = b, y = c): the quotient the fast step forms from an approximate reciprocal y
 *    1 sqrt_and_rsqrt(a): the square root    2 sqrt_and_rsqrt(a): its by-product y ~ 1/sqrt(a)
 *    3 the square root's final residual step for given (x, g, y) = (a, b, c)
 *    4 recip_refined(a)    5 cv_div_nr(a, b) (the -1/x of atan)    6 recip_newton(d = a, y = b)
 *    7 / 8 / 9 component 0 / 1 / 2 of v / |v| for v = (a, b, c) through the efficient pixel kernel's shared reciprocal (unit3)
 *    10 a / b as the pixel kernel divides an index by a constant (div_index, y = recip_chain(b))    11 ... an angle (div_angle)
 *    12 sqrt(a) through the pixel kernel's sqrt_plain (the compiler's chain without its range wrappers)
 *    13 Rust's `a as u32` (saturating, NaN -> 0) as the sky lookup converts its texel coordinates
 * -- the directed hard cases of tests/test_gpu_fast_step.py go through here. */
int curvis_selftest_math3(curvis_ctx *ctx, int op, const double *a, const double *b, const double *c, double *out, size_t n);

/* ONE Euler step (src/metrics.rs:283-297) of the fast kernel per input state, every quotient recorded.
 * states: n x {l, theta, p_l, p_theta, p_phi}; out: n x CURVIS_FAST_STEP_RECORD doubles =
 *   6 x {numerator n, denominator d, shared reciprocal y used, the step's quotient, the IEEE quotient, the remainder
 *        n - d RN(n y), 1 - d y}
 *       (k = 0 r' = l/r [Ellis only], 1 1/r^2, 2 p_phi^2/sin^2, 3 dp_l, 4 cos/(r^2 sin^3), 5 1/(r^2 sin^2); NaN = not formed),
 *   5 new state of the fast step (l, theta, phi - phi0, p_l, p_theta), 5 of the strict step, 1 flag (1 = fast path taken). */
#define CURVIS_FAST_STEP_RECORD 53
int curvis_selftest_fast_step(curvis_ctx *ctx, const curvis_metric *metric, double delta, double max_radius, const double *states,
                              size_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif /* CURVIS_HIP_H */
