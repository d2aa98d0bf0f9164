"""Host orchestration of the video path (mirror of src/rendering.rs + src/interpolation.rs) with the
frames of a video sharded across the GPUs of a node.

Reference surface mirrored:
  Interpolator::{from_file, min_time, max_time, camera_position, camera_forward, camera_up}
                                                              src/interpolation.rs:45-112
  VideoRenderingSystem::times_of_frames                       src/rendering.rs:224-238
  VideoRenderingSystem::update_camera / render                src/rendering.rs:242-327
The reference renders frames one after the other on one thread.  Frames are independent
(src/rendering.rs:291-316: the only shared mutable is the camera, overwritten per frame), so here
frame k goes to rank k mod world_size; each rank renders its shard in batches of `batch` frames
per kernel launch; there is no data-path collective (the two skies are broadcast once, before).
"""
import numpy as np

from . import paths
from .systems import Camera


class InterpolatorPanic(RuntimeError):
    """a panic of the reference's Interpolator (message of the panic, or the index-out-of-bounds)."""


class Interpolator:
    def __init__(self, positions, forward_vectors, up_vectors):
        self.positions = np.asarray(positions, dtype=np.float64)
        self.forward_vectors = np.asarray(forward_vectors, dtype=np.float64)
        self.up_vectors = np.asarray(up_vectors, dtype=np.float64)

    @classmethod
    def from_file(cls, path_to_csv_file):
        return cls(*paths.load_path(path_to_csv_file))

    def min_time(self):
        return float(self.positions[0][0])

    def max_time(self):
        return float(self.positions[len(self.positions) - 1][0])

    def time_indexes_and_frac_from_time(self, t):
        """src/interpolation.rs:63-91, including its off-by-one: the loop leaves (t1, t2) on segment
        (i-1, i) but returns the indices (i, i+1)."""
        if t < self.min_time():
            raise InterpolatorPanic("Interpolation time cannot be smaller than first time in positions[0].")
        if t > self.max_time():
            raise InterpolatorPanic("Interpolation time cannot be greater than last time in positions[0].")
        t1, t2 = self.min_time(), self.max_time()
        i = 0
        while t > self.positions[i][0]:
            t1 = float(self.positions[i][0])
            t2 = float(self.positions[i + 1][0])
            i += 1
        frac = (t - t1) / (t2 - t1)
        return i, i + 1, frac

    def _interp(self, table, t):
        i1, i2, frac = self.time_indexes_and_frac_from_time(t)
        if i2 >= len(table):
            raise InterpolatorPanic("index out of bounds: the len is %d but the index is %d" % (len(table), i2))
        if not (0.0 <= frac <= 1.0):
            raise InterpolatorPanic("frac must be between 0 and 1")
        v1, v2 = table[i1], table[i2]
        return v1 + frac * (v2 - v1)

    def camera_position(self, t):
        return self._interp(self.positions, t)

    def camera_forward(self, t):
        return self._interp(self.forward_vectors, t)

    def camera_up(self, t):
        return self._interp(self.up_vectors, t)


def times_of_frames(min_time, max_time, frame_rate):
    """src/rendering.rs:224-238 (float accumulation, not k*dt)."""
    delta_time = 1.0 / frame_rate
    times = []
    t = min_time
    while t < max_time:
        times.append(t)
        t += delta_time
    return times


def frames_of_rank(n_frames, rank, world_size):
    """frame k -> rank k mod world_size (costs are near-uniform: same ray count per frame)."""
    return list(range(rank, n_frames, world_size))


def rows_of_rank(height, rank, world_size):
    """single image split by rows: rank r renders rows [begin, begin + count), bands differ by at
    most one row and cover [0, height) exactly; ranks beyond the height get an empty band."""
    base, extra = divmod(height, world_size)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def render_image_sharded(context, metric, camera, max_iterations, max_radius, delta, rank=0, world_size=1, dist=None):
    """RelativisticSystem::render_image with the rows of ONE frame split over the ranks; every rank renders its band
    on its own GPU, rank 0 returns the assembled H x W x 3 image (others None) plus the list of per-band stats."""
    H, W = camera.resolution_height, camera.resolution_width
    begin, count = rows_of_rank(H, rank, world_size)
    band, st = (context.render_brute_rows(metric, camera, begin, count, max_iterations, max_radius, delta)
                if count else (np.zeros((0, W, 3), np.uint8), None))
    info = {"rank": rank, "row_begin": begin, "rows": count, "steps": int(st.steps) if st else 0,
            "rays": int(st.rays) if st else 0}
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return band, [info]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (info, band))
    if rank != 0:
        return None, [p[0] for p in parts]
    parts.sort(key=lambda p: p[0]["row_begin"])
    return np.concatenate([p[1] for p in parts], axis=0), [p[0] for p in parts]


class ImageRenderingSettings:
    """ImageRenderingSettings (src/rendering.rs, fields as used by ImageRenderingSystem::new / render :33-117)."""

    def __init__(self, path_to_background_image_1, path_to_background_image_2, path_to_output_folder, output_image_name,
                 camera_position, camera_forward, camera_up, camera_focal_length=15.0, camera_diagonal=43.0,
                 resolution_x=960, resolution_y=540, escape_radius=100.0, max_iterations_propagation=40000,
                 ray_integration_step=0.05, alphas_num=100, max_iterations_sampling=100,
                 sampling_convergence_threshold_1=1e-5, sampling_convergence_threshold_2=1e-5):
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})


class ImageRenderingSystem:
    """ImageRenderingSystem<M> (src/rendering.rs:16-117): two backgrounds + Camera + RelativisticSystem; render()
    calls render_image_efficient with the settings' seven arguments and saves <folder>/<name>.png (an existing
    extension of the name is replaced, PathBuf::with_extension).  mode="brute" renders with the per-pixel
    integrator instead (not in the reference's ImageRenderingSystem)."""

    def __init__(self, metric, image_rendering_settings, context=None, mode="efficient"):
        from .images import load_image_as_spherical_image
        from .systems import RelativisticSystem
        st = self.image_rendering_settings = image_rendering_settings
        self.mode = mode
        image_1 = load_image_as_spherical_image(st.path_to_background_image_1)
        image_2 = load_image_as_spherical_image(st.path_to_background_image_2)
        camera = Camera(st.camera_position, st.camera_forward, st.camera_up, st.camera_focal_length, st.camera_diagonal,
                        st.resolution_x, st.resolution_y)
        self.relativistic_system = RelativisticSystem(metric, image_1, image_2, camera, context=context)

    def render(self):
        import os
        from .images import save_image
        st = self.image_rendering_settings
        folder = str(st.path_to_output_folder)
        if not os.path.exists(folder):
            try:
                os.mkdir(folder)
            except OSError as err:
                raise RuntimeError("Could not create video output folder %r due to error: %s" % (folder, err))
        if self.mode == "brute":
            image = self.relativistic_system.render_image(st.max_iterations_propagation, st.escape_radius, st.ray_integration_step)
        else:
            image = self.relativistic_system.render_image_efficient(
                st.max_iterations_propagation, st.escape_radius, st.ray_integration_step, st.alphas_num,
                st.max_iterations_sampling, st.sampling_convergence_threshold_1, st.sampling_convergence_threshold_2)
        path_of_image = os.path.join(folder, os.path.splitext(st.output_image_name)[0] + ".png")
        save_image(path_of_image, image)
        return path_of_image


class VideoRenderingSettings:
    """VideoRenderingSettings (src/rendering.rs:356-374), field for field."""

    def __init__(self, frame_rate, resolution_x, resolution_y, camera_diagonal, camera_focal_length,
                 filepath_to_camera_path, filepath_to_background_image_1, filepath_to_background_image_2,
                 filepath_to_output_folder, output_video_name="output_video", escape_radius=100.0,
                 max_iterations_propagation=40000, ray_integration_step=0.05, alphas_num=100, max_iterations_sampling=100,
                 sampling_convergence_threshold_1=1e-5, sampling_convergence_threshold_2=1e-5):
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})


class VideoRenderingSystem:
    """VideoRenderingSystem<M> (src/rendering.rs:178-327) over one curvis Context per rank.

    mode="efficient" (default) is what the reference's video loop calls: render_image_efficient with
    alphas_num AND max_iterations_sampling both taken from `sampling_initial_nums` (src/main.rs:91-110 wires
    max_iterations_sampling to sampling_initial_nums) and `sampling_convergence_threshold_1` passed for both
    thresholds (src/rendering.rs:299-307) -- reproduced.  mode="brute" renders every frame with the per-pixel
    integrator (RelativisticSystem::render_image, src/systems.rs:307-330), the path bench.py measures."""

    def __init__(self, metric, context, interpolator, frame_rate, resolution, camera_diagonal, camera_focal_length,
                 escape_radius, max_iterations_propagation, ray_integration_step, rank=0, world_size=1, batch=8,
                 mode="efficient", sampling_initial_nums=100, sampling_convergence_threshold_1=1e-5):
        if mode not in ("efficient", "brute"):
            raise ValueError("mode must be 'efficient' or 'brute'")
        self.metric = metric
        self.context = context
        self.interpolator = interpolator
        self.frame_rate = float(frame_rate)
        self.resolution = tuple(resolution)
        self.camera_diagonal = float(camera_diagonal)
        self.camera_focal_length = float(camera_focal_length)
        self.escape_radius = float(escape_radius)
        self.max_iterations_propagation = int(max_iterations_propagation)
        self.ray_integration_step = float(ray_integration_step)
        self.rank, self.world_size, self.batch = int(rank), int(world_size), max(1, int(batch))
        self.mode = mode
        self.sampling_initial_nums = int(sampling_initial_nums)
        self.sampling_convergence_threshold_1 = float(sampling_convergence_threshold_1)

    @classmethod
    def new(cls, metric, video_rendering_settings, context=None, rank=0, world_size=1, batch=8, mode="efficient"):
        """VideoRenderingSystem::new(metric, video_rendering_settings) (src/rendering.rs:188-221): loads the two
        backgrounds into the context's HBM and the camera path into an Interpolator.  The reference passes
        `alphas_num` and `max_iterations_sampling` separately and `sampling_convergence_threshold_1` twice (:299-307);
        so does this (threshold_2 of the settings is never read, as there)."""
        from .images import load_image_as_spherical_image
        from .systems import default_context
        st = video_rendering_settings
        context = context or default_context()
        context.set_sky(0, load_image_as_spherical_image(st.filepath_to_background_image_1))
        context.set_sky(1, load_image_as_spherical_image(st.filepath_to_background_image_2))
        self = cls(metric, context, Interpolator.from_file(str(st.filepath_to_camera_path)), st.frame_rate,
                   (st.resolution_x, st.resolution_y), st.camera_diagonal, st.camera_focal_length, st.escape_radius,
                   st.max_iterations_propagation, st.ray_integration_step, rank=rank, world_size=world_size, batch=batch,
                   mode=mode, sampling_initial_nums=st.alphas_num,
                   sampling_convergence_threshold_1=st.sampling_convergence_threshold_1)
        self.max_iterations_sampling = int(st.max_iterations_sampling)
        self.video_rendering_settings = st
        return self

    def render_to_folder(self, output_folder=None):
        """VideoRenderingSystem::render's file side (src/rendering.rs:258-327): <folder> created if missing, a
        pre-existing <folder>/tmp removed and recreated (rank 0; with several ranks the caller puts a barrier between
        this call's start and the first frame, e.g. torch.distributed.barrier), this rank's frames written as
        <folder>/tmp/frame_{index}.png.  Returns the per-frame statistics (see render)."""
        import os
        import shutil
        from .images import save_image
        folder = str(output_folder if output_folder is not None else self.video_rendering_settings.filepath_to_output_folder)
        tmp = os.path.join(folder, "tmp")
        if self.rank == 0:
            if not os.path.exists(folder):
                try:
                    os.mkdir(folder)
                except OSError as err:
                    raise RuntimeError("Could not create video output folder %r due to error: %s" % (folder, err))
            if os.path.exists(tmp):
                try:
                    shutil.rmtree(tmp)
                except OSError as err:
                    raise RuntimeError("Could not remove pre-existing tmp folder %r due to error: %s" % (tmp, err))
            try:
                os.mkdir(tmp)
            except OSError as err:
                raise RuntimeError("Could not create tmp output folder %r due to error: %s" % (tmp, err))

        from .images import save_zlib_stream

        def write(index, frame, _stats):
            path = os.path.join(tmp, "frame_%d.png" % index)
            try:
                if isinstance(frame, (bytes, bytearray)):  # a zlib stream made on the device (curvis_ctx_deflate_frames)
                    save_zlib_stream(path, frame, self.resolution[0], self.resolution[1])
                else:
                    save_image(path, frame)
            except Exception as err:
                raise RuntimeError("Could not save image frame %r due to error: %s" % (path, err))
        # the frames are compressed where they are (filter, Huffman coding, Adler-32 as HIP kernels): only the streams cross PCIe
        return self.render(on_frame=write, download=False, streams=True)

    def times_of_frames(self):
        return times_of_frames(self.interpolator.min_time(), self.interpolator.max_time(), self.frame_rate)

    def camera_at(self, t):
        """update_camera (src/rendering.rs:242-253): a fresh Camera with interpolated pose."""
        it = self.interpolator
        return Camera(it.camera_position(t), it.camera_forward(t), it.camera_up(t), self.camera_focal_length,
                      self.camera_diagonal, self.resolution[0], self.resolution[1])

    def _render_batch(self, cams, download):
        if self.mode == "brute":
            return self.context.render_brute(self.metric, cams, self.max_iterations_propagation, self.escape_radius,
                                             self.ray_integration_step, download=download)
        thr1 = self.sampling_convergence_threshold_1
        return self.context.render_efficient(self.metric, cams, self.max_iterations_propagation, self.escape_radius,
                                             self.ray_integration_step, self.sampling_initial_nums,
                                             getattr(self, "max_iterations_sampling", self.sampling_initial_nums), thr1, thr1,
                                             download=download)

    def _prefetch(self, cams):
        """the sampler of a batch that _render_batch will render later, started now on a stream of its own (a no-op where the library
        would not sample on the device: brute mode, small batches, contexts without the entry point such as the tests' stubs)"""
        if cams and self.mode != "brute" and hasattr(self.context, "prefetch_efficient"):
            thr1 = self.sampling_convergence_threshold_1
            self.context.prefetch_efficient(self.metric, cams, self.max_iterations_propagation, self.escape_radius, self.ray_integration_step,
                                            self.sampling_initial_nums, getattr(self, "max_iterations_sampling", self.sampling_initial_nums), thr1, thr1)

    def render(self, on_frame=None, download=True, streams=False):
        """Render this rank's shard, `batch` frames per launch.  on_frame(index, rgb_or_None, stats_dict) is
        called per frame in index order of the shard; with streams=True the frames stay in HBM and on_frame receives each
        frame's finished zlib stream (bytes; Context.deflate_frames) instead of pixels -- or the pixels after all, should a
        batch not compress into the worst-case buffer.  Returns the list of per-frame statistics dicts: the
        early-termination statistics (rays, executed Euler steps, escaped +l / -l, capped, clamped texels) are
        exact PER FRAME -- the kernels keep one set of counters per frame of a launch
        (curvis_ctx_frame_stats); kernel_ms is the frame's share of its launch."""
        times = self.times_of_frames()
        mine = frames_of_rank(len(times), self.rank, self.world_size)
        out = []
        batches = [mine[b0:b0 + self.batch] for b0 in range(0, len(mine), self.batch)]
        cams_of = lambda idx: [self.camera_at(times[k]) for k in idx]  # noqa: E731
        nxt = cams_of(batches[0]) if batches else None
        self._prefetch(nxt)           # efficient mode: the first batch's sampler, then always one batch ahead (curvis_ctx_prefetch_efficient)
        for bi, idx in enumerate(batches):
            cams = nxt
            nxt = cams_of(batches[bi + 1]) if bi + 1 < len(batches) else None
            self._prefetch(nxt)
            rgb, st = self._render_batch(cams, download and not streams)
            frames = list(rgb) if (download and not streams) else [None] * len(cams)
            per = self.context.frame_stats()
            if streams:
                from ._abi import CurvisError
                try:
                    frames, _ = self.context.deflate_frames(self.resolution[0], self.resolution[1], len(cams))
                except CurvisError:
                    frames = list(self.context.download_frames(self.resolution[0], self.resolution[1], len(cams)))
            assert len(per) == len(idx)
            for k, frame, s in zip(idx, frames, per):
                d = dict(frame=k, time=times[k], rank=self.rank, mode=self.mode, batch_frames=len(idx),
                         rays=int(s.rays), steps=int(s.steps), n_pos=int(s.n_pos), n_neg=int(s.n_neg),
                         n_none=int(s.n_none), n_oob=int(s.n_oob), kernel_ms=float(s.kernel_ms),
                         batch_kernel_ms=float(st.kernel_ms))
                out.append(d)
                if on_frame is not None:
                    on_frame(k, frame, d)
        return out


def gather_frame_stats(local_stats, dist=None):
    """all ranks -> rank 0 list ordered by frame index (host-side gather of small python objects)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(local_stats, key=lambda d: d["frame"])
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, local_stats)
    return sorted([d for part in gathered for d in part], key=lambda d: d["frame"])
