"""Settings of the reference's command line (src/settings.rs:22-293), Python side: the six settings groups with
their TOML keys (including the reference's spelling `ray_integration_max_itarations`), `validate()` with the
reference's messages, `normalize()`, `from_toml_file()`, and defaults equal to settings/defaults/*.toml.
`image_rendering_settings()` / `video_rendering_system()` wire them into the rendering systems the way
src/main.rs:14-111 does -- including its two quirks: `max_iterations_sampling` takes `sampling_initial_nums`, and
the video path uses `sampling_convergence_threshold_1` for both thresholds (src/rendering.rs:305-306).

The `curvis` binary (curvis_amd/bin/curvis) implements the same in C++; this module is the host mirror for Python
callers and for the parity tests."""
import math
import os

from . import paths


class SettingsError(ValueError):
    """what the reference returns as Err(String)"""


def _load_toml(path):
    path = str(path)
    if not path.endswith(".toml"):
        raise SettingsError('The file "%s" is not a toml file.' % path)
    try:
        import tomli
    except ImportError:  # Python >= 3.11
        import tomllib as tomli
    try:
        with open(path, "rb") as f:
            return tomli.load(f)
    except OSError:
        raise SettingsError('Could not read file "%s"' % path)
    except Exception as exc:  # tomli.TOMLDecodeError
        raise SettingsError("Could not parse toml file: %s" % exc)


class _Settings:
    FIELDS = ()  # (name, type, default)

    def __init__(self, **kw):
        for name, typ, default in self.FIELDS:
            setattr(self, name, typ(kw.pop(name, default)))
        if kw:
            raise TypeError("unknown setting(s): %s" % ", ".join(sorted(kw)))

    @classmethod
    def from_toml_file(cls, toml_file_path):
        """FromToml::from_toml_file: every key is required (serde: `missing field`), then normalize + validate"""
        table = _load_toml(toml_file_path)
        vals = {}
        for name, typ, _ in cls.FIELDS:
            if name not in table:
                raise SettingsError("missing field `%s`" % name)
            v = table[name]
            if typ is int and (isinstance(v, bool) or not isinstance(v, int) or v < 0 or v > 0xFFFFFFFF):
                raise SettingsError("invalid type or range for `%s`: expected u32" % name)
            if typ is float and (isinstance(v, bool) or not isinstance(v, (int, float))):
                raise SettingsError("invalid type for `%s`: expected a float" % name)
            if typ is str and not isinstance(v, str):
                raise SettingsError("invalid type for `%s`: expected a string" % name)
            vals[name] = v
        s = cls(**vals)
        s.normalize()
        s.validate()
        return s

    def normalize(self):
        pass

    def validate(self):
        pass

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % (n, getattr(self, n)) for n, _, _ in self.FIELDS))


class VideoSettings(_Settings):
    FIELDS = (("video_name", str, "output_video"), ("frame_rate", float, 30.0),
              ("filepath_to_camera_path", str, "paths/path_through.csv"))

    def normalize(self):
        """resolve_path (src/filepaths.rs:42-47): an absolute path is kept, a relative one is joined to the package
        folder -- subdirectories included.  The package folder here is curvis_amd/ and the bundled camera paths live
        in curvis_amd/data/paths/, so the reference's default "paths/path_through.csv" is looked up as
        <package>/paths/... and <package>/data/paths/...; a relative path that exists from the working directory is
        accepted first (an extension shared with the `curvis` binary, host/curvis_cli.cpp resolve_path)."""
        p = self.filepath_to_camera_path
        if not os.path.isabs(p) and not os.path.exists(p):
            root = os.path.dirname(os.path.abspath(__file__))
            if os.path.basename(p) in ("path_orbit.csv", "path_through.csv") and os.path.normpath(p) == os.path.join("paths", os.path.basename(p)):
                paths.ensure_paths()  # the two bundled paths are generated on demand
            for cand in (os.path.join(root, p), os.path.join(root, "data", p)):
                if os.path.exists(cand):
                    p = cand
                    break
            else:
                p = os.path.join(root, p)
        self.filepath_to_camera_path = p

    def validate(self):
        if self.video_name == "":
            raise SettingsError("Video name cannot be an empty string.")
        if not self.filepath_to_camera_path.endswith(".csv"):
            raise SettingsError('The camera path "%s" is not a csv file.' % self.filepath_to_camera_path)
        if not os.path.exists(self.filepath_to_camera_path):
            raise SettingsError('The camera path "%s" does not exist.' % self.filepath_to_camera_path)


class ImageSettings(_Settings):
    FIELDS = (("image_name", str, "output_image"), ("t", float, 0.0), ("l", float, 5.0), ("theta", float, math.pi / 2),
              ("phi", float, 0.0), ("forward_x", float, -1.0), ("forward_y", float, 0.0), ("forward_z", float, 0.0),
              ("up_x", float, 0.0), ("up_y", float, 0.0), ("up_z", float, 1.0))

    def validate(self):
        if self.image_name == "":
            raise SettingsError("Image name cannot be an empty string.")


class CameraSettings(_Settings):
    FIELDS = (("resolution_x", int, 960), ("resolution_y", int, 540), ("diagonal", float, 43.0), ("focal_length", float, 15.0))

    def validate(self):
        if self.resolution_x <= 0:
            raise SettingsError("The resolution in the x direction must be larger than zero.")
        if self.resolution_y <= 0:
            raise SettingsError("The resolution in the y direction must be larger than zero.")
        if self.diagonal <= 0.0:
            raise SettingsError("The diagonal of the camera must be larger than zero.")
        if self.focal_length <= 0.0:
            raise SettingsError("The focal length of the camera must be larger than zero.")


class SimulationSettings(_Settings):
    FIELDS = (("escape_radius", float, 100.0), ("ray_integration_max_itarations", int, 40000),
              ("ray_integration_step", float, 0.05), ("sampling_initial_nums", int, 100), ("sampling_max_iterations", int, 50),
              ("sampling_convergence_threshold_1", float, 1e-5), ("sampling_convergence_threshold_2", float, 1e-5))

    def validate(self):
        if self.escape_radius <= 0.0:
            raise SettingsError("The escape radius must be larger than zero.")
        if self.ray_integration_max_itarations <= 0:
            raise SettingsError("The maximum number of iterations for the ray integration must be larger than zero.")
        if self.ray_integration_step <= 0.0:
            raise SettingsError("The step for the ray integration must be larger than zero.")
        if self.sampling_initial_nums <= 1:
            raise SettingsError("The initial number of samples must be larger than two.")
        if self.sampling_max_iterations <= 0:
            raise SettingsError("The maximum number of iterations for the sampling must be larger than zero.")
        if self.sampling_convergence_threshold_1 <= 0.0:
            raise SettingsError("The first convergence threshold for the sampling must be larger than zero.")
        if self.sampling_convergence_threshold_2 <= 0.0:
            raise SettingsError("The second convergence threshold for the sampling must be larger than zero.")


class EllisMetricSettings(_Settings):
    FIELDS = (("rho", float, 1.0),)

    def validate(self):
        if self.rho <= 0.0:
            raise SettingsError("The density parameter rho must be larger than zero.")

    def metric(self):
        from .systems import EllisMetric
        return EllisMetric(self.rho)


class InterstellarMetricSettings(_Settings):
    FIELDS = (("m", float, 0.1), ("a", float, 1e-4), ("rho", float, 1.0))

    def validate(self):
        if self.m <= 0.0:
            raise SettingsError("The mass parameter m must be larger than zero.")
        if self.a <= 0.0:
            raise SettingsError("The spin parameter a must be larger than zero.")
        if self.rho <= 0.0:
            raise SettingsError("The density parameter rho must be larger than zero.")

    def metric(self):
        from .systems import InterstellarMetric
        return InterstellarMetric(self.m, self.a, self.rho)


def metric_settings_from_toml_file(path):
    """src/cli.rs:233-261: a metric file is tried as Interstellar settings first, then as Ellis settings"""
    try:
        return InterstellarMetricSettings.from_toml_file(path)
    except SettingsError as first:
        try:
            return EllisMetricSettings.from_toml_file(path)
        except SettingsError:
            # src/cli.rs:255-259: neither parse succeeded -> the reference's one message, not the first parser's
            raise SettingsError("Could not read the metric configuration file.") from first


def image_rendering_settings(background_1, background_2, output_folder, image=None, camera=None, simulation=None):
    """setup of ImageRenderingSettings from the settings groups (src/main.rs:14-52): alphas_num AND
    max_iterations_sampling both take sampling_initial_nums"""
    from .rendering import ImageRenderingSettings
    image, camera, simulation = image or ImageSettings(), camera or CameraSettings(), simulation or SimulationSettings()
    for s in (image, camera, simulation):
        s.normalize()
        s.validate()
    return ImageRenderingSettings(
        background_1, background_2, output_folder, image.image_name, (image.t, image.l, image.theta, image.phi),
        (image.forward_x, image.forward_y, image.forward_z), (image.up_x, image.up_y, image.up_z), camera.focal_length,
        camera.diagonal, camera.resolution_x, camera.resolution_y, simulation.escape_radius,
        simulation.ray_integration_max_itarations, simulation.ray_integration_step, simulation.sampling_initial_nums,
        simulation.sampling_initial_nums, simulation.sampling_convergence_threshold_1,
        simulation.sampling_convergence_threshold_2)


def video_rendering_system(metric, context, video=None, camera=None, simulation=None, rank=0, world_size=1, batch=8,
                           mode="efficient"):
    """VideoRenderingSystem from the settings groups (src/main.rs:54-111)"""
    from .rendering import Interpolator, VideoRenderingSystem
    video, camera, simulation = video or VideoSettings(), camera or CameraSettings(), simulation or SimulationSettings()
    for s in (video, camera, simulation):
        s.normalize()
        s.validate()
    return VideoRenderingSystem(
        metric, context, Interpolator.from_file(video.filepath_to_camera_path), video.frame_rate,
        (camera.resolution_x, camera.resolution_y), camera.diagonal, camera.focal_length, simulation.escape_radius,
        simulation.ray_integration_max_itarations, simulation.ray_integration_step, rank=rank, world_size=world_size,
        batch=batch, mode=mode, sampling_initial_nums=simulation.sampling_initial_nums,
        sampling_convergence_threshold_1=simulation.sampling_convergence_threshold_1)
