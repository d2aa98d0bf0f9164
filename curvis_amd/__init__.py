"""curvis_amd -- MI355X-native per-pixel geodesic renderer behind the CurVis interface.

Python mirror of the reference's hot-path surface (src/lib.rs re-exports): `Camera`,
`EllisMetric`, `InterstellarMetric`, `FlatSphericalMetric`, `SphericalImage`,
`RelativisticSystem.render_image`.  All compute goes through the C ABI of
curvis_amd/lib/libcurvis_hip.so (include/curvis_hip.h); there is no CPU path.
"""
from ._abi import CurvisError, LIB_PATH, lib  # noqa: F401
from .systems import (Camera, Context, HostBuffer, DiagonalSphericalMetric, EllisMetric, EscapeAngle,  # noqa: F401
                      FlatSphericalMetric, InterstellarMetric, RelativisticSystem, SphericalImage,
                      compute_escape_angle, compute_photon_trajectory)
from .vectors import Covariance, CovarianceError, RelativisticObject, RelativisticVector  # noqa: F401
from .algebra import Orientation  # noqa: F401
from .images import load_image_as_spherical_image  # noqa: F401
from . import skies  # noqa: F401
from . import images  # noqa: F401

# the re-exports of src/lib.rs:28-37 (the rendering / settings types live in curvis_amd.rendering / .settings and are
# re-exported lazily below: they import this package)
__all__ = ["Camera", "Context", "HostBuffer", "EllisMetric", "InterstellarMetric", "FlatSphericalMetric", "DiagonalSphericalMetric",
           "SphericalImage", "load_image_as_spherical_image", "RelativisticSystem", "RelativisticObject",
           "RelativisticVector", "Covariance", "CovarianceError", "Orientation", "EscapeAngle", "compute_escape_angle",
           "compute_photon_trajectory", "CurvisError", "skies", "images",
           "ImageRenderingSystem", "ImageRenderingSettings", "VideoRenderingSystem", "VideoRenderingSettings",
           "CameraSettings", "VideoSettings", "ImageSettings", "InterstellarMetricSettings", "EllisMetricSettings",
           "SimulationSettings"]

_LAZY = {"ImageRenderingSystem": "rendering", "ImageRenderingSettings": "rendering", "VideoRenderingSystem": "rendering",
         "VideoRenderingSettings": "rendering", "CameraSettings": "settings", "VideoSettings": "settings",
         "ImageSettings": "settings", "InterstellarMetricSettings": "settings", "EllisMetricSettings": "settings",
         "SimulationSettings": "settings"}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
