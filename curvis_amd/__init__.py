"""curvis_amd -- MI355X-native per-pixel geodesic renderer behind the CurVis interface.

Python mirror of the reference's hot-path surface (src/lib.rs re-exports): `Camera`,
`EllisMetric`, `InterstellarMetric`, `FlatSphericalMetric`, `SphericalImage`,
`RelativisticSystem.render_image`.  All compute goes through the C ABI of
curvis_amd/lib/libcurvis_hip.so (include/curvis_hip.h); there is no CPU path.
"""
from ._abi import CurvisError, LIB_PATH, lib  # noqa: F401
from .systems import (Camera, Context, EllisMetric, FlatSphericalMetric, InterstellarMetric,  # noqa: F401
                      RelativisticSystem, SphericalImage)
from . import skies  # noqa: F401
from . import images  # noqa: F401

__all__ = ["Camera", "Context", "EllisMetric", "InterstellarMetric", "FlatSphericalMetric", "SphericalImage",
           "RelativisticSystem", "CurvisError", "skies", "images"]
