"""ctypes binding of libcurvis_hip.so (include/curvis_hip.h).  No fallback: if the shared
library is missing or no gfx950 GPU is visible, this fails loudly."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcurvis_hip.so")

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_CAMERA_OUTSIDE, E_NO_SKY, E_PARALLEL, E_METRIC, E_RCCL, E_SAMPLING, E_IO = range(-1, -11, -1)
METRIC_ELLIS, METRIC_INTERSTELLAR, METRIC_FLAT = 0, 1, 2
RCCL_ID_BYTES = 128


class CurvisError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("curvis error %d: %s" % (code, msg))
        self.code = code


class Metric(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("rho", C.c_double), ("m", C.c_double), ("a", C.c_double)]


class CameraC(C.Structure):
    _fields_ = [("pos", C.c_double * 4), ("rot", C.c_double * 9), ("focal", C.c_double), ("sensor_w", C.c_double),
                ("sensor_h", C.c_double), ("res_x", C.c_uint32), ("res_y", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("steps", C.c_uint64), ("n_pos", C.c_uint64), ("n_neg", C.c_uint64),
                ("n_none", C.c_uint64), ("n_oob", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("integrate_ms", C.c_double), ("shade_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SamplingInfo(C.Structure):
    _fields_ = [("n_samples", C.c_uint32), ("rounds", C.c_uint32), ("calls", C.c_uint64), ("steps", C.c_uint64),
                ("warned_max_iterations", C.c_int32), ("_pad", C.c_int32)]


RAY_DEBUG = np.dtype([("x", "<f8", 4), ("p", "<f8", 4), ("steps", "<u4"), ("code", "<i4"), ("tx", "<u4"),
                      ("ty", "<u4")])

# every symbol include/curvis_hip.h declares: (restype, argtypes)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
SYMBOLS = {
    "curvis_version": (C.c_char_p, []),
    "curvis_last_error": (C.c_char_p, [_vp]),
    "curvis_device_count": (C.c_int, []),
    "curvis_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "curvis_ctx_destroy": (None, [_vp]),
    "curvis_ctx_device_info": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "curvis_ctx_device_status": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "curvis_ctx_set_sky": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, C.c_uint32]),
    "curvis_ctx_set_sky_device": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_int]),
    "curvis_ctx_set_sky_orientation": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "curvis_ctx_bcast_skies": (C.c_int, [_vp, _vp, C.c_int]),
    "curvis_rccl_unique_id": (C.c_int, [_vp]),
    "curvis_ctx_rccl_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "curvis_rccl_comm_destroy": (C.c_int, [_vp]),
    "curvis_ctx_read_sky": (C.c_int, [_vp, C.c_int, C.c_size_t, C.c_size_t, _vp]),
    "curvis_camera_init": (C.c_int, [C.POINTER(CameraC), _dp, _dp, _dp, C.c_double, C.c_double, C.c_uint32, C.c_uint32]),
    "curvis_orientation_init": (C.c_int, [_dp, _dp, _dp, _dp, _dp]),
    "curvis_metric_validate": (C.c_int, [C.POINTER(Metric)]),
    "curvis_metric_functions": (C.c_int, [C.POINTER(Metric), C.c_double, _dp, _dp, _dp]),
    "curvis_metric_tensor": (C.c_int, [C.POINTER(Metric), _dp, _dp, _dp]),
    "curvis_camera_outward_vector": (C.c_int, [C.POINTER(CameraC), C.c_uint32, C.c_uint32, _dp, _dp]),
    "curvis_vector_to_direction": (C.c_int, [C.POINTER(Metric), _dp, _dp, _dp]),
    "curvis_update_relativistic_object": (C.c_int, [C.POINTER(Metric), _dp, _dp, C.c_double]),
    "curvis_sky_texel_index": (C.c_int, [C.c_uint32, C.c_uint32, _dp, _dp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "curvis_render_brute": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_double, C.c_double,
                                      _vp, C.POINTER(Stats)]),
    "curvis_render_brute_rows": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_double, C.c_double, _vp, C.POINTER(Stats)]),
    "curvis_render_brute_debug": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_double,
                                            C.c_double, _vp, _vp, C.POINTER(Stats)]),
    "curvis_render_brute_batch": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_uint32,
                                            C.c_double, C.c_double, _vp, C.POINTER(Stats)]),
    "curvis_render_efficient": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_double,
                                          C.c_double, C.c_uint32, C.c_uint32, C.c_double, C.c_double, _vp,
                                          C.POINTER(Stats)]),
    "curvis_render_efficient_batch": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_uint32,
                                                C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_double,
                                                C.c_double, _vp, C.POINTER(Stats)]),
    "curvis_ctx_prefetch_efficient": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_uint32,
                                                C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_double, C.c_double]),
    "curvis_render_direct": (C.c_int, [_vp, C.POINTER(Metric), C.POINTER(CameraC), C.c_uint32, C.c_double, C.c_double, _vp,
                                       C.POINTER(Stats)]),
    "curvis_ctx_sampling_info": (C.c_int, [_vp, C.c_uint32, C.POINTER(SamplingInfo)]),
    "curvis_ctx_frame_stats": (C.c_int, [_vp, C.c_uint32, C.POINTER(Stats)]),
    "curvis_ctx_samples": (C.c_int, [_vp, C.c_uint32, _dp, _dp, _dp, C.c_size_t]),
    "curvis_compute_escape_angles": (C.c_int, [_vp, C.POINTER(Metric), C.c_double, _dp, C.c_uint32, C.c_double,
                                               C.c_uint32, C.c_double, _dp, C.POINTER(C.c_int32),
                                               C.POINTER(C.c_uint32)]),
    "curvis_new_photon": (C.c_int, [C.POINTER(Metric), _dp, _dp, _dp, _dp]),
    "curvis_photon_trajectories": (C.c_int, [_vp, C.POINTER(Metric), C.c_uint32, _dp, _dp, C.c_uint32, C.c_double,
                                             _dp]),
    "curvis_image_load": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "curvis_image_free": (None, [C.POINTER(C.c_uint8)]),
    "curvis_image_save_rgb8": (C.c_int, [C.c_char_p, _vp, C.c_uint32, C.c_uint32]),
    "curvis_image_save_rgb8_level": (C.c_int, [C.c_char_p, _vp, C.c_uint32, C.c_uint32, C.c_int]),
    "curvis_ctx_deflate_frames": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_size_t, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_double)]),
    "curvis_image_save_zlib_rgb8": (C.c_int, [C.c_char_p, _vp, C.c_size_t, C.c_uint32, C.c_uint32]),
    "curvis_ctx_deflate_frames_crc": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_size_t, C.POINTER(C.c_size_t),
                                                C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "curvis_image_save_zlib_rgb8_crc": (C.c_int, [C.c_char_p, _vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32]),
    "curvis_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "curvis_host_free": (None, [_vp]),
    "curvis_ctx_framebuffer": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "curvis_ctx_download": (C.c_int, [_vp, _vp, C.c_size_t]),
    "curvis_ctx_upload": (C.c_int, [_vp, _vp, C.c_size_t]),
    "curvis_ctx_synchronize": (C.c_int, [_vp]),
    "curvis_ctx_download_wait": (C.c_int, [_vp]),
    "curvis_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "curvis_ctx_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "curvis_device_link": (C.c_int, [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 5),
    "curvis_selftest_math": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, C.c_size_t]),
    "curvis_selftest_math3": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, C.c_size_t]),
    "curvis_selftest_fast_step": (C.c_int, [_vp, C.POINTER(Metric), C.c_double, C.c_double, _dp, C.c_size_t, _dp]),
}

_lib = None


def lib():
    """Load libcurvis_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C curvis_amd/csrc` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, ctx=None):
    if rc != OK:
        msg = lib().curvis_last_error(ctx)
        raise CurvisError(rc, msg.decode() if msg else "")
    return rc


def dptr(a):
    return a.ctypes.data_as(_dp)
