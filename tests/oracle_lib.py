"""ctypes binding of oracle/libcurvis_oracle.so -- TEST INFRASTRUCTURE (checker only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
LIBM, CV = 0, 1
# glibc with sin+cos pairs merged into ONE sincos() call (inside one reference function / throughout the inlined update):
# whether rustc/LLVM merges them in the reference cannot be seen here, so every variant is carried (oracle/curvis_oracle.h)
LIBM_SINCOS, LIBM_SINCOS_INL = 2, 3
GLIBC_FLAVOURS = (LIBM, LIBM_SINCOS, LIBM_SINCOS_INL)
FLAVOUR_NAMES = {LIBM: "CVO_LIBM (sin, cos separate)", CV: "CVO_CV (cv_math.h)", LIBM_SINCOS: "CVO_LIBM_SINCOS (sincos per function)",
                 LIBM_SINCOS_INL: "CVO_LIBM_SINCOS_INL (sincos, update inlined)"}
ELLIS, INTERSTELLAR, FLAT = 0, 1, 2
NOT_ESCAPED, POSITIVE, NEGATIVE, PANIC = 0, 1, -1, -2


class Metric(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("rho", C.c_double), ("m", C.c_double), ("a", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_double * 4), ("rot", C.c_double * 9), ("inv_rot", C.c_double * 9),
                ("fwd", C.c_double * 3), ("up", C.c_double * 3), ("focal", C.c_double),
                ("sensor_w", C.c_double), ("sensor_h", C.c_double), ("res_x", C.c_uint32), ("res_y", C.c_uint32)]


class Sky(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("inv_rot", C.c_double * 9)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "steps", "n_pos", "n_neg", "n_none", "n_oob")]


class Samples(C.Structure):
    _fields_ = [("a", C.POINTER(C.c_double)), ("e", C.POINTER(C.c_double)), ("s", C.POINTER(C.c_double)),
                ("n", C.c_size_t), ("calls", C.c_uint64), ("steps", C.c_uint64), ("rounds", C.c_uint32),
                ("warned_max_iterations", C.c_int)]


class Path(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_double)), ("fwd", C.POINTER(C.c_double)), ("up", C.POINTER(C.c_double)),
                ("n", C.c_size_t)]


RAY_DEBUG = np.dtype([("x", "<f8", 4), ("p", "<f8", 4), ("steps", "<u4"), ("code", "<i4"), ("tx", "<u4"),
                      ("ty", "<u4")])

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "libcurvis_oracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    d, dp, i32, u32 = C.c_double, C.POINTER(C.c_double), C.c_int, C.c_uint32
    vp = C.c_void_p
    MP, CP, SP = C.POINTER(Metric), C.POINTER(Camera), C.POINTER(Sky)

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args

    sig("cvo_orientation_new", i32, [dp, dp, dp, dp, dp])
    sig("cvo_face_towards", None, [dp, dp, dp])
    sig("cvo_rotation_between", i32, [i32, dp, dp, dp])
    sig("cvo_rotation_from_two_vectors", i32, [i32, dp, dp, dp])
    sig("cvo_from_axis_angle", None, [i32, dp, d, dp])
    sig("cvo_normalize_theta_phi", None, [d, d, dp, dp])
    sig("cvo_vector3_from_theta_phi", None, [i32, d, d, dp])
    sig("cvo_theta_phi_from_vector3", None, [i32, dp, dp, dp])
    sig("cvo_mat3_vec", None, [dp, dp, dp])
    sig("cvo_camera_new", i32, [CP, dp, dp, dp, d, d, u32, u32])
    sig("cvo_camera_outward_camera_space", None, [CP, u32, u32, dp])
    sig("cvo_camera_outward_world", None, [CP, u32, u32, dp])
    sig("cvo_metric_r", d, [i32, MP, d])
    sig("cvo_metric_r_squared", d, [i32, MP, d])
    sig("cvo_metric_r_derivative", d, [i32, MP, d])
    sig("cvo_new_photon", None, [i32, MP, dp, dp, dp, dp])
    sig("cvo_update", None, [i32, MP, dp, dp, d])
    sig("cvo_update_memo", None, [i32, MP, dp, dp, d])
    sig("cvo_set_metric_memo", None, [i32])
    sig("cvo_get_metric_memo", i32, [])
    sig("cvo_escape_photon", i32, [i32, MP, dp, dp, d, u32, d, C.POINTER(u32)])
    sig("cvo_vector_to_direction", None, [i32, MP, dp, dp, dp])
    sig("cvo_squared_norm_cov", d, [i32, MP, dp, dp])
    sig("cvo_sky_indices", None, [i32, SP, dp, C.POINTER(u32), C.POINTER(u32)])
    sig("cvo_sky_pixel", i32, [i32, SP, dp, C.POINTER(C.c_uint8)])
    sig("cvo_render_image", i32, [i32, MP, CP, SP, SP, u32, d, d, u32, u32, vp, vp, C.POINTER(Stats)])
    sig("cvo_compute_escape_angle", i32, [i32, MP, d, d, d, u32, d, dp, C.POINTER(u32)])
    sig("cvo_doubly_sample", i32, [i32, MP, d, d, u32, d, d, d, C.c_size_t, C.c_size_t, d, d, C.POINTER(Samples)])
    sig("cvo_samples_free", None, [C.POINTER(Samples)])
    sig("cvo_interp_slice", None, [dp, dp, C.c_size_t, dp, C.c_size_t, dp])
    sig("cvo_render_image_efficient", i32, [i32, MP, CP, SP, SP, u32, d, d, u32, u32, d, d, vp,
                                            C.POINTER(Samples), C.POINTER(Stats)])
    sig("cvo_render_image_direct", i32, [i32, MP, CP, SP, SP, u32, d, d, vp, C.POINTER(Stats)])
    sig("cvo_load_path", i32, [C.c_char_p, C.POINTER(Path)])
    sig("cvo_path_free", None, [C.POINTER(Path)])
    sig("cvo_path_camera", i32, [C.POINTER(Path), d, dp, dp, dp])
    sig("cvo_times_of_frames", C.c_size_t, [d, d, d, dp, C.c_size_t])
    sig("cvo_math_array", None, [i32, i32, dp, dp, dp, C.c_size_t])
    sig("cvo_photon_trajectory", None, [i32, MP, dp, dp, u32, d, dp])
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def vec(*v):
    return np.array(v, dtype=np.float64)


def ellis(rho=1.0):
    return Metric(ELLIS, 0, rho, 0.0, 0.0)


def interstellar(m=0.1, a=1e-4, rho=1.0):
    return Metric(INTERSTELLAR, 0, rho, m, a)


def flat():
    return Metric(FLAT, 0, 0.0, 0.0, 0.0)


def camera(pos=(0.0, 5.0, np.pi / 2, 0.0), fwd=(-1.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), focal=15.0, diag=43.0,
           res=(960, 540)):
    c = Camera()
    rc = lib().cvo_camera_new(C.byref(c), _dp(vec(*pos)), _dp(vec(*fwd)), _dp(vec(*up)), focal, diag, res[0], res[1])
    if rc != 0:
        raise ValueError("cvo_camera_new failed: %d" % rc)
    return c


def sky(rgba, inv_rot=None):
    """rgba: HxWx4 uint8 C-contiguous (kept alive by the returned object)."""
    assert rgba.dtype == np.uint8 and rgba.ndim == 3 and rgba.shape[2] == 4 and rgba.flags.c_contiguous
    s = Sky()
    s.rgba = rgba.ctypes.data
    s.w, s.h = rgba.shape[1], rgba.shape[0]
    m = np.eye(3).ravel() if inv_rot is None else np.asarray(inv_rot, dtype=np.float64).ravel()
    for i in range(9):
        s.inv_rot[i] = m[i]
    s._keep = rgba
    return s


def render_image(fl, metric, cam, sky_pos, sky_neg, max_iter, max_radius, delta, row_begin=0, row_step=1,
                 debug=False):
    W, H = cam.res_x, cam.res_y
    rgb = np.zeros((H, W, 3), dtype=np.uint8)
    dbg = np.zeros((H, W), dtype=RAY_DEBUG) if debug else None
    st = Stats()
    rc = lib().cvo_render_image(fl, C.byref(metric), C.byref(cam), C.byref(sky_pos), C.byref(sky_neg), max_iter,
                                max_radius, delta, row_begin, row_step, rgb.ctypes.data,
                                dbg.ctypes.data if debug else None, C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle panic: %d" % rc)
    return rgb, dbg, st


def render_image_efficient(fl, metric, cam, sky_pos, sky_neg, max_iter, max_radius, delta, alpha_nums,
                           max_it_sampling, thr1, thr2):
    W, H = cam.res_x, cam.res_y
    rgb = np.zeros((H, W, 3), dtype=np.uint8)
    st = Stats()
    smp = Samples()
    rc = lib().cvo_render_image_efficient(fl, C.byref(metric), C.byref(cam), C.byref(sky_pos), C.byref(sky_neg),
                                          max_iter, max_radius, delta, alpha_nums, max_it_sampling, thr1, thr2,
                                          rgb.ctypes.data, C.byref(smp), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle panic: %d" % rc)
    n = smp.n
    samples = dict(a=np.ctypeslib.as_array(smp.a, (n,)).copy(), e=np.ctypeslib.as_array(smp.e, (n,)).copy(),
                   s=np.ctypeslib.as_array(smp.s, (n,)).copy(), calls=smp.calls, steps=smp.steps,
                   rounds=smp.rounds)
    lib().cvo_samples_free(C.byref(smp))
    return rgb, samples, st


def render_image_direct(fl, metric, cam, sky_pos, sky_neg, max_iter, max_radius, delta):
    """"direct" mode (not a reference function): compute_escape_angle for the alpha of every pixel"""
    W, H = cam.res_x, cam.res_y
    rgb = np.zeros((H, W, 3), dtype=np.uint8)
    st = Stats()
    rc = lib().cvo_render_image_direct(fl, C.byref(metric), C.byref(cam), C.byref(sky_pos), C.byref(sky_neg), max_iter,
                                       max_radius, delta, rgb.ctypes.data, C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle panic: %d" % rc)
    return rgb, st


def escape_photon(fl, metric, pos, direction, delta, max_iter, max_radius):
    x = np.zeros(4)
    p = np.zeros(4)
    lib().cvo_new_photon(fl, C.byref(metric), _dp(vec(*pos)), _dp(vec(*direction)), _dp(x), _dp(p))
    steps = C.c_uint32(0)
    code = lib().cvo_escape_photon(fl, C.byref(metric), _dp(x), _dp(p), delta, max_iter, max_radius, C.byref(steps))
    return code, steps.value, x, p


def compute_escape_angle(fl, metric, l, alpha, delta, max_iter, max_radius):
    ang = C.c_double(0.0)
    steps = C.c_uint32(0)
    code = lib().cvo_compute_escape_angle(fl, C.byref(metric), l, alpha, delta, max_iter, max_radius, C.byref(ang),
                                          C.byref(steps))
    return code, ang.value, steps.value


def math_array(fl, op, a, b=None):
    """elementary function `op` (0 sin, 1 cos, 2 atan, 3 acos, 4 log, 5 atan2(a, b)) of flavour fl over an array"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    out = np.empty_like(a)
    lib().cvo_math_array(fl, op, _dp(a), _dp(bb) if bb is not None else None, _dp(out), a.size)
    return out


def photon_trajectory(fl, metric, pos, direction, iterations, delta):
    """compute_photon_trajectory (src/systems.rs:77-92): [iterations, 8] = (x, p_cov) before each Euler step"""
    x = np.zeros(4)
    p = np.zeros(4)
    lib().cvo_new_photon(fl, C.byref(metric), _dp(vec(*pos)), _dp(vec(*direction)), _dp(x), _dp(p))
    out = np.zeros((iterations, 8))
    lib().cvo_photon_trajectory(fl, C.byref(metric), _dp(x), _dp(p), iterations, delta, _dp(out))
    return out


_quad = None


def quad_ulp_errors(op, a, got, b=None):
    """|got - f(a)| in ulps of the exact value, f evaluated in binary128 by libquadmath (oracle/curvis_quad.c)"""
    global _quad
    if _quad is None:
        so = os.path.join(ORACLE_DIR, "libcurvis_quad.so")
        if not os.path.exists(so):
            build()
        _quad = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        _quad.cvq_ulp_errors.restype = None
        _quad.cvq_ulp_errors.argtypes = [C.c_int, dp, dp, dp, dp, C.c_size_t]
    a = np.ascontiguousarray(a, dtype=np.float64)
    got = np.ascontiguousarray(got, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    err = np.empty_like(a)
    _quad.cvq_ulp_errors(op, _dp(a), _dp(bb) if bb is not None else None, _dp(got), _dp(err), a.size)
    return err
