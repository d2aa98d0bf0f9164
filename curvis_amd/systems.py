"""Host-side mirror of the reference's scene types over the C ABI.

Reference surface mirrored (names, argument meaning and error behaviour):
  EllisMetric::new(rho)                      src/metrics.rs:407-414
  InterstellarMetric::new(m, a, rho)         src/metrics.rs:443-459
  FlatSphericalMetric::new()                 src/metrics.rs:496-498
  Camera::new(position, forward_world, up_world, focal_length, sensor_diagonal,
              resolution_width, resolution_height)                src/cameras.rs:79-122
  SphericalImage::new(img, forward, up)      src/images.rs:71-90
  RelativisticSystem::new(metric, background_positive, background_negative, camera)  src/systems.rs:286-288
  RelativisticSystem::render_image(max_iterations, max_radius, delta)                src/systems.rs:307-330
Reference panics surface as CurvisError (or ValueError for constructor argument checks).
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import RCCL_ID_BYTES, CameraC, Metric, Stats, check, dptr, lib
from .vectors import Covariance, CovarianceError, RelativisticObject, RelativisticVector


class DiagonalSphericalMetric:
    """trait DiagonalSphericalMetric (src/metrics.rs:40-349) for the closed set of metrics the kernels implement.
    Required methods r(l), r_squared(l), r_derivative(l) (:40-48) through curvis_metric_functions; the diagonal of
    the metric tensor and the index raising / lowering built on it (:49-219) through curvis_metric_tensor;
    new_photon (:301-334) through curvis_new_photon -- all host-side, in the kernels' arithmetic."""

    def _functions(self, l):
        m = self._c()
        out = (C.c_double * 3)()
        rc = lib().curvis_metric_functions(C.byref(m), float(l), C.cast(C.byref(out, 0), C.POINTER(C.c_double)),
                                           C.cast(C.byref(out, 8), C.POINTER(C.c_double)),
                                           C.cast(C.byref(out, 16), C.POINTER(C.c_double)))
        if rc != 0:
            raise ValueError("invalid metric parameters")
        return out[0], out[1], out[2]

    def r(self, l):
        return self._functions(l)[0]

    def r_squared(self, l):
        return self._functions(l)[1]

    def r_derivative(self, l):
        return self._functions(l)[2]

    def _tensor(self, position_contr):
        if position_contr.covariance != Covariance.Contravariant:  # check_contravariance, src/metrics.rs:9-13
            raise CovarianceError("The position vector must be contravariant.")
        m = self._c()
        pos = np.ascontiguousarray(position_contr.vector, dtype=np.float64)
        g, gi = np.zeros(4), np.zeros(4)
        if lib().curvis_metric_tensor(C.byref(m), dptr(pos), dptr(g), dptr(gi)) != 0:
            raise ValueError("invalid metric parameters")
        return g, gi

    def gii(self, i, position_contr):
        """covariant (i, i) component of the metric tensor (src/metrics.rs:49-79)"""
        if i not in (0, 1, 2, 3):
            raise IndexError("Invalid index for the covariant metric components.")
        return float(self._tensor(position_contr)[0][i])

    def gii_contr(self, i, position_contr):
        """contravariant (i, i) component: gii.powi(-1) (src/metrics.rs:81-104)"""
        if i not in (0, 1, 2, 3):
            raise IndexError("Invalid index for the contravariant metric components.")
        return float(self._tensor(position_contr)[1][i])

    def to_covariant(self, position_contr, vector_contr):
        """src/metrics.rs:148-161: v_i = v^i * g_ii"""
        if vector_contr.covariance == Covariance.Covariant:
            raise CovarianceError("The vector is already covariant.")
        return RelativisticVector(vector_contr.vector * self._tensor(position_contr)[0], Covariance.Covariant)

    def to_contravariant(self, position_contr, vector_cov):
        """src/metrics.rs:190-203: v^i = v_i * g^ii"""
        if vector_cov.covariance == Covariance.Contravariant:
            raise CovarianceError("The vector is already contravariant.")
        return RelativisticVector(vector_cov.vector * self._tensor(position_contr)[1], Covariance.Contravariant)

    def relativistic_vector_to_direction(self, vector, position):
        """src/metrics.rs:339-349: vector at `position` (a covariant one is raised first, :340-342) ->
        tangent-space direction (not normalised; the z component uses frame_field_22, as the reference does)"""
        if position.covariance != Covariance.Contravariant:
            raise CovarianceError("The position vector must be contravariant.")
        if vector.covariance == Covariance.Contravariant:  # used as it is: components times the frame field
            r = np.float64(self.r(position.v(1)))
            return np.array([np.float64(vector.v(1)) * 1.0, np.float64(vector.v(2)) * r, np.float64(vector.v(3)) * r])
        m = self._c()
        pos = np.ascontiguousarray(position.vector, dtype=np.float64)
        p = np.ascontiguousarray(vector.vector, dtype=np.float64)
        out = np.zeros(3)
        check(lib().curvis_vector_to_direction(C.byref(m), dptr(pos), dptr(p), dptr(out)))
        return out

    def update_relativistic_object(self, obj, delta):
        """src/metrics.rs:283-297: ONE forward-Euler step of `obj` in place (host-side, the kernels' arithmetic); a
        contravariant momentum is lowered first (:285-288)"""
        if obj.covariance_x() != Covariance.Contravariant:
            raise CovarianceError("The position vector must be contravariant.")
        if obj.covariance_p() == Covariance.Contravariant:
            obj.momentum = self.to_covariant(obj.position, obj.momentum)
        m = self._c()
        x = np.ascontiguousarray(obj.position.vector, dtype=np.float64).copy()
        p = np.ascontiguousarray(obj.momentum.vector, dtype=np.float64).copy()
        check(lib().curvis_update_relativistic_object(C.byref(m), dptr(x), dptr(p), float(delta)))
        obj.position = RelativisticVector(x, Covariance.Contravariant)
        obj.momentum = RelativisticVector(p, Covariance.Covariant)

    def new_photon(self, position, direction):
        """src/metrics.rs:301-334: a photon at `position` (contravariant t, l, theta, phi) leaving along the
        tangent-space `direction` (normalised here): covariant momentum (1, d0, d1 r(l), d2 r(l) sin(theta))."""
        if position.covariance != Covariance.Contravariant:
            raise CovarianceError("The position vector must be contravariant.")
        m = self._c()
        pos = np.ascontiguousarray(position.vector, dtype=np.float64)
        d = np.ascontiguousarray(direction, dtype=np.float64).reshape(3)
        x, p = np.zeros(4), np.zeros(4)
        check(lib().curvis_new_photon(C.byref(m), dptr(pos), dptr(d), dptr(x), dptr(p)))
        return RelativisticObject(RelativisticVector(x, Covariance.Contravariant), RelativisticVector(p, Covariance.Covariant))


_MetricFunctions = DiagonalSphericalMetric  # earlier name


class EllisMetric(DiagonalSphericalMetric):
    def __init__(self, rho):
        if not rho > 0.0:
            raise ValueError("The rho parameter for Ellis Metrics must be positive.")
        self.rho = float(rho)

    def _c(self):
        return Metric(_abi.METRIC_ELLIS, 0, self.rho, 0.0, 0.0)


class InterstellarMetric(DiagonalSphericalMetric):
    def __init__(self, m, a, rho):
        if not m > 0.0:
            raise ValueError("The mass parameter for Interstellar Metrics must be positive.")
        if not a > 0.0:
            raise ValueError("The angular momentum parameter for Interstellar Metrics must be positive.")
        if not rho > 0.0:
            raise ValueError("The rho parameter for Interstellar Metrics must be positive.")
        self.m, self.a, self.rho = float(m), float(a), float(rho)

    def _c(self):
        return Metric(_abi.METRIC_INTERSTELLAR, 0, self.rho, self.m, self.a)


class FlatSphericalMetric(DiagonalSphericalMetric):
    def _c(self):
        return Metric(_abi.METRIC_FLAT, 0, 0.0, 0.0, 0.0)


def _vec(v, n):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(n))
    return a


class Camera:
    def __init__(self, position, forward_world, up_world, focal_length, sensor_diagonal, resolution_width,
                 resolution_height):
        self._c = CameraC()
        self.position = _vec(position, 4)
        self.forward = _vec(forward_world, 3)
        self.up = _vec(up_world, 3)
        self.focal_length = float(focal_length)
        self.sensor_diagonal = float(sensor_diagonal)
        self.resolution_width = int(resolution_width)
        self.resolution_height = int(resolution_height)
        self._rebuild()

    def _rebuild(self):
        rc = lib().curvis_camera_init(C.byref(self._c), dptr(self.position), dptr(self.forward), dptr(self.up),
                                      self.focal_length, self.sensor_diagonal, self.resolution_width,
                                      self.resolution_height)
        if rc == _abi.E_PARALLEL:
            raise ValueError("Forward and up vectors must not be parallel")
        if rc != 0:
            raise ValueError("invalid camera arguments (focal_length, sensor_diagonal, resolution must be > 0)")

    def update_position(self, new_position):  # src/cameras.rs:135-140
        self.position = _vec(new_position, 4)
        self._rebuild()

    def update_orientation(self, forward_world, up_world):  # src/cameras.rs:143-146
        self.forward = _vec(forward_world, 3)
        self.up = _vec(up_world, 3)
        self._rebuild()

    @property
    def rotation_matrix(self):
        return np.array(self._c.rot[:]).reshape(3, 3)

    def outward_vector_on_camera_space(self, camera_pixel_x, camera_pixel_y):
        """src/cameras.rs:150-164: unit vector through pixel (x, y) in camera space (x forward, y left, z up)"""
        out = np.zeros(3)
        check(lib().curvis_camera_outward_vector(C.byref(self._c), int(camera_pixel_x), int(camera_pixel_y), dptr(out), None))
        return out

    def outward_vector_on_world_space_from_x_y(self, camera_pixel_x, camera_pixel_y):
        """src/cameras.rs:169-172: the same vector in the tangent space of the camera's position"""
        out = np.zeros(3)
        check(lib().curvis_camera_outward_vector(C.byref(self._c), int(camera_pixel_x), int(camera_pixel_y), None, dptr(out)))
        return out

    @property
    def sensor_width(self):
        return self._c.sensor_w

    @property
    def sensor_height(self):
        return self._c.sensor_h


class SphericalImage:
    """Equirectangular background; `img` is HxWx4 (RGBA8) or HxWx3 (RGB8, alpha := 255 as
    DynamicImage::get_pixel does)."""

    def __init__(self, img, forward=None, up=None):
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] not in (3, 4):
            raise ValueError("img must be HxWx3 or HxWx4 uint8")
        if img.shape[2] == 3:
            img = np.concatenate([img, np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=2)
        self.rgba = np.ascontiguousarray(img)
        self.height_pixels, self.width_pixels = self.rgba.shape[:2]
        self.forward = _vec((1.0, 0.0, 0.0) if forward is None else forward, 3)
        self.up = _vec((0.0, 0.0, 1.0) if up is None else up, 3)

    def set_forward_up(self, forward, up):
        self.forward, self.up = _vec(forward, 3), _vec(up, 3)

    def get_pixel(self, x, y):
        """src/images.rs:107-111 (DynamicImage::get_pixel panics outside the image)"""
        if not (0 <= x < self.width_pixels and 0 <= y < self.height_pixels):
            raise IndexError("Image index (%d, %d) out of bounds (%d, %d)" % (x, y, self.width_pixels, self.height_pixels))
        return tuple(int(c) for c in self.rgba[y, x])

    def pixel_index_from_vector3(self, v):
        """texel (x, y) a world-space direction lands on (src/images.rs:115-142), in the kernels' arithmetic;
        IndexError where the reference's get_pixel would panic (x == width or y == height)"""
        from .algebra import Orientation
        inv = np.ascontiguousarray(Orientation(self.forward, self.up).inverse_rotation_matrix().reshape(9))
        vv = _vec(v, 3)
        x, y = C.c_uint32(0), C.c_uint32(0)
        rc = lib().curvis_sky_texel_index(self.width_pixels, self.height_pixels, dptr(inv), dptr(vv), C.byref(x), C.byref(y))
        if rc != 0:
            raise IndexError("Image index (%d, %d) out of bounds (%d, %d)" % (x.value, y.value, self.width_pixels, self.height_pixels))
        return x.value, y.value

    def get_pixel_from_vector3(self, v):
        """src/images.rs:171-174: Rgba of the texel the world-space direction v points at"""
        return self.get_pixel(*self.pixel_index_from_vector3(v))


class HostBuffer:
    """page-locked host memory (curvis_host_alloc) as a uint8 numpy array `.array`: the device-to-host copy of a render
    into it is one DMA transfer instead of a staged copy through pageable memory"""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().curvis_host_alloc(int(nbytes), C.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (int(nbytes),))

    def close(self):
        if self._p:
            self.array = None
            lib().curvis_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU: owns the HIP stream, the two sky textures in HBM and the device framebuffer."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().curvis_ctx_create(int(device), C.byref(self._h)))
        self.device = device
        self._sky_objs = [None, None]  # strong refs: identity check must not suffer id() reuse
        self._async_download = False
        self._async_streams = False
        self._dl_keep = None  # the array the last download went (or is still going) into: see _downloaded

    def close(self):
        if self._h:
            lib().curvis_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mhz = C.c_int(0), C.c_int(0)
        check(lib().curvis_ctx_device_info(self._h, name, 256, C.byref(cus), C.byref(mhz)), self._h)
        return {"name": name.value.decode(), "compute_units": cus.value, "clock_mhz": mhz.value}

    def device_status(self):
        """{"pci_bus_id", "sclk_mhz", "power_w"} of this context's GPU right now (-1 where sysfs does not tell)"""
        buf = C.create_string_buffer(64)
        sclk, power = C.c_int(-1), C.c_int(-1)
        check(lib().curvis_ctx_device_status(self._h, buf, 64, C.byref(sclk), C.byref(power)), self._h)
        return {"pci_bus_id": buf.value.decode(), "sclk_mhz": sclk.value, "power_w": power.value}

    def set_sky(self, which, image):
        check(lib().curvis_ctx_set_sky(self._h, which, image.rgba.ctypes.data, image.width_pixels,
                                       image.height_pixels), self._h)
        check(lib().curvis_ctx_set_sky_orientation(self._h, which, dptr(image.forward), dptr(image.up)), self._h)
        self._sky_objs[which] = image

    def set_sky_device(self, which, dev_ptr, w, h, copy=False, forward=(1.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
        check(lib().curvis_ctx_set_sky_device(self._h, which, C.c_void_p(dev_ptr), w, h, int(copy)), self._h)
        check(lib().curvis_ctx_set_sky_orientation(self._h, which, dptr(_vec(forward, 3)), dptr(_vec(up, 3))),
              self._h)
        self._sky_objs[which] = None

    LINK_TYPES = {0: "same device", 2: "PCIe", 4: "xGMI", -1: "unknown"}

    @staticmethod
    def device_link(device_a, device_b):
        """how two devices of this node are connected (curvis_device_link): {"link", "link_type", "hops", "peer_access",
        "performance_rank", "native_atomics"}; xGMI is point-to-point, 7 links x ~153 GB/s per MI355X"""
        v = [C.c_int(-1) for _ in range(5)]
        check(lib().curvis_device_link(int(device_a), int(device_b), *[C.byref(x) for x in v]))
        lt = v[0].value
        return {"link": Context.LINK_TYPES.get(lt, "HSA link type %d" % lt), "link_type": lt, "hops": v[1].value,
                "peer_access": v[2].value, "performance_rank": v[3].value, "native_atomics": v[4].value}

    @staticmethod
    def rccl_unique_id():
        """ncclGetUniqueId as bytes: rank 0 draws it and ships it to the other processes out of band"""
        buf = (C.c_uint8 * RCCL_ID_BYTES)()
        check(lib().curvis_rccl_unique_id(buf))
        return bytes(buf)

    def rccl_comm_init(self, unique_id, n_ranks, rank):
        """ncclCommInitRank on this context's GPU; returns the communicator handle (destroy with rccl_comm_destroy)"""
        if len(unique_id) != RCCL_ID_BYTES:
            raise ValueError("an RCCL unique id has %d bytes" % RCCL_ID_BYTES)
        buf = (C.c_uint8 * RCCL_ID_BYTES).from_buffer_copy(unique_id)
        comm = C.c_void_p()
        check(lib().curvis_ctx_rccl_comm_init(self._h, buf, int(n_ranks), int(rank), C.byref(comm)), self._h)
        return comm

    @staticmethod
    def rccl_comm_destroy(comm):
        check(lib().curvis_rccl_comm_destroy(comm))

    def bcast_skies(self, comm, root=0):
        """curvis_ctx_bcast_skies: both textures from rank `root` over the communicator (RCCL over xGMI); ranks other
        than the root need no skies beforehand"""
        check(lib().curvis_ctx_bcast_skies(self._h, comm, int(root)), self._h)
        self._sky_objs = [None, None]

    def read_sky(self, which, offset, nbytes):
        """bytes [offset, offset + nbytes) of sky texture `which` as it sits in HBM (RGBA8, row-major)"""
        out = np.empty(int(nbytes), dtype=np.uint8)
        check(lib().curvis_ctx_read_sky(self._h, int(which), int(offset), int(nbytes), out.ctypes.data), self._h)
        return out

    def deflate_frames(self, width, height, n_frames=1, out=None):
        """curvis_ctx_deflate_frames: the frames the last render call left in HBM -> [zlib stream of frame 0, ...] (bytes
        objects) and the HIP-event time of the kernels in ms.  `out`: a uint8 array to receive the streams (a HostBuffer's
        array: page-locked); default: a fresh array of the worst-case size."""
        if out is None:
            out = np.empty(int(n_frames) * ((height * (width * 3 + 1)) * 3 // 2 + 512), dtype=np.uint8)
        offs = (C.c_size_t * (int(n_frames) + 1))()
        ms = C.c_double(0.0)
        check(lib().curvis_ctx_deflate_frames(self._h, int(width), int(height), int(n_frames), out.ctypes.data, out.size, offs,
                                              C.byref(ms)), self._h)
        if self._async_streams:  # this wrapper hands out COPIES of the streams, which have to be there first
            self.download_wait()
        return [out[offs[k]:offs[k + 1]].tobytes() for k in range(int(n_frames))], ms.value

    def deflate_frames_into(self, width, height, n_frames, out):
        """curvis_ctx_deflate_frames_crc straight into the caller's uint8 array (a HostBuffer's: page-locked), nothing copied: returns
        (offsets [n_frames + 1], kernel ms, [crc, ...]).  With option "async_streams" = 1 the bytes may still be on their way when this
        returns -- offsets, trailers' values and CRCs are final --; they are there after download_wait()."""
        offs = (C.c_size_t * (int(n_frames) + 1))()
        ms, valid = C.c_double(0.0), C.c_int(0)
        crc = (C.c_uint32 * int(n_frames))()
        check(lib().curvis_ctx_deflate_frames_crc(self._h, int(width), int(height), int(n_frames), out.ctypes.data, out.size, offs,
                                                  C.byref(ms), crc, C.byref(valid)), self._h)
        self._dl_keep = out
        return list(offs), ms.value, list(crc)

    def deflate_frames_crc(self, width, height, n_frames=1, out=None):
        """curvis_ctx_deflate_frames_crc: as deflate_frames, plus the PNG chunk CRC-32 ("IDAT" + stream) of every frame computed
        on the device: (streams, kernel ms, [crc, ...] or None when the path taken does not compute it)"""
        if out is None:
            out = np.empty(int(n_frames) * ((height * (width * 3 + 1)) * 3 // 2 + 512), dtype=np.uint8)
        offs = (C.c_size_t * (int(n_frames) + 1))()
        ms, valid = C.c_double(0.0), C.c_int(0)
        crc = (C.c_uint32 * int(n_frames))()
        check(lib().curvis_ctx_deflate_frames_crc(self._h, int(width), int(height), int(n_frames), out.ctypes.data, out.size, offs,
                                                  C.byref(ms), crc, C.byref(valid)), self._h)
        if self._async_streams:  # (see deflate_frames)
            self.download_wait()
        return [out[offs[k]:offs[k + 1]].tobytes() for k in range(int(n_frames))], ms.value, (list(crc) if valid.value else None)

    def set_option(self, key, value):
        check(lib().curvis_ctx_set_option(self._h, key.encode(), int(value)), self._h)
        if key == "async_streams":
            self._async_streams = bool(int(value))
        if key == "async_download":
            self._async_download = bool(int(value))
            if not self._async_download:
                self._dl_keep = None  # (the library drains a pending copy when the option is switched off)

    def _downloaded(self, array, callers_buffer):
        """after a render call that was given a host buffer.  With option "async_download" = 1 the copy into `array` may still be
        in flight when the call returns: the context keeps a reference to it until the next download or download_wait -- a
        caller dropping the returned array must not leave the DMA writing into freed memory -- and when the array is one this
        wrapper allocated itself (pageable np.empty, handed straight back to the caller) the copy is waited for right here:
        overlapping it is only meaningful into a caller-owned page-locked buffer, and nobody expects to call download_wait
        for an array they were just handed."""
        if not self._async_download or array is None:
            return
        self._dl_keep = array
        if not callers_buffer:
            self.download_wait()

    def get_option(self, key):
        v = C.c_int64(0)
        check(lib().curvis_ctx_get_option(self._h, key.encode(), C.byref(v)), self._h)
        return v.value

    def framebuffer(self):
        p, n = C.c_void_p(), C.c_size_t(0)
        check(lib().curvis_ctx_framebuffer(self._h, C.byref(p), C.byref(n)), self._h)
        return p.value, n.value

    def upload_frames(self, rgb):
        """curvis_ctx_upload: RGB8 frames (n x H x W x 3 uint8) from host memory into the context's framebuffer"""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        check(lib().curvis_ctx_upload(self._h, rgb.ctypes.data, rgb.size), self._h)

    def download_wait(self):
        """curvis_ctx_download_wait: with option "async_download" = 1, the frames of the last render call that was given
        an output buffer are in host memory when this returns (they also are once the NEXT such call has returned)"""
        check(lib().curvis_ctx_download_wait(self._h), self._h)
        self._dl_keep = None

    def download_frames(self, width, height, n_frames=1):
        """curvis_ctx_download: the frames the last render call left in HBM as an n x H x W x 3 uint8 array"""
        rgb = np.empty((int(n_frames), int(height), int(width), 3), dtype=np.uint8)
        check(lib().curvis_ctx_download(self._h, rgb.ctypes.data, rgb.size), self._h)
        return rgb

    def render_brute(self, metric, cameras, max_iterations, max_radius, delta, download=True, debug=False, out=None):
        """cameras: one Camera or a list (one launch for the whole batch).  Returns (rgb, stats[, dbg]).
        out: a uint8 array of n*H*W*3 bytes to receive the frames (e.g. a HostBuffer: page-locked, one DMA transfer)."""
        single = isinstance(cameras, Camera)
        cams = [cameras] if single else list(cameras)
        n = len(cams)
        W, H = cams[0].resolution_width, cams[0].resolution_height
        arr = (CameraC * n)(*[c._c for c in cams])
        m = metric._c()
        st = Stats()
        if out is not None:
            if out.dtype != np.uint8 or out.size < n * H * W * 3 or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous uint8 array of at least n*H*W*3 bytes")
            rgb = out.reshape(-1)[:n * H * W * 3].reshape(n, H, W, 3)
            download = True
        else:
            rgb = np.empty((n, H, W, 3), dtype=np.uint8) if download else None
        callers_buffer = out is not None
        out = rgb.ctypes.data if download else None
        if debug:
            if n != 1:
                raise ValueError("debug dump is single-frame")
            dbg = np.zeros((H, W), dtype=_abi.RAY_DEBUG)
            check(lib().curvis_render_brute_debug(self._h, C.byref(m), arr, max_iterations, max_radius, delta, out,
                                                  dbg.ctypes.data, C.byref(st)), self._h)
            self._downloaded(rgb, callers_buffer)
            return (rgb[0] if download else None), st, dbg
        check(lib().curvis_render_brute_batch(self._h, C.byref(m), arr, n, max_iterations, max_radius, delta, out,
                                              C.byref(st)), self._h)
        self._downloaded(rgb, callers_buffer)
        if download and single:
            rgb = rgb[0]
        return rgb, st

    def render_brute_rows(self, metric, camera, row_begin, row_count, max_iterations, max_radius, delta, download=True):
        """rows [row_begin, row_begin + row_count) of render_image: (row_count x W x 3 uint8 or None, stats)."""
        W = camera.resolution_width
        m = metric._c()
        st = Stats()
        rgb = np.empty((row_count, W, 3), dtype=np.uint8) if download else None
        check(lib().curvis_render_brute_rows(self._h, C.byref(m), C.byref(camera._c), row_begin, row_count, max_iterations,
                                             max_radius, delta, rgb.ctypes.data if download else None, C.byref(st)),
              self._h)
        self._downloaded(rgb, False)
        return rgb, st

    def render_efficient(self, metric, cameras, max_iterations_propagation, max_radius, delta, alpha_nums,
                         max_iterations_sampling, thr1, thr2, download=True, out=None):
        """render_image_efficient for one Camera or a list (samplers advance in lock step).
        out: as in render_brute (a page-locked HostBuffer array to receive the frames)."""
        single = isinstance(cameras, Camera)
        cams = [cameras] if single else list(cameras)
        n = len(cams)
        W, H = cams[0].resolution_width, cams[0].resolution_height
        arr = (CameraC * n)(*[c._c for c in cams])
        m = metric._c()
        st = Stats()
        if out is not None:
            if out.dtype != np.uint8 or out.size < n * H * W * 3 or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous uint8 array of at least n*H*W*3 bytes")
            rgb, download = out.reshape(-1)[:n * H * W * 3].reshape(n, H, W, 3), True
        else:
            rgb = np.empty((n, H, W, 3), dtype=np.uint8) if download else None
        check(lib().curvis_render_efficient_batch(self._h, C.byref(m), arr, n, max_iterations_propagation, max_radius,
                                                  delta, alpha_nums, max_iterations_sampling, thr1, thr2,
                                                  rgb.ctypes.data if download else None, C.byref(st)), self._h)
        self._downloaded(rgb, out is not None)
        if download and single:
            rgb = rgb[0]
        return rgb, st

    def prefetch_efficient(self, metric, cameras, max_iterations_propagation, max_radius, delta, alpha_nums, max_iterations_sampling,
                           thr1, thr2):
        """curvis_ctx_prefetch_efficient: launch the sampler of a FUTURE render_efficient call (same metric, settings and camera
        radii) now, on a stream of its own; returns at once.  The render call then finds its tables ready."""
        cams = [cameras] if isinstance(cameras, Camera) else list(cameras)
        arr = (CameraC * len(cams))(*[c._c for c in cams])
        m = metric._c()
        check(lib().curvis_ctx_prefetch_efficient(self._h, C.byref(m), arr, len(cams), max_iterations_propagation, max_radius, delta,
                                                  alpha_nums, max_iterations_sampling, thr1, thr2), self._h)

    def render_direct(self, metric, camera, max_iterations, max_radius, delta, download=True, out=None):
        """"direct" mode (not in the reference): compute_escape_angle for the alpha of every pixel instead of sampling and
        interpolating (curvis_render_direct).  Returns (rgb or None, stats)."""
        W, H = camera.resolution_width, camera.resolution_height
        m = metric._c()
        st = Stats()
        if out is not None:
            if out.dtype != np.uint8 or out.size < H * W * 3 or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous uint8 array of at least H*W*3 bytes")
            rgb, download = out.reshape(-1)[:H * W * 3].reshape(H, W, 3), True
        else:
            rgb = np.empty((H, W, 3), dtype=np.uint8) if download else None
        check(lib().curvis_render_direct(self._h, C.byref(m), C.byref(camera._c), max_iterations, max_radius, delta,
                                         rgb.ctypes.data if download else None, C.byref(st)), self._h)
        self._downloaded(rgb, out is not None)
        return rgb, st

    def frame_stats(self, frame=None):
        """Statistics of one frame (or the list for all frames) of the last render call: the per-frame
        early-termination counters (rays, executed steps, escaped +l / -l, capped, out-of-range texels)."""
        def one(k):
            st = Stats()
            check(lib().curvis_ctx_frame_stats(self._h, k, C.byref(st)), self._h)
            return st
        if frame is not None:
            return one(int(frame))
        return [one(k) for k in range(self.get_option("last_frames"))]

    def sampling_info(self, frame=0):
        info = _abi.SamplingInfo()
        check(lib().curvis_ctx_sampling_info(self._h, frame, C.byref(info)), self._h)
        return info

    def samples(self, frame=0):
        n = self.sampling_info(frame).n_samples
        a, e, s = np.zeros(n), np.zeros(n), np.zeros(n)
        check(lib().curvis_ctx_samples(self._h, frame, dptr(a), dptr(e), dptr(s), n), self._h)
        return a, e, s

    def compute_escape_angles_range(self, metric, l, alphas, delta, max_iterations, max_radius):
        """src/systems.rs:265-281: returns (angle[n], space[n] in {+1,-1,0}, steps[n])."""
        alphas = np.ascontiguousarray(alphas, dtype=np.float64)
        n = alphas.size
        angle = np.zeros(n)
        space = np.zeros(n, dtype=np.int32)
        steps = np.zeros(n, dtype=np.uint32)
        m = metric._c()
        check(lib().curvis_compute_escape_angles(self._h, C.byref(m), float(l), dptr(alphas), n, delta, max_iterations,
                                                 max_radius, dptr(angle), space.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 steps.ctypes.data_as(C.POINTER(C.c_uint32))), self._h)
        return angle, space, steps

    def compute_photon_trajectory(self, metric, positions, directions, iterations, delta):
        """src/systems.rs:77-92 for a batch of photons built with new_photon(position, direction):
        returns an array [n, iterations, 8] = (x[4], p_cov[4]) before each step."""
        positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 4)
        directions = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        n = positions.shape[0]
        m = metric._c()
        x0, p0 = np.zeros((n, 4)), np.zeros((n, 4))
        for i in range(n):
            check(lib().curvis_new_photon(C.byref(m), dptr(positions[i]), dptr(directions[i]), dptr(x0[i]), dptr(p0[i])))
        out = np.zeros((n, iterations, 8))
        check(lib().curvis_photon_trajectories(self._h, C.byref(m), n, dptr(x0), dptr(p0), iterations, delta, dptr(out)),
              self._h)
        return out

    def selftest_math(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty_like(a)
        bb = None
        if b is not None:
            bb = np.ascontiguousarray(b, dtype=np.float64)
        check(lib().curvis_selftest_math(self._h, op, dptr(a), dptr(bb) if bb is not None else None, dptr(out),
                                         a.size), self._h)
        return out

    def selftest_math3(self, op, a, b=None, c=None):
        """primitives of the shared-reciprocal Euler step on three inputs (include/curvis_hip.h: op 0 div_with_recip(n, d, y),
        1 / 2 sqrt_and_rsqrt's root / reciprocal root, 3 its last residual step (x, g, y), 4 recip_refined, 5 cv_div_nr,
        6 recip_newton(d, y))"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
        cc = np.ascontiguousarray(c, dtype=np.float64) if c is not None else None
        out = np.empty_like(a)
        check(lib().curvis_selftest_math3(self._h, op, dptr(a), dptr(bb) if bb is not None else None,
                                          dptr(cc) if cc is not None else None, dptr(out), a.size), self._h)
        return out

    FAST_STEP_QUOTIENTS = ("r' = l/r", "1/r^2", "p_phi^2/sin^2", "dp_l = b^2 r'/r^3", "cos/(r^2 sin^3)", "1/(r^2 sin^2)")

    def selftest_fast_step(self, metric, states, delta=0.05, max_radius=100.0):
        """ONE fast Euler step per state (n x {l, theta, p_l, p_theta, p_phi}) with every quotient recorded: returns
        (quot [n, 6, 7] = numerator n, denominator d, shared reciprocal y, the step's quotient, the IEEE quotient, the
        remainder n - d RN(n y), 1 - d y -- NaN rows where a quotient is not formed --, fast [n, 5], strict [n, 5] = the new (l, theta, phi - phi0, p_l, p_theta),
        took_fast [n] bool)"""
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 5)
        n = st.shape[0]
        out = np.zeros((n, 53))
        m = metric._c()
        check(lib().curvis_selftest_fast_step(self._h, C.byref(m), float(delta), float(max_radius), dptr(st), n, dptr(out)),
              self._h)
        return out[:, :42].reshape(n, 6, 7), out[:, 42:47], out[:, 47:52], out[:, 52] != 0.0


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class EscapeAngle:
    """enum EscapeAngle (src/systems.rs:54-58): PositiveSpace(angle) / NegativeSpace(angle) / NotEscaped"""

    __slots__ = ("kind", "angle")

    def __init__(self, kind, angle=None):
        if kind not in ("PositiveSpace", "NegativeSpace", "NotEscaped"):
            raise ValueError("unknown EscapeAngle variant")
        self.kind, self.angle = kind, (None if kind == "NotEscaped" else float(angle))

    def __repr__(self):
        return "EscapeAngle::NotEscaped" if self.angle is None else "EscapeAngle::%s(%r)" % (self.kind, self.angle)

    def __eq__(self, other):
        return isinstance(other, EscapeAngle) and (self.kind, self.angle) == (other.kind, other.angle)


def compute_escape_angle(metric, l, alpha, delta, max_iterations, max_radius, context=None):
    """compute_escape_angle (src/systems.rs:203-261, re-exported by src/lib.rs:37), same argument order: the photon
    at (0, l, pi/2, 0) leaving at angle alpha from the radial direction is integrated on the GPU until it escapes."""
    ctx = context or default_context()
    angle, space, _ = ctx.compute_escape_angles_range(metric, l, [alpha], delta, max_iterations, max_radius)
    if space[0] > 0:
        return EscapeAngle("PositiveSpace", angle[0])
    if space[0] < 0:
        return EscapeAngle("NegativeSpace", angle[0])
    return EscapeAngle("NotEscaped")


def compute_photon_trajectory(photon, metric, iterations, delta, context=None):
    """compute_photon_trajectory (src/systems.rs:77-92, re-exported by src/lib.rs:37), same argument order: the
    states of `photon` BEFORE each of `iterations` Euler steps as a list of RelativisticObjects; like the reference
    (`&mut`), `photon` itself ends up advanced by `iterations` steps.  A contravariant momentum is lowered first, as
    update_relativistic_object does (src/metrics.rs:283-288)."""
    if photon.covariance_x() != Covariance.Contravariant:
        raise CovarianceError("The position vector must be contravariant.")
    if photon.covariance_p() == Covariance.Contravariant:
        photon.momentum = metric.to_covariant(photon.position, photon.momentum)
    iterations = int(iterations)
    if iterations <= 0:
        return []
    ctx = context or default_context()
    m = metric._c()
    x0 = np.ascontiguousarray(photon.position.vector, dtype=np.float64).reshape(1, 4)
    p0 = np.ascontiguousarray(photon.momentum.vector, dtype=np.float64).reshape(1, 4)
    out = np.zeros((1, iterations + 1, 8))  # one more row: the state the reference leaves in `photon`
    check(lib().curvis_photon_trajectories(ctx._h, C.byref(m), 1, dptr(x0), dptr(p0), iterations + 1, float(delta), dptr(out)), ctx._h)
    states = [RelativisticObject(RelativisticVector(row[:4], Covariance.Contravariant),
                                 RelativisticVector(row[4:], Covariance.Covariant)) for row in out[0]]
    photon.position, photon.momentum = states[-1].position, states[-1].momentum
    return states[:-1]


class RelativisticSystem:
    """RelativisticSystem<M> (src/systems.rs:68-73)."""

    def __init__(self, metric, background_positive, background_negative, camera, context=None):
        self.metric = metric
        self.background_positive = background_positive
        self.background_negative = background_negative
        self.camera = camera
        self.context = context if context is not None else default_context()
        self.last_stats = None

    def _bind_skies(self):
        ctx = self.context
        if ctx._sky_objs[0] is not self.background_positive:
            ctx.set_sky(0, self.background_positive)
        if ctx._sky_objs[1] is not self.background_negative:
            ctx.set_sky(1, self.background_negative)

    def render_image(self, max_iterations, max_radius, delta):
        """The per-pixel renderer; returns an HxWx3 uint8 array (DynamicImage::ImageRgb8)."""
        self._bind_skies()
        rgb, st = self.context.render_brute(self.metric, self.camera, max_iterations, max_radius, delta)
        self.last_stats = st
        return rgb

    def render_image_efficient(self, max_iterations_propagation, max_radius, delta, alpha_nums,
                               max_iterations_sampling, sampling_convergence_threshold_1,
                               sampling_convergence_threshold_2):
        """src/systems.rs:333-343: the renderer behind `curvis image` / `curvis video`."""
        self._bind_skies()
        rgb, st = self.context.render_efficient(self.metric, self.camera, max_iterations_propagation, max_radius, delta,
                                                alpha_nums, max_iterations_sampling, sampling_convergence_threshold_1,
                                                sampling_convergence_threshold_2)
        self.last_stats = st
        return rgb

    def render_image_direct(self, max_iterations_propagation, max_radius, delta):
        """NOT in the reference: the image render_image_efficient approximates, with compute_escape_angle evaluated
        for every pixel instead of sampled and interpolated (a quality option; Context.render_direct)."""
        self._bind_skies()
        rgb, st = self.context.render_direct(self.metric, self.camera, max_iterations_propagation, max_radius, delta)
        self.last_stats = st
        return rgb

    def render_image_debug(self, max_iterations, max_radius, delta):
        self._bind_skies()
        rgb, st, dbg = self.context.render_brute(self.metric, self.camera, max_iterations, max_radius, delta,
                                                 debug=True)
        self.last_stats = st
        return rgb, dbg
