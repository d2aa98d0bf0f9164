"""Orientation (src/algebra.rs:8-62, re-exported by src/lib.rs): forward / up of an object in world space and the
rotation that takes its own frame (x forward, z up) to the world, through curvis_orientation_init -- the host code
that builds the matrix the kernels' cameras and skies use."""
import numpy as np

from ._abi import check, dptr, lib


class Orientation:
    def __init__(self, forward, up):
        f = np.ascontiguousarray(forward, dtype=np.float64).reshape(3)
        u = np.ascontiguousarray(up, dtype=np.float64).reshape(3)
        rot, inv, up_out = np.zeros(9), np.zeros(9), np.zeros(3)
        check(lib().curvis_orientation_init(dptr(f), dptr(u), dptr(rot), dptr(inv), dptr(up_out)))
        self._forward = f.copy()         # kept as given, like the reference's field
        self._up = up_out                # rotation * z: orthogonal to forward (src/algebra.rs:30)
        self._rot, self._inv = rot.reshape(3, 3), inv.reshape(3, 3)

    def forward(self):
        return self._forward.copy()

    def up(self):
        return self._up.copy()

    def rotation_matrix(self):
        """object frame -> world (row-major 3x3)"""
        return self._rot.copy()

    def inverse_rotation_matrix(self):
        return self._inv.copy()

    @staticmethod
    def _gemv(m, v):  # nalgebra's order: ((m_i0 x0) + m_i1 x1) + m_i2 x2
        v = np.asarray(v, dtype=np.float64).reshape(3)
        return np.array([(m[i, 0] * v[0] + m[i, 1] * v[1]) + m[i, 2] * v[2] for i in range(3)])

    def to_world(self, v_object):
        return self._gemv(self._rot, v_object)

    def to_object(self, v_world):
        return self._gemv(self._inv, v_world)
