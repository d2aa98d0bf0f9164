"""The C-ABI library loads and exports every symbol include/curvis_hip.h declares; host-only entry
points (camera/orientation/metric validation) agree with the oracle.  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import curvis_amd
from curvis_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "curvis_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(curvis_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    L = _abi.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libcurvis_hip.so does not export %s" % n
        assert n in _abi.SYMBOLS, "%s is not bound in curvis_amd/_abi.py" % n
    assert b"gfx950" in L.curvis_version()


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.Metric) == 32
    assert C.sizeof(_abi.CameraC) == 4 * 8 + 9 * 8 + 3 * 8 + 8
    assert C.sizeof(_abi.Stats) == 80
    assert _abi.RAY_DEBUG.itemsize == 80 and O.RAY_DEBUG.itemsize == 80


@pytest.mark.skipif(_abi.lib().curvis_device_count() > 0, reason="a GPU is present")
def test_no_gpu_fails_loudly():
    with pytest.raises(curvis_amd.CurvisError) as e:
        curvis_amd.Context(0)
    assert e.value.code == _abi.E_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_camera_init_matches_oracle_bitwise():
    rng = np.random.default_rng(3)
    cases = [((0, 5, np.pi / 2, 0), (-1, 0, 0), (0, 0, 1), 15.0, 43.0, (960, 540)),
             ((0, 3, np.pi / 2, 1.0), (-1, 0.2, 0.1), (0.1, 0, 1), 20.0, 43.0, (1920, 1080)),
             ((0, -4, 1.0, 2.0), (1, 1, 0), (-1, -1, 1), 35.0, 30.0, (17, 13))]
    for _ in range(50):
        cases.append(((0, rng.uniform(-9, 9), rng.uniform(0.1, 3), rng.uniform(0, 6)), tuple(rng.uniform(-1, 1, 3)),
                      tuple(rng.uniform(-1, 1, 3)), rng.uniform(5, 50), rng.uniform(10, 60),
                      (int(rng.integers(1, 4000)), int(rng.integers(1, 3000)))))
    for pos, fwd, up, focal, diag, res in cases:
        oc = O.camera(pos, fwd, up, focal, diag, res)
        pc = curvis_amd.Camera(pos, fwd, up, focal, diag, res[0], res[1])
        assert np.array_equal(np.array(oc.rot[:]).view(np.uint64), np.array(pc._c.rot[:]).view(np.uint64))
        assert oc.sensor_w == pc._c.sensor_w and oc.sensor_h == pc._c.sensor_h and oc.focal == pc._c.focal
        assert list(oc.pos) == list(pc._c.pos)


def test_constructor_errors_mirror_reference_panics():
    with pytest.raises(ValueError):
        curvis_amd.Camera((0, 5, 1.5, 0), (1, 0, 0), (2, 0, 0), 15, 43, 64, 36)  # parallel (algebra.rs:19)
    with pytest.raises(ValueError):
        curvis_amd.Camera((0, 5, 1.5, 0), (1, 0, 0), (0, 0, 1), 0.0, 43, 64, 36)  # focal (cameras.rs:92)
    with pytest.raises(ValueError):
        curvis_amd.Camera((0, 5, 1.5, 0), (1, 0, 0), (0, 0, 1), 15, 43, 0, 36)  # resolution (cameras.rs:98)
    with pytest.raises(ValueError):
        curvis_amd.EllisMetric(0.0)
    with pytest.raises(ValueError):
        curvis_amd.InterstellarMetric(0.1, -1.0, 1.0)
    m = _abi.Metric(_abi.METRIC_INTERSTELLAR, 0, 1.0, 0.0, 1e-4)
    assert _abi.lib().curvis_metric_validate(C.byref(m)) == _abi.E_METRIC
