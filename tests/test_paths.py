"""Camera paths of BASELINE configs[3] / [4]: the committed fixtures ARE the reference's files, and the product's own
closed-form generator (curvis_amd/paths.py) is cross-checked against them."""
import os

import numpy as np
import pytest

import refpaths
from curvis_amd import paths

REF = "/root/reference/paths"
NAMES = ("path_orbit.csv", "path_through.csv")


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("name", NAMES)
def test_fixture_is_the_reference_file(name):
    assert refpaths.reference_path_bytes(name) == open(os.path.join(REF, name), "rb").read()


@pytest.mark.parametrize("name", NAMES)
def test_fixture_unpacks_to_the_recorded_bytes(name):
    """runs everywhere (the GPU box has no /root/reference): sha256 of the reference's file, CRLF line ends, 1000 rows x 10"""
    data = open(refpaths.reference_path_file(name), "rb").read()
    assert data == refpaths.reference_path_bytes(name)      # reference_path_bytes checks the recorded sha256
    assert data.count(b"\r\n") == 1000 and not data.endswith(b"\n")
    pos, fwd, up = paths.load_path(refpaths.reference_path_file(name))
    assert pos.shape == (1000, 4) and fwd.shape == (1000, 3) and up.shape == (1000, 3)


def test_generator_against_the_reference_files(tmp_path):
    """The generator is a cross-check and the source of the product's BUNDLED default paths, not the tests' input.
    Orbit: byte-identical.  Fly-through: goes through np.exp / arctan / cos / sin, whose last bit depends on the numpy build
    (SIMD loops) -- measured in the build container (numpy 2.2): 29 of the 1000 rows differ from the reference's file, in the
    forward vector only, by at most 7.8e-16; positions and up vectors are identical."""
    o, t = tmp_path / "o.csv", tmp_path / "t.csv"
    paths.write_orbit(o)
    paths.write_through(t)
    assert o.read_bytes() == refpaths.reference_path_bytes("path_orbit.csv")
    mine = paths.load_path(t)
    ref = paths.load_path(refpaths.reference_path_file("path_through.csv"))
    for k in (0, 2):
        assert np.array_equal(mine[k], ref[k])
    differing = int(np.sum(np.any(mine[1] != ref[1], axis=1)))
    print("path_through.csv: %d of 1000 generated rows differ from the reference's file, max |delta| %.3g" % (
        differing, float(np.max(np.abs(mine[1] - ref[1])))))
    assert differing <= 64 and np.max(np.abs(mine[1] - ref[1])) < 2e-15


def test_bundled_paths_match_generators(tmp_path):
    o, t = tmp_path / "o.csv", tmp_path / "t.csv"
    paths.write_orbit(o)
    paths.write_through(t)
    assert o.read_bytes() == open(paths.path_file("path_orbit.csv"), "rb").read()
    assert t.read_bytes() == open(paths.path_file("path_through.csv"), "rb").read()
