"""Closed-form camera paths for the video configurations (BASELINE.json configs 4 and 5).

The reference ships two CSV camera paths produced by small numpy scripts
(paths/generate_path_orbit.py:4-33, paths/generate_path_through.py:3-53).  These are this
project's own generators for the same closed forms; they produce the BUNDLED default paths
(`paths/path_through.csv` of a default video_settings.toml resolves to them).  Against the
reference's files (tests/test_paths.py): path_orbit.csv is reproduced byte for byte;
path_through.csv goes through np.exp / arctan / cos / sin, whose last bit depends on the numpy
build, and is reproduced up to the forward vector of 29 of its 1000 rows (<= 7.8e-16, measured
in the build container).  Everything that claims parity on configs[3] / [4] -- the tests and
bench.py's video legs -- therefore reads the reference's own bytes, committed as data fixtures
(tests/golden/paths/*.csv.gz, tests/refpaths.py), not these; a user who wants the reference's
poses passes the reference's CSV, like any other camera path.
CSV format (src/csv.rs:24-62): one header line, then `t,l,theta,phi,fx,fy,fz,upx,upy,upz`.
"""
import os

import numpy as np

HEADER = ",".join(["t", "l", "theta", "phi", "fx", "fy", "fz", "upx", "upy", "upz"])
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "paths")


def _fmt(rows):
    return "\n".join([HEADER] + [",".join(str(float(x)) for x in r) for r in rows])


def orbit_rows(l0=3.0, T=60.0, n=1000):
    """circular orbit at l = l0 in the equatorial plane, looking at the wormhole."""
    ts = np.linspace(0, T, n)
    return [(t, l0, np.pi / 2, 2 * np.pi * t / T, -1, 0, 0, 0, 0, 1) for t in ts]


def through_rows(l0=-4.0, l1=4.0, T=20.0, b0=3.0, n=1000):
    """fly through the throat from l0 to l1; forward rotates by the impact-parameter law b(l)."""
    ls = np.linspace(l0, l1, n)
    ts = np.linspace(0, T, n)
    b = b0 * np.exp(-10 * (ls / l0) ** 2)
    alpha = np.pi - np.arctan(b / ls)
    fx = np.sign(ls) * np.cos(alpha)
    fy = np.sign(ls) * np.sin(alpha)
    return [(ts[i], ls[i], np.pi / 2, 0, fx[i], fy[i], 0, 0, 0, 1) for i in range(n)]


# The reference's shipped files use CRLF line ends (BufRead::lines strips both "\n" and "\r\n").
def write_orbit(path, **kw):
    with open(path, "w", newline="\r\n") as f:
        f.write(_fmt(orbit_rows(**kw)))


def write_through(path, **kw):
    with open(path, "w", newline="\r\n") as f:
        f.write(_fmt(through_rows(**kw)))


def load_path(path):
    """src/csv.rs:24-62 load_path: returns (positions[n,4], forward[n,3], up[n,3])."""
    with open(path, newline="") as f:
        lines = [ln[:-1] if ln.endswith("\r") else ln for ln in f.read().split("\n")]
    rows = []
    for i, line in enumerate(lines):
        if i == 0 or (line == "" and i == len(lines) - 1):
            continue
        vals = [float(x) for x in line.split(",")]
        if len(vals) < 10:
            raise ValueError("Could not read all 10 columns at line %d" % (i + 1))
        rows.append(vals[:10])
    a = np.array(rows, dtype=np.float64)
    return a[:, 0:4].copy(), a[:, 4:7].copy(), a[:, 7:10].copy()


def ensure_paths():
    """Generate the two camera paths into DATA_DIR if they are not there yet (they are build products of
    this module, not checked in) and return their file names."""
    os.makedirs(DATA_DIR, exist_ok=True)
    out = {}
    for name, writer in (("path_orbit.csv", write_orbit), ("path_through.csv", write_through)):
        dst = os.path.join(DATA_DIR, name)
        if not os.path.exists(dst):
            tmp = dst + ".tmp.%d" % os.getpid()
            writer(tmp)
            os.replace(tmp, dst)
        out[name] = dst
    return out


def path_file(name):
    """absolute file name of a generated camera path ("path_orbit.csv" / "path_through.csv")."""
    return ensure_paths()[name]


if __name__ == "__main__":
    print(ensure_paths())
