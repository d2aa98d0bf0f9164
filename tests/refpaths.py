"""The REFERENCE's own camera paths, as data fixtures (tests/golden/paths/*.csv.gz).

`paths/path_orbit.csv` and `paths/path_through.csv` are the inputs of BASELINE configs[3] and [4].  They are data the
reference holds, not source: 1000 rows of `t,l,theta,phi,fx,fy,fz,upx,upy,upz` each, CRLF line ends, committed here
byte for byte (gzip, mtime 0) so that every video-config test and bench.py's `video_e2e` run on the reference's BYTES --
not on a regeneration whose last bit depends on the numpy build of the box (curvis_amd/paths.py reproduces the orbit file
exactly and the fly-through file up to 29 rows x <= 7.8e-16: tests/test_paths.py).  No dependency beyond the standard
library, so bench.py can import it too."""
import gzip
import hashlib
import os

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "paths")
SHA256 = {  # of the reference's files (sha256sum /root/reference/paths/*.csv in the build container)
    "path_orbit.csv": "fe3872182c0743643b358bbdc0128145a862a7e54db3bbcf23823ffefec3b5ed",
    "path_through.csv": "747b87a2179125188d3cae3f79cace8571aaaa76ebf3b1f27f18f6a9d0e9ac3d",
}


def reference_path_bytes(name):
    with gzip.open(os.path.join(_DIR, name + ".gz"), "rb") as f:
        data = f.read()
    if hashlib.sha256(data).hexdigest() != SHA256[name]:
        raise RuntimeError("tests/golden/paths/%s.gz does not hold the reference's bytes" % name)
    return data


def reference_path_file(name):
    """file name of an unpacked copy ("path_orbit.csv" / "path_through.csv"); unpacked once, checked every time"""
    dst = os.path.join(_DIR, "_unpacked", name)
    data = reference_path_bytes(name)
    try:
        with open(dst, "rb") as f:
            if f.read() == data:
                return dst
    except OSError:
        pass
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    tmp = "%s.tmp.%d" % (dst, os.getpid())
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, dst)
    return dst
