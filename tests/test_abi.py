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


def c_layout():
    """{struct: {"size": n, "align": a, "fields": [(name, offset, size)]}} from the C header, via tests/abi/abi_layout.c"""
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="curvis_abi_"), "abi_layout")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "abi", "abi_layout.c")], check=True)
    out = {}
    for line in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines():
        st, f, off, size = line.split()
        d = out.setdefault(st, {"fields": []})
        if f == ".":
            d["align"], d["size"] = int(off), int(size)
        else:
            d["fields"].append((f, int(off), int(size)))
    return out


RUST_NAMES = {"curvis_metric": "CurvisMetric", "curvis_camera": "CurvisCamera", "curvis_ray_debug": "CurvisRayDebug",
              "curvis_stats": "CurvisStats", "curvis_sampling_info": "CurvisSamplingInfo"}
CTYPES = {"curvis_metric": _abi.Metric, "curvis_camera": _abi.CameraC, "curvis_stats": _abi.Stats,
          "curvis_sampling_info": _abi.SamplingInfo}


def rust_block():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.search(r"```rust\n(.*?extern \"C\" \{.*?\n\}\n)```", txt, flags=re.S).group(1)


def rust_layout(body):
    """#[repr(C)] layout rules (= the C ABI's) applied to `name: type` fields of a struct body"""
    prim = {"i32": 4, "u32": 4, "f64": 8, "u64": 8, "i64": 8, "u8": 1}
    off, align, fields = 0, 1, []
    for name, ty in re.findall(r"(?:pub\s+)?(\w+)\s*:\s*(\[[^\]]+\]|\w+)", body):
        m = re.match(r"\[(\w+);\s*(\d+)\]", ty)
        a = prim[m.group(1)] if m else prim[ty]
        size = a * int(m.group(2)) if m else a
        off = (off + a - 1) // a * a
        fields.append((name, off, size))
        off += size
        align = max(align, a)
    return {"size": (off + align - 1) // align * align, "align": align, "fields": fields}


def test_struct_offsets_c_header_rust_stub_ctypes_and_table_agree():
    """offset-level ABI check: C header (compiled) == INTEGRATION.md's #[repr(C)] structs == its layout table ==
    curvis_amd/_abi.py's ctypes structures == the numpy record dtypes of the debug dump"""
    lay = c_layout()
    assert lay.pop("CURVIS_ABI_VERSION")["align"] == 1          # printed as "CURVIS_ABI_VERSION . 1 0"
    assert set(lay) == set(RUST_NAMES)
    rust = rust_block()
    table = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = table[table.index("<!-- abi-layout:begin -->"):table.index("<!-- abi-layout:end -->")]
    for cname, L in lay.items():
        body = re.search(r"pub struct %s \{(.*?)\}" % RUST_NAMES[cname], rust, flags=re.S).group(1)
        assert rust_layout(body) == L, (cname, rust_layout(body), L)
        row = [ln for ln in table.splitlines() if ln.startswith("| `%s` |" % cname)][0].split("|")
        assert row[2].strip() == "`%s`" % RUST_NAMES[cname] and int(row[3]) == L["size"]
        got = [(n, int(o), int(z)) for n, o, z in re.findall(r"(\w+) @(\d+) \((\d+)\)", row[4])]
        assert got == L["fields"], (cname, got)
        if cname in CTYPES:
            T = CTYPES[cname]
            assert C.sizeof(T) == L["size"] and C.alignment(T) == L["align"]
            assert [(n, getattr(T, n).offset, getattr(T, n).size) for n, _ in T._fields_] == L["fields"]
    for dt in (_abi.RAY_DEBUG, O.RAY_DEBUG):
        assert dt.itemsize == lay["curvis_ray_debug"]["size"]
        assert [(n, dt.fields[n][1], dt.fields[n][0].itemsize) for n in dt.names] == lay["curvis_ray_debug"]["fields"]


def test_rust_extern_block_declares_every_function_with_the_header_arity():
    hdr = open(os.path.join(ROOT, "include", "curvis_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rust = rust_block()

    def arity(params):
        params = params.strip()
        return 0 if params in ("", "void") else params.count(",") + 1
    c_fns = {n: arity(p) for n, p in re.findall(r"\b(curvis_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr)}
    r_fns = {n: arity(p) for n, p in re.findall(r"pub fn (curvis_[a-z_0-9]+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", rust)}
    assert set(c_fns) == set(declared_symbols())
    assert r_fns == c_fns, {k: (c_fns.get(k), r_fns.get(k)) for k in set(c_fns) | set(r_fns) if c_fns.get(k) != r_fns.get(k)}


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


def test_metric_trait_functions_match_oracle_bitwise():
    """r, r_squared, r_derivative (the required methods of DiagonalSphericalMetric, src/metrics.rs:40-48) through
    the ABI == oracle(cv flavour), bit for bit, incl. the Interstellar throat (|l| <= a) and negative l"""
    rng = np.random.default_rng(8)
    ls = np.concatenate([rng.uniform(-120, 120, 400), [0.0, -0.0, 1e-4, -1e-4, 5e-5, 0.31, -0.31, 5.0, -5.0, 100.0]])
    cases = [(O.ellis(1.0), curvis_amd.EllisMetric(1.0)), (O.ellis(0.37), curvis_amd.EllisMetric(0.37)),
             (O.interstellar(0.1, 1e-4, 1.0), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)),
             (O.interstellar(0.5, 0.2, 2.0), curvis_amd.InterstellarMetric(0.5, 0.2, 2.0)), (O.flat(), curvis_amd.FlatSphericalMetric())]
    for om, pm in cases:
        for l in ls:
            want = (O.lib().cvo_metric_r(O.CV, C.byref(om), l), O.lib().cvo_metric_r_squared(O.CV, C.byref(om), l),
                    O.lib().cvo_metric_r_derivative(O.CV, C.byref(om), l))
            got = (pm.r(l), pm.r_squared(l), pm.r_derivative(l))
            assert np.array_equal(np.array(got).view(np.uint64), np.array(want).view(np.uint64)), (om.kind, l, got, want)
    # KAT-1 of SURVEY.md 8c (glibc values; cv_math agrees on these arguments to the last bit or one ulp)
    assert abs(curvis_amd.EllisMetric(1.0).r(5.0) - 5.0990195135927845) < 1e-15
    assert abs(curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0).r(5.0) - 5.5538415248760264) < 2e-15
    assert curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0).r(5e-5) == 1.0
