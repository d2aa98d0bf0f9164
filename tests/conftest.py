import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # product (HIP library + curvis binary; hipcc cross-compiles without a GPU) and test infrastructure
    # (oracle + host twin, plain gcc) are (re)built on demand -- no-ops when up to date
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "curvis_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "host_twin")], check=True)


@pytest.fixture(scope="session")
def gpu_ctx():
    import curvis_amd
    ctx = curvis_amd.Context(0)  # raises loudly if the HIP library or the GPU is missing
    yield ctx
    ctx.close()


def pytest_sessionstart(session):
    """what the oracle-bound tests will get on this host: printed even with -q (the GPU suite's wall time is oracle time)"""
    tr = session.config.pluginmanager.get_plugin("terminalreporter")
    try:
        import common
        quota, usable, logical = common.host_cpus()
        line = "host CPUs: cgroup quota %s, %d in the affinity mask, %d logical" % ("%.2f" % quota if quota else "none", usable, logical)
    except Exception as exc:  # never fatal: a report line
        line = "host CPUs: unknown (%s)" % exc
    if tr is not None:
        tr.write_line(line)


def pytest_collection_finish(session):
    """the heaviest oracle renders of the selected GPU tests start NOW, on a background thread (tests/common.py oracle_prefetch):
    test modules that have such renders name them in ORACLE_PREFETCH(selected node ids)"""
    names = [item.nodeid for item in session.items]
    if not any("test_gpu_" in n for n in names) or session.config.option.collectonly:
        return
    thunks = []
    for mod in sorted({item.module for item in session.items if hasattr(item, "module")}, key=lambda m: m.__name__):   # the order the files run in
        make = getattr(mod, "ORACLE_PREFETCH", None)
        if make is not None:
            try:
                thunks += list(make(names))
            except Exception as exc:  # noqa: BLE001 -- prefetching is an optimisation, never a reason to fail collection
                print("[oracle prefetch] %s: %s" % (getattr(mod, "__name__", mod), exc))
    if thunks:
        import common
        common.oracle_prefetch(thunks)
