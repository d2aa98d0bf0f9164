import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # test infrastructure (oracle + host twin) is (re)built on demand; both are plain gcc builds
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "host_twin")], check=True)


@pytest.fixture(scope="session")
def gpu_ctx():
    import curvis_amd
    ctx = curvis_amd.Context(0)  # raises loudly if the HIP library or the GPU is missing
    yield ctx
    ctx.close()
