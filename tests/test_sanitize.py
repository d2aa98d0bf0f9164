"""ASan + UBSan pass over the native test infrastructure (SURVEY.md section 5 "race detection / sanitizers": the
reference is safe Rust; the restatements here are C and C++): oracle/curvis_oracle.c and tests/host_twin/twin.cpp
-- i.e. the product's per-ray headers cv_device.h / cv_math.h / cv_efficient.h / cv_sampler.h compiled for x86 --
are built with -fsanitize=address,undefined and driven through both renderers, both flavours, all metrics and
adversarial cameras by tests/sanitize/san_driver.cpp, which also cross-checks oracle(cv) == twin."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "sanitize")


def test_oracle_and_host_twin_are_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", D], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(D, "san_driver")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "sanitize ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
