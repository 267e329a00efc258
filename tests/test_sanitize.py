"""The host half of the library under AddressSanitizer + UndefinedBehaviorSanitizer (tools/san_host.cpp): every built-in
scene lowered and packed, corrupted tables refused, scenes built from six threads at once, image output and input.
The reference has no sanitizer runs; GPU sanitizers are not available on the pool, so this covers
the C++ that runs before a kernel is launched."""
import os
import subprocess

from conftest import ROOT


def test_host_code_is_clean_under_asan_and_ubsan():
    pkg = os.path.join(ROOT, "cuda-raytracer_amd")
    r = subprocess.run(["make", "-C", pkg, "san"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "san_host ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr
