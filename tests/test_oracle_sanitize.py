"""The oracle's C code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU
build only; GPU sanitizers are not available on the pool)."""
import os
import subprocess

from conftest import ROOT


def test_oracle_selftest_under_asan_ubsan():
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, "selftest_asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(d, "selftest_asan")], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "oracle selftest: ok" in out.stdout
