"""The C oracle under AddressSanitizer + UBSan (SURVEY.md section 5 'race detection / sanitizers' row): a C driver
calls every entry point on seeded random inputs with exact-size buffers and cross-checks the bit-vector search
against the plain DP.  CPU only."""
import os
import shutil
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_is_clean_under_asan_and_ubsan():
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    subprocess.check_call(["make", "-s", "-C", ORACLE, "asan_driver"])
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)  # ASan's runtime is linked statically; a foreign preload in front of it only confuses its checks
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    r = subprocess.run([os.path.join(ORACLE, "asan_driver")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asan_driver ok" in r.stdout
