"""The oracle's known-answer, raster-rule, sky and golden tests once more through its sanitizer build
(oracle/Makefile: liboracle_asan.so, -fsanitize=address,undefined), in a child interpreter with the sanitizer runtimes
preloaded: test infrastructure that reads out of bounds or hits undefined behaviour would pin nothing.  CPU only: GPU
sanitizers are not available on the pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_oracle_kat.py", "tests/test_oracle_raster.py", "tests/test_oracle_sky.py", "tests/test_golden.py"]


def runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_suites_are_clean_under_asan_and_ubsan():
    asan, ubsan = runtime("libasan.so"), runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               ARCTIC_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    report = out.stdout + out.stderr
    assert out.returncode == 0, report[-4000:]
    assert "AddressSanitizer" not in report and "runtime error" not in report, report[-4000:]
    assert " passed" in report
