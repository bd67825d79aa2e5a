"""The checker's own C code under sanitizers (CPU only; GPU AddressSanitizer is not available on the pool): the oracle's
restatement of /root/reference/src/correlation.py:9-104, 106-285 and src/models.py:20-35 (oracle/corr_oracle.c) is compiled
together with tests/csrc/oracle_sanitize_driver.c with -fsanitize=address,undefined and run on the edge shapes of the GPU
tests.  An oracle that reads or writes out of bounds would pin the product to garbage; this keeps the test infrastructure honest."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_c_runs_clean_under_asan_and_ubsan(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not found")
    exe = str(tmp_path / "oracle_san")
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(ROOT, "oracle", "corr_oracle.c"), os.path.join(ROOT, "tests", "csrc", "oracle_sanitize_driver.c"), "-o", exe, "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("sanitizer runtimes not installed")
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "under ASan + UBSan" in run.stdout
