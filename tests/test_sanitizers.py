"""CPU sanitizers: `make -C oracle asan` builds the C oracle and the product's host-only
translation unit with -fsanitize=address,undefined; tests/asan_driver.py then runs the golden-vector tests, the
exchange-plan checks, the FASTA reader checks and the .npz writer checks on those builds in a child process with libasan preloaded.  Device
code cannot be sanitized on this pool; it is covered by bit-exact parity against the oracle instead."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_code_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env.update(
        LD_PRELOAD=libasan,
        ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",  # CPython itself never frees everything
        UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=24",
        SKM_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "libkmer_oracle_asan.so"),
        SKM_HOST_ASAN_LIB=os.path.join(ROOT, "oracle", "_build", "libskm_host_asan.so"),
        OMP_NUM_THREADS="4",
    )
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "ASAN_DRIVER_OK" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
