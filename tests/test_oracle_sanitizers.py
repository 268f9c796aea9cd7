"""SURVEY.md section 5 (race detection / sanitizers): the reference has none (single thread, -Werror=pedantic only); the
build's counterpart is the CPU oracle under -fsanitize=address,undefined.  The oracle is the checker of every parity test:
an out-of-bounds read or a signed overflow in it would silently bless wrong device results."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan on this box")
    lib = str(tmp_path / "liboracle_san.so")
    srcs = sorted(glob.glob(os.path.join(ROOT, "oracle", "o_*.c")))
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                           "-std=c99", "-shared", "-o", lib] + srcs + ["-lm"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", SRRG2_ORACLE_LIB=lib,
               UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers_scripts", "oracle_sanitizer_drive.py")],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-4000:]
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
    assert "oracle sanitizer drive ok" in out.stdout
