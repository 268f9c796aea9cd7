"""The C++ host-side mirror (include/srrg2_slam_amd.hpp) and the C++ re-statement of the reference's slice tests
(tests/cpp/test_motion_model_slice.cpp).  CPU: it compiles and links against the C-ABI library with plain g++
(no HIP headers needed by a caller).  gpu: the binary runs the three reference scenarios on the device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "bin", "test_motion_model_slice")
BIN_CYCLE = os.path.join(CPP, "bin", "test_tracker_cycle")
BIN_LOOP = os.path.join(CPP, "bin", "test_loop_closure")


def _build():
    subprocess.check_call(["make", "-C", CPP, "-s"])


def test_cpp_mirror_compiles_and_links():
    _build()
    assert os.path.exists(BIN) and os.path.exists(BIN_CYCLE) and os.path.exists(BIN_LOOP)


@pytest.mark.gpu
def test_cpp_reference_scenarios_on_gpu():
    if not os.path.exists(BIN):
        _build()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASSED" in out.stdout


@pytest.mark.gpu
def test_cpp_tracker_cycle_on_gpu():
    """clip -> align -> merge through SceneClipperBall / MultiAligner3DQR / MergerCorrespondenceHomo (C++ mirror)."""
    if not os.path.exists(BIN_CYCLE):
        _build()
    out = subprocess.run([BIN_CYCLE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASSED" in out.stdout


@pytest.mark.gpu
def test_cpp_loop_closure_on_gpu():
    """select candidates -> batched alignment + accept gates -> relocalize -> validate -> optimize, all through the C++
    mirror (srrg2_slam_amd_loop_closure.hpp): the host drivers of SURVEY.md section 8f rows 1 and 3 in C++."""
    if not os.path.exists(BIN_LOOP):
        _build()
    out = subprocess.run([BIN_LOOP], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASSED" in out.stdout
