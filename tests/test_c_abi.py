"""include/ecc_hip.h from plain C: tests/c/test_c_abi.c compiles with gcc -std=c99 -pedantic against the header and links
libecc_hip.so -- the boundary has no C++ or torch types in it.  CPU: the host-side entry points run and context creation
fails loudly without a device.  GPU: images -> Radon intermediates -> all-pairs metric -> pose-delta evaluation -> many poses per call."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    from epipolarconsistency_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = os.path.join(tmp, "test_c_abi")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "test_c_abi.c"), "-o", exe, "-L" + libdir, "-lecc_hip", "-lm",
           "-Wl,-rpath," + libdir]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_c99_and_host_entry_points_run(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe, "nodevice"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "version 100" in r.stdout and "ij 1 4" in r.stdout and "shard 11 4" in r.stdout


@pytest.mark.gpu
def test_c_program_runs_the_path(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok") and "recomputed 5" in r.stdout
    assert "batched 3" in r.stdout  # ecc_metric_evaluate_pose_deltas / ecc_metric_evaluate_poses from plain C: the sequential bits
