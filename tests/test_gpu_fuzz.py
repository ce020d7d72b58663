"""Randomised parity sweep as a test: 60 random geometries / bin grids / parameters (scripts/fuzz_parity.py) through
the C ABI against the oracle -- on the polynomial (throughput) path mean within 1e-5 (relaxed by 30/sqrt(n_pairs) for
tiny problems, whose per-pair fp32 noise does not average out) and pair values within 2e-3; in the library's default
mode (these problems have <= 78 pairs: the CPU path's own arithmetic) mean and every pair within 1e-5."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_randomised_parity_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "60", "11"],
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 of 60 cases out of tolerance" in r.stdout


@pytest.mark.timeout(600)
def test_randomised_radon_sweep():
    """60 random image sizes / bin grids / filters / post-processes / contents (scripts/fuzz_radon.py): bit-exact."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_radon.py"), "60", "9"],
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 of 60 cases differ" in r.stdout


@pytest.mark.timeout(600)
@pytest.mark.parametrize("script,args,done", [("fuzz_preprocess.py", ["60", "4"], "0 of 60 cases differ"),
                                              ("fuzz_direct.py", ["30", "2"], "0 of 30 cases differ"),
                                              ("fuzz_poses.py", ["80", "3"], "0 of 80 cases differ")])
def test_randomised_sweeps_of_the_widened_rows(script, args, done):
    """Pre-processing (bit-exact), MetricDirect / FBCC (1e-5 for both forms) and the batched pose evaluation (bit-identical to the
    sequential evaluations of the same library: csrc/ecc_poses.hip) on random configurations."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + args, capture_output=True, text=True,
                       timeout=500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert done in r.stdout


@pytest.mark.timeout(900)
def test_default_mode_just_above_the_auto_threshold():
    """Where the polynomial path IS the default and averaging over pairs is weakest: 33 ... 90 views (528 ... 4005 pairs),
    the geometry kinds of the sweep above, library default mode -- the MEAN within 1e-5 of the oracle's with no
    size-dependent slack (scripts/fuzz_auto_threshold.py; measured worst of 60 cases: 1.2e-6).  This is what the value of
    ECC_SAMPLING_AUTO_REFERENCE_PAIRS (include/ecc_hip.h) rests on."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_auto_threshold.py"), "48", "5"],
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 of 48 cases out of tolerance" in r.stdout
