"""CPU-side checks of the product library: the C ABI loads and exports every symbol that
include/ecc_hip.h declares (no compute calls without a GPU), fails loudly without a device, and its
host-side pre-compute (E1/E5, get_ij) is bit-identical to the oracle / reference headers."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ecc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(ecc_[a-z0-9_]+)\s*\(", text, flags=re.I)
    return sorted(set(n for n in names if n != "ecc_ctx" and not n.isupper()))


def test_library_exports_every_declared_symbol():
    from epipolarconsistency_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "libecc_hip.so does not export %s" % name
    # and the python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == declared


def test_version_and_no_device_behaviour():
    from epipolarconsistency_amd import _lib
    import epipolarconsistency_amd as E
    L = _lib.lib()
    assert L.ecc_version() == 100
    if L.ecc_device_count() == 0:
        with pytest.raises(E.EccError) as ei:
            E.Context(0)
        assert ei.value.code == 4 and "no CPU fallback" in str(ei.value)


def test_host_precompute_bitwise_vs_oracle(oracle_mod):
    import epipolarconsistency_amd as E
    from test_oracle_pins import P000, P040, _random_Ps
    for P in [P000, P040] + _random_Ps(60, seed=21):
        assert np.array_equal(E.host_pinvT(P), oracle_mod.pinvT(P))
        assert np.array_equal(E.host_source_position(P), oracle_mod.source_position(P))
        if oracle_mod.ref() is not None:
            assert np.array_equal(E.host_pinvT(P), oracle_mod.pinvT(P, use_ref=True))
            assert np.array_equal(E.host_source_position(P), oracle_mod.source_position(P, use_ref=True))
        assert E.host_object_radius(P, 640, 480) == oracle_mod.object_radius(P, 640, 480)


def test_get_ij_closed_form(oracle_mod):
    import epipolarconsistency_amd as E
    for n in (2, 3, 7, 64, 400):
        N = n * (n - 1) // 2
        idx = range(N) if N < 3000 else list(range(0, N, 97)) + [N - 1, N - 2]
        for ij in idx:
            assert E.get_ij(ij, n) == oracle_mod.get_ij(ij, n)
    n = 30000  # closed form stays exact far beyond the reference's `short` indices
    N = n * (n - 1) // 2
    assert E.get_ij(N - 1, n) == (n - 2, n - 1) and E.get_ij(0, n) == (0, 1) and E.get_ij(n - 1, n) == (1, 2)


def test_slab_layout_size():
    import epipolarconsistency_amd as E
    assert E.slab_floats(768, 768) == 770 * 800
    assert E.slab_floats(96, 80) == 98 * 96


def test_metric_helper_functions():
    """estimateAngularRange / estimateAngularStep / estimateIsoCenter (ref: EpipolarConsistency.cpp:8-68) are host
    functions of the library; no device needed."""
    import numpy as np
    from epipolarconsistency_amd import api, synthetic, geometry
    Ps = synthetic.short_scan(12, 256, 256, 1.2)
    # all principal rays of the circular scan pass through the origin
    O = api.estimateIsoCenter(Ps)
    assert np.abs(O[:3]).max() < 1e-6 and O[3] == 1.0
    shifted = [P @ geometry.rigid_transform(tx=-7.0, ty=2.0, tz=3.5) for P in Ps]  # world moved: centre at (7,-2,-3.5)
    assert np.allclose(api.estimateIsoCenter(shifted)[:3], [7.0, -2.0, -3.5], atol=1e-6)
    # baseline distance d of two sources at sid on a circle, angle phi apart: sid*cos(phi/2); kappa_max = asin(r/d)
    sid, r = 744.3, 100.0
    C0, C1 = geometry.camera_center(Ps[0])[:3], geometry.camera_center(Ps[3])[:3]
    d = np.linalg.norm(np.cross(C0, C1)) / np.linalg.norm(C1 - C0)
    a, b = api.estimateAngularRange(Ps[0], Ps[3], r)
    assert abs(b - np.arcsin(r / d)) < 1e-12 and a == -b
    a, b = api.estimateAngularRange(Ps[0], Ps[3], 2 * sid)
    assert abs(b - np.pi / 2) < 1e-15
    rad = max(api.estimateObjectRadius(Ps[0], 256, 256), api.estimateObjectRadius(Ps[3], 256, 256))
    a, b = api.estimateAngularRange(Ps[0], Ps[3], rad)
    assert abs(api.estimateAngularStep(Ps[0], Ps[3], 256, 256) - 2 * (b - a) / np.sqrt(2 * 256 ** 2)) < 1e-15


def test_host_line_to_sample_dtr_matches_oracle(oracle_mod):
    """ecc_host_line_to_sample_dtr (ref: lineToSampleDtr, EpipolarConsistencyCommon.hxx:152-171) is bit-identical to the
    oracle's restatement, which tests/test_oracle_pins.py pins against the reference header."""
    import ctypes as C
    from epipolarconsistency_amd import _lib
    rng = np.random.default_rng(17)
    L = _lib.lib()
    for _ in range(400):
        line = rng.normal(0, 1, 3).astype(np.float32) * np.float32(rng.choice([1.0, 300.0]))
        range_t = float(np.float32(rng.uniform(50, 2000)))
        want, folded = oracle_mod.line_to_sample_dtr(line, range_t)
        got = line.copy()
        f = L.ecc_host_line_to_sample_dtr(C.c_void_p(got.ctypes.data), C.c_float(range_t))
        assert np.array_equal(got[:2], want[:2]) and bool(f) == folded


def test_rccl_is_bound_at_run_time_and_makes_an_id():
    """ecc_comm_unique_id: the library dlopens RCCL on first use (no link-time dependency: `ldd` shows none) and rank 0's
    128-byte communicator id comes back non-trivial; creating the communicator itself needs a device (GPU suite)."""
    import ctypes as C
    import subprocess
    from epipolarconsistency_amd import _lib
    L = _lib.lib()
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed
    buf = (C.c_char * 128)()
    rc = L.ecc_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != 0:  # a box without any librccl.so.1: the call says so and nothing else is affected
        assert rc == 5 and b"librccl" in L.ecc_last_error()
        return
    assert any(b != b"\x00" for b in buf)
    assert L.ecc_comm_unique_id(None) == 1  # ECC_ERR_INVALID_ARGUMENT
    assert L.ecc_comm_destroy(None) == 0


def test_pose_diff_finds_the_moved_views_whatever_the_thread_count(tmp_path):
    """csrc/ecc_pose_diff.h (host only): the comparison that turns full matrix sets into (moved view, matrix) lists for the
    batched pose evaluation (ecc_metric_evaluate_poses; ref for the caller's pattern: Gui/Visualization.h:59-112) -- 0, 1, 2, 7,
    33 and all views moved, strides 1-3, one thread against eight.  The same driver runs under ThreadSanitizer in
    scripts/sanitize.sh."""
    import subprocess
    exe = os.path.join(str(tmp_path), "pose_diff")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "tsan_pose_diff.cpp"), "-lpthread",
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
