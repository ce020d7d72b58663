"""The header-only C++ adapter (epipolarconsistency_amd/cpp/EpipolarConsistencyHip.hxx): compiles with
g++ against include/ecc_hip.h (CPU check) and, on the GPU box, behaves like the reference classes."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "epipolarconsistency_amd")


def _build(tmp_path):
    exe = os.path.join(tmp_path, "test_adapter")
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "cpp"),
           os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp"), "-L" + PKG, "-lecc_hip", "-L/opt/rocm/lib",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_adapter_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path))
    assert os.path.exists(exe)
    # without arguments the driver exits with its usage code and touches no device
    assert subprocess.run([exe]).returncode == 2


@pytest.mark.gpu
def test_adapter_matches_oracle(tmp_path, oracle_mod, small_scan):
    s = small_scan
    exe = _build(str(tmp_path))
    n = 4
    imgs = np.ascontiguousarray(s["imgs"][:n], np.float32)
    Ps = s["Ps"][:n]
    ipath, ppath = os.path.join(tmp_path, "imgs.bin"), os.path.join(tmp_path, "Ps.bin")
    imgs.tofile(ipath)
    oracle_mod.pack_Ps(Ps).tofile(ppath)
    out = subprocess.run([exe, ipath, str(n), str(s["n_u"]), str(s["n_v"]), str(s["n_alpha"]), str(s["n_t"]), ppath],
                         check=True, capture_output=True, text=True).stdout
    val = {k: v for k, v in re.findall(r"^(\w+) (.*)$", out, flags=re.M)}
    dtrs = s["dtrs"][:n]
    want = oracle_mod.evaluate_all(Ps, dtrs, s["n_u"], s["n_v"])
    assert abs(float(val["radius"]) - oracle_mod.object_radius(Ps[0], s["n_u"], s["n_v"])) < 1e-9
    assert abs(float(val["mean"]) - want["mean"]) < 1e-5 * want["mean"]
    c10, c01 = re.match(r"([\d.eE+-]+) cost01 ([\d.eE+-]+)", val["cost10"]).groups()
    assert abs(float(c10) - want["pairs"][0]) < 2e-4 * want["pairs"][0] and float(c01) == -1.0
    idx = np.array([[0, 2, 0, 2], [0, 3, 0, 3], [2, 3, 2, 3]], np.int32)
    sub = oracle_mod.evaluate_pairs(Ps, dtrs, s["n_u"], s["n_v"], idx)
    assert abs(float(val["subset"]) - sub["mean"]) < 5e-5 * sub["mean"]
    f = val["dtr1"].split()
    assert int(f[0]) == s["n_alpha"] * s["n_t"] and float(f[1]) == float(dtrs[1].reshape(-1)[1234])
    assert f[3:5] == [str(s["n_alpha"]), str(s["n_t"])] and f[6:8] == [str(s["n_u"]), str(s["n_v"])]
    # host tex2D / sample of the adapter's dtr class against the Python mirror of the same reference functions
    import epipolarconsistency_amd as E
    ctx = E.Context(0)
    d1 = E.RadonIntermediate.from_host(ctx, dtrs[1], s["n_u"], s["n_v"])
    d1.readback()
    line = np.array([0.6, -0.8, -30.0], np.float32)
    smp = d1.sample(line)
    f = [np.float32(x) for x in val["hostsample"].split()]
    assert f[0] == np.float32(d1.tex2D(0.25, 0.75)) and f[1] == np.float32(smp) and f[2] == line[0] and f[3] == line[1]
    d1.close()
    ctx.close()
    e7 = oracle_mod.evaluate_for_image_pair(Ps, dtrs, 0, 2, s["n_u"], s["n_v"])
    f = val["pair02"].split()
    assert abs(float(f[0]) - e7["ecc"]) < 1e-4 * e7["ecc"]
    assert [int(x) for x in f[1:4]] == [len(e7["kappas"])] * 3
    assert np.float32(f[4]) == e7["kappas"][0] and abs(float(f[5]) - e7["radon0"][0, 0]) < 2e-6
    f = val["direct"].split()
    dsum = oracle_mod.direct_evaluate(Ps, imgs)
    dpair = oracle_mod.direct_pair(Ps[0], Ps[2], imgs[0], imgs[2], 0.0, oracle_mod.object_radius(Ps[0], s["n_u"], s["n_v"]))
    assert abs(float(f[0]) - dsum["sum"]) < 1e-5 * dsum["sum"] and abs(float(f[1]) - dpair["metric"]) < 1e-5 * dpair["metric"]
    assert int(f[2]) == len(dpair["kappas"])
    r50 = oracle_mod.evaluate_all(Ps, dtrs, s["n_u"], s["n_v"], object_radius_mm=50.0)
    assert abs(float(val["mean_r50"]) - r50["mean"]) < 1e-5 * r50["mean"]
