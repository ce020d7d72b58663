"""bench.py's host-side helpers: the closed-form bilinear-fetch count of a Radon intermediate (the algorithmic LDS
bytes of `roofline_radon`) against the oracle's exact count, and the PMC summary reader."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_radon_fetch_count_matches_oracle(oracle_mod):
    b = _bench()
    rng = np.random.default_rng(0)
    for (n_v, n_u), (n_alpha, n_t) in (((96, 128), (96, 80)), ((128, 128), (96, 96)), ((61, 47), (33, 29))):
        img = rng.uniform(0, 1, size=(n_v, n_u)).astype(np.float32)
        _, exact = oracle_mod.radon(img, n_alpha, n_t, count_fetches=True)
        got = b.radon_fetches_per_image(n_u, n_v, n_alpha, n_t)
        assert abs(got - exact) <= 1e-3 * exact, (got, exact)
    assert b.n_kappa_auto(1024, 1024, 768) == 1448 and b.n_kappa_auto(512, 512, 768) == 724


def test_pmc_summary_reader():
    b = _bench()
    p = b.load_pmc("pairs_kernel<true, false>")
    assert p is not None and p["SQ_INSTS_VALU"] > 1e8 and p["FETCH_SIZE"] > 1e5 and p["_tag"]
    for name in ("radon_kernel<true, false>", "radon_kernel<true, true>"):  # exact and contracted arithmetic
        r = b.load_pmc(name)
        assert r is not None and 0 < r["SQ_LDS_BANK_CONFLICT"] < r["SQ_LDS_IDX_ACTIVE"] and r["SQ_INSTS_VALU"] > 1e10
    assert b.load_pmc("radon_kernel<true, true>")["SQ_INSTS_VALU"] < 0.8 * b.load_pmc("radon_kernel<true, false>")["SQ_INSTS_VALU"]
    assert b.load_pmc("no_such_kernel") is None


def test_power_sampler_without_and_with_hwmon(tmp_path, monkeypatch):
    """bench.py's socket-power / engine-clock sampler: silent (None everywhere) where the amdgpu hwmon files do not exist -- this
    container --, and a plain reader of power1_input (microwatts) / freq1_input (Hz) where they do (a fake card here)."""
    import glob as _glob
    import time
    b = _bench()
    real_glob = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat, **kw: [] if pat.startswith("/sys/class/drm") else real_glob(pat, **kw))
    s = b.PowerSampler()
    assert s.dir is None and s.once() is None and s.cap_w() is None
    assert s.start() is s and s.stop() is None
    card = tmp_path / "card7" / "device" / "hwmon" / "hwmon3"
    card.mkdir(parents=True)
    (card / "power1_input").write_text("1398000000\n")
    (card / "freq1_input").write_text("2215000000\n")
    (card / "power1_cap").write_text("1400000000\n")
    (tmp_path / "card7" / "device" / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 2400Mhz *\n")
    monkeypatch.setattr(_glob, "glob", lambda pat, **kw: [str(card)] if pat.startswith("/sys/class/drm") else real_glob(pat, **kw))
    s = b.PowerSampler()
    assert s.dir == str(card) and s.once() == (1398.0, 2215.0) and s.cap_w() == 1400.0 and s.rated_mhz() == 2400.0
    s.start()
    time.sleep(0.05)
    r = s.stop()
    assert r and r["samples"] >= 3 and abs(r["avg_w"] - 1398.0) < 1e-9 and abs(r["sclk_mhz_avg"] - 2215.0) < 1e-9
    # one card visible: it is the device's whatever the bus id says; several cards and none matches: no reading rather than
    # another card's
    assert b.PowerSampler(pci_bus="0000:ff").dir == str(card)
    card2 = tmp_path / "card9" / "device" / "hwmon" / "hwmon4"
    card2.mkdir(parents=True)
    (card2 / "power1_input").write_text("250000000\n")
    (card2 / "freq1_input").write_text("500000000\n")
    monkeypatch.setattr(_glob, "glob", lambda pat, **kw: [str(card), str(card2)] if pat.startswith("/sys/class/drm") else real_glob(pat, **kw))
    assert b.PowerSampler(pci_bus="0000:ff").dir is None
    assert b.PowerSampler(pci_bus="card9").dir == str(card2)  # (the match is a substring of the device's real path)
