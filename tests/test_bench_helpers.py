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
