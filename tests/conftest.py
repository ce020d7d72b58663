import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()  # builds oracle/libecc_oracle.so if missing
    return oracle


def make_small_scan(n=8, n_u=128, n_v=128, pixel_mm=2.464, seed=1234, extent=30.0, rmin=8.0, rmax=25.0):
    """Synthetic short scan used by the parity tests (same generator as the benchmark configs)."""
    from epipolarconsistency_amd import synthetic
    Ps = synthetic.short_scan(n, n_u, n_v, pixel_mm)
    phantom = synthetic.sphere_phantom(seed=seed, extent_mm=extent, rmin=rmin, rmax=rmax)
    imgs = synthetic.projections_numpy(Ps, n_u, n_v, phantom)
    return Ps, imgs


@pytest.fixture(scope="session")
def small_scan(oracle_mod):
    """8 views, 128x128, 96x96 Radon bins: images, matrices and ORACLE dtrs."""
    Ps, imgs = make_small_scan()
    dtrs = [oracle_mod.radon(im, 96, 96) for im in imgs]
    return dict(Ps=Ps, imgs=imgs, dtrs=dtrs, n_u=128, n_v=128, n_alpha=96, n_t=96)


@pytest.fixture(scope="session")
def gpu_ctx():
    import epipolarconsistency_amd as E
    ctx = E.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(autouse=True, scope="session")
def _fixed_sampling_mode():
    """The library's default sampling mode (ECC_SAMPLING_AUTO) switches small evaluations (<= 512 pairs) to the CPU
    path's own arithmetic.  Most GPU tests are small on purpose and are there to hold the THROUGHPUT path (fitted
    polynomials, per-sample fallback) against the oracle, so new metrics default to "polynomial" here; tests of the
    other modes select them with setSampling() (tests/test_configs.py, tests/test_gpu_sampling_modes.py)."""
    import epipolarconsistency_amd as E
    E.MetricRadonIntermediate.default_sampling = "polynomial"
    yield
    E.MetricRadonIntermediate.default_sampling = None
