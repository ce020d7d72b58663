"""The experiment hooks of earlier versions were environment variables read inside the product (ECC_POLY_TOL changed the
arithmetic of every pair silently).  They are explicit ecc_debug_* calls now: the old variables must change nothing, the
calls must do what they say."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OLD_HOOKS = {"ECC_POLY_TOL": "1e-3", "ECC_SMALL_MAX_PAIRS": "0", "ECC_SMALL_DEBUG": "1", "ECC_RESULT_WAIT": "stream",
             "ECC_QUAD_COPIES": "1"}

WORKER = r'''
import hashlib, json, sys
sys.path.insert(0, %r)
import numpy as np
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
out = {}
ctx = E.Context(0)
for n, mode in ((12, "polynomial"), (12, "auto"), (40, "polynomial"), (40, "auto")):
    Ps = synthetic.short_scan(n, 128, 128, 2.464)
    imgs = synthetic.projections_numpy(Ps, 128, 128, synthetic.sphere_phantom(extent_mm=30, rmin=8, rmax=25))
    dtrs = E.RadonIntermediate.compute_batch(ctx, imgs, 96, 96)
    m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling(mode)
    mean = m.evaluate()
    s, vals = m.evaluate_range(0, n * (n - 1) // 2, want_pairs=True)
    out["%%d %%s" %% (n, mode)] = [mean.hex(), float(s).hex(), hashlib.sha256(vals.tobytes()).hexdigest()]
    m.close()
print(json.dumps(out))
''' % ROOT


def _digests(env):
    p = subprocess.run([sys.executable, "-c", WORKER], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([t for t in p.stdout.splitlines() if t.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_the_old_environment_hooks_change_nothing():
    clean = {k: v for k, v in os.environ.items() if k not in OLD_HOOKS}
    want = _digests(clean)
    got = _digests(dict(clean, **OLD_HOOKS))
    assert got == want
    assert len(want) == 4


@pytest.mark.gpu
def test_debug_calls_do_what_the_variables_did(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import _lib
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setSampling("polynomial")
    base = m.evaluate_range(0, 28, want_pairs=True)
    deg0 = [p["degree"] for p in m.debug_polynomials(0, 28)]
    # the one-launch path off by its size bound, and waiting for the stream instead of polling: the same bits
    m.debugSetSmallEvalBound(0)
    a = m.evaluate_range(0, 28, want_pairs=True)
    assert a[0] == base[0] and np.array_equal(a[1], base[1])
    m.debugSetSmallEvalBound(-1)
    _lib.check(_lib.lib().ecc_debug_set_result_polling(0))
    try:
        b = m.evaluate_range(0, 28, want_pairs=True)
        assert m.evaluate() == base[0] / 28
    finally:
        _lib.check(_lib.lib().ecc_debug_set_result_polling(1))
    assert b[0] == base[0] and np.array_equal(b[1], base[1])
    # a loose economisation bound lowers degrees and moves values a little; the default restores the bits
    m.debugSetPolyTolerance(1e-3)
    deg1 = [p["degree"] for p in m.debug_polynomials(0, 28)]
    c = m.evaluate_range(0, 28, want_pairs=True)
    assert sum(deg1) < sum(deg0)
    assert not np.array_equal(c[1], base[1]) and np.allclose(c[1], base[1], rtol=5e-2)
    m.debugSetPolyTolerance(2e-8)
    d = m.evaluate_range(0, 28, want_pairs=True)
    assert d[0] == base[0] and np.array_equal(d[1], base[1])
    with pytest.raises(E.EccError):
        m.debugSetPolyTolerance(-1.0)
    m.debugSetPolyTolerance(2e-8)
    m.close()
