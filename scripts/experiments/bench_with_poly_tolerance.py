#!/usr/bin/env python3
"""bench.py with ecc_debug_set_poly_tolerance(POLY_TOL bins) on every metric (the bound on what lowering a pair's polynomial degree
may cost; default 2e-8): POLY_TOL=1e-7 python scripts/experiments/bench_with_poly_tolerance.py --no-live-pmc --no-power"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import epipolarconsistency_amd as E  # noqa: E402

TOL = float(os.environ.get("POLY_TOL", "2e-8"))
_init = E.MetricRadonIntermediate.__init__


def _init_tol(self, *a, **k):
    _init(self, *a, **k)
    self.debugSetPolyTolerance(TOL)


E.MetricRadonIntermediate.__init__ = _init_tol
import bench  # noqa: E402

bench.main()
