#!/usr/bin/env python3
"""bench.py with the opt-in row-quad copies switched on for every context (ecc_debug_set_quad_copies before the metric is made):
the exact part of the kappa_max > pi/4 pairs samples four-rows-per-line copies (4x the slab memory).  Same arguments as bench.py.
    python scripts/experiments/bench_with_quads.py --no-cpu-baseline --no-live-pmc --no-power"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import epipolarconsistency_amd as E  # noqa: E402

_init = E.Context.__init__


def _init_with_quads(self, *a, **k):
    _init(self, *a, **k)
    self.debugSetQuadCopies(True)


E.Context.__init__ = _init_with_quads
import bench  # noqa: E402

bench.main()
