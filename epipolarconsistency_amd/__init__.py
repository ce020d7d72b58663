"""epipolarconsistency_amd -- MI355X-native epipolar-consistency hot path.

Radon intermediates + all-pairs ECC behind the reference's RadonIntermediate /
MetricRadonIntermediate interface (aaichert/EpipolarConsistency).  The compute lives in
libecc_hip.so (hand-written HIP for gfx950, C ABI in include/ecc_hip.h); this package is the thin
host mirror.  Importing the package does not load the library; the first API call does and fails
loudly if it has not been built.
"""
from ._lib import (EccError, FILTER_DERIVATIVE, FILTER_NONE, FILTER_RAMP, POST_IDENTITY, POST_LOGARITHM,
                   POST_SQUARE_ROOT)
from . import geometry
from .api import (Context, Group, GroupMetricRadonIntermediate, MetricDirect, MetricRadonIntermediate, pair_shard, pair_shards_balanced, PreProccess, RadonIntermediate, get_ij, host_object_radius, host_pinvT,
                  host_source_position, pack_projection_matrices, slab_floats, estimateAngularRange, estimateAngularStep,
                  estimateIsoCenter, estimateObjectRadius)

__all__ = ["Context", "Group", "GroupMetricRadonIntermediate", "pair_shard", "pair_shards_balanced", "MetricDirect", "PreProccess", "RadonIntermediate", "MetricRadonIntermediate", "EccError", "get_ij", "slab_floats", "pack_projection_matrices", "host_pinvT",
           "host_source_position", "host_object_radius", "FILTER_DERIVATIVE", "FILTER_RAMP", "FILTER_NONE",
           "POST_IDENTITY", "POST_SQUARE_ROOT", "POST_LOGARITHM", "estimateAngularRange", "estimateAngularStep",
           "estimateIsoCenter", "estimateObjectRadius"]
