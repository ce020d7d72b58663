"""Builds libecc_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

`python -m epipolarconsistency_amd.build` or build_library() from __graft_entry__.build().
hipcc cross-compiles without a GPU.  -ffp-contract=off: the Radon kernel's arithmetic is specified
as unfused IEEE binary32 (bit parity with the oracle); the host geometry is specified in unfused
binary64.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libecc_hip.so")
SOURCES = ["radon_kernel.hip", "ramp_kernel.hip", "preprocess_kernel.hip", "direct_kernel.hip", "pairs_kernel.hip", "small_eval_kernel.hip", "geometry_kernel.hip", "ecc_capi.hip", "ecc_radon_api.hip", "ecc_metric_api.hip", "ecc_preprocess_api.hip", "ecc_direct_api.hip", "ecc_evaluate.hip", "ecc_poses.hip",
           "ecc_exchange.cpp", "ecc_group.cpp", "ecc_rccl.cpp"]  # .cpp: host-only code (no device code), still built by hipcc for the HIP headers
HEADERS = ["ecc_layout.h", "ecc_host_geometry.h", "ecc_sampling.h", "ecc_worker_pool.h", "ecc_pose_diff.h", "ecc_slab_tile.h", "ecc_pairs_device.h", "ecc_capi_internal.h", os.path.join("..", "..", "include", "ecc_hip.h")]
# radon_kernel.hip: the SLP vectoriser packs the two samples of the derivative pair into v_pk_*_f32 pairs, which
# cost two issue slots each on gfx950 (no gain, scripts/micro/valu_rate.hip) plus ~12 v_mov per iteration to
# arrange operands -- scalar code is ~15 % faster there; the pair kernel gains 7 % the same way (0.548 -> 0.512 ms).
# direct_kernel.hip (round 5): the same walker as the Radon kernel was still built WITH the vectoriser -- 9.2-9.4 -> 7.45-7.55 ms per
# 496-pair evaluation of 1024^2 images without it (A/B/A/B on one box, bit-identical sums; unrolling its tile loop on top: nothing).
PER_SOURCE_FLAGS = {"radon_kernel.hip": ["-fno-slp-vectorize"], "pairs_kernel.hip": ["-fno-slp-vectorize"],
                    "small_eval_kernel.hip": ["-fno-slp-vectorize"], "direct_kernel.hip": ["-fno-slp-vectorize"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, extra_flags=()):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc] + FLAGS + PER_SOURCE_FLAGS.get(s, []) + list(extra_flags) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lrt", "-lpthread", "-ldl", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
