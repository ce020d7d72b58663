"""The collective north_star names, executed: the sharded evaluation path over a ONE-rank RCCL group on the one GPU a
test box has (ref: the sum at EpipolarConsistencyRadonIntermediate.cpp:216-224 is the path's only exchange).

The worker (tests/rccl_one_rank_worker.py) runs in a child process: RCCL communicator set-up, all_gather_into_tensor of the
Radon-intermediate stack, ecc_metric_evaluate_range_async -> all_reduce -> publish_scalar_kernel -> poll, the same with the
all-reduce issued by the library on a communicator of its own (ecc_comm_*), and the gathered cost image; every result must have the bits of the plain single-device call."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sharded_evaluation_over_a_one_rank_rccl_group():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=540)
    assert p.returncode == 0, p.stderr[-4000:]
    line = [t for t in p.stdout.splitlines() if t.startswith("{")][-1]
    r = json.loads(line)
    assert r["backend"] == "nccl" and r["world"] == 1 and r["probe"] == 41.0
    assert len(r["cases"]) == 3
    for c in r["cases"]:
        assert c["same_stack"], c
        # bit-identical: the collective adds one rank's sum to nothing
        assert c["publish"] == c["want"] and c["item"] == c["want"], c
        assert c["publish_moved"] == c["want_moved"] and c["want_moved"] != c["want"], c
        assert abs(c["allreduce_mean"] - c["want_moved"]) <= 4e-16 * abs(c["want_moved"]), c
        assert c["range_sum_over_pairs"] == c["want"], c
        assert c["cost_image_equal"] and c["cost_image_nonzero"] > 0, c
        # the all-reduce issued by the library itself (ecc_comm_*, ecc_metric_evaluate_range_allreduce): the same bits
        assert c["native"] == c["want"] and c["native_moved"] == c["want_moved"], c
        assert abs(c["native_parts_mean"] - c["want"]) <= 4e-16 * abs(c["want"]), c
