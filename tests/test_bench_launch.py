"""bench.py --gpus N starts its own N ranks when it is not already running under torchrun (the
driver's form for N = 1, `python bench.py --gpus N ...`, must never measure one rank and call it N).  The sharded
operation: ref EpipolarConsistencyRadonIntermediate.cpp:166-225 (the only cross-pair step is the mean, :216-224).
--launch-check rehearses the launch alone (gloo, no GPU), so this runs on the CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_without_torchrun_starts_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen_by_collective_backend"] == 2


def test_a_dead_rank_ends_the_job_non_zero():
    r = _run(["--gpus", "2", "--launch-check"], ECC_BENCH_FAIL_RANK="1")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]  # and no result line


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--launch-check"], WORLD_SIZE="1", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0 and json.loads(r.stdout)["n_gpus"] == 1


def test_without_a_gpu_the_ranks_fail_loudly_and_so_does_the_parent():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc"])
    assert r.returncode != 0 and "needs a GPU" in r.stderr


def test_launcher_hands_out_distinct_local_ranks_and_shared_devices_trip():
    """what can be verified about N > 1 without a second GPU -- the self-launcher gives every rank its own
    LOCAL_RANK (= its device), and the check that runs before anything is measured refuses ranks that report the same device."""
    r = _run(["--gpus", "3", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert sorted(d["config"]["local_ranks"]) == [0, 1, 2]
    r = _run(["--gpus", "2", "--launch-check"], ECC_BENCH_FAKE_SHARED_DEVICE="1")
    assert r.returncode != 0 and "ranks share devices" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    r = _run(["--gpus", "2", "--launch-check", "--single-device"], ECC_BENCH_FAKE_SHARED_DEVICE="1")  # the one-GPU rehearsal form
    assert r.returncode == 0


def test_ranks_agree_on_the_rccl_exchange_before_anybody_joins_it():
    """Advisor, round 5: a rank that cannot bind RCCL returns from ecc_comm_create at once and the others would wait inside
    ncclCommInitRank for ever.  The ranks therefore agree first (MIN over `ecc_comm_available`); one rank without RCCL turns
    the library's exchange off on ALL ranks."""
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    a = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])["config"]["rccl_agreement"]
    assert len(a) == 2 and len({x["agreed"] for x in a}) == 1 and all(x["agreed"] == all(y["local"] for y in a) for x in a)
    r = _run(["--gpus", "2", "--launch-check"], ECC_BENCH_FAKE_NO_RCCL_RANK="1")
    assert r.returncode == 0, r.stderr[-2000:]
    a = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])["config"]["rccl_agreement"]
    assert [x["local"] for x in sorted(a, key=lambda x: x["rank"])][1] is False and not any(x["agreed"] for x in a)
