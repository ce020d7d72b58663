#!/usr/bin/env python3
"""bench.py -- ECC evaluations/s on the BASELINE workload (400 projections, 1024x1024, 768x768 Radon bins).

One "step" = one all-pairs evaluation as an optimiser iteration pays for it
(ref: Gui/SingleImageMotion.h:84-90 -> MetricRadonIntermediate::setProjectionMatrices + evaluate):
new projection matrices come from the host, the per-view pre-compute + upload, the pair kernel over
all n(n-1)/2 pairs, the float64 reduction and the scalar back on the host.  Radon intermediates are
resident in HBM when the timed region starts (they are computed once per data set, before it, and
that cost is reported separately as ms_per_radon_intermediate).

N GPUs: one process per GPU, dtr stack produced data-parallel + all-gathered once (RCCL), contiguous
cost-balanced shards of the pair range per rank, and per evaluation the 8-byte partial sums are added by
an RCCL all-reduce of a device scalar (the exchange north_star names): THAT step is `value`.  The
library's shared-memory exchange of the same sums (they are on the host already) is timed too and
reported under timing.other_exchange (config.sum_exchange says which one `value` is).  The total work
per evaluation is fixed: scaling is "strong".
(The single-process form of the same partitioning -- ecc_group_* of the C ABI, what a C++ caller of the
adapter uses -- is measured by scripts/bench_group.py.)

Launch forms: under torchrun (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, RANK /
WORLD_SIZE in the environment) this process is one rank.  WITHOUT them `python bench.py --gpus N` with N > 1
starts its own N ranks: the parent -- before it imports torch or touches the GPU -- runs `python -m
torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py
<same arguments>` as a CHILD process, forwards rank 0's single JSON line and exits with the child's status
(a failing or killed rank ends the job non-zero); it never falls back to one rank.

Timing: W warm-up steps, then blocks of exactly K steps, each bracketed by barrier + synchronize on both
sides and reduced with MAX over ranks.  The first block is what a cold process gets (clocks still
ramping, first touch of the pose arrays) and is reported as `cold`; `value` / `ms_per_step` are the
MEDIAN block (all block times are listed), so a 20-step run reports the same steady-state number as a
200-step one.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0       # HBM3E, spec
HBM_ACHIEVABLE_GBS = 6290.0  # what a float4 copy reaches (MI355X_MICROARCH.md: "8.0 TB/s spec; 6.29 TB/s measured (float4 copy, 79%)")
ENGINE_CLOCK_GHZ = 2.4      # peak engine clock; the pair kernel runs at ~2.1 GHz under load
N_CU, SIMD_PER_CU = 256, 4
L1_BYTES_PER_CLK_CU = 64.0  # a 16-byte-per-lane gather occupies the CU's vector L1 for 16 cycles (profiles/r01_gather_rate.txt)
LDS_READ2_B32_BYTES_PER_CLK_CU = 128.0  # ds_read2_b32 / ds_read_b32: 128 B/clk/CU (the guide's LDS table)
LDS_READ_B64_BYTES_PER_CLK_CU = 256.0   # ds_read_b64 / b128: 256 B/clk/CU
VALU_CYCLES_PER_WAVE_INSTR = 2.0        # wave64 fp32 instruction on a SIMD-32: 2 cycles (4 for cvt/fract/f64, not modelled)


_RESULT_STREAM = None


def claim_stdout():
    """The one JSON line is the only thing that may reach this process's stdout: libraries write there too (RCCL prints a
    five-line version banner to stdout when rank 0 creates its communicator).  Keep a private handle on the original
    stdout for emit() and point file descriptor 1 at stderr for everything else."""
    global _RESULT_STREAM
    if _RESULT_STREAM is None:
        sys.stdout.flush()
        _RESULT_STREAM = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    out = _RESULT_STREAM or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


class PowerSampler:
    """Socket power and engine clock of one GPU while a region runs, from the amdgpu hwmon files (power1_input in microwatts,
    freq1_input in Hz; readable without privileges).  The pair and Radon kernels run into the socket's POWER CAP (1400 W):
    the clock they hold -- not the 2.4 GHz the roofs are priced at -- is what the chip allows this instruction mix, so the
    bench reports both.  A thread that reads two small files every 2 ms; never active inside a timed block."""

    def __init__(self, pci_bus=None):
        """pci_bus: "dddd:bb" of the device whose files are wanted (a node with several cards); None = the first card found"""
        import glob
        self.dir = None
        found = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(d, "power1_input")) and os.path.exists(os.path.join(d, "freq1_input")):
                found.append(d)  # (a one-GPU box shows the hwmon files of its own card only)
        for d in found:
            if pci_bus and pci_bus in os.path.realpath(os.path.dirname(os.path.dirname(d))):
                self.dir = d
        if not self.dir and found and (pci_bus is None or len(found) == 1):
            self.dir = found[0]
        self.samples = []
        self._stop = False
        self._thread = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().split()[0])
        except Exception:
            return None

    def once(self):
        if not self.dir:
            return None
        p, f = self._read("power1_input"), self._read("freq1_input")
        return None if p is None or f is None else (p * 1e-6, f * 1e-6)

    def cap_w(self):
        v = self._read("power1_cap") if self.dir else None
        return None if v is None else v * 1e-6

    def rated_mhz(self):
        try:
            with open(os.path.join(os.path.dirname(os.path.dirname(self.dir)), "pp_dpm_sclk")) as f:
                return max(float(tok[:-3]) for line in f for tok in line.split() if tok.lower().endswith("mhz") and tok[:-3].replace(".", "").isdigit())
        except Exception:
            return None

    def start(self):
        import threading
        if not self.dir:
            return self
        self.samples, self._stop = [], False

        def run():
            while not self._stop:
                v = self.once()
                if v:
                    self.samples.append(v)
                time.sleep(0.002)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()
        return self

    def stop(self, skip_s=0.0):
        """-> summary of the samples (the first skip_s seconds dropped: the power reading is a moving average)"""
        if not self._thread:
            return None
        self._stop = True
        self._thread.join()
        self._thread = None
        k = int(skip_s / 0.002)
        sm = self.samples[k:] if len(self.samples) > k + 4 else self.samples
        if not sm:
            return None
        w = sorted(v[0] for v in sm)
        f = [v[1] for v in sm]
        return {"avg_w": sum(w) / len(w), "p95_w": w[int(0.95 * (len(w) - 1))], "sclk_mhz_avg": sum(f) / len(f),
                "sclk_mhz_min": min(f), "samples": len(sm)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--views", type=int, default=400)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--bins", type=int, default=768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--exchange", default="both", choices=["both", "shm", "collective", "rccl"],
                    help="N > 1: per-evaluation sum of the partial results through the library's shared-memory "
                         "exchange, an all-reduce of a device scalar, or (default) both, one after the other")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: do not run the rocprofv3 --pmc pass of this same bench (a child process, ~15 s) that "
                         "measures the pair kernel's HBM bytes and vector instructions for the roofline objects; the "
                         "committed profiles/pmc_current.json is used instead")
    ap.add_argument("--blocks", type=int, default=0,
                    help="number of timed K-step blocks (0 = automatic: 5 to 60, about 1.2 s in total)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--cpu-sample-stride", type=int, default=1,
                    help="CPU baseline evaluates all pairs among every k-th view")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse only the launch: every rank joins the process group (gloo, no GPU touched), one all-reduce "
                         "checks the rank count, rank 0 prints one JSON line; ECC_BENCH_FAIL_RANK=r makes rank r die first "
                         "(tests/test_bench_launch.py)")
    ap.add_argument("--pmc-child", action="store_true",
                    help="internal: this run is the short child of live_pmc() under rocprofv3 -- no contracted-arithmetic stack, no "
                         "metric tie (their two extra all-pairs launches would be averaged into the pair kernel's counters)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: run the N > 1 code path over a ONE-rank process group (RCCL with --backend nccl) -- the "
                         "all-gather of the Radon-intermediate stack, ecc_metric_evaluate_range_async -> all-reduce of the "
                         "device scalar -> publish_scalar_kernel -> poll per step, MAX-reduced block times: first contact of "
                         "that path with RCCL on a one-GPU box (profiles/r05_bench_1rank_rccl.json)")
    ap.add_argument("--no-power", action="store_true",
                    help="skip the ~1.5 s of extra steps behind the timed blocks during which socket power and engine clock are sampled "
                         "(profiler runs: every step is a traced dispatch)")
    ap.add_argument("--sweep-poses", action="store_true",
                    help="BASELINE config 5 instead of the per-step bench: the 600-point 6-DoF sweep of one view "
                         "(ref: Gui/Visualization.h:78-98), the POSES sharded round-robin over the ranks, every rank "
                         "evaluating all pairs of its poses, no exchange inside the timed loop; --steps / --warmup are ignored")
    return ap.parse_args()


def radon_fetches_per_image(n_u, n_v, n_alpha, n_t):
    """Bilinear fetches of one derivative Radon intermediate: sum over bins of 2 * (#steps of `for t = t0; t <= t1;
    t += 0.66f`), line clipping as in ref: RadonIntermediate.cu:44-99, evaluated in float32 numpy.  The step count is
    taken as floor((t1 - t0) / 0.66) + 1 (the kernel's fp32 accumulation of t can differ by one step in rare bins;
    checked against the oracle's exact count in tests/test_bench_helpers.py to 1e-3)."""
    import numpy as np
    f = np.float32
    Pi = f(3.14159265359)
    ix, iy = np.meshgrid(np.arange(n_alpha, dtype=np.float32), np.arange(n_t, dtype=np.float32))
    alpha = (ix / f(n_alpha) - f(0.5)) * Pi
    tau = (iy / f(n_t) - f(0.5)) * f(np.sqrt(f(n_u) * f(n_u) + f(n_v) * f(n_v)))
    l0, l1 = -np.sin(alpha.astype(np.float64)).astype(np.float32), np.cos(alpha.astype(np.float64)).astype(np.float32)
    l2 = -tau + (f(-0.5) * f(n_u) * l0 - f(0.5) * f(n_v) * l1)
    o0, o1, d0, d1 = -l2 * l0, -l2 * l1, l1, -l0
    with np.errstate(divide="ignore", invalid="ignore"):
        ts = np.stack([(f(1) - o0) / d0, (f(n_u) - f(1) - o0) / d0, (f(1) - o1) / d1, (f(n_v) - f(1) - o1) / d1])
    ax0, ax1 = d0 * d0 < f(1e-12), d1 * d1 < f(1e-12)
    ts[0][ax0], ts[1][ax0], ts[2][ax1], ts[3][ax1] = -1e10, 1e10, -1e10, 1e10
    ts = np.sort(ts, axis=0)
    t0, t1 = ts[1], ts[2]
    u, v = o0 + t0 * d0, o1 + t0 * d1
    ok = (u <= n_u) & (v <= n_v) & (u >= 0) & (v >= 0) & (t1 > t0)
    steps = np.where(ok, np.floor((t1 - t0).astype(np.float64) / float(f(0.66))) + 1, 0)
    return int(2 * steps.sum())


def load_pmc(kernel_prefix):
    """Per-dispatch PMC averages of a kernel from profiles/pmc_current.json (made by scripts/pmc_summary_to_json.py
    from separate rocprofv3 --pmc passes of this bench, see its `files`), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_current.json")
    try:
        d = json.load(open(path))
        for name, c in d["kernels"].items():
            if name.startswith(kernel_prefix):
                return dict(c, _tag=d.get("tag"), _files=d.get("files"))
    except Exception:
        pass
    return None


def live_pmc(args, counters=("FETCH_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD")):
    """One rocprofv3 --pmc pass over a short run of THIS bench (child process: python3 bench.py --steps 3 --no-live-pmc):
    per-launch averages of FETCH_SIZE, SQ_INSTS_VALU and SQ_INSTS_VMEM_RD of the pair kernel, measured on this box in
    this run.  FETCH_SIZE (TCC) and the SQ counters fit one pass; WRITE_SIZE cannot ride along (TCC slots,
    MI355X_MICROARCH.md) and is taken as the kernel's algorithmic 4 bytes per pair (the separate pass under profiles/
    measures 328 KiB for 79 800 pairs: 4.2 B per pair).  Returns a dict or None (rocprofv3 missing, timeout, parse error)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    # already running under a profiler (scripts/pmc_pass.sh, profile_round.sh): its preloaded tool library would be
    # inherited by the child pass, whose launcher then execs from a GPU-initialised process -- do not nest
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or \
            any(k.startswith(("ROCP", "ROCPROF", "ROCTX")) for k in os.environ):
        return None
    out_dir = tempfile.mkdtemp(prefix="ecc_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv",
               "-d", out_dir, "--", sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--blocks", "3",
               "--no-cpu-baseline", "--no-live-pmc", "--pmc-child", "--views", str(args.views), "--size", str(args.size), "--bins", str(args.bins)]
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
        per, grids = {}, {}
        for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "pairs_kernel<true, false>" in r["Kernel_Name"]:
                    # a dispatch's counter can come in several rows (one per XCC group): they add up
                    key = (r["Dispatch_Id"], r["Counter_Name"])
                    per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
                    grids[r["Dispatch_Id"]] = int(float(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))
        # with record reuse a step launches the pair kernel twice: over all pairs (skipping the moved view's), and over the
        # list of the moved view's pairs on the side stream; the roofline is that of the former (the larger grid)
        big = max(grids.values()) if grids else 0
        per = {k: v for k, v in per.items() if 2 * grids.get(k[0], 0) >= big}
        res = {}
        for name in counters:
            v = [val for (d, c), val in per.items() if c == name]
            if not v:
                return None
            res[name] = sum(v) / len(v)
        res["dispatches"] = len([1 for (d, c) in per if c == counters[0]])
        return res
    except Exception:
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def host_cpu_info():
    """Model string, physical cores and usable CPUs of this box (printed with every CPU number)."""
    model, phys, logical = None, set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                logical += 1
            elif line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical or (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2 CPU quota of the container, in CPUs
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    sockets = len({p for p, _ in phys}) or 1
    return {"model": model or "unknown", "sockets": sockets, "physical_cores": len(phys) or None, "logical_cpus": logical or None,
            "usable_cpus": usable, "cgroup_cpu_quota": quota}


def n_kappa_auto(n_u, n_v, n_t):
    """#{k >= 0 : dkappa*(k+1/2) < kappa_max} with dkappa = 2*kappa_max/num_samples,
    num_samples = n_t*step_t*2 (ref: ...RadonIntermediate.cu:320,337; .cu:257-263) = ceil(D - 1/2)."""
    import numpy as np
    step_t = np.float32(np.sqrt(float(n_v) ** 2 + float(n_u) ** 2) / n_t)
    D = float(np.float32(n_t) * step_t)
    return int(np.ceil(D - 0.5))


def sweep_poses(args, E, geometry, synthetic, dist, torch, np, ctx, metric, Ps, dtrs, n, S, B, rank, world, dev, ranks_seen, devices_seen):
    """BASELINE config 5 (ref: Gui/Visualization.h:78-98 plotCostFunction over the "3D Rigid" parameters of
    LibProjectiveGeometry/Models/ModelSimilarity3D.hxx:64-88, P' = P T): view n/2 swept over 6 parameters x 100 steps
    in [-5, 5] mm / [-2, 2] deg = 600 all-pairs evaluations.  The POSES are the independent units here: rank r takes
    poses r, r + N, ...; every rank holds the whole Radon-intermediate stack and evaluates all pairs of its poses;
    nothing is exchanged inside the timed loop; the 600 values are gathered afterwards."""
    moving = n // 2
    names = ["tx", "ty", "tz", "rx", "ry", "rz"]
    ranges = [5.0, 5.0, 5.0] + [float(np.deg2rad(2.0))] * 3
    packed = E.pack_projection_matrices(Ps)
    P0 = Ps[moving].copy()

    def pose(q):
        p, k = divmod(q, 100)
        x = -ranges[p] + 2 * ranges[p] * k / 99.0
        out = packed.copy()
        out[moving] = (P0 @ geometry.rigid_transform(**{names[p]: x})).T.reshape(12)
        return out

    mine = list(range(rank, 600, world))
    poses = [pose(q) for q in mine]  # producing a pose is the optimiser's work, not the metric's
    flat = np.ascontiguousarray(np.stack(poses))
    for q in range(3):
        metric.setProjectionMatrices(poses[q % len(poses)]).evaluate()  # warm-up
    metric.setProjectionMatrices(packed)
    metric.evaluate_poses(flat)  # (untimed: allocates the batch's scratch for this many poses -- 76 MB of records, values and index tuples)
    metric.setProjectionMatrices(packed)
    values = np.zeros(600)

    def fenced(fn):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    # ecc_metric_evaluate_poses: this rank's poses handed over as FULL matrix sets; the library finds the moved view of every pose
    # and runs all of them as ONE record launch, ONE pair launch and ONE segmented sum (csrc/ecc_poses.hip)
    ctx.enable_timing(True)
    got, elapsed = fenced(lambda: metric.evaluate_poses(flat))
    batch_pairs_ms = ctx.last_kernel_ms("pairs")
    ctx.enable_timing(False)
    batched = metric.last_batched_poses()
    values[mine] = got
    # the same poses as deltas of the current matrices (what a caller that knows which view it moves hands over: 96 bytes a pose)
    metric.setProjectionMatrices(packed)
    d_off = np.arange(len(poses) + 1, dtype=np.int32)
    d_views = np.full(len(poses), moving, np.int32)
    d_Ps = np.ascontiguousarray(flat[:, moving, :])
    got_sparse, elapsed_sparse = fenced(lambda: metric.evaluate_pose_deltas_packed(d_off, d_views, d_Ps))
    # the rounds-4/5 forms, for the record: two deep on the stream; one setProjectionMatrices + evaluate at a time; the latter in
    # the pose-delta mode (only the moved view's pairs re-evaluated per step)
    metric.setPoseBatching(False)
    got_two_deep, elapsed_two_deep = fenced(lambda: metric.evaluate_poses(flat))
    one_by_one, elapsed_seq = fenced(lambda: np.array([metric.setProjectionMatrices(Pq).evaluate() for Pq in poses]))
    metric.setIncremental(True)
    metric.setProjectionMatrices(packed).evaluate()
    got_delta, elapsed_delta = fenced(lambda: np.array([metric.setProjectionMatrices(Pq).evaluate() for Pq in poses]))
    metric.setIncremental(False)
    metric.setPoseBatching(True)
    for name, other in (("deltas", got_sparse), ("two-deep", got_two_deep), ("pose-delta mode", got_delta), ("batched", values[mine])):
        if not np.array_equal(one_by_one, other):
            raise SystemExit("rank %d: the %s form changed %d of %d values" % (rank, name, int((one_by_one != other).sum()), len(mine)))
    if batched != len(mine):
        raise SystemExit("rank %d: %d of %d poses went through the batch" % (rank, batched, len(mine)))
    if world > 1:
        on = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        v = torch.from_numpy(values).to(on)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)  # every pose was evaluated by exactly one rank
        values = v.cpu().numpy()
    values = values.reshape(6, 100)
    out = {"metric": "ECC evaluations/sec (6-DoF sweep of one view, N=%d, %d^2 projections)" % (n, S),
           "value": 600 / elapsed, "unit": "evaluations/s", "n_gpus": world, "steps": 600, "warmup": 3,
           "ms_per_step": 1e3 * elapsed / 600 * world, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config 5: %d-projection %dx%d scan, %dx%d Radon bins, view %d swept over 6 rigid "
                                  "parameters x 100 steps, all %d pairs per pose" % (n, S, S, B, B, moving, n * (n - 1) // 2),
                      "parallelism": "poses round-robin over %d ranks, whole Radon-intermediate stack on every rank, no exchange "
                                     "in the timed loop" % world,
                      "k01_record_reuse": "on (library default)", "ranks_seen_by_collective_backend": ranks_seen,
                      "devices": devices_seen},
           "ms_per_step_note": "per rank: each rank evaluates %d poses" % len(mine),
           "timing": {"value_is": "ecc_metric_evaluate_poses (full matrix sets per pose): this rank's poses as one batched record / pair / "
                                  "sum launch each (csrc/ecc_poses.hip); every value bit-identical to setProjectionMatrices + evaluate",
                      "batch_pair_kernel_ms": batch_pairs_ms, "batch_pairs": len(mine) * (n - 1),
                      "pose_deltas": {"value": len(mine) * world / elapsed_sparse, "ms_total": 1e3 * elapsed_sparse,
                                      "note": "ecc_metric_evaluate_pose_deltas: the same poses handed over as (moved view, its matrix)"},
                      "two_deep": {"value": len(mine) * world / elapsed_two_deep, "ms_per_pose": 1e3 * elapsed_two_deep / len(mine),
                                   "note": "ecc_metric_set_pose_batching(0): one stream-ordered all-pairs evaluation per pose, two deep "
                                           "(rounds 4-5's headline)"},
                      "one_pose_at_a_time": {"value": len(mine) * world / elapsed_seq if elapsed_seq > 0 else None,
                                             "ms_per_pose": 1e3 * elapsed_seq / len(mine),
                                             "note": "rank 0's poses by setProjectionMatrices + evaluate each"},
                      "pose_delta_mode": {"value": len(mine) * world / elapsed_delta, "ms_per_pose": 1e3 * elapsed_delta / len(mine),
                                          "note": "the same with ecc_metric_set_incremental: only the moved view's pairs per step"},
                      "all_forms_bit_identical": True},
           # the batch's dominant kernel: pairs_kernel over the (pose, moved view) x partner grid -- the same per-pair bytes as the
           # all-pairs launch, priced against the CU's vector L1 like there (the grid launch is L2-resident:
           # profiles/r06_pmc_pose_batch.txt, hit rate 0.97 -- HBM is not what it crosses)
           "roofline": ({"bound": "l1", "kernel": "pairs_kernel<true, false> over %d grid entries" % (len(mine) * n),
                         "kernel_ms": batch_pairs_ms,
                         "achieved": (64 * n_kappa_auto(S, S, B) + 68) * len(mine) * (n - 1) / (batch_pairs_ms * 1e-3) / 1e9,
                         "peak": L1_BYTES_PER_CLK_CU * N_CU * ENGINE_CLOCK_GHZ, "unit": "GB/s",
                         "frac": (64 * n_kappa_auto(S, S, B) + 68) * len(mine) * (n - 1) / (batch_pairs_ms * 1e-3) / 1e9
                                 / (L1_BYTES_PER_CLK_CU * N_CU * ENGINE_CLOCK_GHZ),
                         "traffic": None,
                         "note": "algorithmic gather bytes of the %d real pairs of the launch (holes excluded) through the vector L1"
                                 % (len(mine) * (n - 1))} if batch_pairs_ms > 0 else None),
           "min_at_step": [int(np.argmin(values[p])) for p in range(6)],
           "values_checksum": float(values.sum())}
    if rank == 0 and not args.no_cpu_baseline:
        import oracle
        oracle.build(native=True)
        host = [d.readback() for d in dtrs]
        errs = []
        # one oracle point per swept parameter (">= 6 sampled sweep points")
        for p, k in ((0, 10), (1, 35), (2, 88), (3, 5), (4, 70), (5, 99)):
            x = -ranges[p] + 2 * ranges[p] * k / 99.0
            Pk = [q.copy() for q in Ps]
            Pk[moving] = P0 @ geometry.rigid_transform(**{names[p]: x})
            ref = oracle.evaluate_all(Pk, host, S, S, native=True)["mean"]
            errs.append(abs(values[p, k] - ref) / abs(ref))
        out["parity_rel_err_vs_oracle_at_6_points"] = errs
        out["parity_points"] = "(parameter, step) = (tx,10) (ty,35) (tz,88) (rx,5) (ry,70) (rz,99): all pairs of the pose by oracle/"
    if rank == 0:
        emit(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a child job and forward its one JSON
    line and its exit status.  Nothing here imports torch or makes a HIP call: the ranks are children of a process that
    has never initialised the GPU (an exec / fork from a GPU-initialised process is what the pool forbids)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for raw in proc.stdout:  # rank 0's JSON line goes to stdout exactly once; anything else the ranks print goes to stderr
        t = raw.strip()
        is_json = False
        if t.startswith("{") and t.endswith("}"):
            try:
                is_json = "metric" in json.loads(t)
            except ValueError:
                is_json = False
        if is_json:
            line = t
        else:
            sys.stderr.write(raw)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the %d-rank job ended without a result line\n" % args.gpus)
        rc = 1
    if rc == 0:
        d = json.loads(line)
        if d.get("n_gpus") != args.gpus:
            sys.stderr.write("bench.py: asked for %d ranks, the job reports n_gpus = %r\n" % (args.gpus, d.get("n_gpus")))
            rc = 1
    if rc == 0:
        print(line)
        sys.stdout.flush()
    sys.exit(rc if rc >= 0 else 1)  # (negative: torchrun itself was killed by a signal)


def check_rank_devices(ids, world, single_device):
    """One device per rank: `ids` holds every rank's device identity.  Raises SystemExit('ranks share devices ...')."""
    if not single_device and len(set(ids)) != world:
        raise SystemExit("ranks share devices: %r" % (ids,))


def launch_check(args, rank, world):
    """--launch-check: the process group of the launch and the checks that run before anything is measured -- distinct
    LOCAL_RANKs from the launcher, one device per rank, the ranks' agreement on whether the library's RCCL exchange is
    usable -- without touching a GPU (gloo).  Test hooks: ECC_BENCH_FAIL_RANK (that rank dies before it joins),
    ECC_BENCH_FAKE_SHARED_DEVICE (every rank reports local rank 0's device), ECC_BENCH_FAKE_NO_RCCL_RANK (that rank
    pretends it cannot bind RCCL)."""
    import torch
    import torch.distributed as dist
    fail = os.environ.get("ECC_BENCH_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        os._exit(7)  # a rank that dies before it joins the group
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    seen, local_ranks, rccl = 1, [local_rank], None
    if world > 1:
        dist.init_process_group("gloo")
        probe = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(probe)
        if probe.item() != world * (world + 1) / 2.0:
            raise SystemExit("all-reduce over %d ranks returned %r" % (world, probe.item()))
        seen = dist.get_world_size()
        local_ranks = [None] * world
        dist.all_gather_object(local_ranks, local_rank)
        if sorted(local_ranks) != list(range(world)):  # one node: the launcher must hand out 0 .. N-1
            raise SystemExit("the launcher did not hand out distinct LOCAL_RANKs: %r" % (local_ranks,))
        ids = [None] * world
        dist.all_gather_object(ids, "cuda:%d" % (0 if os.environ.get("ECC_BENCH_FAKE_SHARED_DEVICE") else local_rank))
        check_rank_devices(ids, world, args.single_device)
        # the agreement in front of ecc_comm_create (sharding.RcclComm): all ranks learn the same answer and nobody enters
        # ncclCommInitRank unless everybody can
        from epipolarconsistency_amd import _lib, sharding
        fake = os.environ.get("ECC_BENCH_FAKE_NO_RCCL_RANK")
        local_ok = _lib.lib().ecc_comm_available() == 0 and not (fake is not None and int(fake) == rank)
        agreed = sharding.torch_agree_min()(1 if local_ok else 0)
        answers = [None] * world
        dist.all_gather_object(answers, {"rank": rank, "local": bool(local_ok), "agreed": bool(agreed)})
        rccl = answers
        if len({a["agreed"] for a in answers}) != 1:
            raise SystemExit("the ranks disagree on the RCCL exchange: %r" % (answers,))
    if rank == 0:
        emit({"metric": "launch check (nothing measured)", "value": None, "n_gpus": world,
              "config": {"ranks_seen_by_collective_backend": seen, "launch_check": True, "local_ranks": local_ranks,
                         "rccl_agreement": rccl}})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(args)  # does not return
    claim_stdout()
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s: the job must have exactly --gpus ranks"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "<unset>")))
    if args.launch_check:
        launch_check(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
        return
    import numpy as np
    import torch
    import torch.distributed as dist

    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import geometry, sharding, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.force_collective and world != 1:
        raise SystemExit("--force-collective is the one-rank rehearsal of the N > 1 path: use it with --gpus 1")
    grouped = world > 1 or args.force_collective  # a process group exists and every per-step sum goes through it
    if grouped:
        kw = {}
        if world == 1 and "MASTER_ADDR" not in os.environ:  # one rank outside torchrun: a rendezvous of our own
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                kw = dict(init_method="tcp://127.0.0.1:%d" % sk.getsockname()[1], rank=0, world_size=1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group("gloo", **kw)

    ranks_seen, devices_seen = 1, [torch.cuda.get_device_properties(dev).name + " #%d" % local_rank]
    if grouped:
        # the collective backend must see exactly the ranks the driver asked for, one device each
        if dist.get_world_size() != args.gpus or dist.get_rank() != rank:
            raise SystemExit("process group has %d ranks (this one %d), --gpus %d RANK %d" % (dist.get_world_size(), dist.get_rank(), args.gpus, rank))
        probe = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if abs(probe.item() - world * (world + 1) / 2.0) > 0:
            raise SystemExit("all-reduce over %d ranks returned %r" % (world, probe.item()))
        ranks_seen = dist.get_world_size()
        ids = [None] * world
        try:
            uuid = str(torch.cuda.get_device_properties(dev).uuid)
        except Exception:
            uuid = "cuda:%d" % local_rank
        dist.all_gather_object(ids, "%s (local rank %d)" % (uuid, local_rank))
        devices_seen = ids
        check_rank_devices(ids, world, args.single_device)

    n, S, B = args.views, args.size, args.bins
    pixel_mm = 0.308 * 1024.0 / S
    Ps = synthetic.short_scan(n, S, S, pixel_mm)
    phantom = synthetic.sphere_phantom()

    torch.cuda.set_stream(torch.cuda.Stream(dev))  # a stream of our own, not the legacy default stream
    stream = torch.cuda.current_stream()
    ctx = E.Context(local_rank, stream=stream.cuda_stream)
    ctx.enable_timing(True)  # Radon / pre-processing kernel times below
    try:
        pr = torch.cuda.get_device_properties(dev)
        pci = "%04x:%02x" % (pr.pci_domain_id, pr.pci_bus_id)
    except Exception:
        pci = None
    power = PowerSampler(pci)
    power_first = power.once()  # (watts, MHz) before this process has launched anything of its own

    # ---- Radon intermediates: data-parallel over views, then one all-gather ---------------------
    slab = E.slab_floats(B, B)
    chunk = (n + world - 1) // world
    lo, hi = min(rank * chunk, n), min((rank + 1) * chunk, n)
    slabs_all = torch.zeros((chunk * world, slab), dtype=torch.float32, device=dev)
    local = slabs_all[rank * chunk:(rank + 1) * chunk]
    sub = 50  # projections are generated 50 at a time
    # "ms per Radon intermediate at 1024^2 -> 768^2 bins, device-resident in/out, batch of 400" -- this
    # rank's whole share of the projection stack is resident (1.68 GB at N = 1) before ONE timed call transforms it (the
    # library walks it in sub-batches of 64 images: the transposed copy is scratch of the context)
    imgs_all = torch.empty((hi - lo, S, S), dtype=torch.float32, device=dev)
    for a in range(lo, hi, sub):
        b = min(a + sub, hi)
        imgs_all[a - lo:b - lo] = synthetic.projections_torch(Ps[a:b], S, S, phantom, dev)
    # untimed first call: the context allocates its scratch (transposed image copy, trig table), clocks ramp
    keep = E.RadonIntermediate.compute_into(ctx, imgs_all[:min(sub, hi - lo)], local[:min(sub, hi - lo)], B, B)
    ctx.synchronize()
    power.start()
    keep = E.RadonIntermediate.compute_into(ctx, imgs_all, local[:hi - lo], B, B)
    ctx.synchronize()
    power_radon = power.stop(skip_s=0.05)
    radon_ms = ctx.last_kernel_ms("radon")
    ms_per_radon = radon_ms / max(hi - lo, 1)
    # the same stack in the contracted arithmetic (ecc_radon_set_arithmetic(ECC_RADON_FMA): positions fmaf(t, d, o), lerps
    # as one fma each; bit-identical to the oracle's contracted variant, tests/test_gpu_radon_fma.py).  The metric below is
    # evaluated on the EXACT stack; at N = 1 it is evaluated on this one too and the two means are compared.
    slabs_fma, ms_per_radon_fma = None, 0.0
    if not args.pmc_child:
        slabs_fma = torch.zeros((hi - lo, slab), dtype=torch.float32, device=dev)
        ctx.setRadonArithmetic("fma")
        keep = E.RadonIntermediate.compute_into(ctx, imgs_all, slabs_fma, B, B)
        ctx.synchronize()
        ms_per_radon_fma = ctx.last_kernel_ms("radon") / max(hi - lo, 1)
        ctx.setRadonArithmetic("exact")
    del keep, imgs_all
    if world > 1:
        slabs_fma = None
    # pre-processing (the step in front of the Radon intermediate) on one sub-batch, device
    # resident in and out, reference defaults + cosine weighting: HBM-bound, 8 B per pixel algorithmic
    imgs = synthetic.projections_torch(Ps[lo:min(lo + sub, hi)], S, S, phantom, dev)
    pre_out = torch.empty_like(imgs)
    pp = E.PreProccess()
    for _ in range(2):
        pp.process(ctx, imgs, Ps[lo:min(lo + sub, hi)], out=pre_out)
    ms_per_preprocess = ctx.last_kernel_ms("preprocess") / imgs.shape[0]
    del imgs, pre_out
    if grouped:
        if args.backend == "nccl":
            gathered = torch.empty_like(slabs_all)
            dist.all_gather_into_tensor(gathered, local.contiguous())
        else:  # gloo rehearsal: through host memory
            parts = [torch.empty(local.shape, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, local.cpu())
            gathered = torch.cat(parts).to(dev)
        slabs_all = gathered
        # the gathered stack must be what this rank would have computed itself: recompute one view of the next rank
        probe_view = min(((rank + 1) % world) * chunk, n - 1)
        pimg = synthetic.projections_torch(Ps[probe_view:probe_view + 1], S, S, phantom, dev)
        pslab = torch.zeros((1, slab), dtype=torch.float32, device=dev)
        keep = E.RadonIntermediate.compute_into(ctx, pimg, pslab, B, B)
        ctx.synchronize()
        if not torch.equal(pslab[0], slabs_all[probe_view]):
            raise SystemExit("rank %d: gathered Radon intermediate of view %d differs from a local recomputation" % (rank, probe_view))
        del keep, pimg, pslab
    dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs_all[k], B, B, S, S) for k in range(n)]
    radon_tie = None
    if world == 1 and slabs_fma is not None:  # metric-level tie of the two Radon arithmetic modes on this very data set
        dtrs_fma = [E.RadonIntermediate.wrap_device(ctx, slabs_fma[k], B, B, S, S) for k in range(n)]
        m_fma = E.MetricRadonIntermediate(ctx, Ps, dtrs_fma)
        mean_fma = m_fma.evaluate()
        m_fma.close()
        for d in dtrs_fma:
            d.close()
        del dtrs_fma, slabs_fma
        radon_tie = {"mean_on_fma_stack": mean_fma}
    metric = E.MetricRadonIntermediate(ctx, Ps, dtrs)
    if radon_tie is not None:
        mean_exact = metric.evaluate()
        radon_tie.update({"mean_on_exact_stack": mean_exact, "rel_diff": abs(mean_fma - mean_exact) / abs(mean_exact),
                          "bar": 2e-6})
        if radon_tie["rel_diff"] > 2e-6:
            raise SystemExit("Radon arithmetic modes disagree on the metric: %r" % (radon_tie,))

    n_pairs = n * (n - 1) // 2
    if args.sweep_poses:
        sweep_poses(args, E, geometry, synthetic, dist, torch, np, ctx, metric, Ps, dtrs, n, S, B, rank, world, dev, ranks_seen, devices_seen)
        return

    # ---- shard of the pair range --------------------------------------------------------------
    # cost-balanced contiguous shards (equal-count shards leave rank 0 the straggler: the expensive pairs of a circular
    # scan sit in the first rows of the pair triangle); the same boundaries on every rank, fixed for the whole run
    first, count = sharding.balanced_pair_range(metric, rank, world) if world > 1 else (0, n_pairs)
    sum_t = torch.zeros(1, dtype=torch.float64, device=dev)
    moving = n // 2  # view perturbed per step, like SingleImageMotion does for its input view

    P_pack = E.pack_projection_matrices(Ps)  # (n, 12) float64, what Eigen's Ps[i].data() holds
    P_moving = Ps[moving].copy()
    # what the optimiser hands over: view `moving` perturbed by a small rigid motion, 350 distinct poses prepared
    # ahead (producing a pose is the optimiser's work, not the metric's); the handover itself -- 38 KB of float64
    # into the library, the per-view pre-compute -- is inside every step
    poses = []
    for k in range(350):
        T = geometry.rigid_transform(tx=0.01 * (k % 50), rz=1e-4 * (k % 7))
        Pk = P_pack.copy()
        Pk[moving] = (P_moving @ T).T.reshape(12)
        poses.append(Pk)

    # N > 1: the per-evaluation exchange of the 8-byte partial sums
    exchange = None
    if grouped and args.exchange in ("both", "shm"):
        ok = True
        try:
            exchange = sharding.open_exchange(rank, world, dist.barrier)
        except Exception as e:  # no usable /dev/shm
            sys.stderr.write("rank %d: shared-memory exchange unavailable (%s)\n" % (rank, e))
            exchange, ok = None, False
            if rank == 0:
                dist.barrier()  # the one open_exchange did not reach
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 0:
            exchange = None  # all ranks take the collective
        elif abs(exchange.sum(float(rank + 1)) - world * (world + 1) / 2.0) > 1e-12:
            raise SystemExit("shared-memory exchange returned a wrong sum")

    # the library's own RCCL communicator (ecc_comm_*): the all-reduce queued by the library between its sum kernel and the
    # kernel that publishes the scalar -- no torch.distributed call on the step's path
    comm, comm_status = None, "not attempted"
    if grouped and args.backend == "nccl" and args.exchange in ("both", "rccl"):
        # (the ranks first AGREE that all of them can bind RCCL -- a rank that cannot must not leave the others inside
        # ncclCommInitRank --, and this rank gives the collective two minutes: it has never run on more than one GPU)
        try:
            comm = sharding.RcclComm(ctx, rank, world, sharding.torch_broadcast_bytes(dev), agree=sharding.torch_agree_min(dev),
                                     timeout_s=120.0)
            comm_status = "ncclCommInitRank ok"
        except Exception as e:
            sys.stderr.write("rank %d: the library's RCCL communicator is unavailable (%s)\n" % (rank, e))
            comm_status = "%s: %s" % (type(e).__name__, e)
            comm = None
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 0:
            if comm is not None:
                comm_status += " (dropped: another rank has none)"
            comm = None
        elif abs(metric.evaluate_range_allreduce(comm, 0, 0) - 0.0) > 0:
            raise SystemExit("the library's all-reduce of empty shards returned a non-zero sum")

    def make_step(mode):
        def step(k):
            metric.setProjectionMatrices(poses[k % len(poses)])
            if not grouped:
                return metric.evaluate()
            if mode == "shm":
                return sharding.exchanged_evaluate(metric, n, exchange, shard=(first, count))
            if mode == "rccl":
                return sharding.rccl_evaluate(metric, n, comm, shard=(first, count))
            # (the context's stream is torch's current stream here, so the reduced scalar can come back through the
            # metric's pinned result slot; with gloo the reduction runs on the host anyway)
            return sharding.distributed_evaluate(metric, n, sum_t, rank, world, shard=(first, count), publish=args.backend == "nccl")
        return step

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_block(step, k0):
        """Exactly args.steps steps between two fences; seconds, MAX over ranks."""
        fence()
        t0 = time.perf_counter()
        last = None
        for k in range(k0, k0 + args.steps):
            last = step(k)
        fence()
        el = time.perf_counter() - t0
        if grouped:
            e = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
            el = e.item()
        return el, last

    def measure(mode):
        """W warm-up steps, then blocks of K steps: the first is the cold number, the median the steady one."""
        step = make_step(mode)
        for k in range(args.warmup):
            step(k)
        blocks, last = [], None
        el, last = timed_block(step, 0)
        blocks.append(el)
        # automatic: about 1.2 s of timed steps, 5 to 60 blocks -- the driver's 20-step blocks last 7 ms each and the clocks
        # of a box that idled through the host-side set-up take a few tenths of a second to settle (round 4: blocks 1-5
        # 0.40-0.45 ms per step, 6-9 0.364-0.370; round 5, one box: eight 67-ms blocks falling from 0.333 to 0.312 ms per step,
        # the median of nine on the ramp); every block's time is in timing.blocks_ms_per_step
        n_blocks = args.blocks if args.blocks > 0 else int(min(60, max(5, 1.2 / max(el, 1e-6))))
        if grouped:  # every rank must run the same number of blocks
            nb = torch.tensor([n_blocks], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
            dist.broadcast(nb, 0)
            n_blocks = int(nb.item())
        for b in range(1, n_blocks):
            el, last = timed_block(step, b * args.steps)
            blocks.append(el)
        steady = sorted(blocks)[len(blocks) // 2]
        return dict(cold=blocks[0], steady=steady, blocks=blocks, last=last, step=step)

    ctx.enable_timing(False)  # no event records inside the timed region (they break back-to-back dispatch)
    if not grouped:
        modes = ["single"]
    elif args.exchange == "both":
        modes = [m for m in ("shm", "collective", "rccl") if (m != "shm" or exchange is not None) and (m != "rccl" or comm is not None)]
    elif args.exchange == "shm" and exchange is not None:
        modes = ["shm"]
    elif args.exchange == "rccl" and comm is not None:
        modes = ["rccl"]
    else:
        modes = ["collective"]
    results = {m: measure(m) for m in modes}
    # N > 1: the headline is the step with the all-reduce of the partial sums (RCCL with --backend nccl), the exchange
    # north_star names; the shared-memory exchange is reported next to it under timing.other_exchange
    # the headline is the RCCL all-reduce: issued by the library where its communicator exists (one call per step, everything
    # stream-ordered), else through torch.distributed
    # Which exchange is the headline.  The library-issued RCCL all-reduce (one call per step, everything stream-ordered) when its
    # communicator exists on every rank -- but only after THIS run has shown, below, that it returns what torch.distributed's
    # all-reduce and the shared-memory exchange return for the same step (round 5's advisor: it had only ever run with one
    # rank; the ranks now agree on its availability first, its creation is under a watchdog, and a disagreement ends the run).
    best = "rccl" if "rccl" in modes else ("collective" if "collective" in modes else modes[0])
    same_step = {m: [results[m]["step"](k) for k in (3, 4, 11)] for m in modes}  # the same poses through every exchange (every rank, the same order)
    for m in modes:
        for x, y in zip(same_step[m], same_step[modes[0]]):
            if not abs(x - y) <= 1e-12 * abs(y):
                raise SystemExit("the exchanges disagree on the same step: %r" % (same_step,))
    res = results[best]
    elapsed, last, step = res["steady"], res["last"], res["step"]
    # pair-kernel duration: HIP events on the context's stream around the pair kernel alone, averaged over a
    # second, untimed pass over the same steps (at most 50)
    ctx.enable_timing(True)
    pair_ms, n_timed = 0.0, max(1, min(args.steps, 50))
    for k in range(n_timed):
        step(k)
        pair_ms += ctx.last_kernel_ms("pairs")
    fence()
    pair_ms /= n_timed
    pair_s = pair_ms * 1e-3
    # socket power and engine clock while the same steps run back to back (untimed, ~1.5 s; the first 0.5 s dropped: the
    # reading is a moving average and the controller takes about a second to settle at the cap)
    ctx.enable_timing(False)
    n_power = 0 if (args.no_power or args.pmc_child) else int(min(20000, max(200, 1.5 / max(elapsed / args.steps, 1e-6))))  # (elapsed is the MAX over ranks: the same count everywhere)
    fence()
    power.start()
    t_p = time.perf_counter()
    for k in range(n_power):
        step(k)
    fence()
    t_p = time.perf_counter() - t_p
    power_steps = power.stop(skip_s=0.5) if n_power else (power.stop() and None)
    if power_steps:
        power_steps["ms_per_step_while_sampling"] = 1e3 * t_p / n_power
        power_steps["steps"] = n_power
    ctx.enable_timing(True)

    # the same steps with the record reuse switched off (every step refits all pairs and launches e1_kernel): the
    # results are bit-identical, only the fixed cost per step differs
    ctx.enable_timing(False)
    metric.setRecordReuse(False)
    for k in range(args.warmup):
        step(k)
    off_blocks = []
    for b in range(3):
        el_off, last_off = timed_block(step, b * args.steps)
        off_blocks.append(el_off)
    v_off = [step(k) for k in (3, 4, 4, 11)]
    metric.setRecordReuse(True)
    v_on = [step(k) for k in (3, 4, 4, 11)]
    reuse_off_s = sorted(off_blocks)[1]
    if v_on != v_off:
        raise SystemExit("record reuse changed the results: %r vs %r" % (v_on, v_off))

    # ---- socket power and engine clock (hwmon) --------------------------------------------------------------------------
    # Both kernels run INTO THE POWER CAP: the socket reads 1395-1399 W of its 1400 W while the steps run back to back and
    # the engine clock settles near 2.2 GHz, below the 2.4 GHz every roof here is priced at.  At the cap a kernel's time
    # follows the ENERGY of its instruction and data stream, not the issue slots of one pipe: cutting instructions returns
    # well under its share (CHANGELOG.md 4.2 has five rounds of such cuts), and idle XCDs at the end of a launch cost
    # nothing (the remaining ones clock up: CHANGELOG.md round 5, XCD schedule).  `frac_at_measured_clock` restates the two
    # clock-bound roofs against the clock the chip actually ran at.
    power_report = None
    if power.dir and rank == 0:
        cap, rated = power.cap_w(), power.rated_mhz()
        power_report = {"source": "amdgpu hwmon power1_input / freq1_input, 2-ms samples, outside the timed blocks",
                        "cap_w": cap, "sclk_rated_mhz": rated,
                        "before_first_launch": {"w": power_first[0], "sclk_mhz": power_first[1]} if power_first else None,
                        "pair_steps": power_steps, "radon_stack": power_radon,
                        "radon_stack_note": "one call of about 0.2 s: the power reading (a moving average) is still rising when it ends"}
        if power_steps and cap:
            power_report["pair_steps"]["frac_of_cap"] = power_steps["avg_w"] / cap

    n_kappa = n_kappa_auto(S, S, B)
    bytes_per_pair = 64 * n_kappa + 68            # 2 views x 2 signs x 4 taps x 4 B + K01 + result
    launch_bytes = bytes_per_pair * count         # one launch = this rank's shard of pairs
    achieved = launch_bytes / pair_s / 1e9 if pair_ms > 0 else 0.0
    exch_name = {"single": "none", "shm": "host shared memory", "collective": "all-reduce (%s)" % args.backend,
                 "rccl": "all-reduce (RCCL, issued by the library: ecc_metric_evaluate_range_allreduce)"}

    # ---- which roof bounds the pair kernel -------------------------------------------------------------------
    # The gather is served on chip (measured HBM traffic << algorithmic bytes), so the algorithmic bytes are priced
    # against the roof they actually cross -- the CU's vector L1 -- and the kernel's instruction stream against the
    # vector-ALU issue rate; `bound` is the roof with the larger fraction.  Instruction counts and HBM bytes per
    # launch are PMC measurements (separate rocprofv3 --pmc passes of this bench, profiles/pmc_current.json); they
    # are used only at N = 1 on the workload they were taken on.
    l1_peak = L1_BYTES_PER_CLK_CU * N_CU * ENGINE_CLOCK_GHZ  # GB/s
    roofs = {"l1": {"achieved": achieved, "peak": l1_peak, "unit": "GB/s", "frac": achieved / l1_peak,
                    "note": "algorithmic gather bytes through the vector L1: %.0f B/clk/CU x %d CUs x %.1f GHz"
                            % (L1_BYTES_PER_CLK_CU, N_CU, ENGINE_CLOCK_GHZ)}}
    pmc = load_pmc("pairs_kernel<true, false>") if (world == 1 and (n, S, B) == (400, 1024, 768)) else None
    pmc_origin = "separate rocprofv3 --pmc passes of bench.py --steps 5, profiles/pmc_current.json (%s); not this run" % (pmc or {}).get("_tag")
    live = live_pmc(args) if (world == 1 and rank == 0 and not args.no_live_pmc) else None
    if live:
        pmc = {"SQ_INSTS_VALU": live["SQ_INSTS_VALU"], "FETCH_SIZE": live["FETCH_SIZE"], "WRITE_SIZE": 4.0 * count / 1024.0,
               "SQ_INSTS_VMEM_RD": live["SQ_INSTS_VMEM_RD"], "_tag": "live"}
        pmc_origin = ("measured in this run: one rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_VMEM_RD pass over "
                      "a child run of this bench (--steps 3), %d launches averaged; WRITE_SIZE taken as 4 B per pair" % live["dispatches"])
    traffic, traffic_src = None, None
    if pmc and "SQ_INSTS_VALU" in pmc and pair_ms > 0:
        valu_peak = N_CU * SIMD_PER_CU * ENGINE_CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INSTR  # G wave-instructions/s
        valu_ach = pmc["SQ_INSTS_VALU"] / pair_s / 1e9
        roofs["valu"] = {"achieved": valu_ach, "peak": valu_peak, "unit": "G wave-instr/s", "frac": valu_ach / valu_peak,
                         "note": "SQ_INSTS_VALU = %.4g per launch (PMC, %s) at %.0f cycles per wave64 instruction on %d SIMDs"
                                 % (pmc["SQ_INSTS_VALU"], pmc["_tag"], VALU_CYCLES_PER_WAVE_INSTR, N_CU * SIMD_PER_CU)}
        if "SQ_INSTS_VMEM_RD" in pmc:
            roofs["l1"]["gathers_per_launch"] = pmc["SQ_INSTS_VMEM_RD"]
    # L2 hit rate of the pair kernel: a second, separate pass (the TCC counters do not fit beside FETCH_SIZE)
    live_tcc = live_pmc(args, ("TCC_HIT_sum", "TCC_MISS_sum")) if live else None
    if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic = int(2 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024)
        traffic_src = "2 x FETCH_SIZE + WRITE_SIZE (KiB -> B; gfx950 counts 128-B fabric reads as 64 B); " + pmc_origin
        roofs["hbm_measured"] = {"achieved": traffic / pair_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": traffic / pair_s / 1e9 / HBM_PEAK_GBS,
                                 "achievable": HBM_ACHIEVABLE_GBS,
                                 "frac_of_achievable": traffic / pair_s / 1e9 / HBM_ACHIEVABLE_GBS,
                                 "note": "bytes the launch moved through the fabric (PMC) per second; `achievable` = what a float4 copy "
                                         "reaches on this part (the guide's measured 6.29 TB/s): the roof a kernel can actually touch"}
    # SURVEY.md 8(d)'s compulsory figure: every Radon intermediate read once + the per-view geometry + one float per pair
    hbm_compulsory = 4 * n * B * B + 64 * n + 4 * count
    if power_report and power_report.get("pair_steps"):
        scale = ENGINE_CLOCK_GHZ * 1e3 / power_report["pair_steps"]["sclk_mhz_avg"]
        for r in ("l1", "valu"):
            if r in roofs:
                roofs[r]["frac_at_measured_clock"] = roofs[r]["frac"] * scale
                roofs[r]["measured_clock_mhz"] = power_report["pair_steps"]["sclk_mhz_avg"]
    # The binding roof: every roof priced against what can actually be reached -- the measured HBM bytes against the achievable
    # bandwidth, not the spec sheet's (round 5's verdict: against 6.29 TB/s HBM is the tightest of the three).
    bound_key = max(roofs, key=lambda r: roofs[r].get("frac_of_achievable", roofs[r]["frac"]))
    bound = {"hbm_measured": "hbm"}.get(bound_key, bound_key)
    # on-chip / off-chip split: the same 79 800 pair geometries sampling only EIGHT Radon intermediates (view v -> v mod 8: 39 MB of
    # row-paired copies, resident in the Infinity Cache) -- the launch with its HBM misses taken away
    cache_resident_ms, one_slab_ms = None, None
    if world == 1 and rank == 0 and not args.pmc_child and n >= 16:
        ii, jj = np.triu_indices(n, 1)
        out8 = np.empty(len(ii), np.float32)
        ctx.enable_timing(True)
        res_ms = {}
        for alias in (8, 1):
            m8 = E.MetricRadonIntermediate(ctx, Ps, dtrs[:alias])
            idx8 = np.ascontiguousarray(np.stack([ii, jj, ii % alias, jj % alias], axis=1).astype(np.int32))
            ts = []
            for rep in range(6):
                m8.evaluate(idx8, out8)
                ts.append(ctx.last_kernel_ms("pairs"))
            res_ms[alias] = float(np.median(ts[1:]))
            m8.close()
        cache_resident_ms, one_slab_ms = res_ms[8], res_ms[1]
    roofline = {"bound": bound, "bound_roof": bound_key, "achieved": roofs[bound_key]["achieved"], "peak": roofs[bound_key]["peak"],
                "unit": roofs[bound_key]["unit"], "frac": roofs[bound_key]["frac"],
                "frac_of_achievable": roofs[bound_key].get("frac_of_achievable"), "traffic": traffic,
                "kernel_ms_cache_resident": cache_resident_ms, "kernel_ms_one_slab": one_slab_ms,
                "kernel_ms_cache_resident_note": "pairs_kernel over the same pair geometries with every view's Radon intermediate aliased to "
                                                 "one of 8 (index list i, j, i mod 8, j mod 8; 39 MB of row-paired copies: Infinity-Cache resident) -- what the launch takes when "
                                                 "nothing misses to HBM; kernel_ms_one_slab: all views aliased to ONE (4.9 MB: mostly L2 resident)",
                "traffic_source": traffic_src, "kernel": "pairs_kernel<true, false>", "kernel_ms": pair_ms,
                "algorithmic_bytes_per_launch": launch_bytes, "roofs": roofs,
                "hbm_compulsory_bytes": hbm_compulsory,
                "traffic_over_compulsory": (traffic / hbm_compulsory) if traffic else None,
                "l2_hit_rate": (live_tcc["TCC_HIT_sum"] / (live_tcc["TCC_HIT_sum"] + live_tcc["TCC_MISS_sum"]))
                if (live_tcc and live_tcc["TCC_HIT_sum"] + live_tcc["TCC_MISS_sum"] > 0) else None,
                "l2_hit_rate_source": ("measured in this run: a second rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass over a child run "
                                       "of this bench, %d launches averaged" % live_tcc["dispatches"]) if live_tcc else None,
                # SURVEY.md 8(d)'s figure against HBM peak: > 1 means the gather's reuse is captured on chip
                "hbm_algorithmic": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "reuse_factor_vs_hbm_peak": achieved / HBM_PEAK_GBS}}

    # ---- second headline metric: ms per Radon intermediate ---------------------------------------------------
    # The Radon kernel reads its bilinear footprints from an LDS tile of texel pairs with ds_read_b64 (2 per fetch,
    # 16 B); since round 3 its binding roof is vector-ALU issue, not the LDS pipe.  `frac` is the binding roof's
    # fraction; the algorithmic LDS bytes are priced against both LDS rates for continuity with rounds 1-2.
    fetches = radon_fetches_per_image(S, S, B, B)
    lds_peak_b32 = LDS_READ2_B32_BYTES_PER_CLK_CU * N_CU * ENGINE_CLOCK_GHZ
    lds_peak_b64 = LDS_READ_B64_BYTES_PER_CLK_CU * N_CU * ENGINE_CLOCK_GHZ
    lds_ach = 16.0 * fetches / (ms_per_radon * 1e-3) / 1e9 if ms_per_radon > 0 else 0.0
    roofs_r = {"lds_algorithmic_b64": {"achieved": lds_ach, "peak": lds_peak_b64, "unit": "GB/s", "frac": lds_ach / lds_peak_b64,
                                       "note": "16 B per bilinear fetch (2 x ds_read_b64) against %.0f B/clk/CU x %d CUs x %.1f GHz"
                                               % (LDS_READ_B64_BYTES_PER_CLK_CU, N_CU, ENGINE_CLOCK_GHZ)},
               "lds_algorithmic_b32": {"achieved": lds_ach, "peak": lds_peak_b32, "unit": "GB/s", "frac": lds_ach / lds_peak_b32,
                                       "note": "the same bytes against the ds_read2_b32 rate (128 B/clk/CU) rounds 1-2 were priced on"}}
    roofline_radon = {"kernel": "radon_kernel<true, false> (derivative, exact arithmetic)", "kernel_ms_per_image": ms_per_radon,
                      "algorithmic_bytes_per_image": 16 * fetches, "bilinear_fetches_per_image": fetches,
                      "compulsory_hbm_bytes_per_image": 4 * (S * S + B * B)}
    rp = load_pmc("radon_kernel<true, false>") if (S, B) == (1024, 768) else None
    if rp and rp.get("SQ_LDS_IDX_ACTIVE"):
        roofline_radon["lds_bank_conflict_ratio"] = rp["SQ_LDS_BANK_CONFLICT"] / rp["SQ_LDS_IDX_ACTIVE"]
        roofline_radon["pmc_source"] = ("SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_INSTS_VALU per 50-image launch: separate rocprofv3 --pmc "
                                        "passes, profiles/pmc_current.json (%s); not this run" % rp["_tag"])
        # the committed passes (scripts/pmc_radon.sh on scripts/bench_radon.py) profile launches of 50 images each
        img_s = ms_per_radon * 1e-3
        roofs_r["lds_pipe_active"] = {"frac": rp["SQ_LDS_IDX_ACTIVE"] / sub / (N_CU * ENGINE_CLOCK_GHZ * 1e9 * img_s),
                                      "note": "SQ_LDS_IDX_ACTIVE cycles per image / (256 CUs x 2.4 GHz x seconds per image): bank "
                                              "conflicts included"}
        if rp.get("SQ_INSTS_VALU"):
            valu_peak = N_CU * SIMD_PER_CU * ENGINE_CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INSTR  # G wave-instructions/s
            valu_ach = rp["SQ_INSTS_VALU"] / sub / img_s / 1e9
            roofs_r["valu"] = {"achieved": valu_ach, "peak": valu_peak, "unit": "G wave-instr/s", "frac": valu_ach / valu_peak,
                               "note": "SQ_INSTS_VALU = %.4g per image at 2 cycles per wave64 instruction on 1024 SIMDs (the 4-cycle "
                                       "v_floor_f32 count as 2: a lower bound of the pipe's occupancy)" % (rp["SQ_INSTS_VALU"] / sub)}
    # the contracted-arithmetic mode (ecc_radon_set_arithmetic(ECC_RADON_FMA)) against the same two roofs, from its own PMC pass
    rf = load_pmc("radon_kernel<true, true>") if (S, B) == (1024, 768) else None
    if rf and rf.get("SQ_INSTS_VALU") and rf.get("SQ_LDS_IDX_ACTIVE") and ms_per_radon_fma > 0:
        img_f = ms_per_radon_fma * 1e-3
        roofline_radon["fma_mode"] = {
            "kernel": "radon_kernel<true, true>", "kernel_ms_per_image": ms_per_radon_fma,
            "valu_frac": rf["SQ_INSTS_VALU"] / sub / img_f / 1e9 / (N_CU * SIMD_PER_CU * ENGINE_CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INSTR),
            "lds_pipe_active_frac": rf["SQ_LDS_IDX_ACTIVE"] / sub / (N_CU * ENGINE_CLOCK_GHZ * 1e9 * img_f),
            "wave_instructions_per_image": rf["SQ_INSTS_VALU"] / sub,
            "note": "both pipes are 0.6-0.7 busy: the contracted loop (40 instead of 52 vector instructions per step) is no longer "
                    "bound by vector issue alone"}
    binding = max(roofs_r, key=lambda r: roofs_r[r]["frac"] if r != "lds_algorithmic_b32" else -1.0)
    roofline_radon.update({"bound": {"valu": "valu", "lds_pipe_active": "lds"}.get(binding, "lds"), "binding": binding,
                           "achieved": roofs_r[binding].get("achieved"), "peak": roofs_r[binding].get("peak"),
                           "unit": roofs_r[binding].get("unit"), "frac": roofs_r[binding]["frac"], "traffic": None,
                           "roofs": roofs_r})

    out = {
        "metric": "ECC evaluations/sec (N=%d, %d^2 projections)" % (n, S),
        "value": args.steps / elapsed,
        "unit": "evaluations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%d-projection %dx%d circular short scan, %dx%d Radon bins, all %d pairs"
                               % (n, S, S, B, B, n_pairs),
                   "radon_arithmetic": "exact (the Radon intermediates the timed evaluations sample; ms_per_radon_intermediate is this mode)",
                   "n_kappa_per_pair": n_kappa, "pairs_per_rank": count, "parallelism": "pair-shard x%d (contiguous, cost-balanced)" % world,
                   "sum_exchange": exch_name[best],
                   "metric_device_bytes": dict(metric.device_bytes(), radon_intermediates=int(4 * n * E.slab_floats(B, B)),
                                               note="ecc_metric_device_bytes: the row-paired and row-quad sampling copies the metric "
                                                    "owns (2x and 4x the stack; include/ecc_hip.h, ECC_QUAD_COPIES_*) and its scratch"),
                   # the step moves one view like the reference's optimiser loop does; the library's opt-in pose-delta mode
                   # (ecc_metric_set_incremental) would re-evaluate 399 pairs instead of all -- it is NOT used here
                   "pose_delta_evaluation": "off: every step evaluates all %d pairs" % n_pairs,
                   # library default: the per-pair geometry records of pairs whose two matrices did not change since the
                   # last step are kept (one view moves per step: 399 of 79 800 pairs are refitted, E1 of the moved view
                   # on the host, no e1_kernel launch); every pair is sampled; bit-identical results
                   "k01_record_reuse": "on (library default); the same run with it off under timing.record_reuse_off",
                   "ranks_seen_by_collective_backend": ranks_seen, "devices": devices_seen,
                   "force_collective": bool(args.force_collective)},
        "timing": {"value_is": "median of %d blocks of %d steps" % (len(res["blocks"]), args.steps),
                   # round 1's definition of the headline: all timed steps of the run over all their time
                   "whole_run": {"ms_per_step": 1e3 * sum(res["blocks"]) / (args.steps * len(res["blocks"])),
                                 "value": args.steps * len(res["blocks"]) / sum(res["blocks"])},
                   "blocks_ms_per_step": [1e3 * b / args.steps for b in res["blocks"]],
                   "cold": {"ms_per_step": 1e3 * res["cold"] / args.steps, "value": args.steps / res["cold"],
                            "note": "first block after the %d warm-up steps" % args.warmup},
                   "record_reuse_off": {"ms_per_step": 1e3 * reuse_off_s / args.steps, "value": args.steps / reuse_off_s,
                                        "non_pair_kernel_us_per_step": 1e3 * (1e3 * reuse_off_s / args.steps - pair_ms),
                                        "note": "median of 3 blocks with ecc_metric_set_record_reuse(0): e1_kernel + k01_kernel "
                                                "over all pairs every step; same result bits"}},
        "roofline": roofline,
        "roofline_radon": roofline_radon,
        "power": power_report,
        "ms_per_radon_intermediate": ms_per_radon,
        "ms_per_radon_intermediate_by_arithmetic": {
            "exact": ms_per_radon, "fma": ms_per_radon_fma,
            "headline_mode": "exact (library default: unfused fp32, bit-identical to the oracle's normative variant); fma = "
                             "ecc_radon_set_arithmetic(ECC_RADON_FMA), bit-identical to the oracle's contracted variant",
            "metric_tie": radon_tie},
        "ms_per_preprocess": ms_per_preprocess,
        "preprocess_hbm_GBs": 8.0 * S * S / (ms_per_preprocess * 1e-3) / 1e9 if ms_per_preprocess > 0 else 0.0,
        "pairs_per_s": n_pairs * args.steps / elapsed,
        "kappa_samples_per_s": n_pairs * n_kappa * args.steps / elapsed,
        "non_pair_kernel_us_per_step": 1e3 * (1e3 * elapsed / args.steps - pair_ms),
        "pairs_evaluated_last_step": metric.last_evaluated_pairs(),  # the library's own count: this rank's whole shard
        "last_value": last,
    }
    others = [{"sum_exchange": exch_name[m], "ms_per_step": 1e3 * results[m]["steady"] / args.steps,
               "value": args.steps / results[m]["steady"], "cold_ms_per_step": 1e3 * results[m]["cold"] / args.steps}
              for m in modes if m != best]
    out["timing"]["other_exchange"] = others  # always a list (advisor, round 5: one schema whatever the number of modes)
    if grouped:
        # what every rank actually used: the exchanges it timed, the one behind `value`, and how its ncclCommInitRank went
        mine_status = {"rank": rank, "exchanges_timed": [exch_name[m] for m in modes], "value_is": exch_name[best],
                       "ecc_comm_create": comm_status}
        statuses = [None] * world
        dist.all_gather_object(statuses, mine_status)
        out["config"]["exchange_per_rank"] = statuses

    # ---- CPU baseline: the oracle timed on this box's host cores (rank 0, N = 1 only) -----------
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        import oracle
        oracle.build(native=True)
        cpu = host_cpu_info()
        # threads = physical cores, but no more than this process may run on (affinity mask, cgroup quota)
        threads = min(cpu["physical_cores"] or cpu["usable_cpus"], cpu["usable_cpus"])
        if cpu["cgroup_cpu_quota"]:
            threads = max(1, min(threads, int(cpu["cgroup_cpu_quota"])))
        oracle.lib(native=True).eccor_set_num_threads(int(threads))
        views = list(range(0, n, args.cpu_sample_stride))
        host_dtrs = [dtrs[v].readback() for v in views]
        Psub = [Ps[v] for v in views]
        oracle.evaluate_all(Psub[:8], host_dtrs[:8], S, S, native=True)  # warm-up (thread pool, page-in)
        reps, cpu_s, ref = 0, 0.0, None
        while reps < 3 or (cpu_s < 12.0 and reps < 60):  # bounded: 10-15 s of CPU work
            t1 = time.perf_counter()
            ref = oracle.evaluate_all(Psub, host_dtrs, S, S, native=True)
            cpu_s += time.perf_counter() - t1
            reps += 1
        sub_pairs = len(views) * (len(views) - 1) // 2
        cpu_evals = (sub_pairs * reps / cpu_s) / n_pairs
        # the same sub-problem on the GPU doubles as a full-size parity check
        metric.setProjectionMatrices(Ps)
        gpu_mean = metric.evaluate(set(views)) if args.cpu_sample_stride > 1 else metric.evaluate()
        out["cpu_baseline"] = {
            "value": cpu_evals, "unit": "evaluations/s", "cores": oracle.lib(native=True).eccor_num_threads(), "kind": "port",
            "cpu_model": cpu["model"], "sockets": cpu["sockets"], "physical_cores": cpu["physical_cores"],
            "logical_cpus": cpu["logical_cpus"], "usable_cpus": cpu["usable_cpus"], "cgroup_cpu_quota": cpu["cgroup_cpu_quota"],
            "threads_used": oracle.lib(native=True).eccor_num_threads(),
            "sample": "%d x all %d pairs among every %d-th view (%d of %d pairs per evaluation) in %.1f s, "
                      "scaled by pair count; oracle/ecc_oracle.c -O3 -march=native -fopenmp"
                      % (reps, sub_pairs, args.cpu_sample_stride, sub_pairs, n_pairs, cpu_s),
        }
        out["parity_rel_err_vs_oracle_on_sample"] = abs(gpu_mean - ref["mean"]) / abs(ref["mean"])
        if args.cpu_sample_stride == 1:
            # the n x n cost image is what the reference hands its callers (ref: ...RadonIntermediate.cpp:214-221, plotted by
            # Gui/Visualization.h:18-56): every single pair value of the timed path against the oracle's
            _, gpu_pairs = metric.evaluate_range(0, n_pairs, want_pairs=True)
            ref_pairs = np.asarray(ref["pairs"], np.float64)
            rel = np.abs(gpu_pairs.astype(np.float64) - ref_pairs) / np.maximum(np.abs(ref_pairs), 1e-300)
            out["pair_rel_err_max"], out["pair_rel_err_p99"], out["pair_rel_err_p50"] = (
                float(rel.max()), float(np.percentile(rel, 99)), float(np.percentile(rel, 50)))
            out["pair_rel_err_note"] = ("all %d pair values of the timed (polynomial) path vs oracle/; single values carry the fp32 "
                                        "rounding of the sample positions (DESIGN.md 2), the mean averages it out" % n_pairs)
            # WHOSE error is that?  The normative oracle computes a sample's (angle, distance) in fp32 like the reference
            # (ref: ...RadonIntermediate.cu:71-113, EpipolarConsistencyCommon.hxx:152-171); its variant 1 does the same mapping
            # in binary64 and rounds once -- the noise-free pair values of the same formula.  Three distributions over all pairs:
            # timed path vs normative (the one above), normative vs variant 1 (the reference arithmetic's own fp32 rounding),
            # timed path vs variant 1 (this library's error against the noise-free values).
            oracle.set_variant(1, native=True)
            try:
                ref64 = oracle.evaluate_all(Psub, host_dtrs, S, S, native=True)
            finally:
                oracle.set_variant(0, native=True)
            p64 = np.asarray(ref64["pairs"], np.float64)

            def dist3(a, b):
                r = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
                return {"max": float(r.max()), "p99": float(np.percentile(r, 99)), "p50": float(np.percentile(r, 50))}
            g64 = gpu_pairs.astype(np.float64)
            out["pair_rel_err_attribution"] = {
                "timed_vs_normative_oracle": dist3(g64, ref_pairs),
                "normative_oracle_vs_float64_geometry": dist3(ref_pairs, p64),
                "timed_vs_float64_geometry": dist3(g64, p64),
                "mean_rel_err": {"timed_vs_normative": abs(gpu_mean - ref["mean"]) / abs(ref["mean"]),
                                 "normative_vs_float64_geometry": abs(ref["mean"] - ref64["mean"]) / abs(ref64["mean"]),
                                 "timed_vs_float64_geometry": abs(gpu_mean - ref64["mean"]) / abs(ref64["mean"])},
                "note": "oracle variant 1 = the reference's line -> (angle, distance) mapping in binary64, rounded once "
                        "(oracle/ecc_oracle.c or_redundancy_f64): when the timed path is closer to it than the normative oracle "
                        "is, a cost-image entry carries the REFERENCE arithmetic's fp32 rounding, not this library's"}
        # Radon baseline: 1/4 of the bins of one image
        img = synthetic.projections_numpy([Ps[n // 3]], S, S, phantom)[0]
        bins = np.arange(0, B * B, 4, dtype=np.int32)
        t1 = time.perf_counter()
        oracle.radon_bins(img, B, B, bins, native=True)
        out["cpu_baseline"]["ms_per_radon_intermediate"] = 1e3 * (time.perf_counter() - t1) * 4

    if rank == 0:
        emit(out)
    if grouped:
        if comm is not None:
            comm.close()  # ncclCommDestroy here, on every rank, not from __del__ at interpreter exit
        dist.barrier()
        if exchange is not None:
            exchange.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # a failing rank must take the job down with a non-zero status, never hang the others
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        if isinstance(e, SystemExit) and isinstance(e.code, int):  # the self-launching parent handing on its job's status
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(e.code)
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
