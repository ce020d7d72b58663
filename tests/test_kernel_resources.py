"""The register budgets the launch design of the pair path depends on, read from the built library's code objects (no GPU):
scripts/kernel_resources.py parses the offload bundles and the AMDGPU metadata notes of libecc_hip.so.

  * pairs_kernel: at most 96 scalar registers (+16 the hardware keeps per wave: 7 x 112 <= 800) and at most 72
    vector registers (7 x 72 <= 512) -- seven waves per SIMD; no scratch (8 bytes of it are a reload per trip inside the sampling
    loops: +14 % per launch); and not fewer than 65 vector registers in the sum-of-squares kernels (see the test).
  * k01_kernel<8 / 16>, k01_patched_kernel (the refit of a moved view's pairs on the side stream): at most 80 vector registers,
    the hole one retiring pairs_kernel workgroup leaves (72 + the 8 that were free); with 126 the refit found no room until the
    all-pairs launch had drained and the step paid for it in full (CHANGELOG, round 5).
  * pairs_split_kernel<*, 4> (the moved pairs' own launch beside the all-pairs launch): at most 80 as well."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resources():
    pytest.importorskip("msgpack")  # scripts/kernel_resources.py decodes the AMDGPU metadata notes with it
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = os.path.join(ROOT, "epipolarconsistency_amd", "libecc_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libecc_hip.so not built")
    return mod, mod.kernels(lib)


def test_pair_kernel_runs_seven_waves_per_simd():
    mod, ks = _resources()
    main = mod.find(ks, "12pairs_kernelILb")
    assert len(main) == 4
    for name, k in main.items():
        assert k[".sgpr_count"] <= 96, (name, k[".sgpr_count"])
        assert k[".vgpr_count"] <= 72, (name, k[".vgpr_count"])
        # (the sum-of-squares kernels: below 65 registers the compiler's scheduler has aimed at eight waves per SIMD, which the
        # scalar registers do not allow, and given up the overlap of a trip's eight gathers for it: 7 % per launch, CHANGELOG round 5)
        if "ELb0EEE" in name:
            assert k[".vgpr_count"] >= 65, (name, k[".vgpr_count"])
        assert k[".private_segment_fixed_size"] == 0, name
        assert k[".group_segment_fixed_size"] == 0, name


def test_side_stream_kernels_fit_beside_the_pair_kernel():
    mod, ks = _resources()
    wide = {**mod.find(ks, "10k01_kernelILi8E"), **mod.find(ks, "10k01_kernelILi16E"), **mod.find(ks, "18k01_patched_kernel")}
    assert len(wide) == 4
    for name, k in wide.items():
        assert k[".vgpr_count"] <= 80, (name, k[".vgpr_count"])
        assert k[".private_segment_fixed_size"] == 0, name
    split4 = mod.find(ks, "18pairs_split_kernel", "ELi4E")
    assert len(split4) == 2
    for name, k in split4.items():
        assert k[".vgpr_count"] <= 80, (name, k[".vgpr_count"])
        assert k[".private_segment_fixed_size"] == 0, name


def test_every_kernel_of_the_library_is_listed():
    mod, ks = _resources()
    for part in ("radon_kernel", "k01_kernel", "pairs_kernel", "small_eval_kernel", "sum_pairs_kernel", "e1_kernel", "direct_lines_kernel",
                 "preprocess_kernel", "ramp_kernel", "publish_scalar_kernel"):
        assert mod.find(ks, part), part
