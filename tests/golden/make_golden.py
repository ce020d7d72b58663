#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run from the repo root, in the
container that has /root/reference):

  example_pair_256.npz  the reference's example pair (config/example_data/proj000.nrrd, proj040.nrrd;
                        data files, not source) box-averaged 4x4 to 256x190 with P' = diag(1/4,1/4,1) P
                        (the MATLAB demo rescales the same way, ref: matlab/ecc_demo.m:10-17), plus the
                        oracle's outputs on it: K01, pair value, dtr checksums and sparse dtr samples.
  synthetic8_128.npz    8-view 128x128 synthetic scan (tests/conftest.py:make_small_scan): oracle pair
                        values, mean, K01s and per-dtr checksums.

  example_pair_native.npz  BASELINE config 1 at its native size: the same two data files as they
                        are, 1024x760 float32 (stored losslessly; mostly zero outside the object, 1.8 MB compressed),
                        their projection matrices, and the oracle's outputs at 768x768 bins: K01, pair value, N_kappa,
                        object radius, dtr checksums and 512 sparse dtr samples per view.

  example_pair_256_rows.npz  the rows of SURVEY.md 8(f) on the example pair (see widened_rows()).
  variants_128.npz      SURVEY.md 8c (a) and (d): every Radon filter / post-process on one 128x96 image at 96x80
                        bins; index-list, subset and user-parameter variants of the metric on the 8-view set.

  radon_contract.npz    round 4: the CONTRACTED arithmetic variant of the Radon intermediate (eccor_set_radon_contract(1),
                        what ecc_radon_set_arithmetic(ECC_RADON_FMA) is held to bit for bit) on the three data sets above:
                        dtr checksums, sparse samples, and the metric evaluated on those dtrs next to the exact variant's.

The reference has no golden vectors for this path: these pin the ORACLE against
regressions and travel to the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from epipolarconsistency_amd import nrrd  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/config/example_data"


def checksum(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum(),
                     float(np.bitwise_xor.reduce(a.view(np.uint32).reshape(-1)))])


def example_pair():
    imgs, Ps = [], []
    for name in ("proj000.nrrd", "proj040.nrrd"):
        img, _, meta = nrrd.read(os.path.join(REF, name))
        P = nrrd.parse_matrix(meta["Projection Matrix"])
        small = img.reshape(190, 4, 256, 4).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)
        imgs.append(small)
        Ps.append(np.diag([0.25, 0.25, 1.0]) @ P)
    imgs = np.stack(imgs)
    n_u, n_v, n_alpha, n_t = 256, 190, 192, 192
    dtrs = [oracle.radon(im, n_alpha, n_t) for im in imgs]
    res = oracle.evaluate_all(Ps, dtrs, n_u, n_v, want_K01=True)
    rng = np.random.default_rng(42)
    bins = rng.integers(0, n_alpha * n_t, size=256).astype(np.int32)
    np.savez_compressed(
        os.path.join(HERE, "example_pair_256.npz"), images=imgs, Ps=np.stack(Ps), n_alpha=n_alpha, n_t=n_t,
        K01=res["K01s"][0], pair_value=res["pairs"][0], mean=res["mean"], n_kappa=res["n_kappa"],
        object_radius=oracle.object_radius(Ps[0], n_u, n_v),
        dtr_checksums=np.stack([checksum(d) for d in dtrs]), sample_bins=bins,
        dtr_samples=np.stack([d.reshape(-1)[bins] for d in dtrs]))
    print("example pair: value %.9g, n_kappa %d" % (res["pairs"][0], res["n_kappa"]))


def example_pair_native():
    imgs, Ps = [], []
    for name in ("proj000.nrrd", "proj040.nrrd"):
        img, _, meta = nrrd.read(os.path.join(REF, name))
        imgs.append(np.ascontiguousarray(img, np.float32))
        Ps.append(nrrd.parse_matrix(meta["Projection Matrix"]))
    imgs = np.stack(imgs)
    n_u, n_v, n_alpha, n_t = 1024, 760, 768, 768
    dtrs = [oracle.radon(im, n_alpha, n_t) for im in imgs]
    res = oracle.evaluate_all(Ps, dtrs, n_u, n_v, want_K01=True)
    rng = np.random.default_rng(43)
    bins = rng.integers(0, n_alpha * n_t, size=512).astype(np.int32)
    np.savez_compressed(
        os.path.join(HERE, "example_pair_native.npz"), images=imgs, Ps=np.stack(Ps), n_alpha=n_alpha, n_t=n_t,
        K01=res["K01s"][0], pair_value=res["pairs"][0], mean=res["mean"], n_kappa=res["n_kappa"],
        object_radius=oracle.object_radius(Ps[0], n_u, n_v),
        dtr_checksums=np.stack([checksum(d) for d in dtrs]), sample_bins=bins,
        dtr_samples=np.stack([d.reshape(-1)[bins] for d in dtrs]))
    print("example pair (native 1024x760): value %.9g, n_kappa %d" % (res["pairs"][0], res["n_kappa"]))


def synthetic8():
    from conftest import make_small_scan
    Ps, imgs = make_small_scan()
    dtrs = [oracle.radon(im, 96, 96) for im in imgs]
    res = oracle.evaluate_all(Ps, dtrs, 128, 128, want_K01=True)
    np.savez_compressed(
        os.path.join(HERE, "synthetic8_128.npz"), Ps=np.stack(Ps), image_checksums=np.stack([checksum(i) for i in imgs]),
        dtr_checksums=np.stack([checksum(d) for d in dtrs]), pairs=res["pairs"], mean=res["mean"], K01s=res["K01s"],
        n_kappa=res["n_kappa"])
    print("synthetic8: mean %.9g" % res["mean"])


def widened_rows():
    """example_pair_256_rows.npz: oracle outputs of the SURVEY.md 8(f) rows on the same example pair (its images and
    matrices are read from example_pair_256.npz): ramp-filtered dtr, pre-processed image (defaults + -log +
    cosine weight), evaluateForImagePair, useCorrelation, MetricDirect (derivative and FBCC form)."""
    g = np.load(os.path.join(HERE, "example_pair_256.npz"))
    imgs, Ps = g["images"], list(g["Ps"])
    n_u, n_v, n_alpha, n_t = 256, 190, int(g["n_alpha"]), int(g["n_t"])
    bins = g["sample_bins"]
    ramp = oracle.radon(imgs[0], n_alpha, n_t, filter=1)
    pre = oracle.preprocess(imgs[1] + 1.0, Ps[1], apply_log=True, scale=0.01)
    dtrs = [oracle.radon(im, n_alpha, n_t) for im in imgs]
    e7 = oracle.evaluate_for_image_pair(Ps, dtrs, 0, 1, n_u, n_v)
    oracle.set_use_corr(True)
    corr = oracle.evaluate_all(Ps, dtrs, n_u, n_v)["pairs"][0]
    oracle.set_use_corr(False)
    radius = oracle.object_radius(Ps[0], n_u, n_v)
    d = oracle.direct_pair(Ps[0], Ps[1], imgs[0], imgs[1], 0.0, radius)
    f = oracle.direct_pair(Ps[0], Ps[1], imgs[0], imgs[1], 0.0, radius, fbcc=True)
    pix = np.random.default_rng(7).integers(0, n_u * n_v, size=256)
    np.savez_compressed(
        os.path.join(HERE, "example_pair_256_rows.npz"),
        ramp_checksum=checksum(ramp), ramp_samples=ramp.reshape(-1)[bins],
        pre_checksum=checksum(pre), pre_pixels=pix, pre_samples=pre.reshape(-1)[pix],
        e7_ecc=e7["ecc"], e7_n=len(e7["kappas"]), e7_samples0=e7["samples0"][::8], e7_samples1=e7["samples1"][::8],
        corr_value=corr, direct_metric=d["metric"], direct_n=len(d["kappas"]), direct_samples0=d["samples0"][::16],
        direct_samples1=d["samples1"][::16], fbcc_metric=f["metric"], fbcc_samples0=f["samples0"][::16])
    print("widened rows: e7 %.9g corr %.9g direct %.9g fbcc %.9g" % (e7["ecc"], corr, d["metric"], f["metric"]))


def variants():
    """variants_128.npz: SURVEY.md 8c fixtures (a) and (d): one 128x96 synthetic image and its dtr at 96x80 bins for
    every filter / post-process the Radon kernel has, and the index-list / subset / user-parameter variants of the
    metric on the 8-view set."""
    from conftest import make_small_scan
    Ps, imgs = make_small_scan()
    img = np.ascontiguousarray(imgs[3][16:112, :], np.float32)  # 96 rows x 128 columns
    rng = np.random.default_rng(5)
    bins = rng.integers(0, 96 * 80, size=256).astype(np.int32)
    radon = {}
    for name, (f, post) in dict(deriv=(0, 0), deriv_sqrt=(0, 1), deriv_log=(0, 2), plain=(2, 0), ramp=(1, 0)).items():
        d = oracle.radon(img, 96, 80, filter=f, post=post)
        radon["radon_%s_checksum" % name] = checksum(d)
        radon["radon_%s_samples" % name] = d.reshape(-1)[bins]
    dtrs = [oracle.radon(im, 96, 96) for im in imgs]
    idx = np.array([[0, 1, 0, 1], [2, 5, 2, 5], [7, 3, 7, 3], [1, 6, 4, 2], [4, 5, 5, 4], [6, 0, 6, 0]], np.int32)
    r_idx = oracle.evaluate_pairs(Ps, dtrs, 128, 128, idx)
    views = [1, 2, 4, 6, 7]
    idx_sub = np.array([(a, b, a, b) for k, a in enumerate(views) for b in views[k + 1:]], np.int32)
    r_sub = oracle.evaluate_pairs(Ps, dtrs, 128, 128, idx_sub)
    r_par = oracle.evaluate_all(Ps, dtrs, 128, 128, object_radius_mm=25.0, dkappa=0.004)
    np.savez_compressed(
        os.path.join(HERE, "variants_128.npz"), image=img, bins=bins, idx=idx, idx_pairs=r_idx["pairs"],
        idx_mean=r_idx["mean"], subset=np.array(views, np.int32), subset_mean=r_sub["mean"], param_pairs=r_par["pairs"],
        param_mean=r_par["mean"], param_n_kappa=r_par["n_kappa"], **radon)
    print("variants: idx mean %.9g subset mean %.9g param mean %.9g" % (r_idx["mean"], r_sub["mean"], r_par["mean"]))


def radon_contract():
    out = {}
    for tag in ("example_pair_256", "example_pair_native"):
        g = np.load(os.path.join(HERE, tag + ".npz"))
        imgs, Ps = g["images"], list(g["Ps"])
        n_v, n_u = imgs[0].shape
        n_alpha, n_t = int(g["n_alpha"]), int(g["n_t"])
        dtrs = [oracle.radon(im, n_alpha, n_t, contract=True) for im in imgs]
        exact = [oracle.radon(im, n_alpha, n_t) for im in imgs]
        assert all(np.array_equal(checksum(d), c) for d, c in zip(exact, g["dtr_checksums"]))  # the exact variant is what it was
        res = oracle.evaluate_all(Ps, dtrs, n_u, n_v)
        k = "pair256" if tag.endswith("256") else "native"
        out[k + "_dtr_checksums"] = np.stack([checksum(d) for d in dtrs])
        out[k + "_dtr_samples"] = np.stack([d.reshape(-1)[g["sample_bins"]] for d in dtrs])
        out[k + "_mean"] = res["mean"]
        out[k + "_mean_exact"] = float(g["mean"])
        out[k + "_max_bin_dev_rel"] = max(np.abs(a - b).max() for a, b in zip(dtrs, exact)) / max(np.abs(b).max() for b in exact)
        print("%s: contracted mean %.9g, exact %.9g (rel %.2e), max bin deviation / max|dtr| %.2e"
              % (tag, res["mean"], float(g["mean"]), abs(res["mean"] - float(g["mean"])) / abs(float(g["mean"])), out[k + "_max_bin_dev_rel"]))
    from conftest import make_small_scan
    Ps, imgs = make_small_scan()
    dtrs = [oracle.radon(im, 96, 96, contract=True) for im in imgs]
    res = oracle.evaluate_all(Ps, dtrs, 128, 128)
    out["synthetic8_dtr_checksums"] = np.stack([checksum(d) for d in dtrs])
    out["synthetic8_mean"] = res["mean"]
    out["synthetic8_pairs"] = res["pairs"]
    v = np.load(os.path.join(HERE, "variants_128.npz"))
    for name, (f, post) in dict(deriv=(0, 0), deriv_sqrt=(0, 1), deriv_log=(0, 2), plain=(2, 0), ramp=(1, 0)).items():
        d = oracle.radon(v["image"], 96, 80, filter=f, post=post, contract=True)
        out["variants_%s_checksum" % name] = checksum(d)
        out["variants_%s_samples" % name] = d.reshape(-1)[v["bins"]]
    np.savez_compressed(os.path.join(HERE, "radon_contract.npz"), **out)
    print("radon_contract: synthetic8 mean %.9g" % res["mean"])


if __name__ == "__main__":
    if "--only-contract" in sys.argv:
        radon_contract()
        sys.exit(0)
    if os.path.isdir(REF):
        example_pair()
        example_pair_native()
    synthetic8()
    widened_rows()
    variants()
    radon_contract()
