"""Synthetic cone-beam projections of a sphere phantom.

Eight (by default) homogeneous spheres; the projection value of a pixel is the analytic line
integral (chord length x density) along the ray through the pixel centre, multiplied by the cosine
weight sdd/sqrt(u'^2+v'^2+sdd^2) that the reference's pre-processing applies
(ref: code/LibEpipolarConsistency/Gui/PreProccess.cpp:146-166).  Data generator only -- not part
of the hot path.  `projections_numpy` is the canonical (float64 -> float32) definition;
`projections_torch` evaluates the same formula on a torch device for the full-size benchmark.
"""
import numpy as np

from . import geometry


def sphere_phantom(n_spheres=8, seed=1234, extent_mm=60.0, rmin=10.0, rmax=40.0):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-extent_mm, extent_mm, size=(n_spheres, 3))
    radii = rng.uniform(rmin, rmax, size=n_spheres)
    dens = rng.uniform(0.5, 1.5, size=n_spheres)
    return centers, radii, dens


def short_scan(n_proj, n_u, n_v, pixel_mm, sid=744.3, sdd=1088.15476, span_deg=200.0):
    """Circular short scan with the example data's source/detector distances.  The n_proj views
    cover [0, span_deg] inclusive (increment span/(n-1)): with the reference generator's
    increment max_angle/n (makeCircularTrajectory) a 200 deg / 400 view scan contains 40 view pairs
    exactly 180 deg apart, whose baseline passes through the world origin -- computeK01 then
    divides 0/0 (ref: EpipolarConsistencyCommon.hxx:122,126) and the mean becomes NaN in the
    reference as well.  The inclusive span has no such pair for the benchmark sizes."""
    max_angle = span_deg * n_proj / max(n_proj - 1, 1)
    return geometry.make_circular_trajectory(n_proj, sid, sdd, n_u, n_v, max_angle, pixel_mm)


def _ray_setup(P):
    P = np.asarray(P, dtype=np.float64).reshape(3, 4)
    M = P[:, :3]
    Minv = np.linalg.inv(M)
    C = -Minv @ P[:, 3]
    # principal point and focal length in px for the cosine weight
    m3 = M[2]
    pp = M @ m3
    pp = pp[:2] / pp[2]
    n3 = np.linalg.norm(m3)
    fu = np.linalg.norm(np.cross(M[0], m3)) / (n3 * n3)
    return Minv, C, pp, fu


def projections_numpy(Ps, n_u, n_v, phantom):
    centers, radii, dens = phantom
    out = np.zeros((len(Ps), n_v, n_u), np.float32)
    u = np.arange(n_u, dtype=np.float64)  # pixel (i,j) centre sits at (u,v)=(i,j), ref: RadonIntermediate.cu:97-99
    v = np.arange(n_v, dtype=np.float64)
    uu, vv = np.meshgrid(u, v)
    for k, P in enumerate(Ps):
        Minv, C, pp, f = _ray_setup(P)
        d = (Minv[:, 0, None, None] * uu + Minv[:, 1, None, None] * vv + Minv[:, 2, None, None])
        d = d / np.sqrt((d * d).sum(0))
        acc = np.zeros((n_v, n_u))
        for c, r, rho in zip(centers, radii, dens):
            oc = C - c
            b = d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2]
            disc = b * b - (oc @ oc - r * r)
            acc += rho * 2.0 * np.sqrt(np.maximum(disc, 0.0))
        w = f / np.sqrt((uu - pp[0]) ** 2 + (vv - pp[1]) ** 2 + f * f)
        out[k] = (acc * w).astype(np.float32)
    return out


def projections_torch(Ps, n_u, n_v, phantom, device, out=None):
    """Same formula on a torch device (float64 math, float32 result), one view at a time."""
    import torch
    centers, radii, dens = phantom
    n = len(Ps)
    if out is None:
        out = torch.empty((n, n_v, n_u), dtype=torch.float32, device=device)
    u = torch.arange(n_u, dtype=torch.float64, device=device)
    v = torch.arange(n_v, dtype=torch.float64, device=device)
    vv, uu = torch.meshgrid(v, u, indexing="ij")
    for k, P in enumerate(Ps):
        Minv, C, pp, f = _ray_setup(P)
        Mi = torch.tensor(Minv, dtype=torch.float64, device=device)
        d = Mi[:, 0, None, None] * uu + Mi[:, 1, None, None] * vv + Mi[:, 2, None, None]
        d = d / torch.sqrt((d * d).sum(0))
        acc = torch.zeros((n_v, n_u), dtype=torch.float64, device=device)
        for c, r, rho in zip(centers, radii, dens):
            oc = C - c
            b = d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2]
            disc = b * b - (float(oc @ oc) - r * r)
            acc += rho * 2.0 * torch.sqrt(torch.clamp(disc, min=0.0))
        w = f / torch.sqrt((uu - pp[0]) ** 2 + (vv - pp[1]) ** 2 + f * f)
        out[k] = (acc * w).to(torch.float32)
    return out
