"""Projection-matrix helpers (float64 numpy) needed on either side of the hot path: the synthetic
circular trajectories of the benchmark configs and the 6-DoF rigid perturbation of config 5.

These restate, for data generation only, the semantics of
  ref: code/HeaderOnly/Utils/Projtable.hxx:138-165 (makeCircularTrajectory),
  ref: code/LibProjectiveGeometry/CameraOpenGL.hxx:11-31 (cameraPerspective, cameraLookAt),
  ref: code/LibProjectiveGeometry/ProjectionMatrix.cpp:11-18,133-146 (normalize, makeProjectionMatrix),
  ref: code/LibProjectiveGeometry/Models/ModelSimilarity3D.hxx:64-88 ("3D Rigid" parameters).
"""
import numpy as np


def normalize_projection_matrix(P):
    P = np.asarray(P, dtype=np.float64).reshape(3, 4)
    norm_m3 = np.linalg.norm(P[2, :3])
    if np.linalg.det(P[:, :3]) < 0:
        norm_m3 = -norm_m3
    return P / norm_m3


def camera_perspective(fovy_rad, width, height):
    tanfov2 = 2.0 * np.tan(0.5 * fovy_rad)
    K = np.eye(3)
    K[0, 0] = K[1, 1] = height / tanfov2
    K[0, 2] = 0.5 * width
    K[1, 2] = 0.5 * height
    return K


def make_projection_matrix(K, R, t):
    P = np.zeros((3, 4))
    P[:, :3] = K @ R
    P[:, 3] = K @ t
    return normalize_projection_matrix(P)


def camera_look_at(K, eye, center, up=(0.0, 1.0, 0.0)):
    eye = np.asarray(eye, float)
    center = np.asarray(center, float)
    up = np.asarray(up, float)
    fwd = center - eye
    fwd = fwd / np.linalg.norm(fwd)
    left = np.cross(up, fwd)
    left = left / np.linalg.norm(left)
    up = np.cross(fwd, left)
    R = np.stack([left, up, -fwd])
    return make_projection_matrix(K, R, -R @ eye)


def make_circular_trajectory(n_proj, sid, sdd, n_u, n_v, max_angle_deg, pixel_spacing):
    """n_proj 3x4 matrices of a circular C-arm scan about the y axis (after the 90 deg x-rotation
    the reference applies), primary angle i*max_angle/n_proj."""
    fovy = np.arctan(n_v * pixel_spacing / sdd)
    K = camera_perspective(fovy, n_u, n_v)
    T = np.eye(4)
    c, s = np.cos(0.5 * np.pi), np.sin(0.5 * np.pi)
    T[1, 1] = c; T[2, 2] = c; T[1, 2] = -s; T[2, 1] = s
    Ps = []
    for i in range(n_proj):
        a = i * (max_angle_deg / n_proj) / 180.0 * np.pi
        P = camera_look_at(K, (sid * np.cos(a), 0.0, sid * np.sin(a)), (0.0, 0.0, 0.0))
        Ps.append(normalize_projection_matrix(P @ T))
    return Ps


def rigid_transform(tx=0.0, ty=0.0, tz=0.0, rx=0.0, ry=0.0, rz=0.0):
    """T = [Rx(rx) Ry(ry) Rz(rz), t]; the perturbed view is P' = P @ T."""
    def rot(axis, a):
        c, s = np.cos(a), np.sin(a)
        R = np.eye(3)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        R[i, i] = c; R[j, j] = c; R[i, j] = -s; R[j, i] = s
        return R
    T = np.eye(4)
    if rx != 0 or ry != 0 or rz != 0:
        T[:3, :3] = rot(0, rx) @ rot(1, ry) @ rot(2, rz)
    T[:3, 3] = (tx, ty, tz)
    return T


def camera_center(P):
    """Homogeneous null vector of P scaled to w = 1 (float64)."""
    P = np.asarray(P, dtype=np.float64).reshape(3, 4)
    C = np.array([np.linalg.det(P[:, [1, 2, 3]]), -np.linalg.det(P[:, [0, 2, 3]]),
                  np.linalg.det(P[:, [0, 1, 3]]), -np.linalg.det(P[:, [0, 1, 2]])])
    return C / C[3] if abs(C[3]) > 1e-12 else C


def fundamental_matrix(P0, P1):
    """F = [e1]x P1 P0^+  (ref: code/LibProjectiveGeometry/ProjectionMatrix.cpp:148-163);
    used by tests as an independent cross-check of the epipolar-line pencil."""
    P0 = np.asarray(P0, float).reshape(3, 4)
    P1 = np.asarray(P1, float).reshape(3, 4)
    e1 = P1 @ camera_center(P0)
    e1x = np.array([[0, e1[2], -e1[1]], [-e1[2], 0, e1[0]], [e1[1], -e1[0], 0]])
    return e1x @ P1 @ np.linalg.pinv(P0)
