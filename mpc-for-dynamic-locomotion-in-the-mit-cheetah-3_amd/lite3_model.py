"""Analytic leg kinematics of the Lite3 (geometry data read off lite3_urdf/urdf/Lite3.urdf:44-124 of the reference:
joint origins and axes; no code).  Replaces, for the caller-side torque map, what the reference asks DART for
(`getLinearJacobian(...)[:, 6:9]` etc., src/main.py:205-210).  Legs FL, FR, HL, HR; joints HipX, HipY, Knee.
"""
from __future__ import annotations

import numpy as np

LEGS = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")
_HIPX = np.array([[0.1745, 0.062, 0.0], [0.1745, -0.062, 0.0], [-0.1745, 0.062, 0.0], [-0.1745, -0.062, 0.0]])
_HIPY = np.array([[0.0, 0.0985, 0.0], [0.0, -0.0985, 0.0], [0.0, 0.0985, 0.0], [0.0, -0.0985, 0.0]])
_KNEE = np.array([0.0, 0.0, -0.20])
_FOOT = np.array([0.0, 0.0, -0.21])
_AX_X = np.array([-1.0, 0.0, 0.0])      # HipX axis
_AX_Y = np.array([0.0, -1.0, 0.0])      # HipY and Knee axes


def _rot(axis, angle):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def leg_fk_jac(leg: int, q):
    """Foot position and 3x3 linear Jacobian d p / d q in the torso frame, q = (HipX, HipY, Knee) in rad."""
    R1 = _rot(_AX_X, q[0]); R2 = R1 @ _rot(_AX_Y, q[1]); R3 = R2 @ _rot(_AX_Y, q[2])
    p1 = _HIPX[leg]
    p2 = p1 + R1 @ _HIPY[leg]
    p3 = p2 + R2 @ _KNEE
    pf = p3 + R3 @ _FOOT
    J = np.stack([np.cross(R1 @ _AX_X, pf - p1), np.cross(R2 @ _AX_Y, pf - p2), np.cross(R3 @ _AX_Y, pf - p3)], axis=1)
    return pf, J


def leg_ik(leg: int, p_body, q0=(0.0, -1.0, 1.6), iters=30):
    """Newton inverse kinematics (knee-bent branch selected by the start value)."""
    q = np.array(q0, dtype=float)
    for _ in range(iters):
        p, J = leg_fk_jac(leg, q)
        e = np.asarray(p_body, float) - p
        if np.abs(e).max() < 1e-12:
            break
        q = q + np.linalg.solve(J + 1e-9 * np.eye(3), e)
    return q


def world_jacobians(R_body, q_all):
    """{leg: 3x3 world-frame linear Jacobian block} for joint angles q_all[4,3] (src/main.py:205-210)."""
    return {LEGS[k]: R_body @ leg_fk_jac(k, q_all[k])[1] for k in range(4)}
