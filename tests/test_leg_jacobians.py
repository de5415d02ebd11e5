"""Batched leg kinematics (include/mpcqp.h, mpcqp_leg_jacobians): what the reference's caller asks DART for before its torque map
(src/main.py:205-210).  Three independent statements of the same map are compared: the device kernel (composed Rodrigues rotations,
analytic cross-product columns), the CPU checker (series exponential + Richardson-extrapolated central differences of the forward
kinematics) and the host model `mpcqp.lite3_model` (numpy, analytic).  Geometry: lite3_urdf/urdf/Lite3.urdf:44-124 (data)."""
import ctypes

import numpy as np
import pytest

import mpcqp
from mpcqp import lite3_model


def _angles(B, seed):
    rng = np.random.default_rng(seed)
    q = np.stack([rng.uniform(-0.5, 0.5, (B, 4)), rng.uniform(-1.6, -0.3, (B, 4)), rng.uniform(0.6, 2.2, (B, 4))], axis=2)
    # random torso orientations (QR of a Gaussian matrix, determinant fixed to +1)
    R = np.linalg.qr(rng.normal(size=(B, 3, 3)))[0]
    R[:, :, 0] *= np.sign(np.linalg.det(R))[:, None]
    return q, R


def _host_model(q, R=None):
    B = q.shape[0]
    J = np.zeros((B, 4, 3, 3)); P = np.zeros((B, 4, 3))
    for b in range(B):
        for l in range(4):
            p, Jb = lite3_model.leg_fk_jac(l, q[b, l])
            J[b, l] = Jb if R is None else R[b] @ Jb
            P[b, l] = p if R is None else R[b] @ p
    return J, P


def test_geometry_struct_and_defaults(oracle_lib):
    for lib in (mpcqp.product_library(), oracle_lib):
        g = mpcqp._capi.MpcQpLegGeometry()
        assert lib.lib.mpcqp_default_leg_geometry(ctypes.byref(g)) == 0
        assert g.size == ctypes.sizeof(mpcqp._capi.MpcQpLegGeometry) == 8 + 8 * (24 + 12)
        assert np.allclose(np.array(g.hip_x), lite3_model._HIPX) and np.allclose(np.array(g.hip_y), lite3_model._HIPY)
        assert list(g.knee) == list(lite3_model._KNEE) and list(g.foot) == list(lite3_model._FOOT)
        assert list(g.axis_x) == list(lite3_model._AX_X) and list(g.axis_y) == list(lite3_model._AX_Y)


def test_checker_agrees_with_the_host_model(oracle_lib):
    q, R = _angles(64, 5)
    eng = mpcqp.Engine(oracle_lib, oracle_lib.default_config())
    for rot in (None, R):
        J, P = eng.leg_jacobians_host(q, rot)
        Jm, Pm = _host_model(q, rot)
        assert np.abs(P - Pm).max() <= 1e-13 and np.abs(J - Jm).max() <= 1e-9     # (differences: h^4 truncation ~ 1e-12 + rounding / h)
    # a geometry of the caller's own: scaled links, axes given unnormalised
    g = mpcqp._capi.MpcQpLegGeometry(); oracle_lib.lib.mpcqp_default_leg_geometry(ctypes.byref(g))
    g.knee[2] = -0.35; g.axis_y[1] = -2.0
    J2, P2 = eng.leg_jacobians_host(q, None, g)
    assert np.abs(P2 - P).max() > 0.05
    bad = mpcqp._capi.MpcQpLegGeometry()        # size field left at zero
    with pytest.raises(mpcqp.MpcQpError):
        eng.leg_jacobians_host(q, None, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("io", ["f64", "f32"])
def test_device_leg_jacobians(oracle_lib, io):
    import torch
    B = 3000
    q, R = _angles(B, 11)
    sol = mpcqp.MPCBatch(io_dtype=io)
    tq = torch.as_tensor(q, dtype=sol.tdtype, device="cuda").contiguous(); tR = torch.as_tensor(R, dtype=sol.tdtype, device="cuda").contiguous()
    ref = mpcqp.Engine(oracle_lib, oracle_lib.default_config())
    tol = 1e-9 if io == "f64" else 2e-6
    for rot, trot in ((None, None), (R, tR)):
        J, P = sol.leg_jacobians(tq, trot)
        torch.cuda.synchronize()
        Jc, Pc = ref.leg_jacobians_host(q, rot)
        Jm, Pm = _host_model(q[:200], None if rot is None else rot[:200])
        assert np.abs(J.cpu().numpy() - Jc).max() <= tol and np.abs(P.cpu().numpy() - Pc).max() <= tol
        assert np.abs(J.cpu().numpy()[:200] - Jm).max() <= (1e-12 if io == "f64" else 2e-6)
    # the whole caller-side chain on the device: joint angles -> Jacobians -> tau = J^T (-f)  (src/main.py:205-214)
    u = torch.as_tensor(np.random.default_rng(2).normal(0, 30, (B, sol.N, 12)), dtype=sol.tdtype, device="cuda")
    J, _ = sol.leg_jacobians(tq, tR, want_foot=False)
    tau = sol.torque_map(u, J)
    torch.cuda.synchronize()
    Jm, _ = _host_model(q[:100], R[:100])
    want = np.einsum("blaq,bla->blq", Jm, -u.cpu().numpy()[:100, 0].reshape(100, 4, 3).astype(np.float64))
    assert np.abs(tau.cpu().numpy()[:100] - want).max() <= (1e-10 if io == "f64" else 2e-4)
    # a geometry of the caller's own, and the empty batch
    g = mpcqp._capi.MpcQpLegGeometry(); mpcqp.product_library().lib.mpcqp_default_leg_geometry(ctypes.byref(g))
    g.knee[2] = -0.35; g.axis_y[1] = -2.0
    J2, P2 = sol.leg_jacobians(tq, None, geometry=g)
    torch.cuda.synchronize()
    Jc2, Pc2 = ref.leg_jacobians_host(q, None, g)
    assert np.abs(J2.cpu().numpy() - Jc2).max() <= tol and np.abs(P2.cpu().numpy() - Pc2).max() <= tol
    J0, P0 = sol.leg_jacobians(tq[:0].contiguous(), None)
    assert tuple(J0.shape) == (0, 4, 3, 3)
