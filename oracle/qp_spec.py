"""CPU restatement (numpy, float64) of the convex-MPC QP that the reference builds in ``src/mpc.py``.

TEST INFRASTRUCTURE ONLY.  Nothing on the product path may import this module: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use ``oracle/`` (as the checker).

PARITY STATUS: the *problem data* this module produces are pinned against the reference's committed
run log (inputs reproduced exactly, see tests/test_oracle_pinning.py).  The *solver* the reference calls
(CasADi ``Opti('conic')`` -> OSQP, ``src/mpc.py:49-55,258``) is an un-vendored, un-pinned third-party
dependency that is absent from this image, so solver outputs are "parity unpinned" against OSQP itself;
they are instead certified by an explicit KKT check of the QP that ``mpc.py`` defines
(``kkt_report`` below), which is a stronger statement than agreement with OSQP's default-tolerance
output (the committed log is visibly unconverged).

Every function cites the reference lines it restates.  State x = [Theta(3) p(3) omega(3) v(3) g]
(src/mpc.py:61,189-198); force vector per stage u = [f_FL f_FR f_HL f_HR] (src/mpc.py:273-278).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

NX = 13
NU = 12
LEGS = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")

# src/mpc.py:121-134 (cost weights; the 13th, on the gravity state, is 0.0) and :121 (force weight 0.0)
W_REF = np.array([1e4, 2.7e4, 1e4, 2.7e5, 2.7e5, 2.7e5, 1e4, 1e4, 1e4, 1.6e4, 1.6e4, 1.6e4, 0.0])


@dataclass
class QPConfig:
    """Constants hard-coded in the reference (src/mpc.py:45-46,71-76) + the knobs the build adds."""
    N: int = 10
    delta: float = 0.03                      # params['world_time_step'] (src/mpc.py:31)
    m: float = 8.885                         # src/mpc.py:71
    Ibody_inv: tuple = (1.0 / 0.24, 1.0, 1.0)  # src/mpc.py:73-76
    w: np.ndarray = field(default_factory=lambda: W_REF.copy())
    alpha: float = 0.0                       # force regulariser; the reference has 0.0 (src/mpc.py:121)
    f_min: float = 3.0                       # src/mpc.py:45
    f_max: float = 100.0                     # src/mpc.py:46
    disc: str = "euler"                      # 'euler' = reference (src/mpc.py:117); 'zoh' = closed-form expm


def skew(v):
    """src/utils.py:43-56 (compute_skew)."""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rot_z(yaw):
    """src/mpc.py:64-69."""
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def build_A(yaw):
    """Continuous single-rigid-body A, src/mpc.py:86-96 (Rz, not Rz^T, multiplies omega)."""
    A = np.zeros((NX, NX))
    A[0:3, 6:9] = rot_z(yaw)
    A[3:6, 9:12] = np.eye(3)
    A[11, 12] = 1.0
    return A


def build_B(yaw, r_legs, cfg: QPConfig):
    """Stage input matrix, src/mpc.py:98-107; r_legs[4,3] = foot - com lever arms (src/mpc.py:218-239)."""
    Rz = rot_z(yaw)
    I_hat_inv = Rz @ np.diag(cfg.Ibody_inv) @ Rz.T       # src/mpc.py:78
    B = np.zeros((NX, NU))
    for j in range(4):
        B[6:9, 3 * j:3 * j + 3] = I_hat_inv @ skew(r_legs[j])
        B[9:12, 3 * j:3 * j + 3] = np.eye(3) / cfg.m
    return B


def discretise(A, B, delta, mode):
    """'euler': X+ = X + delta (A X + B U) (src/mpc.py:117).  'zoh': exact, because A^3 = 0 and A^2 B = 0."""
    I = np.eye(NX)
    if mode == "euler":
        return I + delta * A, delta * B
    if mode == "zoh":
        return I + delta * A + 0.5 * delta**2 * (A @ A), (delta * I + 0.5 * delta**2 * A) @ B
    raise ValueError(mode)


def make_xdes(roll0, pitch0, yaw_start, com_start, v_ref, omega_ref, g, N, delta):
    """Reference trajectory, src/mpc.py:202-214."""
    xd = np.zeros((N + 1, NX))
    xd[:, 0] = roll0
    xd[:, 1] = pitch0
    xd[:, 8] = omega_ref
    xd[:, 9:12] = v_ref
    xd[:, 12] = g
    xd[0, 2] = yaw_start
    xd[0, 3:6] = com_start
    for i in range(1, N + 1):
        xd[i, 2] = xd[i - 1, 2] + omega_ref * delta
        xd[i, 3:6] = xd[i - 1, 3:6] + np.asarray(v_ref) * delta
    return xd


# ----------------------------------------------------------------------------------------------------
# Sparse multiple-shooting form, literal to src/mpc.py:58-173 (duplicates and trivial rows included).
# Variable order: z = [X[:,0], ..., X[:,N], U[:,0], ..., U[:,N-1]]  (column-major like CasADi's vec()).
# ----------------------------------------------------------------------------------------------------
def sparse_qp(x0, r, contact, xdes, mu, cfg: QPConfig, labels=False):
    """Returns P, q, c0, Ac, lo, hi with objective 1/2 z'Pz + q'z + c0 and lo <= Ac z <= hi.

    x0[13]; r[N,4,3]; contact[N,4] in {0,1} (1 = stance, swing = 1-contact, src/mpc.py:248-254);
    xdes[N+1,13]; mu scalar.  With ``labels=True`` a seventh value names the row class of every constraint row
    ("x0", "dyn_<state row>", "swing", "gpin", "fz_lo", "fz_hi", "fric_y", "fric_x") for row-class-wise checks.
    """
    N = cfg.N
    nX = NX * (N + 1)
    nz = nX + NU * N
    ix = lambda k: slice(NX * k, NX * (k + 1))
    iu = lambda k: slice(nX + NU * k, nX + NU * (k + 1))
    yaw = x0[2]                                            # src/mpc.py:64
    A = build_A(yaw)
    w = np.asarray(cfg.w, float)

    P = np.zeros((nz, nz))
    q = np.zeros(nz)
    c0 = 0.0
    for k in range(N + 1):                                 # src/mpc.py:121-134 (k = 0 term included)
        P[ix(k), ix(k)] = 2.0 * np.diag(w)
        q[ix(k)] = -2.0 * w * xdes[k]
        c0 += float(np.sum(w * xdes[k] ** 2))
    for k in range(N):
        P[iu(k), iu(k)] = 2.0 * cfg.alpha * np.eye(NU)

    rows, lo, hi, lab = [], [], [], []

    def add(row, l, h, name):
        rows.append(row); lo.append(l); hi.append(h); lab.append(name)

    for i in range(NX):                                    # src/mpc.py:113
        row = np.zeros(nz); row[i] = 1.0
        add(row, x0[i], x0[i], "x0")
    for k in range(N):                                     # src/mpc.py:116-117
        Ad, Bd = discretise(A, build_B(yaw, r[k], cfg), cfg.delta, cfg.disc)
        for i in range(NX):
            row = np.zeros(nz)
            row[ix(k + 1)][i] = 1.0
            row[ix(k)] -= Ad[i]
            row[iu(k)] -= Bd[i]
            add(row, 0.0, 0.0, "dyn_%d" % i)
    swing = 1.0 - np.asarray(contact, float)
    for k in range(N):                                     # src/mpc.py:139-144
        for j in range(4):
            for a in range(3):
                row = np.zeros(nz); row[iu(k)][3 * j + a] = swing[k, j]
                add(row, 0.0, 0.0, "swing")
    g = xdes[0, 12]
    INF = 1e20
    for k in range(N):
        row = np.zeros(nz); row[ix(k)][12] = 1.0           # src/mpc.py:149
        add(row, g, g, "gpin")
        for j in range(4):                                 # src/mpc.py:151-157
            cond = 1.0 - swing[k, j]
            row = np.zeros(nz); row[iu(k)][3 * j + 2] = cond
            add(row, cond * cfg.f_min, INF, "fz_lo")
            add(row.copy(), -INF, cond * cfg.f_max, "fz_hi")
        for j in range(4):                                 # src/mpc.py:159-165 (y), written twice
            for _dup in range(2):
                row = np.zeros(nz); row[iu(k)][3 * j + 1] = 1.0; row[iu(k)][3 * j + 2] = mu
                add(row, 0.0, INF, "fric_y")               # -mu fz <= fy
                row = np.zeros(nz); row[iu(k)][3 * j + 1] = 1.0; row[iu(k)][3 * j + 2] = -mu
                add(row, -INF, 0.0, "fric_y")              # fy <= mu fz
        for j in range(4):                                 # src/mpc.py:167-173 (x), written twice
            for _dup in range(2):
                row = np.zeros(nz); row[iu(k)][3 * j + 0] = 1.0; row[iu(k)][3 * j + 2] = mu
                add(row, 0.0, INF, "fric_x")
                row = np.zeros(nz); row[iu(k)][3 * j + 0] = 1.0; row[iu(k)][3 * j + 2] = -mu
                add(row, -INF, 0.0, "fric_x")
    out = (P, q, c0, np.array(rows), np.array(lo), np.array(hi))
    return out + (np.array(lab),) if labels else out


# ----------------------------------------------------------------------------------------------------
# Condensed form (the build's re-formulation of the same optimal-control problem): X = Sx x0 + Su U.
# ----------------------------------------------------------------------------------------------------
def condense(x0, r, contact, cfg: QPConfig):
    """Prediction matrices by plain recursion on (Ad, Bd_k); swing-leg columns of Bd are zeroed because
    those forces are pinned to 0 by src/mpc.py:139-144 (so they cannot influence X)."""
    N = cfg.N
    yaw = x0[2]
    A = build_A(yaw)
    Sx = np.zeros((N + 1, NX, NX))
    Su = np.zeros((N + 1, NX, N * NU))
    Sx[0] = np.eye(NX)
    for k in range(N):
        Ad, Bd = discretise(A, build_B(yaw, r[k], cfg), cfg.delta, cfg.disc)
        Bd = Bd * np.repeat(np.asarray(contact[k], float), 3)[None, :]
        Sx[k + 1] = Ad @ Sx[k]
        Su[k + 1] = Ad @ Su[k]
        Su[k + 1][:, NU * k:NU * (k + 1)] += Bd
    return Sx.reshape(-1, NX), Su.reshape(-1, N * NU)


def condensed_qp(x0, r, contact, xdes, mu, cfg: QPConfig):
    """H, g, c0 with J(U) = 1/2 U'HU + g'U + c0 == the reference cost (src/mpc.py:121-134) + alpha |U|^2,
    and the unique constraint rows of src/mpc.py:138-173: per leg-stage, rows
      [fz], [fx - mu fz], [fx + mu fz], [fy - mu fz], [fy + mu fz]
    with bounds stance: [f_min,f_max], (-inf,0], [0,inf), (-inf,0], [0,inf); swing: all [0,0]."""
    N = cfg.N
    n = N * NU
    Sx, Su = condense(x0, r, contact, cfg)
    Wd = np.tile(np.asarray(cfg.w, float), N + 1)
    e0 = Sx @ x0 - xdes.reshape(-1)
    H = 2.0 * (Su.T @ (Wd[:, None] * Su)) + 2.0 * cfg.alpha * np.eye(n)
    g = 2.0 * Su.T @ (Wd * e0)
    c0 = float(e0 @ (Wd * e0))
    INF = np.inf
    G = np.zeros((5 * 4 * N, n))
    lo = np.zeros(5 * 4 * N)
    hi = np.zeros(5 * 4 * N)
    for k in range(N):
        for j in range(4):
            b = 5 * (4 * k + j)
            c = NU * k + 3 * j
            G[b + 0, c + 2] = 1.0
            G[b + 1, c + 0] = 1.0; G[b + 1, c + 2] = -mu
            G[b + 2, c + 0] = 1.0; G[b + 2, c + 2] = mu
            G[b + 3, c + 1] = 1.0; G[b + 3, c + 2] = -mu
            G[b + 4, c + 1] = 1.0; G[b + 4, c + 2] = mu
            if contact[k][j]:
                lo[b:b + 5] = [cfg.f_min, -INF, 0.0, -INF, 0.0]
                hi[b:b + 5] = [cfg.f_max, 0.0, INF, 0.0, INF]
    return H, g, c0, G, lo, hi, Sx, Su


def predict_states(x0, U, r, contact, cfg: QPConfig):
    """Roll the discrete dynamics forward (src/mpc.py:110-117): returns X[N+1,13]."""
    N = cfg.N
    yaw = x0[2]
    A = build_A(yaw)
    X = np.zeros((N + 1, NX))
    X[0] = x0
    U = np.asarray(U).reshape(N, NU)
    for k in range(N):
        Ad, Bd = discretise(A, build_B(yaw, r[k], cfg), cfg.delta, cfg.disc)
        X[k + 1] = Ad @ X[k] + Bd @ (U[k] * np.repeat(np.asarray(contact[k], float), 3))
    return X


def objective(X, U, xdes, cfg: QPConfig):
    """src/mpc.py:121-134 evaluated numerically (+ alpha |U|^2)."""
    w = np.asarray(cfg.w, float)
    return float(np.sum(w[None, :] * (X - xdes) ** 2) + cfg.alpha * np.sum(np.asarray(U) ** 2))


def net_wrench(U, r, contact, cfg: QPConfig):
    """Per-stage net force and moment about the com (unique even when alpha = 0): [N,6]."""
    N = cfg.N
    U = np.asarray(U).reshape(N, 4, 3) * np.asarray(contact, float)[:, :, None]
    F = U.sum(axis=1)
    M = np.cross(np.asarray(r), U).sum(axis=1)
    return np.concatenate([F, M], axis=1)


# ----------------------------------------------------------------------------------------------------
# Reference solver for the condensed QP: OSQP-style ADMM (Stellato et al. 2020, alg. 1) in float64,
# followed by an active-set polish.  Small and slow; the C oracle (oracle/mpcqp_oracle.c) is the fast one.
# ----------------------------------------------------------------------------------------------------
def admm_solve(H, g, G, lo, hi, rho=0.1, sigma=1e-6, relax=1.6, max_iter=20000, eps=1e-10, adapt=True):
    n = H.shape[0]
    eq = (hi - lo) < 1e-12
    rho_vec = np.where(eq, 1e3 * rho, rho)
    u = np.zeros(n); z = np.zeros(G.shape[0]); y = np.zeros(G.shape[0])

    def factor(rv):
        M = H + sigma * np.eye(n) + G.T @ (rv[:, None] * G)
        return np.linalg.cholesky(M)

    L = factor(rho_vec)
    it = 0
    rp = rd = np.inf
    for it in range(1, max_iter + 1):
        rhs = sigma * u - g + G.T @ (rho_vec * z - y)
        ut = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        zt = G @ ut
        u = relax * ut + (1 - relax) * u
        zr = relax * zt + (1 - relax) * z
        znew = np.clip(zr + y / rho_vec, lo, hi)
        y = y + rho_vec * (zr - znew)
        z = znew
        if it % 25 == 0 or it == max_iter:
            Gu = G @ u
            rp = np.max(np.abs(Gu - z))
            rd = np.max(np.abs(H @ u + g + G.T @ y))
            sp = max(np.max(np.abs(Gu)), np.max(np.abs(z)), 1e-12)
            sd = max(np.max(np.abs(H @ u)), np.max(np.abs(G.T @ y)), np.max(np.abs(g)), 1e-12)
            if rp <= eps * (1 + sp) and rd <= eps * (1 + sd):
                break
            if adapt and it % 100 == 0:
                ratio = np.sqrt((rp / sp) / max(rd / sd, 1e-30))
                if ratio > 5 or ratio < 0.2:
                    rho = float(np.clip(rho * ratio, 1e-6, 1e6))
                    rho_vec = np.where(eq, 1e3 * rho, rho)
                    L = factor(rho_vec)
    return u, z, y, it, rp, rd


def polish(H, g, G, lo, hi, u, y, tol=1e-7, rounds=3):
    """Active-set polish: solve the equality-constrained QP on the rows ADMM marks active, refine."""
    n = H.shape[0]
    for _ in range(rounds):
        Gu = G @ u
        act_lo = (y < -tol) | ((hi - lo < 1e-12))
        act_hi = (y > tol) & ~act_lo
        idx = np.where(act_lo | act_hi)[0]
        b = np.where(act_lo[idx], lo[idx], hi[idx])
        Ga = G[idx]
        k = len(idx)
        K = np.block([[H, Ga.T], [Ga, np.zeros((k, k))]])
        rhs = np.concatenate([-g, b])
        reg = np.diag(np.concatenate([1e-9 * np.ones(n), -1e-9 * np.ones(k)]))
        sol = np.linalg.solve(K + reg, rhs)
        for _r in range(5):                               # iterative refinement against the unregularised K
            sol = sol + np.linalg.solve(K + reg, rhs - K @ sol)
        u_new = sol[:n]
        y_new = np.zeros_like(y)
        y_new[idx] = sol[n:]
        Gu = G @ u_new
        feas = np.all(Gu >= lo - 1e-8) and np.all(Gu <= hi + 1e-8)
        sign_ok = np.all(y_new[act_hi] >= -1e-8) and np.all(y_new[act_lo & ~(hi - lo < 1e-12)] <= 1e-8)
        if feas and sign_ok:
            return u_new, y_new, True
        u, y = u_new, y_new
    return u, y, False


def kkt_report(H, g, G, lo, hi, u, y):
    """Explicit optimality certificate of the QP: stationarity, primal feasibility, dual sign, complementarity."""
    Gu = G @ u
    stat = float(np.max(np.abs(H @ u + g + G.T @ y)))
    prim = float(max(np.max(np.maximum(lo - Gu, 0.0)), np.max(np.maximum(Gu - hi, 0.0))))
    yp, ym = np.maximum(y, 0.0), np.minimum(y, 0.0)
    fin_hi = np.isfinite(hi); fin_lo = np.isfinite(lo)
    dual_sign = float(max(np.max(np.where(fin_hi, 0.0, yp)), np.max(np.where(fin_lo, 0.0, -ym))))
    hi_f, lo_f = np.where(fin_hi, hi, Gu), np.where(fin_lo, lo, Gu)      # (an infinite bound has no complementarity term: no 0 * inf)
    comp = float(max(np.max(yp * (hi_f - Gu)), np.max(-ym * (Gu - lo_f))))
    return {"stationarity": stat, "primal": prim, "dual_sign": dual_sign, "complementarity": comp}
