"""numpy prototype of the stage-wise (Riccati) solve of the wrench-space system  S y = b,  S = K^-1 + E  (E block diagonal, 6 x 6 per
stage) -- design study for csrc/mpcqp_stage.h (long horizons: the reference's committed N = 60, src/main.py:37).  Not on the product
path, not a test.

K = C' (2W) C with C: wrench sequence v -> deviation states z_k = (P_k, Q_k), k = 1..N, of six decoupled double integrators
    z_{k+1} = Phi z_k + Gam v_k,  Phi = [[I, d I], [0, I]],  Gam = [th d^2 I ; d I],  z_0 = 0
so  S y = b  is the two-point boundary problem   v = K^-1 y,  v_k + E_k y_k = b_k,  y_k = Gam' lam_{k+1},  lam_k = Phi' lam_{k+1} + 2W z_k,
solved with the ansatz lam_k = Pi_k z_k + pi_k (Pi: 12 x 12 per stage):
    Psi = Gam' Pi+ Gam,  Z = (Psi^-1 + E)^-1,  Ehat = E - E Z E,  L = Pi+ Gam Ehat,   F = I - L Gam'
    Pi_k = 2W + Phi' (Pi+ - L (Pi+ Gam)') Phi
    backward  pi_k = Phi' F (Pi+ Gam b_k + pi_{k+1})
    forward   z_{k+1} = F' (Phi z_k + Gam (b_k - E_k Gam' pi_{k+1}))
    y_k = Gam' (Pi_{k+1} z_{k+1} + pi_{k+1})
"""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tools"))
import mpcqp            # noqa: E402
import qp_spec as S     # noqa: E402
from wrench_proto import Wrench   # noqa: E402


def riccati_factor(E, wP, wQ, d, th, dtype=np.float64):
    """E [N,6,6] -> per-stage factors."""
    N = E.shape[0]
    I6 = np.eye(6)
    Phi = np.block([[I6, d * I6], [np.zeros((6, 6)), I6]])
    Gam = np.vstack([th * d * d * I6, d * I6])
    W2 = 2 * np.diag(np.concatenate([wP, wQ]))
    Pi = W2.copy()                                   # Pi_N
    PG = np.zeros((N, 12, 6)); L = np.zeros((N, 12, 6))
    for k in range(N - 1, -1, -1):
        pg = Pi @ Gam
        Psi = Gam.T @ pg
        Z = np.linalg.inv(np.linalg.inv(Psi) + E[k])
        Eh = E[k] - E[k] @ Z @ E[k]
        l = pg @ Eh
        PG[k], L[k] = pg, l
        M = Pi - l @ pg.T
        M = 0.5 * (M + M.T)
        Pi = W2 + Phi.T @ M @ Phi
    return {"PG": PG.astype(dtype), "L": L.astype(dtype), "E": E.astype(dtype), "Phi": Phi, "Gam": Gam}


def riccati_solve(f, b):
    """b [N,6] -> y [N,6] with S y = b."""
    PG, L, E, Phi, Gam = f["PG"], f["L"], f["E"], f["Phi"], f["Gam"]
    N = b.shape[0]
    dt = PG.dtype
    pi = np.zeros((N + 1, 12), dt)
    for k in range(N - 1, 0, -1):
        s = PG[k] @ b[k].astype(dt) + pi[k + 1]
        s = s - L[k] @ (Gam.T.astype(dt) @ s)
        pi[k] = Phi.T.astype(dt) @ s
    z = np.zeros((N + 1, 12), dt)
    y = np.zeros((N, 6), dt)
    for k in range(N):
        dk = b[k].astype(dt) - E[k] @ (Gam.T.astype(dt) @ pi[k + 1])
        t = Phi.astype(dt) @ z[k] + Gam.astype(dt) @ dk
        z[k + 1] = t - Gam.astype(dt) @ (L[k].T @ t)
        y[k] = PG[k].T @ z[k + 1] + Gam.T.astype(dt) @ pi[k + 1]
    return y


def main():
    rng = np.random.default_rng(0)
    for N, d, disc in ((10, 0.03, "euler"), (10, 0.03, "zoh"), (60, 0.01, "euler"), (60, 0.01, "zoh"), (20, 0.03, "euler")):
        b = mpcqp.synth.make_batch(4, N, d, 3, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0))
        cfg = S.QPConfig(N=N, delta=d, alpha=1e-2, disc=disc)
        W = Wrench(b, cfg)
        th = 0.5 if disc == "zoh" else 0.0
        wP, wQ = np.array(cfg.w[0:6]), np.array(cfg.w[6:12])
        for i in range(4):
            for rho, alpha in ((1.0, 1e-2), (20.0, 1e-2), (0.0, 1e-2), (0.0, 1e-4)):     # rho = 0: a polish-like system, D = 2 alpha
                mu = b["mu"][i]
                dd = np.zeros(12 * N)
                for j in range(N):
                    for l in range(4):
                        if b["contact"][i, j, l]:
                            base = 2 * alpha + (1e-6 if rho > 0 else 0.0)
                            dd[12 * j + 3 * l: 12 * j + 3 * l + 3] = [base + 2 * rho, base + 2 * rho, base + rho * (1 + 4 * mu * mu)]
                Dinv = np.where(dd > 0, 1.0 / np.maximum(dd, 1e-300), 0.0)
                E = np.zeros((N, 6, 6))
                for j in range(N):
                    Tj = W.T[i, j]
                    E[j] = (Tj * Dinv[12 * j:12 * j + 12]) @ Tj.T
                Sfull = W.Kinv_full().copy()
                for j in range(N):
                    Sfull[6 * j:6 * j + 6, 6 * j:6 * j + 6] += E[j]
                rhs = rng.normal(size=(N, 6))
                y_ref = np.linalg.solve(Sfull, rhs.reshape(-1)).reshape(N, 6)
                f = riccati_factor(E, wP, wQ, d, th)
                y = riccati_solve(f, rhs)
                err = np.abs(y - y_ref).max() / np.abs(y_ref).max()
                f32 = {k: (v.astype(np.float32) if k in ("PG", "L", "E") else v) for k, v in f.items()}
                y32 = riccati_solve(f32, rhs)
                err32 = np.abs(y32 - y_ref).max() / np.abs(y_ref).max()
                assert err < 1e-8, (N, disc, rho, alpha, err)
                print(f"N={N} {disc} QP {i} rho={rho} alpha={alpha}: Riccati vs dense rel err {err:.1e}; fp64 factor rounded to fp32 + fp32 solve {err32:.1e}; cond(S) {np.linalg.cond(Sfull):.1e}")


if __name__ == "__main__":
    main()
