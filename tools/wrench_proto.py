"""numpy prototype of the wrench-space (Woodbury) form of the engine's two linear systems -- design study for
csrc/mpcqp_wrench.h.  Not on the product path, not a test.  Checks the closed forms against oracle/qp_spec.py and
measures ADMM-iterations / polish-steps trade-offs with fp32 matrix arithmetic emulated in numpy.

H = 2 alpha I + T' K T,   T = blockdiag(T_j) (6 x 12 per stage: forces -> [Rz tau ; a]),   K = (+)_q K_q  (N x N per
wrench component q, constant per configuration when w_omega is isotropic in xy):
    K_q = 2 (wP_q c1 + wQ_q c0),  c0 = d^2 (N - max(j,j')),  c1 = d^4 sum_{k>max} (k-1-j+th)(k-1-j'+th)
M = D + T'KT  =>  M^-1 = D^-1 - D^-1 T' (K^-1 + T D^-1 T')^-1 T D^-1     (60 x 60 system instead of 120 x 120)
"""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp          # noqa: E402
import qp_spec as S   # noqa: E402


def tables(N, d, th, w):
    j = np.arange(N)
    mx = np.maximum(j[:, None], j[None, :])
    c0 = d * d * (N - mx)
    c1 = np.zeros((N, N))
    for a in range(N):
        for b in range(N):
            k = np.arange(max(a, b) + 1, N + 1)
            c1[a, b] = d ** 4 * np.sum((k - 1 - a + th) * (k - 1 - b + th))
    wP = np.array([w[0], w[1], w[2], w[3], w[4], w[5]])
    wQ = np.array([w[6], w[7], w[8], w[9], w[10], w[11]])
    assert w[6] == w[7], "omega weights must be isotropic in xy"
    K = np.stack([2 * (wP[q] * c1 + wQ[q] * c0) for q in range(6)])   # [6,N,N]
    Kinv = np.linalg.inv(K)
    return K, Kinv, wP, wQ


class Wrench:
    """Batched problem data in wrench form (float64)."""

    def __init__(self, b, cfg: S.QPConfig):
        self.cfg = cfg
        N, d = cfg.N, cfg.delta
        th = 0.5 if cfg.disc == "zoh" else 0.0
        x0, r, ct, xd, mu = (np.asarray(b[k], float) for k in ("x0", "r", "contact", "xdes", "mu"))
        B = len(x0)
        self.B, self.N, self.mu, self.ct = B, N, mu, ct
        yaw = x0[:, 2]
        c, s = np.cos(yaw), np.sin(yaw)
        Rz = np.zeros((B, 3, 3)); Rz[:, 0, 0] = c; Rz[:, 0, 1] = -s; Rz[:, 1, 0] = s; Rz[:, 1, 1] = c; Rz[:, 2, 2] = 1
        Ib = np.diag(cfg.Ibody_inv)
        Ihat = Rz @ Ib @ Rz.transpose(0, 2, 1)
        RI = Rz @ Ihat                                             # tau' = Rz Ihat^-1 (r x f)
        sk = np.zeros((B, N, 4, 3, 3))
        sk[..., 0, 1] = -r[..., 2]; sk[..., 0, 2] = r[..., 1]; sk[..., 1, 0] = r[..., 2]
        sk[..., 1, 2] = -r[..., 0]; sk[..., 2, 0] = -r[..., 1]; sk[..., 2, 1] = r[..., 0]
        Bl = np.einsum("bij,bnljk->bnlik", RI, sk) * ct[..., None, None]      # [B,N,4,3,3]
        T = np.zeros((B, N, 6, 12))
        for l in range(4):
            T[:, :, 0:3, 3 * l:3 * l + 3] = Bl[:, :, l]
            for a in range(3):
                T[:, :, 3 + a, 3 * l + a] = ct[:, :, l] / cfg.m
        self.T = T
        self.K, self.Kinv, wP, wQ = tables(N, d, th, cfg.w)
        # free response and targets per channel, stages k = 1..N
        k = np.arange(1, N + 1)[None, :, None]
        g = x0[:, 12][:, None]
        rzw0 = np.einsum("bij,bj->bi", Rz, x0[:, 6:9])
        eP = np.zeros((B, N, 6)); eQ = np.zeros((B, N, 6))
        eP[:, :, 0:3] = x0[:, None, 0:3] + k * d * rzw0[:, None, :] - xd[:, 1:, 0:3]
        p0 = x0[:, None, 3:6] + k * d * x0[:, None, 9:12]
        p0[:, :, 2] += d * d * g * (k[:, :, 0] * (k[:, :, 0] - 1) * 0.5 + th * k[:, :, 0])
        eP[:, :, 3:6] = p0 - xd[:, 1:, 3:6]
        eQ[:, :, 0:3] = rzw0[:, None, :] - np.einsum("bij,bkj->bki", Rz, xd[:, 1:, 6:9])
        v0 = x0[:, None, 9:12] + 0 * k
        v0[:, :, 2] += k[:, :, 0] * d * g
        eQ[:, :, 3:6] = v0 - xd[:, 1:, 9:12]
        # gamma_jq = 2 sum_{k>j} [wP d^2 (k-1-j+th) eP_kq + wQ d eQ_kq]
        jj = np.arange(N)[:, None]; kk = np.arange(1, N + 1)[None, :]
        on = (kk > jj).astype(float)
        Wp = on * (kk - 1 - jj + th)
        self.gamma = 2 * (d * d * np.einsum("jk,bkq->bjq", Wp, eP) * wP + d * np.einsum("jk,bkq->bjq", on, eQ) * wQ)
        self.g = np.einsum("bjqi,bjq->bji", T, self.gamma).reshape(B, 12 * N)

    def Tu(self, u):
        return np.einsum("bjqi,bji->bjq", self.T, u.reshape(self.B, self.N, 12))

    def Tt(self, w):
        return np.einsum("bjqi,bjq->bji", self.T, w).reshape(self.B, 12 * self.N)

    def Kw(self, w):
        return np.einsum("qjk,bkq->bjq", self.K, w)

    def grad(self, u):
        return 2 * self.cfg.alpha * u + self.Tt(self.Kw(self.Tu(u)) + self.gamma)

    def Kinv_full(self):
        N = self.N
        M = np.zeros((6 * N, 6 * N))
        for q in range(6):
            M[q::6, q::6] = self.Kinv[q]
        return M


def check_against_spec(B=6):
    b = mpcqp.synth.config3(B)
    for disc in ("euler", "zoh"):
        cfg = S.QPConfig(N=10, delta=0.03, alpha=1e-2, disc=disc)
        W = Wrench(b, cfg)
        for i in range(B):
            H, g, *_ = S.condensed_qp(b["x0"][i], b["r"][i], b["contact"][i], b["xdes"][i], b["mu"][i], cfg)
            Tf = np.zeros((60, 120))
            for j in range(10):
                Tf[6 * j:6 * j + 6, 12 * j:12 * j + 12] = W.T[i, j]
            Kf = np.linalg.inv(W.Kinv_full())
            H2 = 2 * cfg.alpha * np.eye(120) + Tf.T @ Kf @ Tf
            # swing columns of the spec's H carry only 2 alpha (their Bd columns are zeroed): same here
            assert np.abs(H - H2).max() <= 1e-9 * np.abs(H).max(), np.abs(H - H2).max()
            assert np.abs(g - W.g[i]).max() <= 1e-9 * np.abs(g).max()
    print("wrench form == literal condensed form (H, g), euler + zoh")


def solve_batch(W: Wrench, K_admm=100, rho=1.0, sigma=1e-6, relax=1.6, mdt=np.float32, polish_max=8, verbose=True,
                max_rounds=4, pdt=None):
    """ADMM(K) in `mdt` with the Woodbury solve, then primal-dual active-set polish steps with f64 gradient refinement."""
    cfg = W.cfg
    B, N = W.B, W.N
    n = 12 * N
    mu = W.mu[:, None, None]
    st = W.ct.astype(bool)                                          # [B,N,4]
    pdt = pdt or mdt
    a2 = 2 * cfg.alpha

    def woodbury_factory(dinv, Z=None, mdt=mdt):
        Kinv = W.Kinv_full().astype(mdt)
        T = W.T.astype(mdt)
        """dinv [B,N,4,3]; Z optional [B,N,4,3,3] reduced->force map.  Returns solve(rhs_reduced[B,N,4,3]) -> reduced."""
        Tl = T.reshape(B, N, 6, 4, 3)
        A = Tl if Z is None else np.einsum("bjqlc,bjlcd->bjqld", Tl, Z.astype(mdt))
        E = np.einsum("bjqlc,bjlc,bjplc->bjqp", A, dinv.astype(mdt), A)
        Smat = np.broadcast_to(Kinv, (B, 6 * N, 6 * N)).copy()
        for j in range(N):
            Smat[:, 6 * j:6 * j + 6, 6 * j:6 * j + 6] += E[:, j]
        Sinv = np.linalg.inv(Smat.astype(np.float64)).astype(mdt)   # (sweep in fp32 on the device; inverse rounded to mdt here)

        def solve(rhs):
            a = (dinv * rhs).astype(mdt)
            bvec = np.einsum("bjqlc,bjlc->bjq", A, a).reshape(B, 6 * N)
            c = np.einsum("bij,bj->bi", Sinv, bvec).reshape(B, N, 6)
            return (a - dinv.astype(mdt) * np.einsum("bjqlc,bjq->bjlc", A, c)).astype(mdt)
        return solve

    gq = W.g.reshape(B, N, 4, 3)
    fmin, fmax = cfg.f_min, cfg.f_max
    BIG = 1e30
    lo = np.zeros((B, N, 4, 5)); hi = np.zeros((B, N, 4, 5))
    lo[..., 0] = np.where(st, fmin, 0); hi[..., 0] = np.where(st, fmax, 0)
    for rr in (1, 3):
        lo[..., rr] = np.where(st, -BIG, 0); hi[..., rr] = 0
    for rr in (2, 4):
        lo[..., rr] = 0; hi[..., rr] = np.where(st, BIG, 0)

    def Gu(u):   # [B,N,4,3] -> [B,N,4,5]
        fx, fy, fz = u[..., 0], u[..., 1], u[..., 2]
        m = mu * fz
        return np.stack([fz, fx - m, fx + m, fy - m, fy + m], axis=-1)

    def Gt(v):
        m = mu
        return np.stack([v[..., 1] + v[..., 2], v[..., 3] + v[..., 4], v[..., 0] + m * (-v[..., 1] + v[..., 2] - v[..., 3] + v[..., 4])], axis=-1)

    u = np.zeros((B, N, 4, 3)); z = np.zeros((B, N, 4, 5)); y = np.zeros((B, N, 4, 5))
    done = np.zeros(B, bool); u_fin = np.zeros((B, N, 4, 3))
    iters = np.zeros(B, int); psteps = np.zeros(B, int)
    rho_b = np.full(B, rho)
    for rnd in range(max_rounds):
        dg = np.stack([np.full((B, N, 4), 1.0), np.full((B, N, 4), 1.0), np.ones((B, N, 4))], axis=-1)
        dg = a2 + sigma + rho_b[:, None, None, None] * np.stack([2 * np.ones((B, N, 4)), 2 * np.ones((B, N, 4)),
                                                                   (1 + 4 * mu ** 2) * np.ones((B, N, 4))], axis=-1)
        dinv = np.where(st[..., None], 1.0 / dg, 0.0)
        solve = woodbury_factory(dinv)
        rb = rho_b[:, None, None, None]
        for it in range(K_admm):
            rhs = (sigma * u - gq + Gt(rb * z - y)).astype(mdt)
            ut = solve(rhs).astype(np.float64)
            u = relax * ut + (1 - relax) * u
            zr = relax * Gu(ut) + (1 - relax) * z
            zn = np.clip(zr + y / rb, lo, hi)
            y = y + rb * (zr - zn)
            z = zn
        iters[~done] += K_admm
        # polish steps (all QPs in lockstep; finished ones are frozen)
        pu, py = u.copy(), y.copy()
        for ps in range(polish_max):
            fx, fy, fz = pu[..., 0], pu[..., 1], pu[..., 2]
            m = mu
            zs = np.where(py[..., 0] + (fz - fmax) > 0, 1, np.where(py[..., 0] + (fz - fmin) < 0, -1, 0))
            g1, g2, g3, g4 = fx - m * fz, fx + m * fz, fy - m * fz, fy + m * fz
            hx, lx = py[..., 1] + g1 > 0, py[..., 2] + g2 < 0
            xs = np.where(hx & lx, np.where(g1 > -g2, 1, -1), np.where(hx, 1, np.where(lx, -1, 0)))
            hy, ly = py[..., 3] + g3 > 0, py[..., 4] + g4 < 0
            ys = np.where(hy & ly, np.where(g3 > -g4, 1, -1), np.where(hy, 1, np.where(ly, -1, 0)))
            zs = np.where(st, zs, 0); xs = np.where(st, xs, 0); ys = np.where(st, ys, 0)
            ez = st & (zs == 0); ex = st & (xs == 0); ey = st & (ys == 0)
            Z = np.zeros((B, N, 4, 3, 3))
            Z[..., 0, 0] = ex; Z[..., 1, 1] = ey; Z[..., 2, 2] = ez
            Z[..., 0, 2] = xs * m * ez; Z[..., 1, 2] = ys * m * ez
            F = np.where(zs > 0, fmax, fmin)
            up = np.zeros((B, N, 4, 3))
            fixed = st & (zs != 0)
            up[..., 2] = np.where(fixed, F, 0); up[..., 0] = np.where(fixed, xs * m * F, 0); up[..., 1] = np.where(fixed, ys * m * F, 0)
            dr = np.stack([a2 * np.ones_like(m * fz), a2 * np.ones_like(fz), a2 * (1 + m * m * ((xs != 0).astype(float) + (ys != 0)))], axis=-1)
            en = np.stack([ex, ey, ez], axis=-1)
            dinv_r = np.where(en, 1.0 / dr, 0.0)
            solve_r = woodbury_factory(dinv_r, Z, mdt=pdt)
            v = np.stack([np.where(ex, fx, 0), np.where(ey, fy, 0), np.where(ez, fz, 0)], axis=-1)
            stat_hist = []
            for rf in range(12):
                uc = up + np.einsum("bjlcd,bjld->bjlc", Z, v)
                gr = W.grad(uc.reshape(B, n)).reshape(B, N, 4, 3)
                rg = np.einsum("bjlcd,bjlc->bjld", Z, gr)
                stat = np.abs(rg).reshape(B, -1).max(axis=1)
                stat_hist.append(stat)
                v = v + solve_r((-rg).astype(pdt)).astype(np.float64)
                v = np.where(en, v, 0)
            uc = up + np.einsum("bjlcd,bjld->bjlc", Z, v)
            gr = W.grad(uc.reshape(B, n)).reshape(B, N, 4, 3)
            stat = np.abs(np.einsum("bjlcd,bjlc->bjld", Z, gr)).reshape(B, -1).max(axis=1)
            yn = np.zeros((B, N, 4, 5))
            zacc = gr[..., 2].copy()
            yn[..., 1] = np.where(xs > 0, -gr[..., 0], 0); yn[..., 2] = np.where(xs < 0, -gr[..., 0], 0)
            yn[..., 3] = np.where(ys > 0, -gr[..., 1], 0); yn[..., 4] = np.where(ys < 0, -gr[..., 1], 0)
            zacc += m * (-yn[..., 1] + yn[..., 2] - yn[..., 3] + yn[..., 4])
            yn[..., 0] = np.where(zs != 0, -zacc, 0)
            gg = Gu(uc)
            pv = np.maximum(np.maximum(lo - gg, gg - hi), 0).reshape(B, -1).max(axis=1)
            dv = np.maximum.reduce([-yn[..., 1], yn[..., 2], -yn[..., 3], yn[..., 4],
                                    np.where(zs > 0, -yn[..., 0], 0), np.where(zs < 0, yn[..., 0], 0)])
            dv = np.where(st, dv, 0).reshape(B, -1).max(axis=1)
            gmax = np.abs(W.g).max(axis=1)
            usc = np.maximum(1, np.abs(uc).reshape(B, -1).max(axis=1))
            ok = (pv <= 1e-7 * usc) & (dv <= np.minimum(1e-5 + 1e-9 * gmax, a2 * 1e-5 * usc)) & (stat <= np.minimum(1e-5 + 1e-8 * gmax, a2 * 2e-5 * usc))
            newly = ok & ~done
            u_fin[newly] = uc[newly]
            psteps[~done] += 1
            done |= ok
            pu, py = uc, yn
            if verbose and ps == 0 and rnd == 0:
                sh = np.array(stat_hist)
                ratio = sh[1:] / np.maximum(sh[:-1], 1e-300)
                print("   refinement contraction (median / 90%% / max over QPs, rounds 1..3):",
                      [(float(np.median(ratio[i])), float(np.percentile(ratio[i], 90)), float(ratio[i].max())) for i in range(3)])
            if done.all():
                break
        if verbose:
            print(f"  round {rnd}: solved {done.mean():.4f}  mean iters {iters.mean():.1f}  mean polish steps {psteps.mean():.2f}")
        if done.all():
            break
    return u_fin.reshape(B, N, 12), done, iters, psteps


if __name__ == "__main__":
    check_against_spec()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    b = mpcqp.synth.config3(B)
    cfg = S.QPConfig(N=10, delta=0.03, alpha=1e-2)
    W = Wrench(b, cfg)
    lib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
    eng = mpcqp.Engine(lib, lib.default_config(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30))
    ref = eng.solve_batch_host(b["x0"], b["r"], b["contact"], b["xdes"], b["mu"])
    for K in (100, 60, 40, 25):
        print(f"K = {K} (fp32 matrices)")
        u, done, it, ps = solve_batch(W, K_admm=K)
        err = np.abs(u - ref["u"]).reshape(B, -1).max(axis=1) / np.maximum(np.abs(ref["u"]).reshape(B, -1).max(axis=1), 1)
        print(f"  solved {done.mean():.4f}; max rel err of solved {err[done].max():.2e}; cost model: iters*1 + psteps*25 = {(it + 25 * ps).mean():.1f}")
