"""Offline study (numpy, fp64, CPU) of the solver POLICY on the QPs that end a launch of 4096: two-legged support at low friction
(tools/tail_study.py: the launch is as long as the few QPs that need three or four ADMM rounds, each alone on its SIMD from t = 0).
Emulates the engine's round structure on the dense condensed QP of oracle/qp_spec.py with the swing variables eliminated -- OSQP ADMM
blocks with the early rho check, primal-dual active-set polish steps with the patience rule, rho adaptation between rounds -- and
prices a solve with the measured lone-wave costs (tools/hardest.py): 0.40 us per ADMM iteration, 12 us per fp32 build + sweep,
19 us per fp64 build + sweep, 4 us per polish step.  Not on the product path, not a test.
usage: python tools/policy_study.py [n_hard] [n_easy]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp            # noqa: E402
import qp_spec as S     # noqa: E402

US_ITER, US_SWEEP32, US_SWEEP64, US_STEP, US_LEG = 0.40, 12.0, 19.0, 4.0, 3.7   # (a changed leg-stage: up to six rank-one updates)


class QP:
    """Reduced (stance-only) dense QP + per-leg structure."""

    def __init__(self, b, i, cfg):
        x0, r, c, xd, mu = b["x0"][i], b["r"][i], b["contact"][i], b["xdes"][i], float(b["mu"][i])
        H, g, c0, G, lo, hi, *_ = S.condensed_qp(x0, r, c, xd, mu, cfg)
        st = np.asarray(c, bool).reshape(-1)                     # [4N] leg-stages
        vi = np.repeat(st, 3); ri = np.repeat(st, 5)
        self.H, self.g, self.G, self.lo, self.hi = H[np.ix_(vi, vi)], g[vi], G[np.ix_(ri, vi)], lo[ri], hi[ri]
        self.mu, self.nl = mu, int(st.sum())
        self.fmin, self.fmax, self.alpha = cfg.f_min, cfg.f_max, cfg.alpha
        self.gmax = max(np.abs(self.g).max(), 1e-30)
        self.GtG = self.G.T @ self.G

    def ratio(self, u, z, y):
        Gu = self.G @ u
        rp = np.abs(Gu - z).max(); rd = np.abs(self.H @ u + self.g + self.G.T @ y).max()
        npn = max(np.abs(Gu).max(), np.abs(z).max(), 1e-12)
        ndn = max(np.abs(self.H @ u).max(), np.abs(self.G.T @ y).max(), self.gmax, 1e-12)
        return np.sqrt((rp / npn) / max(rd / ndn, 1e-30)), rp, rd

    def admm(self, u, z, y, rho, K, sigma=1e-6, relax=1.6, Minv=None):
        if Minv is None:
            Minv = np.linalg.inv(self.H + sigma * np.eye(len(u)) + rho * self.GtG)
        lo, hi = np.where(np.isfinite(self.lo), self.lo, -1e30), np.where(np.isfinite(self.hi), self.hi, 1e30)
        for _ in range(K):
            ut = Minv @ (sigma * u - self.g + self.G.T @ (rho * z - y))
            u = relax * ut + (1 - relax) * u
            zr = relax * (self.G @ ut) + (1 - relax) * z
            zn = np.clip(zr + y / rho, lo, hi)
            y = y + rho * (zr - zn)
            z = zn
        return u, z, y, Minv

    def rule(self, pu, py):
        """Active set of the polish on (pu, py): per leg (zs, xs, ys) in {-1, 0, 1}."""
        u = pu.reshape(-1, 3); y = py.reshape(-1, 5); m = self.mu
        fx, fy, fz = u[:, 0], u[:, 1], u[:, 2]
        zs = np.where(y[:, 0] + (fz - self.fmax) > 0, 1, np.where(y[:, 0] + (fz - self.fmin) < 0, -1, 0))
        g1, g2, g3, g4 = fx - m * fz, fx + m * fz, fy - m * fz, fy + m * fz
        hx, lx = y[:, 1] + g1 > 0, y[:, 2] + g2 < 0
        xs = np.where(hx & lx, np.where(g1 > -g2, 1, -1), np.where(hx, 1, np.where(lx, -1, 0)))
        hy, ly = y[:, 3] + g3 > 0, y[:, 4] + g4 < 0
        ys = np.where(hy & ly, np.where(g3 > -g4, 1, -1), np.where(hy, 1, np.where(ly, -1, 0)))
        return np.stack([zs, xs, ys], axis=1)

    def step(self, aset):
        """Equality-constrained QP on the active set -> candidate (u, y) and its KKT violations."""
        nl, m = self.nl, self.mu
        rows, rhs = [], []
        for L in range(nl):
            zs, xs, ys = aset[L]
            if zs != 0: rows.append(5 * L); rhs.append(self.fmax if zs > 0 else self.fmin)
            if xs != 0: rows.append(5 * L + (1 if xs > 0 else 2)); rhs.append(0.0)
            if ys != 0: rows.append(5 * L + (3 if ys > 0 else 4)); rhs.append(0.0)
        n = len(self.g)
        A = self.G[rows]
        KKT = np.block([[self.H, A.T], [A, np.zeros((len(rows), len(rows)))]])
        sol = np.linalg.solve(KKT + 1e-13 * np.eye(n + len(rows)), np.concatenate([-self.g, rhs]))
        u = sol[:n]; y = np.zeros(5 * nl); y[rows] = sol[n:]
        Gu = self.G @ u
        lo, hi = self.lo, self.hi
        pv = max(np.maximum(lo - Gu, Gu - hi).max(), 0.0)
        yy = y.reshape(-1, 5)
        dv = np.maximum.reduce([-yy[:, 1], yy[:, 2], -yy[:, 3], yy[:, 4], np.where(aset[:, 0] > 0, -yy[:, 0], 0), np.where(aset[:, 0] < 0, yy[:, 0], 0)]).max()
        dv = max(dv, 0.0)
        usc = max(1.0, np.abs(u).max())
        ok = pv <= 1e-7 * usc and dv <= min(1e-5 + 1e-9 * self.gmax, 2 * self.alpha * 1e-5 * usc)
        return u, y, pv, dv, ok


def solve(qp, pol):
    """One QP under a policy dict -> (solved, iters, sweeps32, rebuilds64, steps, rounds, updated leg-stages)."""
    n = len(qp.g)
    u, z, y = np.zeros(n), np.zeros(5 * qp.nl), np.zeros(5 * qp.nl)
    rho = pol.get("rho0", 1.0)
    iters = sw32 = sw64 = steps = legs = 0
    hard = False
    for rnd in range(pol.get("max_rounds", 12)):
        if iters >= pol["max_iter"]:
            break
        K = pol["first_block"] if rnd == 0 else pol["block"]
        K = min(K, pol["max_iter"] - iters)
        if rnd == 0 and pol.get("early", True):
            u, z, y, Minv = qp.admm(u, z, y, rho, min(25, K)); sw32 += 1
            rt, _, _ = qp.ratio(u, z, y)
            if rt > pol.get("adapt_thr", 6.0):
                rho = min(rho * rt, pol.get("rho_max", 30.0)); hard = True
                K = max(K, min(int(pol.get("hard_factor", 2) * K), pol["max_iter"] - iters))
                u, z, y, Minv = qp.admm(u, z, y, rho, K - 25); sw32 += 1
            else:
                u, z, y, _ = qp.admm(u, z, y, rho, K - 25, Minv=Minv)
        else:
            u, z, y, _ = qp.admm(u, z, y, rho, K); sw32 += 1
        iters += K
        # polish round
        budget = (pol.get("hard_polish", 2) if hard else 1) * pol["polish_max"]
        last = iters >= pol["max_iter"]
        pu, py = u.copy(), y.copy()
        vprev = vprev2 = np.inf
        nstall = 0; cheap_used = 0
        seen = []
        aset = qp.rule(pu, py)
        in_row = 0
        for ps in range(budget):
            key = aset.tobytes()
            if key in seen:
                break
            seen.append(key)
            uc, yn, pv, dv, ok = qp.step(aset)
            steps += 1
            if ps == 0 or not incr:
                sw64 += 1; in_row = 0
            if ok:
                return True, iters, sw32, sw64, steps, rnd + 1, legs
            v = pv + dv / max(qp.gmax, 1.0) * 100.0
            one_sided = min(pv, dv) <= 1e-9
            stalled = not (v < 0.5 * vprev)
            alternating = one_sided and 2 <= ps < 4 and v < 0.5 * vprev2
            if ps >= 1 and stalled and not alternating:
                nstall += 1
            vprev2, vprev = vprev, v
            pu, py = uc, yn
            new = qp.rule(pu, py)
            nchg = int((new != aset).any(axis=1).sum())
            incr = in_row < 12 and nchg <= 8
            if incr: in_row += 1; legs += nchg
            aset = new
            if nstall >= pol["patience"] and not last:
                if incr and nchg <= pol.get("cheap_legs", 8) and cheap_used < pol.get("cheap", 0):
                    cheap_used += 1
                else:
                    break
        if pol.get("restart_from_polish") and np.all(np.isfinite(pu)):   # the next block starts from the last polish candidate
            lo_, hi_ = np.where(np.isfinite(qp.lo), qp.lo, -1e30), np.where(np.isfinite(qp.hi), qp.hi, 1e30)
            u, z, y = pu.copy(), np.clip(qp.G @ pu, lo_, hi_), py.copy()
        # rho adaptation for the next round
        rt, rp, rd = qp.ratio(u, z, y)
        mode = pol.get("adapt", "osqp")
        if mode == "osqp":
            if np.isfinite(rt) and (rt > 2 or rt < 0.5):
                rho = min(max(rho * rt, 1e-4), 1e4)
        elif mode == "up_only":
            if np.isfinite(rt) and rt > 2:
                rho = min(max(rho * rt, 1e-4), 1e4)
        elif mode == "keep":
            pass
        elif mode == "damped":
            if np.isfinite(rt) and (rt > 2 or rt < 0.5):
                rho = min(max(rho * np.sqrt(rt), 1e-4), 1e4)
    return False, iters, sw32, sw64, steps, pol.get("max_rounds", 12), legs


def cost(res):
    ok, it, s32, s64, st, rn, lg = res
    return it * US_ITER + s32 * US_SWEEP32 + s64 * US_SWEEP64 + st * US_STEP + lg * US_LEG


def main():
    n_hard = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n_easy = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cfg = S.QPConfig(N=10, delta=0.03, alpha=1e-2)
    allg = ("trot", "pronk", "amble", "gallop")
    hard, easy = [], []
    for seed in (20250809, 1, 3):
        b = mpcqp.synth.make_batch(4096, 10, 0.03, seed, allg, (0.3, 0.5, 0.7, 1.0))
        gid = b["gait_ids"]; mu = b["mu"]
        cand = np.where((gid == 2) & (mu <= 0.5))[0]
        scr = []
        base0 = dict(first_block=70, block=100, max_iter=400, polish_max=4, patience=1, cheap=0, adapt="osqp")
        for i in cand:
            q = QP(b, i, cfg)
            r = solve(q, base0)
            scr.append((r[5], r[1], i, q))
        scr.sort(key=lambda t: (-t[0], -t[1]))
        print(f"seed {seed}: {len(cand)} amble mu<=0.5 candidates; rounds hist {np.bincount([t[0] for t in scr]).tolist()}; hardest ids {[int(t[2]) for t in scr[:8]]} iters {[t[1] for t in scr[:8]]}", flush=True)
        hard += [t[3] for t in scr[: n_hard // 3]]
        idx_e = np.where(~((gid == 2) & (mu <= 0.5)))[0][: n_easy // 3]
        easy += [QP(b, i, cfg) for i in idx_e]
    base = dict(first_block=70, block=100, max_iter=400, polish_max=4, patience=1, cheap=0, adapt="osqp")
    r3 = {**base, "cheap": 3, "cheap_legs": 3}
    r3f = {**r3, "hard_factor": 3}
    policies = {
        "r03 final": r3f,
        "r03 final + restart from polish": {**r3f, "restart_from_polish": True},
        "engine (r02)": base,
        "r03: cheap 3 on <= 3 legs": r3,
        "r03 + hard block x3": {**r3, "hard_factor": 3},
        "r03 + hard block x1.5": {**r3, "hard_factor": 1.5},
        "r03 + later blocks 70": {**r3, "block": 70},
        "r03 + later blocks 140": {**r3, "block": 140},
        "r03 + rho_max 60": {**r3, "rho_max": 60.0},
        "r03 + thr 4": {**r3, "adapt_thr": 4.0},
        "r03 + thr 9": {**r3, "adapt_thr": 9.0},
        "r03 + cheap 6 on <= 2": {**base, "cheap": 6, "cheap_legs": 2},
        "r03 + hard polish x3": {**r3, "hard_polish": 3},
        "r03 + damped adapt": {**r3, "adapt": "damped"},
    }
    for name, pol in policies.items():
        t0 = time.time()
        line = f"{name:26s}"
        for tag, qps in (("hard", hard), ("easy", easy)):
            res = [solve(q, pol) for q in qps]
            c = np.array([cost(r) for r in res]); ok = np.array([r[0] for r in res])
            it = np.array([r[1] for r in res]); st = np.array([r[4] for r in res]); rn = np.array([r[5] for r in res])
            line += f" | {tag}: solved {ok.mean():.3f} cost mean {c.mean():6.1f} p90 {np.percentile(c, 90):6.1f} max {c.max():6.1f} us; iters {it.mean():5.1f} steps {st.mean():4.2f} rounds {rn.mean():4.2f} (>=3: {(rn >= 3).sum()})"
        print(line + f"  [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
