"""Offline study (numpy, fp64, CPU; not on the product path, not a test): Anderson acceleration of the ADMM blocks.
The engine's round structure (tools/policy_study.py: OSQP ADMM blocks with the early rho check, primal-dual active-set polish with
the patience rules, rho adaptation between rounds) on the dense condensed QP of oracle/qp_spec.py, with the ADMM iterate
extrapolated every p iterations from the last m + 1 of them (type-II Anderson acceleration of v -> f^p(v), v = z + y / rho the
pre-projection variable, or of (z, y / rho)), priced with the measured lone-wave costs.  On the QPs that end a launch of 4096
(two-legged support at mu <= 0.5) the worst solve of a batch goes 375 -> 245..260 us (m = 3, p = 5), every QP solved, the class
mean 135 -> 112 us; fp32 history and regulariser 1e-6 change nothing.  What the device then measured is in DESIGN.md section 5.
usage: python tools/accel_study.py [seed ...]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tools"))
import mpcqp, qp_spec as S
import policy_study as P

F32 = np.float32

def admm_aa(qp, u, z, y, rho, K, Minv, m, p, f32=False, reg=1e-8, sigma=1e-6, relax=1.6, gcap=None, onv=False):
    if onv: return admm_aav(qp, u, z, y, rho, K, Minv, m, p, f32, reg, sigma, relax, gcap)
    """K iterations; AA(m) on f^p on x = (z, y / rho).  History starts empty.  Returns u, z, y, n_aa_steps."""
    lo, hi = np.where(np.isfinite(qp.lo), qp.lo, -1e30), np.where(np.isfinite(qp.hi), qp.hi, 1e30)
    nz = len(z)
    hx, hf = [], []
    x = np.concatenate([z, y / rho]); naa = 0
    dt = F32 if f32 else np.float64
    for it in range(1, K + 1):
        ut = Minv @ (sigma * u - qp.g + qp.G.T @ (rho * z - y))
        u = relax * ut + (1 - relax) * u
        zr = relax * (qp.G @ ut) + (1 - relax) * z
        zn = np.clip(zr + y / rho, lo, hi)
        y = y + rho * (zr - zn); z = zn
        if f32: u = u.astype(F32).astype(np.float64); z = z.astype(F32).astype(np.float64); y = y.astype(F32).astype(np.float64)
        if m > 0 and it % p == 0 and it < K:       # (no extrapolation on the block's last iterate: the polish gets a genuine ADMM iterate)
            fx = np.concatenate([z, y / rho]).astype(dt); r = fx - x.astype(dt)
            hx.append(fx); hf.append(r)
            if len(hf) > m + 1: hx.pop(0); hf.pop(0)
            if len(hf) >= 2:
                dF = np.stack([hf[i + 1] - hf[i] for i in range(len(hf) - 1)], axis=1)
                dX = np.stack([hx[i + 1] - hx[i] for i in range(len(hx) - 1)], axis=1)
                A = (dF.T @ dF).astype(dt); bb = (dF.T @ r).astype(dt)
                A = A + dt(reg) * np.trace(A) * np.eye(A.shape[0], dtype=dt)
                try:
                    gam = np.linalg.solve(A.astype(np.float64), bb.astype(np.float64)).astype(dt)
                except np.linalg.LinAlgError:
                    gam = None
                ok = gam is not None and np.all(np.isfinite(gam)) and (gcap is None or np.abs(gam).sum() <= gcap)
                if ok:
                    xa = (fx - dX @ gam).astype(np.float64)
                    z, y = xa[:nz], rho * xa[nz:]; naa += 1
                    x = xa
                else:
                    hx.clear(); hf.clear(); x = fx.astype(np.float64)
            else:
                x = fx.astype(np.float64)
    return u, z, y, naa

def admm_aav(qp, u, z, y, rho, K, Minv, m, p, f32, reg, sigma, relax, gcap):
    lo, hi = np.where(np.isfinite(qp.lo), qp.lo, -1e30), np.where(np.isfinite(qp.hi), qp.hi, 1e30)
    hx, hf = [], []
    dt = F32 if f32 else np.float64
    x = (z + y / rho).astype(dt); naa = 0
    for it in range(1, K + 1):
        ut = Minv @ (sigma * u - qp.g + qp.G.T @ (rho * z - y))
        u = relax * ut + (1 - relax) * u
        zr = relax * (qp.G @ ut) + (1 - relax) * z
        t = zr + y / rho
        zn = np.clip(t, lo, hi)
        y = rho * (t - zn); z = zn
        if f32: u = u.astype(F32).astype(np.float64); z = z.astype(F32).astype(np.float64); y = y.astype(F32).astype(np.float64); t = t.astype(F32).astype(np.float64)
        if m > 0 and it % p == 0 and it < K:
            fx = t.astype(dt); r = fx - x
            hx.append(fx); hf.append(r)
            if len(hf) > m + 1: hx.pop(0); hf.pop(0)
            if len(hf) >= 2:
                dF = np.stack([hf[i + 1] - hf[i] for i in range(len(hf) - 1)], axis=1)
                dX = np.stack([hx[i + 1] - hx[i] for i in range(len(hx) - 1)], axis=1)
                A = (dF.T @ dF).astype(dt); bb = (dF.T @ r).astype(dt)
                A = A + dt(reg) * np.trace(A) * np.eye(A.shape[0], dtype=dt)
                try: gam = np.linalg.solve(A.astype(np.float64), bb.astype(np.float64)).astype(dt)
                except np.linalg.LinAlgError: gam = None
                ok = gam is not None and np.all(np.isfinite(gam)) and (gcap is None or np.abs(gam).sum() <= gcap)
                if ok:
                    xa = (fx - dX @ gam).astype(dt)
                    va = xa.astype(np.float64)
                    z = np.clip(va, lo, hi); y = rho * (va - z); naa += 1
                    x = xa
                else:
                    hx.clear(); hf.clear(); x = fx
            else:
                x = fx
    return u, z, y, naa

def solve(qp, pol):
    n = len(qp.g)
    u, z, y = np.zeros(n), np.zeros(5 * qp.nl), np.zeros(5 * qp.nl)
    rho = pol.get("rho0", 1.0)
    m, p, f32 = pol.get("m", 0), pol.get("p", 5), pol.get("f32", False)
    aa_from = pol.get("aa_from_round", 0)
    iters = sw32 = sw64 = steps = legs = naa = 0
    hard = False
    sigma = 1e-6
    def inv(rho): return np.linalg.inv(qp.H + sigma * np.eye(n) + rho * qp.GtG)
    for rnd in range(pol.get("max_rounds", 12)):
        if iters >= pol["max_iter"]: break
        K = pol["first_block"] if rnd == 0 else pol["block"]
        K = min(K, pol["max_iter"] - iters)
        mm = m if rnd >= aa_from else 0
        if rnd == 0:
            Minv = inv(rho)
            u, z, y, a = admm_aa(qp, u, z, y, rho, min(25, K), Minv, mm, p, f32, gcap=pol.get("gcap"), onv=pol.get("onv", False), reg=pol.get("reg", 1e-8)); naa += a; sw32 += 1
            rt, _, _ = qp.ratio(u, z, y)
            if rt > pol.get("adapt_thr", 6.0):
                rho = min(rho * rt, pol.get("rho_max", 30.0)); hard = True
                K = max(K, min(int(pol.get("hard_factor", 2) * K), pol["max_iter"] - iters))
                Minv = inv(rho); sw32 += 1
            u, z, y, a = admm_aa(qp, u, z, y, rho, K - 25, Minv, mm, p, f32, gcap=pol.get("gcap"), onv=pol.get("onv", False), reg=pol.get("reg", 1e-8)); naa += a
        else:
            u, z, y, a = admm_aa(qp, u, z, y, rho, K, inv(rho), mm, p, f32, gcap=pol.get("gcap"), onv=pol.get("onv", False), reg=pol.get("reg", 1e-8)); naa += a; sw32 += 1
        iters += K
        budget = (pol.get("hard_polish", 2) if hard else 1) * pol["polish_max"]
        last = iters >= pol["max_iter"]
        pu, py = u.copy(), y.copy()
        vprev = vprev2 = np.inf; nstall = 0; cheap_used = 0; seen = []
        aset = qp.rule(pu, py); in_row = 0; incr = False
        for ps in range(budget):
            key = aset.tobytes()
            if key in seen: break
            seen.append(key)
            uc, yn, pv, dv, ok = qp.step(aset); steps += 1
            if ps == 0 or not incr: sw64 += 1; in_row = 0
            if ok: return True, iters, sw32, sw64, steps, rnd + 1, legs, naa
            v = pv + dv / max(qp.gmax, 1.0) * 100.0
            one_sided = min(pv, dv) <= 1e-9
            stalled = not (v < 0.5 * vprev)
            alternating = one_sided and 2 <= ps < 4 and v < 0.5 * vprev2
            if ps >= 1 and stalled and not alternating: nstall += 1
            vprev2, vprev = vprev, v
            pu, py = uc, yn
            new = qp.rule(pu, py)
            nchg = int((new != aset).any(axis=1).sum())
            incr = in_row < 12 and nchg <= 8
            if incr: in_row += 1; legs += nchg
            aset = new
            if nstall >= pol["patience"] and not last:
                if incr and nchg <= pol.get("cheap_legs", 8) and cheap_used < pol.get("cheap", 0): cheap_used += 1
                else: break
        rt, rp, rd = qp.ratio(u, z, y)
        if np.isfinite(rt) and (rt > 2 or rt < 0.5): rho = min(max(rho * rt, 1e-4), 1e4)
    return False, iters, sw32, sw64, steps, pol.get("max_rounds", 12), legs, naa

US_AA = 0.30   # one extrapolation step (~120 instructions of a lone wave)
def cost(res):
    ok, it, s32, s64, st, rn, lg, naa = res
    return it * P.US_ITER + s32 * P.US_SWEEP32 + s64 * P.US_SWEEP64 + st * P.US_STEP + lg * P.US_LEG + naa * US_AA


def main():
    cfg = S.QPConfig(N=10, delta=0.03, alpha=1e-2)
    allg = ("trot", "pronk", "amble", "gallop")
    base = dict(first_block=70, block=100, max_iter=400, polish_max=4, patience=1, cheap=3, cheap_legs=3, hard_factor=3)
    pols = {
        "r03 policy, plain ADMM": base,
        "AA m3 p5 on (z, y/rho)": {**base, "m": 3, "p": 5},
        "AA m3 p5 on v": {**base, "m": 3, "p": 5, "onv": True},
        "AA m2 p5 on v": {**base, "m": 2, "p": 5, "onv": True},
        "AA m3 p5 on v, fp32 history": {**base, "m": 3, "p": 5, "onv": True, "f32": True, "reg": 1e-6},
        "AA m3 p3 on v": {**base, "m": 3, "p": 3, "onv": True},
        "AA m3 p8 on v": {**base, "m": 3, "p": 8, "onv": True},
        "AA m3 p5 on v, rounds >= 1 only": {**base, "m": 3, "p": 5, "onv": True, "aa_from_round": 1},
        "AA m3 p5 on v, first block 60, flagged x1": {**base, "m": 3, "p": 5, "onv": True, "first_block": 60, "hard_factor": 1},
    }
    seeds = [int(x) for x in sys.argv[1:]] or [20250809, 1, 2, 3]
    tot = {k: [] for k in pols}
    for seed in seeds:
        b = mpcqp.synth.make_batch(4096, 10, 0.03, seed, allg, (0.3, 0.5, 0.7, 1.0))
        cand = np.where((b["gait_ids"] == 2) & (b["mu"] <= 0.5))[0]
        qps = [P.QP(b, i, cfg) for i in cand]
        print(f"seed {seed}: {len(qps)} amble QPs at mu <= 0.5", flush=True)
        for name, pol in pols.items():
            res = [solve(q, pol) for q in qps]
            c = np.array([cost(r) for r in res]); ok = np.array([r[0] for r in res]); it = np.array([r[1] for r in res]); rn = np.array([r[5] for r in res])
            tot[name].append((c.max(), c.mean(), ok.mean()))
            print(f"  {name:44s} solved {ok.mean():.4f} cost mean {c.mean():6.1f} p99 {np.percentile(c, 99):6.1f} max {c.max():6.1f} us | iters mean {it.mean():6.1f} | rounds {np.bincount(np.minimum(rn, 6)).tolist()}", flush=True)
    print("mean over the seeds of (worst solve, class mean, solved):")
    for k, v in tot.items():
        v = np.array(v); print(f"  {k:44s} max {v[:, 0].mean():6.1f} mean {v[:, 1].mean():6.1f} us solved {v[:, 2].mean():.4f}")


if __name__ == "__main__":
    main()
