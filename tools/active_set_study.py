"""Developer study (CPU, numpy): when does the active set guessed from the ADMM iterate stop changing, and is it right?

Replays the engine's fixed-rho ADMM (rho = 1, sigma = 1e-6, relax = 1.6) on a sample of the bench workload and applies
the polish's primal-dual guess rule to every iterate.  Prints, per stopping rule "guess unchanged for S iterations",
the mean stopping iteration and the fraction of QPs whose guess at that point equals the guess at iteration 400.
"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp
import qp_spec as Q

NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = 400
batch = mpcqp.synth.config3(4096)
cfg = Q.QPConfig(N=10, delta=0.03, alpha=1e-2)
rng = np.random.default_rng(0)
idx = rng.choice(4096, NQ, replace=False)

def guess(u, y, lo, hi, mu):
    """The polish's rule (mpcqp_wrench.h, w_polish_rule) per leg-stage -> tuple of (zs, xs, ys)."""
    out = np.zeros((40, 3), dtype=np.int8)
    U = u.reshape(40, 3); Y = y.reshape(40, 5)
    st = hi.reshape(40, 5)[:, 0] > 0
    fz = U[:, 2]
    out[:, 0] = np.where(Y[:, 0] + (fz - hi.reshape(40, 5)[:, 0]) > 0, 1, np.where(Y[:, 0] + (fz - lo.reshape(40, 5)[:, 0]) < 0, -1, 0))
    for a, (ra, rb) in enumerate(((1, 2), (3, 4))):
        g1 = U[:, a] - mu * fz; g2 = U[:, a] + mu * fz
        h = Y[:, ra] + g1 > 0; l = Y[:, rb] + g2 < 0
        out[:, 1 + a] = np.where(h & l, np.where(g1 > -g2, 1, -1), np.where(h, 1, np.where(l, -1, 0)))
    out[~st] = 0
    return out

stops = {S: [] for S in (5, 10, 15, 20, 30)}
good = {S: [] for S in stops}
first_final = []
for b in idx:
    x0, r, c, xd, mu = batch["x0"][b], batch["r"][b], batch["contact"][b], batch["xdes"][b], float(batch["mu"][b])
    H, g, c0, G, lo, hi, Sx, Su = Q.condensed_qp(x0, r, c, xd, mu, cfg)
    n = H.shape[0]
    rho, sigma, relax = 1.0, 1e-6, 1.6
    Minv = np.linalg.inv(H + sigma * np.eye(n) + rho * G.T @ G)
    u = np.zeros(n); z = np.zeros(G.shape[0]); y = np.zeros(G.shape[0])
    sig = []
    for it in range(T):
        ut = Minv @ (sigma * u - g + G.T @ (rho * z - y))
        zt = G @ ut
        u = relax * ut + (1 - relax) * u
        zr = relax * zt + (1 - relax) * z
        zn = np.clip(zr + y / rho, lo, hi)
        y = y + rho * (zr - zn); z = zn
        sig.append(guess(u, y, lo, hi, mu).tobytes())
    final = sig[-1]
    ff = T
    while ff > 0 and sig[ff - 1] == final: ff -= 1
    first_final.append(ff)
    for S in stops:
        run = 0; stop = T - 1
        for it in range(25, T):
            run = run + 1 if sig[it] == sig[it - 1] else 0
            if run >= S: stop = it; break
        stops[S].append(stop); good[S].append(sig[stop] == final)
ff = np.array(first_final)
print(f"{NQ} QPs: iteration from which the guess equals the final (it 400) guess: median {np.median(ff):.0f} mean {ff.mean():.1f} "
      f"p75 {np.percentile(ff, 75):.0f} p90 {np.percentile(ff, 90):.0f};  <=100: {(ff <= 100).mean():.2f}  <=140: {(ff <= 140).mean():.2f}")
for S in stops:
    st = np.array(stops[S]); gd = np.array(good[S])
    print(f"stop when unchanged for {S:2d} iterations: mean stop {st.mean():6.1f}  median {np.median(st):4.0f}  p90 {np.percentile(st, 90):4.0f}  guess==final {gd.mean():.3f}")
for K in (60, 80, 100, 120, 140, 200):
    print(f"fixed K={K}: guess==final {np.mean(ff <= K):.3f}")
