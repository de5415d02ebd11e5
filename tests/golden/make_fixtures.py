"""Generates the committed golden fixtures.  Runs ONLY in the build container (needs /root/reference).

What it does:
  1. reads the reference's committed run log with the non-executing reader (inert_log_reader.py) and exports
     plain arrays                                                        -> ref_log.npz
  2. imports the reference's planner glue (src/footstep_planner.py, src/foot_trajectory_generator.py,
     src/utils.py:compute_skew) with stub `casadi` / `dartpy` modules -- those two are imported by utils.py
     but not used by the planner -- and records plans / phases / swing trajectories for several gaits
                                                                         -> planner_golden.npz
  3. replays the parameter construction of MPC.solve (src/mpc.py:176-254) with the REFERENCE planner objects
     at selected ticks of the logged run, N in {10, 20, 60}              -> qp_inputs.npz
  4. solves those QPs with the CPU oracle (oracle/libmpcqp_oracle.so, KKT-verified) -> qp_optima.npz

src/mpc.py itself cannot be imported (casadi + its osqp plugin are absent from the image), so the QP
*outputs* of the reference are only available as the (unconverged) forces in the log; they are exported in
ref_log.npz as a loose sanity target, not as a parity target.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
LEGS = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from inert_log_reader import read_log  # noqa: E402


def export_log():
    L = read_log(os.path.join(REF, "simulation_log.pkl"))
    tp = L["TRACKING PERFORMANCE"]
    out = {
        "mpc_freq": np.float64(L["mpc_freq"]),
        "actual": np.array(tp["actual"], float),               # [1000,12]
        "desired": np.array(tp["desired"], float),             # [1000,12]
        "feet_actual": np.stack([np.array(L["FEET POS"][l]["actual"], float) for l in LEGS], axis=1),  # [1000,4,3]
        "feet_des": np.stack([np.array(L["FEET POS"][l]["des"], float) for l in LEGS], axis=1),
        "forces": np.stack([np.stack([np.array(L["FORCES"][l][a], float) for a in "xyz"], axis=1) for l in LEGS],
                           axis=1).reshape(1000, 12),          # [1000,12] stage-0 GRFs per tick
        "time": np.array(L["time array"], float),
    }
    sp = L["sim_params"]
    for k in ("g", "h", "step_height", "ss_duration", "ds_duration", "world_time_step", "total_steps", "N", "dof",
              "theta_dot", "log_samples"):
        out["param_" + k] = np.float64(sp[k])
    out["param_mu"] = np.float64(sp["µ"])
    out["param_first_swing"] = np.array(sp["first_swing"], float)
    out["param_v_com_ref"] = np.array(sp["v_com_ref"], float)
    for i, p in enumerate(L["MPC PREDICTIONS"]):
        out[f"pred{i}_t"] = np.float64(p["time step"])
        out[f"pred{i}_state"] = np.array(p["predicted_state"], float)     # [12,61]
        out[f"pred{i}_desired"] = np.array(p["desired_state"], float)     # [12,61]
        out[f"pred{i}_fz"] = np.array(p["predicted forces"], float)       # [4,60]
    return out


def import_reference_planner():
    import matplotlib
    matplotlib.use("Agg")
    sys.dont_write_bytecode = True
    cas = types.ModuleType("casadi"); cas.MX = object; cas.DM = object
    sys.modules.setdefault("casadi", cas)
    sys.modules.setdefault("dartpy", types.ModuleType("dartpy"))
    sys.path.insert(0, REF)
    from footstep_planner import FootstepPlanner
    from foot_trajectory_generator import FootTrajectoryGenerator
    from utils import compute_skew
    return FootstepPlanner, FootTrajectoryGenerator, compute_skew


def params_from_log(log, **over):
    p = {
        "g": float(log["param_g"]), "h": float(log["param_h"]), "step_height": float(log["param_step_height"]),
        "ss_duration": int(log["param_ss_duration"]), "ds_duration": int(log["param_ds_duration"]),
        "world_time_step": float(log["param_world_time_step"]), "total_steps": int(log["param_total_steps"]),
        "first_swing": log["param_first_swing"].astype(int).copy(), "µ": float(log["param_mu"]),
        "N": int(log["param_N"]), "dof": int(log["param_dof"]), "v_com_ref": log["param_v_com_ref"].copy(),
        "theta_dot": float(log["param_theta_dot"]), "log_samples": int(log["param_log_samples"]),
    }
    p.update(over)
    return p


def initial_from_log(log):
    ini = {l: log["feet_actual"][0, j].copy() for j, l in enumerate(LEGS)}
    ini.update(roll=float(log["actual"][0, 0]), pitch=float(log["actual"][0, 1]), yaw=float(log["actual"][0, 2]),
               com_position=log["actual"][0, 3:6].copy())
    return ini


def plan_arrays(planner):
    pos = np.array([[np.asarray(s["pos"][l], float) for l in LEGS] for s in planner.plan])  # [S,4,3]
    ang = np.array([np.atleast_1d(np.asarray(s["ang"], float))[-1] for s in planner.plan])  # yaw per step
    fid = np.array([np.asarray(s["feet_id"], int) for s in planner.plan])
    return pos, ang, fid


def planner_goldens(log):
    FootstepPlanner, FootTrajectoryGenerator, compute_skew = import_reference_planner()
    out = {}
    gaits = {"gallop": [0, 0, 1, 1], "trot": [1, 0, 0, 1], "amble": [1, 0, 1, 0], "pronk": [0, 0, 0, 0]}
    T = 400
    for name, fs in gaits.items():
        for tag, extra in (("", {}), ("_turn", {"theta_dot": 0.3, "v_com_ref": np.array([0.1, 0.02, 0.0])})):
            params = params_from_log(log, first_swing=np.array(fs), **extra)
            pl = FootstepPlanner(initial_from_log(log), params, show=False)
            pos, ang, fid = plan_arrays(pl)
            out[f"{name}{tag}_plan_pos"] = pos
            out[f"{name}{tag}_plan_ang"] = ang
            out[f"{name}{tag}_plan_feet_id"] = fid
            out[f"{name}{tag}_step_index"] = np.array([pl.get_step_index_at_time(t) for t in range(T)])
            out[f"{name}{tag}_phase"] = np.array([pl.get_phase_at_time(t) for t in range(T)])
            # swing trajectories (fresh planner: the generator mutates plan[...]['feet_id'], ftg.py:53-54)
            pl2 = FootstepPlanner(initial_from_log(log), params, show=False)
            tg = FootTrajectoryGenerator(pl2, params)
            traj = np.zeros((T, 4, 3, 6))
            for t in range(T):
                for j, l in enumerate(LEGS):
                    d = tg.generate_feet_trajectories_at_time(t, l)
                    traj[t, j, 0], traj[t, j, 1], traj[t, j, 2] = d["pos"], d["vel"], d["acc"]
            out[f"{name}{tag}_swing_traj"] = traj
            out[f"{name}{tag}_plan_feet_id_after_traj"] = np.array([np.asarray(s["feet_id"], int) for s in pl2.plan])
    rng = np.random.default_rng(7)
    vs = rng.normal(size=(8, 3))
    out["skew_in"] = vs
    out["skew_out"] = np.array([compute_skew(v) for v in vs])

    # closed-loop replay of the logged desired foot positions (src/main.py:152-167, 225-234)
    params = params_from_log(log)
    pl = FootstepPlanner(initial_from_log(log), params, show=False)
    tg = FootTrajectoryGenerator(pl, params)
    des = np.zeros((1000, 4, 3))
    phase = np.zeros((1000, 4), dtype=np.uint8)     # contact[0] of the QP solved at tick t (src/mpc.py:249-252, i = 0)
    for t in range(1000):
        phase[t] = pl.get_phase_at_time(t)          # queried by MPC.solve before the swing controller runs (src/main.py:155-162)
        si = pl.get_step_index_at_time(t)
        gait = list(pl.plan[si]["feet_id"])
        for j, l in enumerate(LEGS):
            if gait[j] == 1:
                des[t, j] = pl.plan[si]["pos"][l]
            else:
                p = tg.generate_feet_trajectories_at_time(t, l)["pos"][3:].copy()
                if p[2] < 0:
                    p[2] = 0
                des[t, j] = p
    out["replay_feet_des"] = des
    out["replay_phase"] = phase
    out["replay_feet_des_maxerr_vs_log"] = np.float64(np.abs(des - log["feet_des"]).max())
    return out


def qp_inputs(log, ticks=(0, 14, 15, 20, 25, 80, 150, 299, 300, 999), horizons=(10, 20, 60)):
    """Parameter construction of MPC.solve (src/mpc.py:176-254) with the reference's own planner objects."""
    FootstepPlanner, FootTrajectoryGenerator, compute_skew = import_reference_planner()
    out = {"ticks": np.array(ticks), "horizons": np.array(horizons)}
    for N in horizons:
        X0, XD, R, C = [], [], [], []
        for t in ticks:
            params = params_from_log(log, N=N)
            delta = params["world_time_step"]
            pl = FootstepPlanner(initial_from_log(log), params, show=False)   # fresh per fixture
            tg = FootTrajectoryGenerator(pl, params)
            v_com_gait = params["v_com_ref"]; omega = params["theta_dot"]
            if pl.get_step_index_at_time(t) == params["total_steps"] - 1:      # src/mpc.py:181-183
                v_com_gait = params["v_com_ref"] * 0; omega = params["theta_dot"] * 0
            x0 = np.concatenate([log["actual"][t], [params["g"]]])             # src/mpc.py:190-198
            xd = np.zeros((13, N + 1))                                         # src/mpc.py:202-214
            xd[:2, :] = np.array([[log["actual"][0, 0]], [log["actual"][0, 1]]])
            xd[8, :] = omega
            xd[9:12, :] = v_com_gait.reshape(3, 1)
            xd[12, :] = params["g"]
            xd[2, 0] = log["desired"][t, 2]                                    # yaw_start at this tick
            xd[3:6, 0] = log["desired"][t, 3:6]                                # com_pos_start at this tick
            for i in range(1, N + 1):
                xd[2, i] = xd[2, i - 1] + omega * delta
                xd[3:6, i] = xd[3:6, i - 1] + v_com_gait * delta

            def update_r_num(time, leg, next_com):                            # src/mpc.py:306-318
                gait = pl.get_phase_at_time(time)
                if pl.is_swing(leg, gait) == 1:
                    return tg.generate_feet_trajectories_at_time(time, leg)["pos"][3:] - next_com
                step = pl.get_step_index_at_time(time)
                return np.asarray(pl.plan[step]["pos"][leg], float) - next_com

            r = np.zeros((N, 4, 3))
            cur = [log["feet_actual"][t, j] - log["actual"][t, 3:6] for j in range(4)]   # src/mpc.py:223-226
            for i in range(N):                                                 # src/mpc.py:228-239
                for j in range(4):
                    r[i, j] = cur[j]
                cur = [update_r_num(t + i + 1, l, xd[3:6, i + 1]) for l in LEGS]
            contact = np.array([pl.get_phase_at_time(t + i) for i in range(N)], dtype=np.uint8)  # src/mpc.py:249-252
            X0.append(x0); XD.append(xd.T.copy()); R.append(r); C.append(contact)
        out[f"N{N}_x0"] = np.array(X0); out[f"N{N}_xdes"] = np.array(XD)
        out[f"N{N}_r"] = np.array(R); out[f"N{N}_contact"] = np.array(C)
    out["delta"] = np.float64(log["param_world_time_step"]); out["mu"] = np.float64(log["param_mu"])
    return out


def qp_optima(qpi):
    import mpcqp
    import qp_spec as S
    lib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
    out = {}
    for N in (10, 20, 60):
        x0, xd, r, c = (qpi[f"N{N}_{k}"] for k in ("x0", "xdes", "r", "contact"))
        mu = np.full(len(x0), float(qpi["mu"]))
        for alpha, tag in ((0.0, "a0"), (1e-2, "a1e-2"), (1e-4, "a1e-4")):
            cfg = lib.default_config(N=N, delta=float(qpi["delta"]), alpha=alpha, eps_abs=1e-10, eps_rel=1e-10,
                                     max_iter=200000, polish_max=30)
            eng = mpcqp.Engine(lib, cfg)
            sol = eng.solve_batch_host(x0, r, c, xd, mu)
            pc = S.QPConfig(N=N, delta=float(qpi["delta"]), alpha=alpha)
            J = np.array([S.objective(sol["X"][i], sol["u"][i], xd[i], pc) for i in range(len(x0))])
            Wn = np.array([S.net_wrench(sol["u"][i], r[i], c[i], pc) for i in range(len(x0))])
            out[f"N{N}_{tag}_u"] = sol["u"]; out[f"N{N}_{tag}_X"] = sol["X"]; out[f"N{N}_{tag}_J"] = J
            out[f"N{N}_{tag}_wrench"] = Wn; out[f"N{N}_{tag}_status"] = sol["status"]
            print(f"oracle optima N={N} alpha={alpha}: status {sol['status']}, iters {sol['iters']}")
    return out


if __name__ == "__main__":
    log = export_log()
    np.savez_compressed(os.path.join(HERE, "ref_log.npz"), **log)
    pg = planner_goldens(log)
    print("closed-loop desired-feet replay vs log: max abs err", float(pg["replay_feet_des_maxerr_vs_log"]))
    np.savez_compressed(os.path.join(HERE, "planner_golden.npz"), **pg)
    qpi = qp_inputs(log)
    np.savez_compressed(os.path.join(HERE, "qp_inputs.npz"), **qpi)
    if "--no-optima" not in sys.argv:
        np.savez_compressed(os.path.join(HERE, "qp_optima.npz"), **qp_optima(qpi))
    for f in ("ref_log.npz", "planner_golden.npz", "qp_inputs.npz", "qp_optima.npz"):
        p = os.path.join(HERE, f)
        if os.path.exists(p):
            print(f, os.path.getsize(p) // 1024, "KiB")
