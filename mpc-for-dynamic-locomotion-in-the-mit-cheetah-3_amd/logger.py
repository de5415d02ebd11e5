"""Run log with the reference's schema (src/logger.py:22-46), so plot tooling written against it keeps working.

Persistence is ``numpy.savez`` (no pickle): ``save_log`` flattens the nested dict into path-keyed arrays.
"""
from __future__ import annotations

import numpy as np

LEGS = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")
JOINTS = ("HipX", "HipY", "Knee")


class Logger:
    def __init__(self, initial):
        self.log = {
            "mpc_freq": 0,
            "sim_params": initial["params"],
            "total_sim_steps": initial["total_sim_steps"],
            "time array": [],
            "FEET POS": {l: {"actual": [], "des": []} for l in LEGS},
            "MPC PREDICTIONS": [],
            "TRACKING PERFORMANCE": {"actual": [], "desired": []},
            "FORCES": {l: {"x": [], "y": [], "z": []} for l in LEGS},
            "CONTROL EFFORT": {l: {f"{l[:2]}_{j}": [] for j in JOINTS} for l in LEGS},
        }

    def log_feet_data(self, actual, des, leg_name):                 # src/logger.py:49-51
        self.log["FEET POS"][leg_name]["actual"].append(actual)
        self.log["FEET POS"][leg_name]["des"].append(des)

    def log_mpc_predictions(self, x_log, x_des, forces_pred, t):    # src/logger.py:53-57
        self.log["MPC PREDICTIONS"].append({"time step": t, "predicted_state": x_log, "desired_state": x_des,
                                            "predicted forces": forces_pred})

    def log_tracking_data(self, actual, des):                       # src/logger.py:59-61
        self.log["TRACKING PERFORMANCE"]["actual"].append(actual)
        self.log["TRACKING PERFORMANCE"]["desired"].append(des)

    def save_log(self, filename="simulation_log.npz"):
        flat = {}

        def walk(prefix, node):
            if isinstance(node, dict):
                for k, v in node.items():
                    walk(f"{prefix}/{k}" if prefix else str(k), v)
            elif isinstance(node, list) and node and isinstance(node[0], dict):
                for i, v in enumerate(node):
                    walk(f"{prefix}/{i}", v)
            else:
                flat[prefix] = np.asarray(node)
        walk("", self.log)
        np.savez_compressed(filename, **flat)

    def load_log(self, filename="simulation_log.npz"):
        data = np.load(filename, allow_pickle=False)
        return {k: data[k] for k in data.files}

    def print_log_info(self):
        print(f"mpc_freq : {self.log['mpc_freq']},\nsim_params : {self.log['sim_params']},\n"
              f"total_sim_steps : {self.log['total_sim_steps']}")
