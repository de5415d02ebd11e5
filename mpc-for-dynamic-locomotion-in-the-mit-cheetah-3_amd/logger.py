"""Run log with the reference's schema (src/logger.py:22-46), so plot tooling written against it keeps working.

Two persistence formats, chosen by the file extension:
* ``.pkl`` -- the reference's own format (src/logger.py:64-72): ``pickle.dump`` of the nested ``log`` dict, protocol 4, lists of
  floats / numpy arrays exactly where the reference puts them, so ``Logger.load_log`` / ``plot.py``-style consumers
  (src/plot.py:12-83) read a file written here unchanged.  ``load_log`` of a ``.pkl`` goes through an unpickler that only admits
  numpy's array reconstruction (no arbitrary callables).
* ``.npz`` -- ``numpy.savez`` of the flattened dict (path-keyed arrays), loadable with ``allow_pickle=False``.
"""
from __future__ import annotations

import pickle

import numpy as np

LEGS = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")
JOINTS = ("HipX", "HipY", "Knee")


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Plain containers, numbers, strings and numpy arrays / scalars -- nothing else is constructed."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
                ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"log files may only hold numpy arrays, not {module}.{name}")


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

    @classmethod
    def from_rollout(cls, params, actual, desired, forces, mpc_freq=0.0):
        """One robot's rows of a device roll-out (MPCBatch.rollout: actual / desired / forces [T,12]) in the reference's layout."""
        T = len(actual)
        lg = cls({"params": params, "total_sim_steps": T})
        lg.log["mpc_freq"] = float(mpc_freq)
        lg.log["time array"] = list(range(T))
        for t in range(T):
            lg.log_tracking_data(np.asarray(actual[t], float).tolist(), np.asarray(desired[t], float))
            for k, leg in enumerate(LEGS):
                for a, ax in enumerate("xyz"):
                    lg.log["FORCES"][leg][ax].append(float(forces[t][3 * k + a]))
        return lg

    def save_log(self, filename="simulation_log.pkl"):
        if str(filename).endswith(".pkl"):                           # src/logger.py:64-66
            with open(filename, "wb") as f:
                pickle.dump(self.log, f, protocol=4)
            return
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

    def load_log(self, filename="simulation_log.pkl"):
        if str(filename).endswith(".pkl"):                           # src/logger.py:69-72
            with open(filename, "rb") as f:
                self.log = _ArraysOnlyUnpickler(f).load()
            return self.log
        data = np.load(filename, allow_pickle=False)
        return {k: data[k] for k in data.files}

    def print_log_info(self):
        print(f"mpc_freq : {self.log['mpc_freq']},\nsim_params : {self.log['sim_params']},\n"
              f"total_sim_steps : {self.log['total_sim_steps']}")
