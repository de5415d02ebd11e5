"""ctypes binding of the C-ABI declared in include/mpcqp.h.

The product path loads ``csrc/libmpcqp.so`` (hand-written HIP for gfx950) and nothing else: if that library is
missing or cannot be loaded, importing the engine raises -- there is no CPU fallback.  The binding itself is
generic over the library path because the CPU checker under ``oracle/`` exports the same symbols with host
pointers; only tests / smoke / bench's cpu_baseline leg ever pass that path in.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int32, c_int64, c_uint32, c_void_p

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_PKG_DIR, "csrc", "libmpcqp.so")

# include/mpcqp.h constants
STATUS_UNSOLVED, STATUS_SOLVED_POLISHED, STATUS_SOLVED_ADMM, STATUS_MAX_ITER, STATUS_NONFINITE = 0, 1, 2, 3, -1
DISC_EULER, DISC_ZOH = 0, 1
DTYPE_F32, DTYPE_F64 = 0, 1
PREC_F32, PREC_MIXED, PREC_F64 = 0, 1, 2
FLAG_POLISH, FLAG_WARM_START, FLAG_GENERAL_KERNEL, FLAG_NATURAL_ORDER, FLAG_WARM_SHIFT, FLAG_TILE_KERNEL, FLAG_NO_TIMING = 1, 2, 4, 8, 16, 32, 64
FLAG_STAGE_KERNEL = 128

EXPORTED_SYMBOLS = (
    "mpcqp_version", "mpcqp_default_config", "mpcqp_create", "mpcqp_destroy", "mpcqp_solve_batch",
    "mpcqp_solve_batch_gait", "mpcqp_solve_batch_gait_steps", "mpcqp_torque_map", "mpcqp_default_leg_geometry", "mpcqp_leg_jacobians", "mpcqp_last_kernel_ms", "mpcqp_last_error", "mpcqp_reserve", "mpcqp_rollout",
)


class MpcQpLegGeometry(ctypes.Structure):
    """Mirror of struct MpcQpLegGeometry (include/mpcqp.h)."""
    _fields_ = [("size", c_uint32), ("reserved", c_uint32), ("hip_x", c_double * 3 * 4), ("hip_y", c_double * 3 * 4),
                ("knee", c_double * 3), ("foot", c_double * 3), ("axis_x", c_double * 3), ("axis_y", c_double * 3)]


class MpcQpConfig(ctypes.Structure):
    """Mirror of ``struct MpcQpConfig`` (include/mpcqp.h)."""
    _fields_ = [
        ("size", c_uint32), ("N", c_int32), ("delta", c_double), ("m", c_double),
        ("Ibody_inv", c_double * 3), ("w", c_double * 13), ("alpha", c_double),
        ("f_min", c_double), ("f_max", c_double), ("disc", c_int32), ("dtype", c_int32),
        ("precision", c_int32), ("flags", c_uint32), ("rho", c_double), ("sigma", c_double),
        ("relax", c_double), ("max_iter", c_int32), ("check_every", c_int32),
        ("eps_abs", c_double), ("eps_rel", c_double), ("polish_max", c_int32), ("device", c_int32),
        ("first_block", c_int32), ("incr_legs", c_int32), ("listed_max", c_int32), ("adapt_thr", c_float), ("alpha_floor", c_double),
        ("polish_patience", c_int32), ("polish_cheap_steps", c_int32), ("polish_cheap_legs", c_int32), ("hard_block_x10", c_int32), ("polish_last_patience", c_int32), ("accel", c_int32), ("accel_restart", c_int32), ("reserved0", c_int32),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class MpcQpError(RuntimeError):
    pass


class Library:
    """A loaded shared object exporting the mpcqp C-ABI."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise MpcQpError(f"mpcqp library not found: {path} (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        self.path = path
        self.lib = ctypes.CDLL(path)
        L = self.lib
        L.mpcqp_version.restype = c_uint32
        L.mpcqp_default_config.argtypes = [ctypes.POINTER(MpcQpConfig)]
        L.mpcqp_default_config.restype = c_int32
        L.mpcqp_create.argtypes = [ctypes.POINTER(MpcQpConfig), ctypes.POINTER(c_void_p)]
        L.mpcqp_create.restype = c_int32
        L.mpcqp_destroy.argtypes = [c_void_p]
        L.mpcqp_reserve.argtypes = [c_void_p, c_int64]
        L.mpcqp_rollout.argtypes = [c_void_p, c_int64, c_int32, c_int32] + [c_void_p] * 12
        L.mpcqp_rollout.restype = ctypes.c_int
        L.mpcqp_reserve.restype = ctypes.c_int
        L.mpcqp_destroy.restype = c_int32
        L.mpcqp_solve_batch.argtypes = [c_void_p, c_int64] + [c_void_p] * 11
        L.mpcqp_solve_batch.restype = c_int32
        L.mpcqp_solve_batch_gait.argtypes = [c_void_p, c_int64] + [c_void_p] * 13
        L.mpcqp_solve_batch_gait.restype = c_int32
        L.mpcqp_solve_batch_gait_steps.argtypes = [c_void_p, c_int64, c_int32] + [c_void_p] * 13
        L.mpcqp_solve_batch_gait_steps.restype = c_int32
        L.mpcqp_torque_map.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        L.mpcqp_torque_map.restype = c_int32
        L.mpcqp_default_leg_geometry.argtypes = [ctypes.POINTER(MpcQpLegGeometry)]
        L.mpcqp_default_leg_geometry.restype = c_int32
        L.mpcqp_leg_jacobians.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, ctypes.POINTER(MpcQpLegGeometry), c_void_p, c_void_p, c_void_p]
        L.mpcqp_leg_jacobians.restype = c_int32
        L.mpcqp_last_kernel_ms.argtypes = [c_void_p, ctypes.POINTER(c_float)]
        L.mpcqp_last_kernel_ms.restype = c_int32
        L.mpcqp_last_error.argtypes = [c_void_p]
        L.mpcqp_last_error.restype = c_char_p

    def version(self) -> int:
        return int(self.lib.mpcqp_version())

    def default_config(self, **overrides) -> MpcQpConfig:
        cfg = MpcQpConfig()
        rc = self.lib.mpcqp_default_config(ctypes.byref(cfg))
        if rc != 0:
            raise MpcQpError(f"mpcqp_default_config failed: {rc}")
        # the ADMM block length is tuned per horizon: 100 iterations at N = 10 (the C default), 200 at N = 20 (measured:
        # +20 % throughput and 100 % instead of 98.8 % solved, tools/n20_knobs.py); explicit overrides win
        n = int(overrides.get("N", cfg.N))
        stage = n not in (10, 20) or bool(int(overrides.get("flags", cfg.flags)) & FLAG_STAGE_KERNEL)
        if n != 10 and "check_every" not in overrides:   # scale the library's own N = 10 defaults
            # (stage-wise engine, any other horizon: an iteration costs two recursions of N steps there and a polish attempt one
            #  factorisation, so shorter blocks pay -- 2 N: 120 at the reference's N = 60, tools/stage_sweep.py: 79.9 k QP/s on the
            #  logged ticks against 29.3 k at 10 N, 38.2 k against 20.3 k on a synthetic mixed batch, 100 % solved at both; with the
            #  Anderson-accelerated first block 5 N / 3 -- 100 at N = 60 -- tools/stage_accel.py, profiles/r03_stage_accel.txt:
            #  logged ticks 133.6 -> 143.5 k QP/s MIXED, 89.8 -> 133.4 k F64)
            cfg.check_every = max(50, (5 * n) // 3) if stage else max(1, cfg.check_every * n // 10)
            if "max_iter" not in overrides:
                cfg.max_iter = max(400 if stage else 1, cfg.max_iter * n // 10)
        # ... and so is the polish budget per round: twice the leg-stages, twice the steps (N = 20, eight batches of 4096: 14 -> 3
        # QPs left at the iteration cap at the same rate, tools/adapt_sweep.py; the steps inside a round update the inverse)
        if n != 10 and "polish_max" not in overrides:
            cfg.polish_max = max(4, cfg.polish_max * min(n, 20) // 10)
        for k, v in overrides.items():
            if k in ("w", "Ibody_inv"):
                arr = getattr(cfg, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                if not hasattr(cfg, k):
                    raise AttributeError(f"MpcQpConfig has no field {k!r}")
                setattr(cfg, k, v)
        return cfg


class Engine:
    """One handle = one configuration on one device / stream.  Not thread-safe (include/mpcqp.h)."""

    def __init__(self, library: Library, cfg: MpcQpConfig):
        self.library = library
        self.cfg = cfg
        self._h = c_void_p()
        rc = library.lib.mpcqp_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            raise MpcQpError(f"mpcqp_create failed with code {rc}")

    def close(self):
        if self._h:
            self.library.lib.mpcqp_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        return (self.library.lib.mpcqp_last_error(self._h) or b"").decode()

    def reserve(self, B):
        """Pre-size the batch-dependent workspace so that no later solve of at most B QPs allocates or synchronises."""
        rc = self.library.lib.mpcqp_reserve(self._h, int(B))
        if rc != 0:
            raise MpcQpError(f"mpcqp_reserve failed with code {rc}: {self.last_error()}")

    def solve_batch_ptr(self, B, x0, r, contact, xdes, mu, u_out, X_out, status, iters, res, stream=0):
        """Raw call: every argument is an integer address (device memory for the product library)."""
        rc = self.library.lib.mpcqp_solve_batch(self._h, int(B), x0, r, contact, xdes, mu, u_out, X_out or None,
                                                status, iters, res or None, stream or None)
        if rc != 0:
            raise MpcQpError(f"mpcqp_solve_batch failed with code {rc}: {self.last_error()}")

    def solve_batch_gait_ptr(self, B, x0, ref, feet0, footholds, gait, feet_id, mu, u_out, X_out, status, iters, res, stream=0):
        """Raw call of the gait entry point (addresses; device memory for the product library)."""
        rc = self.library.lib.mpcqp_solve_batch_gait(self._h, int(B), x0, ref, feet0, footholds, gait, feet_id, mu, u_out,
                                                     X_out or None, status, iters, res or None, stream or None)
        if rc != 0:
            raise MpcQpError(f"mpcqp_solve_batch_gait failed with code {rc}: {self.last_error()}")

    def solve_batch_gait_steps_ptr(self, B, S, x0, ref, feet0, footholds, gait, feet_id, mu, u_out, X_out, status, iters, res, stream=0):
        """Raw call of the gait entry point with S plan steps per robot (footholds [B,S,4,3], feet_id [B,S,4])."""
        rc = self.library.lib.mpcqp_solve_batch_gait_steps(self._h, int(B), int(S), x0, ref, feet0, footholds, gait, feet_id, mu, u_out,
                                                           X_out or None, status, iters, res or None, stream or None)
        if rc != 0:
            raise MpcQpError(f"mpcqp_solve_batch_gait_steps failed with code {rc}: {self.last_error()}")

    def rollout_ptr(self, B, T, S, x, ref, plan_pos, plan_feet_id, plan_meta, tick, mu, actual, desired, forces, solved, stream=0):
        """Raw call of the closed-loop roll-out (include/mpcqp.h, mpcqp_rollout); every argument is an integer address."""
        rc = self.library.lib.mpcqp_rollout(self._h, int(B), int(T), int(S), x, ref, plan_pos, plan_feet_id, plan_meta, tick, mu,
                                            actual or None, desired or None, forces or None, solved or None, stream or None)
        if rc != 0:
            raise MpcQpError(f"mpcqp_rollout failed with code {rc}: {self.last_error()}")

    def rollout_host(self, x, ref, plan_pos, plan_feet_id, plan_meta, tick, mu, T):
        """Oracle convenience (host memory, float64): returns the advanced (x, ref, tick) and the per-tick logs."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64).copy()
        x, ref, mu = f(x), f(ref), f(mu)
        pos = f(plan_pos); fid = np.ascontiguousarray(plan_feet_id, dtype=np.uint8)
        meta = np.ascontiguousarray(plan_meta, dtype=np.int32); tick = np.ascontiguousarray(tick, dtype=np.int32).copy()
        B, S = pos.shape[0], pos.shape[1]
        actual = np.zeros((B, T, 12)); desired = np.zeros((B, T, 12)); forces = np.zeros((B, T, 12)); solved = np.zeros(B, np.int32)
        self.rollout_ptr(B, T, S, x.ctypes.data, ref.ctypes.data, pos.ctypes.data, fid.ctypes.data, meta.ctypes.data, tick.ctypes.data,
                         mu.ctypes.data, actual.ctypes.data, desired.ctypes.data, forces.ctypes.data, solved.ctypes.data)
        return {"x": x, "ref": ref, "tick": tick, "actual": actual, "desired": desired, "forces": forces, "solved": solved}

    def torque_map_ptr(self, B, u, jac, tau, stream=0):
        rc = self.library.lib.mpcqp_torque_map(self._h, int(B), u, jac, tau, stream or None)
        if rc != 0:
            raise MpcQpError(f"mpcqp_torque_map failed with code {rc}: {self.last_error()}")

    def leg_jacobians_ptr(self, B, q, rot, jac, foot=0, geometry=None, stream=0):
        rc = self.library.lib.mpcqp_leg_jacobians(self._h, int(B), q, rot or None, ctypes.byref(geometry) if geometry is not None else None,
                                                  jac, foot or None, stream or None)
        if rc != 0:
            raise MpcQpError(f"mpcqp_leg_jacobians failed with code {rc}: {self.last_error()}")

    def leg_jacobians_host(self, q, rot=None, geometry=None):
        """Host-memory call (the CPU checker): q [B,4,3], rot [B,3,3] or None -> (jac [B,4,3,3], foot [B,4,3]), float64."""
        q = np.ascontiguousarray(q, dtype=np.float64); B = q.shape[0]
        rot = None if rot is None else np.ascontiguousarray(rot, dtype=np.float64)
        jac = np.zeros((B, 4, 3, 3)); foot = np.zeros((B, 4, 3))
        self.leg_jacobians_ptr(B, q.ctypes.data, 0 if rot is None else rot.ctypes.data, jac.ctypes.data, foot.ctypes.data, geometry)
        return jac, foot

    def last_kernel_ms(self) -> float:
        ms = c_float()
        rc = self.library.lib.mpcqp_last_kernel_ms(self._h, ctypes.byref(ms))
        if rc != 0:
            raise MpcQpError(f"mpcqp_last_kernel_ms failed with code {rc}: {self.last_error()}")
        return float(ms.value)

    # host-pointer convenience (numpy); valid for libraries that take host memory
    def solve_batch_host(self, x0, r, contact, xdes, mu, want_X=True):
        N = self.cfg.N
        ft = np.float64 if self.cfg.dtype == DTYPE_F64 else np.float32
        x0 = np.ascontiguousarray(x0, dtype=ft); r = np.ascontiguousarray(r, dtype=ft)
        xdes = np.ascontiguousarray(xdes, dtype=ft); mu = np.ascontiguousarray(mu, dtype=ft)
        contact = np.ascontiguousarray(contact, dtype=np.uint8)
        B = x0.shape[0]
        assert x0.shape == (B, 13) and r.shape == (B, N, 4, 3) and contact.shape == (B, N, 4)
        assert xdes.shape == (B, N + 1, 13) and mu.shape == (B,)
        u = np.zeros((B, N, 12), ft)
        X = np.zeros((B, N + 1, 13), ft) if want_X else None
        status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); res = np.zeros((B, 2), np.float32)
        self.solve_batch_ptr(B, x0.ctypes.data, r.ctypes.data, contact.ctypes.data, xdes.ctypes.data, mu.ctypes.data,
                             u.ctypes.data, X.ctypes.data if want_X else None, status.ctypes.data, iters.ctypes.data,
                             res.ctypes.data)
        return {"u": u, "X": X, "status": status, "iters": iters, "res": res}


    def solve_batch_gait_host(self, g, want_X=True):
        """Host-pointer convenience for the gait entry point; `g` is a dict as produced by synth.make_gait_batch (footholds [B,S,4,3])."""
        N = self.cfg.N
        ft = np.float64 if self.cfg.dtype == DTYPE_F64 else np.float32
        a = {k: np.ascontiguousarray(g[k], dtype=ft) for k in ("x0", "ref", "feet0", "footholds", "mu")}
        gait = np.ascontiguousarray(g["gait"], dtype=np.int32); fid = np.ascontiguousarray(g["feet_id"], dtype=np.uint8)
        B, S = a["x0"].shape[0], a["footholds"].shape[1]
        u = np.zeros((B, N, 12), ft); X = np.zeros((B, N + 1, 13), ft) if want_X else None
        status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); res = np.zeros((B, 2), np.float32)
        self.solve_batch_gait_steps_ptr(B, S, a["x0"].ctypes.data, a["ref"].ctypes.data, a["feet0"].ctypes.data, a["footholds"].ctypes.data,
                                        gait.ctypes.data, fid.ctypes.data, a["mu"].ctypes.data, u.ctypes.data,
                                        X.ctypes.data if want_X else None, status.ctypes.data, iters.ctypes.data, res.ctypes.data)
        return {"u": u, "X": X, "status": status, "iters": iters, "res": res}


_product = None


def product_library() -> Library:
    """The HIP engine.  Raises (loudly) when the extension has not been built -- never falls back."""
    global _product
    if _product is None:
        # PyTorch-ROCm carries its own HIP runtime: it has to be in the process BEFORE libmpcqp.so pulls in /opt/rocm's, or the
        # one loaded second sees no device (mpcqp_create: MPCQP_ENODEV after `build()` + `smoke()` in one process, measured).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _product = Library(PRODUCT_LIB)
    return _product
