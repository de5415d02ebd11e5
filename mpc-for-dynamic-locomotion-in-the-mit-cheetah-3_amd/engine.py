"""Batched solve surface over the HIP engine (torch is used only for device memory and streams).

``MPCBatch.solve_batch`` is the B>1 counterpart of the reference's ``MPC.solve`` (src/mpc.py:176-303): it
takes the operator tuple that the reference pushes into CasADi with seven ``opt.set_value`` calls
(src/mpc.py:242-255) in compact batched form and returns ``sol.value(U)`` / ``sol.value(X)``
(src/mpc.py:265-268) for every instance, plus per-QP status / iteration / residual outputs.
"""
from __future__ import annotations

import numpy as np

from . import _capi


def _torch():
    import torch
    return torch


class MPCBatch:
    """One engine handle on one GPU.  All tensors live on ``cuda:<device>``; nothing is copied to the host."""

    def __init__(self, N=10, delta=0.03, device=0, io_dtype="f32", precision="mixed", warm_start=False, warm_shift=False,
                 **overrides):
        """``warm_start=True`` sets MPCQP_FLAG_WARM_START: every solve is seeded with the forces already in the output
        buffer -- the previous solve's solution unless ``u_init`` is passed -- like ``opt.set_initial(U, sol.value(U))``
        in the reference (src/mpc.py:270-271)."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise _capi.MpcQpError("MPCBatch needs a GPU: the mpcqp engine has no CPU path")
        lib = _capi.product_library()
        if warm_start:
            overrides["flags"] = int(overrides.get("flags", _capi.FLAG_POLISH)) | _capi.FLAG_WARM_START
            if warm_shift:   # the buffer holds the previous control tick's solution: the engine shifts it (and its duals)
                overrides["flags"] |= _capi.FLAG_WARM_SHIFT
        self.warm_start = bool(warm_start)
        cfg = lib.default_config(N=N, delta=delta, device=device,
                                 dtype={"f32": _capi.DTYPE_F32, "f64": _capi.DTYPE_F64}[io_dtype],
                                 precision={"f32": _capi.PREC_F32, "mixed": _capi.PREC_MIXED, "f64": _capi.PREC_F64}[precision],
                                 **overrides)
        self.N, self.delta = N, delta
        self.device = torch.device("cuda", device)
        self.tdtype = torch.float32 if io_dtype == "f32" else torch.float64
        self.engine = _capi.Engine(lib, cfg)
        self.cfg = cfg
        self._out = {}

    def upload(self, batch):
        """Host numpy batch (mpcqp.synth layout) -> resident device tensors."""
        torch = _torch()
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=self.tdtype).to(self.device).contiguous()
        return {"x0": f(batch["x0"]), "r": f(batch["r"]), "xdes": f(batch["xdes"]), "mu": f(batch["mu"]),
                "contact": torch.as_tensor(np.ascontiguousarray(batch["contact"], dtype=np.uint8)).to(self.device).contiguous()}

    def _outputs(self, B, want_X, st=None):
        torch = _torch()
        key = (B, want_X)
        if key not in self._out:   # allocated once per batch size, reused afterwards
            N = self.N
            # The zero fill below is a kernel on torch's CURRENT stream; the solve that reads the buffer (warm start) or writes it runs
            # on `st`, which may be a non-blocking side stream that nothing orders after it: fill on `st` itself.
            with torch.cuda.stream(st if st is not None else torch.cuda.current_stream(self.device)):
                self._out[key] = {
                    "u": torch.zeros((B, N, 12), dtype=self.tdtype, device=self.device),   # zeros = "no guess" for a warm-started engine
                    "X": torch.empty((B, N + 1, 13), dtype=self.tdtype, device=self.device) if want_X else None,
                    "status": torch.empty(B, dtype=torch.int32, device=self.device),
                    "iters": torch.empty(B, dtype=torch.int32, device=self.device),
                    "res": torch.empty((B, 2), dtype=torch.float32, device=self.device),
                }
        return self._out[key]

    def _seed(self, out, u_init, st):
        if u_init is None:
            return
        if not self.warm_start:
            raise ValueError("u_init needs an engine created with warm_start=True")
        if tuple(u_init.shape) != tuple(out["u"].shape) or u_init.dtype != self.tdtype or u_init.device != self.device:
            raise ValueError(f"u_init must be a {tuple(out['u'].shape)} {self.tdtype} tensor on {self.device}")
        if u_init.data_ptr() != out["u"].data_ptr():
            with _torch().cuda.stream(st):
                out["u"].copy_(u_init)

    def solve_batch(self, x0, r, contact, xdes, mu, want_X=False, stream=None, u_init=None):
        """Asynchronous on ``stream`` (default: torch's current stream); results valid after a stream sync."""
        torch = _torch()
        N = self.N
        B = int(x0.shape[0])
        for t, shape, dt in ((x0, (B, 13), self.tdtype), (r, (B, N, 4, 3), self.tdtype), (contact, (B, N, 4), torch.uint8),
                             (xdes, (B, N + 1, 13), self.tdtype), (mu, (B,), self.tdtype)):
            if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"operand mismatch: expected {shape} {dt} contiguous on {self.device}, got "
                                 f"{tuple(t.shape)} {t.dtype} on {t.device}")
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        out = self._outputs(B, want_X, st)
        self._seed(out, u_init, st)
        self.engine.solve_batch_ptr(B, x0.data_ptr(), r.data_ptr(), contact.data_ptr(), xdes.data_ptr(), mu.data_ptr(),
                                    out["u"].data_ptr(), out["X"].data_ptr() if want_X else None, out["status"].data_ptr(),
                                    out["iters"].data_ptr(), out["res"].data_ptr(), st.cuda_stream)
        return out

    def upload_gait(self, g):
        """Host numpy gait descriptors (mpcqp.synth.make_gait_batch layout) -> resident device tensors."""
        torch = _torch()
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=self.tdtype).to(self.device).contiguous()
        return {"x0": f(g["x0"]), "ref": f(g["ref"]), "feet0": f(g["feet0"]), "footholds": f(g["footholds"]), "mu": f(g["mu"]),
                "gait": torch.as_tensor(np.ascontiguousarray(g["gait"], dtype=np.int32)).to(self.device).contiguous(),
                "feet_id": torch.as_tensor(np.ascontiguousarray(g["feet_id"], dtype=np.uint8)).to(self.device).contiguous()}

    def solve_batch_gait(self, x0, ref, feet0, footholds, gait, feet_id, mu, want_X=False, stream=None, u_init=None):
        """Gait entry point: contact masks, stance lever arms and x_des are generated on the device (include/mpcqp.h) from S plan steps
        per robot (footholds [B,S,4,3], feet_id [B,S,4]; S = 2 is the original two-step form)."""
        torch = _torch()
        N = self.N
        B = int(x0.shape[0])
        S = int(footholds.shape[1]) if footholds.dim() == 4 else -1
        for t, shape, dt in ((x0, (B, 13), self.tdtype), (ref, (B, 10), self.tdtype), (feet0, (B, 4, 3), self.tdtype),
                             (footholds, (B, S, 4, 3), self.tdtype), (gait, (B, 4), torch.int32), (feet_id, (B, S, 4), torch.uint8),
                             (mu, (B,), self.tdtype)):
            if S < 1 or tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"operand mismatch: expected {shape} {dt} contiguous on {self.device}, got "
                                 f"{tuple(t.shape)} {t.dtype} on {t.device}")
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        out = self._outputs(B, want_X, st)
        self._seed(out, u_init, st)
        self.engine.solve_batch_gait_steps_ptr(B, S, x0.data_ptr(), ref.data_ptr(), feet0.data_ptr(), footholds.data_ptr(), gait.data_ptr(),
                                               feet_id.data_ptr(), mu.data_ptr(), out["u"].data_ptr(),
                                               out["X"].data_ptr() if want_X else None, out["status"].data_ptr(),
                                               out["iters"].data_ptr(), out["res"].data_ptr(), st.cuda_stream)
        return out

    def rollout(self, x, ref, plan_pos, plan_feet_id, plan_meta, tick, mu, T, log=True, stream=None):
        """Closed-loop roll-out of B robots over T control ticks on the device (include/mpcqp.h, mpcqp_rollout): `x`, `ref` and `tick`
        are advanced IN PLACE; returns the per-tick logs (the reference log's TRACKING PERFORMANCE actual / desired rows and stage-0
        FORCES, src/logger.py:22-46) and the per-robot count of solved ticks."""
        torch = _torch()
        B, S = int(plan_pos.shape[0]), int(plan_pos.shape[1])
        for t, shape, dt in ((x, (B, 13), self.tdtype), (ref, (B, 10), self.tdtype), (plan_pos, (B, S, 4, 3), self.tdtype),
                             (plan_feet_id, (B, S, 4), torch.uint8), (plan_meta, (B, 4), torch.int32), (tick, (B,), torch.int32),
                             (mu, (B,), self.tdtype)):
            if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"operand mismatch: expected {shape} {dt} contiguous on {self.device}, got {tuple(t.shape)} {t.dtype} on {t.device}")
        mk = lambda: torch.empty((B, T, 12), dtype=self.tdtype, device=self.device) if log else None
        actual, desired, forces = mk(), mk(), mk()
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        # (the advance kernel initialises `solved` at the first tick, on `st`; a zero fill here would run on torch's current stream)
        solved = (torch.empty if T > 0 else torch.zeros)(B, dtype=torch.int32, device=self.device)
        p = lambda t: t.data_ptr() if t is not None else None
        self.engine.rollout_ptr(B, T, S, x.data_ptr(), ref.data_ptr(), plan_pos.data_ptr(), plan_feet_id.data_ptr(), plan_meta.data_ptr(),
                                tick.data_ptr(), mu.data_ptr(), p(actual), p(desired), p(forces), solved.data_ptr(), st.cuda_stream)
        return {"actual": actual, "desired": desired, "forces": forces, "solved": solved}

    def torque_map(self, u, jac, stream=None):
        """tau[B,4,3] = J^T (-f) of the stage-0 forces (src/main.py:212-214); jac[B,4,3,3] world-frame leg Jacobians."""
        torch = _torch()
        B = int(u.shape[0])
        if tuple(jac.shape) != (B, 4, 3, 3) or jac.dtype != self.tdtype or u.dtype != self.tdtype or not jac.is_contiguous():
            raise ValueError("jac must be a contiguous [B,4,3,3] tensor of the engine's dtype")
        tau = torch.empty((B, 4, 3), dtype=self.tdtype, device=self.device)
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.engine.torque_map_ptr(B, u.data_ptr(), jac.data_ptr(), tau.data_ptr(), st.cuda_stream)
        return tau

    def leg_jacobians(self, q, rot=None, geometry=None, want_foot=True, stream=None):
        """World-frame 3x3 linear Jacobian block of each foot w.r.t. its leg's joints (what src/main.py:205-210 asks DART for), on the
        device: q [B,4,3] joint angles (HipX, HipY, Knee per leg), rot [B,3,3] torso orientation or None -> (jac [B,4,3,3], foot [B,4,3])."""
        torch = _torch()
        B = int(q.shape[0])
        if tuple(q.shape) != (B, 4, 3) or q.dtype != self.tdtype or not q.is_contiguous():
            raise ValueError("q must be a contiguous [B,4,3] tensor of the engine's dtype")
        if rot is not None and (tuple(rot.shape) != (B, 3, 3) or rot.dtype != self.tdtype or not rot.is_contiguous()):
            raise ValueError("rot must be a contiguous [B,3,3] tensor of the engine's dtype")
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(st):
            jac = torch.empty((B, 4, 3, 3), dtype=self.tdtype, device=self.device)
            foot = torch.empty((B, 4, 3), dtype=self.tdtype, device=self.device) if want_foot else None
        self.engine.leg_jacobians_ptr(B, q.data_ptr(), rot.data_ptr() if rot is not None else 0, jac.data_ptr(),
                                      foot.data_ptr() if foot is not None else 0, geometry, st.cuda_stream)
        return jac, foot

    def last_kernel_ms(self):
        return self.engine.last_kernel_ms()
