"""mpcqp -- batched convex-MPC QP engine for quadruped locomotion on AMD MI355X (gfx950).

The package directory carries the upstream project's name and is therefore not a valid Python identifier;
``import mpcqp`` (the one-file loader at the repository root) imports it under the name ``mpcqp``.
Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C-ABI), the ctypes binding, the host-side
mirror of the reference's ``MPC.solve`` surface and its planner glue, and the synthetic batch generators.
"""
from . import _capi, dist, engine, footstep_planner, synth  # noqa: F401
from .engine import MPCBatch  # noqa: F401
from ._capi import (DISC_EULER, DISC_ZOH, DTYPE_F32, DTYPE_F64, FLAG_NATURAL_ORDER, FLAG_POLISH, FLAG_TILE_KERNEL, FLAG_NO_TIMING, FLAG_STAGE_KERNEL, FLAG_WARM_SHIFT, FLAG_WARM_START, PREC_F32,  # noqa: F401
                    PREC_F64, PREC_MIXED, Engine, Library, MpcQpConfig, MpcQpError, product_library)

__all__ = ["Engine", "Library", "MPCBatch", "MpcQpConfig", "MpcQpError", "product_library", "synth"]
