import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
ORACLE_SO = os.path.join(REPO, "oracle", "libmpcqp_oracle.so")
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU checker (oracle/).  Built on demand with gcc; test infrastructure only."""
    import mpcqp
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "libmpcqp_oracle.so"])
    return mpcqp.Library(ORACLE_SO)


@pytest.fixture(scope="session")
def oracle_solve(oracle_lib):
    import mpcqp

    def solve(batch, N=10, delta=0.03, **kw):
        kw.setdefault("eps_abs", 1e-10); kw.setdefault("eps_rel", 1e-10)
        kw.setdefault("max_iter", 100000); kw.setdefault("polish_max", 30)
        eng = mpcqp.Engine(oracle_lib, oracle_lib.default_config(N=N, delta=delta, **kw))
        return eng.solve_batch_host(batch["x0"], batch["r"], batch["contact"], batch["xdes"], batch["mu"])
    return solve


@pytest.fixture(scope="session")
def golden():
    return {k: np.load(os.path.join(GOLDEN, k + ".npz")) for k in ("ref_log", "planner_golden", "qp_inputs", "qp_optima")}


def rel_err(u, ur):
    """SURVEY.md section 8(c) parity definition: |u - u_ref|_inf / |u_ref|_inf per QP (floor 1 N for all-zero QPs)."""
    u = np.asarray(u, float).reshape(len(u), -1); ur = np.asarray(ur, float).reshape(len(ur), -1)
    return np.abs(u - ur).max(axis=1) / np.maximum(np.abs(ur).max(axis=1), 1.0)
