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


_BENCH2 = {"proc": None, "result": None}
_BENCH1N = {"proc": None, "result": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-rank bench rehearsal (tests/test_gpu_configs45.py): the two-rank `bench.py` job is started HERE, as a fresh child
    # process, before this process makes its first HIP call (torch.cuda.device_count() does not initialise the GPU).
    expr = config.getoption("-m", default="") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch
            have = torch.cuda.device_count() > 0
        except Exception:  # noqa: BLE001
            have = False
        if have:
            env = dict(os.environ, MPCQP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
            # weak mode (4096 QPs per rank, one draw per rank), then strong mode (one batch of 6001 QPs in contiguous, ragged shards):
            # both in ONE child shell, one after the other (at most two ranks of the rehearsal on the card at a time)
            run2 = (f"{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port %d "
                    f"{os.path.join(REPO, 'bench.py')} --gpus 2 --steps 3 --warmup 1 --allgather --no-cpu-baseline --no-breakdown %s")
            _BENCH2["proc"] = subprocess.Popen(["bash", "-c", (run2 % (29517, "")) + " && " + (run2 % (29519, "--global-batch 6001"))],
                                               cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
            # ... and the RCCL ("nccl") branch of the same script with a single rank: process group on the device, barrier,
            # all-reduce and all-gather of device tensors -- every collective call the 8-GPU run makes.
            env1 = dict(os.environ, MPCQP_BENCH_BACKEND="nccl", MPCQP_BENCH_DIST1="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
            _BENCH1N["proc"] = subprocess.Popen(
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", "29518", os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--allgather",
                 "--no-cpu-baseline", "--no-breakdown"], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env1)


def bench2_result():
    """(returncode, stdout, stderr) of the two-rank bench child started in pytest_configure."""
    if _BENCH2["result"] is None:
        p = _BENCH2["proc"]
        if p is None:
            pytest.skip("two-rank bench child was not started (no GPU at configure time)")
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        _BENCH2["result"] = (p.returncode, out, err)
    return _BENCH2["result"]


def bench1_nccl_result():
    """(returncode, stdout, stderr) of the one-rank RCCL bench child started in pytest_configure."""
    if _BENCH1N["result"] is None:
        p = _BENCH1N["proc"]
        if p is None:
            pytest.skip("one-rank RCCL bench child was not started (no GPU at configure time)")
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        _BENCH1N["result"] = (p.returncode, out, err)
    return _BENCH1N["result"]


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
