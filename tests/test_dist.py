"""N>1 path on CPU: world_size-2 gloo.  Each rank solves its shard (the CPU oracle stands in for the local GPU solve --
the sharding / gather logic under test is backend-agnostic) and the all-gathered stage-0 GRFs must equal a
single-process solve of the whole batch, in batch order, for even and ragged splits."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mpcqp
from conftest import ORACLE_SO
from mpcqp.dist import all_gather_stage0, shard_batch, shard_bounds


def test_shard_bounds_partition():
    for B in (0, 1, 7, 8, 4096, 65536, 65537):
        for G in (1, 2, 3, 8):
            spans = [shard_bounds(B, G, g) for g in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = mpcqp.synth.config3(B)
        mine = shard_batch(batch, world, rank)
        lib = mpcqp.Library(ORACLE_SO)
        eng = mpcqp.Engine(lib, lib.default_config(max_iter=4000))
        sol = eng.solve_batch_host(mine["x0"], mine["r"], mine["contact"], mine["xdes"], mine["mu"], want_X=False)
        g = all_gather_stage0(torch.from_numpy(sol["u"]), B)
        if rank == 0:
            q.put(g.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [16, 13])
def test_two_rank_gloo_allgather_matches_single_process(oracle_lib, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    batch = mpcqp.synth.config3(B)
    eng = mpcqp.Engine(oracle_lib, oracle_lib.default_config(max_iter=4000))
    ref = eng.solve_batch_host(batch["x0"], batch["r"], batch["contact"], batch["xdes"], batch["mu"], want_X=False)
    assert got.shape == (B, 12)
    assert np.array_equal(got, ref["u"][:, 0, :])
