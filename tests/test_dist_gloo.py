"""CPU, world_size 2, gloo: the graph-sharded path (shard -> local solve -> ONE all_gather).
The local solve is played by the CPU twin here (tests may use the oracle); on GPUs the same
``solve_sharded`` wraps the HIP engine (bench.py, tests/test_gpu_api.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, num_graphs=24):
    import torch.distributed as dist
    from distgcn_amd import datagen, parallel
    from oracle import ctwin
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hb = datagen.ba_test2_batch(num_graphs)
    layers = datagen.random_model(3, 32)
    calls = []

    def local(sub):
        calls.append(sub.num_graphs)
        r = ctwin.solve(sub, layers)
        return {"state": r["state"], "totals": r["totals"], "rounds": r["rounds"]}

    res = parallel.solve_sharded(hb, local)
    full = ctwin.solve(hb, layers)
    ok = (np.array_equal(res["state"], full["state"]) and np.array_equal(res["rounds"], full["rounds"])
          and np.array_equal(res["totals"], full["totals"]))
    lo, hi = parallel.shard_ranges(hb, world)[rank]
    ok = ok and calls == ([hi - lo] if hi > lo else [])  # a rank without graphs does not call the solver
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_solve_world2():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


@pytest.mark.timeout(300)
def test_sharded_solve_with_an_empty_shard():
    """Fewer graphs than ranks: one rank owns nothing, still joins the gather, everyone gets the full result."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out, 1), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}
