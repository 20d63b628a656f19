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


def _run_bench_driver(extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tests", "_bench_gloo_driver.py"), "--backend", "gloo", "--steps", "2",
           "--warmup", "1", "--layers", "3", "--cpu-seconds", "0", "--no-spmm-probe", "--no-e2e"] + extra
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    return r, [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(300)
def test_bench_gpus2_launches_two_ranks_weak():
    """``bench.py --gpus 2`` with no launcher around it starts two ranks itself; the JSON line reports what the
    collective library saw and what the end-of-step gather (membership + totals + rounds) delivered."""
    from distgcn_amd import datagen
    from oracle import ctwin
    r, lines = _run_bench_driver(["--gpus", "2", "--graphs", "6", "--nodes", "60", "--p", "0.1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1  # rank 0 only
    out = lines[0]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["job_graphs"] == 12
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["ranks_counted_by_all_reduce"] == 2
    assert sorted(p[0] for p in d["rank_device_pairs"]) == [0, 1]
    g = d["gathered_last_step"]
    layers = datagen.random_model(3, 32)
    want_members, want_weight = 0, 0.0
    for rank in range(2):
        ref = ctwin.solve(datagen.er_batch(6, 60, 0.1, first_index=rank * 6), layers)
        want_members += int((ref["state"] == 1).sum())
        want_weight += float(ref["totals"].sum())
    assert g["graphs"] == 12 and g["set_members"] == want_members and g["own_slot_matches_own_result"]
    assert abs(g["total_weight"] - want_weight) < 1e-9


@pytest.mark.timeout(300)
def test_bench_strong_scaling_shards_one_batch():
    """--scaling strong: ONE BA batch sharded by parallel.shard_ranges (unequal shards, one buffer layout)."""
    from distgcn_amd import datagen
    from oracle import ctwin
    r, lines = _run_bench_driver(["--gpus", "2", "--graphs", "25", "--family", "ba", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = lines[0]
    assert out["scaling"] == "strong" and out["config"]["job_graphs"] == 25 and out["n_gpus"] == 2
    ref = ctwin.solve(datagen.ba_test2_batch(25), datagen.random_model(3, 32))
    g = out["dist"]["gathered_last_step"]
    assert g["graphs"] == 25 and g["set_members"] == int((ref["state"] == 1).sum())
    assert abs(g["total_weight"] - float(ref["totals"].sum())) < 1e-9


@pytest.mark.timeout(300)
def test_bench_rejects_a_world_that_differs_from_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.timeout(300)
def test_nccl_rank_without_a_device_fails_loudly():
    """On a box with fewer GPUs than ranks a rank must refuse to start (never print n_gpus: 1 for --gpus 2)."""
    from distgcn_amd import parallel
    os.environ.update(RANK="1", WORLD_SIZE="2", LOCAL_RANK="1", MASTER_PORT=str(_free_port()))
    try:
        with pytest.raises(RuntimeError, match="needs GPU 1"):
            parallel.init_rank_group("nccl")
    finally:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
            os.environ.pop(k, None)
