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


@pytest.mark.timeout(600)
def test_bench_gpus8_c4_preflight_through_torchrun():
    """The driver's own 8-GPU command line - ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 8 --config C4`` - on eight gloo ranks with the twin as the engine: the whole
    4 000-graph BA batch cut into eight unequal shards (strong scaling), ONE gather per step, and every field of the JSON
    line the driver parses.  Then the same batch on ONE rank: the same sets, the same total weight."""
    import json
    import subprocess
    import sys
    from distgcn_amd import datagen, parallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    common = ["--backend", "gloo", "--steps", "2", "--warmup", "1", "--layers", "3", "--cpu-seconds", "0", "--no-spmm-probe",
              "--no-e2e", "--config", "C4"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "_bench_gloo_driver.py"), "--gpus", "8"] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    out = lines[0]
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["job_graphs"] == 4000
    assert out["unit"] == "graphs/s" and out["higher_is_better"] is True and out["steps"] == 2 and out["warmup"] == 1
    assert out["value"] == pytest.approx(4000 * 2 / (out["ms_per_step"] * 2e-3), rel=1e-6)  # whole-job aggregate
    assert "C4" in out["config"]["workload"] and out["config"]["parallelism"] == "graph-sharded x8"
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 8 and d["ranks_counted_by_all_reduce"] == 8
    assert sorted(p[0] for p in d["rank_device_pairs"]) == list(range(8)) and len({tuple(p) for p in d["rank_device_pairs"]}) == 8
    g8 = d["gathered_last_step"]
    assert g8["graphs"] == 4000 and g8["own_slot_matches_own_result"] and g8["status_bits"] == 0
    # the shards are unequal (balanced on sum(nnz + N), not on the graph count)
    ranges = parallel.shard_ranges_from_sizes(__import__("bench").graph_sizes(__import__("bench").parse(["--config", "C4"])), 8)
    assert len({hi - lo for lo, hi in ranges}) > 1 and ranges[0][0] == 0 and ranges[-1][1] == 4000
    # one rank, same batch (strong scaling: the job does not change with the rank count)
    r1, l1 = _run_bench_driver(["--gpus", "1", "--force-dist", "--config", "C4"])
    assert r1.returncode == 0, r1.stderr[-2000:]
    g1 = l1[0]["dist"]["gathered_last_step"]
    assert l1[0]["n_gpus"] == 1 and l1[0]["dist"]["world_size"] == 1
    assert g1["graphs"] == 4000 and g1["set_members"] == g8["set_members"] and abs(g1["total_weight"] - g8["total_weight"]) < 1e-6


@pytest.mark.timeout(300)
def test_bench_gpus8_with_empty_shards():
    """Eight ranks, five graphs: three ranks own nothing, still join every collective; the line reports all five graphs."""
    from distgcn_amd import datagen
    from oracle import ctwin
    r, lines = _run_bench_driver(["--gpus", "8", "--graphs", "5", "--family", "ba", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = lines[0]
    assert out["n_gpus"] == 8 and out["dist"]["world_size"] == 8 and out["dist"]["ranks_counted_by_all_reduce"] == 8
    ref = ctwin.solve(datagen.ba_test2_batch(5), datagen.random_model(3, 32))
    g = out["dist"]["gathered_last_step"]
    assert g["graphs"] == 5 and g["set_members"] == int((ref["state"] == 1).sum())
    assert abs(g["total_weight"] - float(ref["totals"].sum())) < 1e-9
