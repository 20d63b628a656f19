"""Multi-GPU: shard a batch of independent conflict graphs over ranks, one gather at the end.

The reference has no multi-process path; graphs are independent, so the batch shards by
graph with NO data-path collective.  Each rank solves its contiguous range of graphs; the only
exchange is ONE ``all_gather`` of a packed byte buffer per batch (membership bytes + float64 totals +
int32 rounds), padded to the largest shard.  Backend "nccl" is RCCL on ROCm (intra-node xGMI); "gloo"
runs the same code on CPU tensors (used by the tests).  C4-sized payloads are ~1 MB: latency-bound,
which is why it is a single call per batch and never per graph.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .batch import HostBatch


def shard_ranges(hb: HostBatch, world: int) -> List[Tuple[int, int]]:
    """Contiguous graph ranges [lo, hi) per rank, balanced on sum(nnz_g + N_g) (BA graphs vary 30x)."""
    sizes = np.diff(hb.graph_ptr).astype(np.int64)
    nnz = (hb.row_ptr[hb.graph_ptr[1:]] - hb.row_ptr[hb.graph_ptr[:-1]]).astype(np.int64)
    return shard_ranges_from_sizes(list(zip(sizes, nnz)), world)


def shard_ranges_from_sizes(sizes_nnz, world: int) -> List[Tuple[int, int]]:
    """The same cut from (vertices, entries) pairs alone: a rank that knows every graph's size can find its range
    without holding the other ranks' graphs."""
    B = len(sizes_nnz)
    if world <= 1:
        return [(0, B)]
    arr = np.asarray(sizes_nnz, dtype=np.int64).reshape(B, 2)
    cost = np.cumsum(arr[:, 0] + arr[:, 1])
    total = int(cost[-1]) if B else 0
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.searchsorted(cost, target, side="left")) if B else 0)
    cuts.append(B)
    for i in range(1, len(cuts)):  # monotone
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _pack(state: np.ndarray, totals: np.ndarray, rounds: np.ndarray, cap_nodes: int, cap_graphs: int) -> np.ndarray:
    buf = np.zeros(cap_graphs * 12 + cap_nodes, dtype=np.uint8)
    g = totals.size
    buf[:g * 8] = np.ascontiguousarray(totals, dtype=np.float64).view(np.uint8)
    buf[cap_graphs * 8:cap_graphs * 8 + g * 4] = np.ascontiguousarray(rounds, dtype=np.int32).view(np.uint8)
    buf[cap_graphs * 12:cap_graphs * 12 + state.size] = state
    return buf


def solve_sharded(hb: HostBatch, solve_fn: Callable[[HostBatch], Dict[str, np.ndarray]], group=None,
                  device=None) -> Dict[str, np.ndarray]:
    """Every rank calls this with the SAME full host batch.  ``solve_fn(sub_batch)`` returns numpy
    ``state`` (uint8 per vertex), ``totals`` (float64 per graph), ``rounds`` (int32 per graph) for
    the rank's shard.  Returns the assembled result for the whole batch on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ranges = shard_ranges(hb, world)
    lo, hi = ranges[rank]
    sub = hb.subset(lo, hi)
    res = solve_fn(sub) if hi > lo else {"state": np.zeros(0, np.uint8), "totals": np.zeros(0), "rounds": np.zeros(0, np.int32)}
    if world == 1:
        return {k: np.asarray(res[k]) for k in ("state", "totals", "rounds")}
    node_counts = [int(hb.graph_ptr[b] - hb.graph_ptr[a]) for a, b in ranges]
    cap_nodes = (max(node_counts) + 15) & ~15
    cap_graphs = max(b - a for a, b in ranges)
    mine = _pack(np.asarray(res["state"], np.uint8), np.asarray(res["totals"]), np.asarray(res["rounds"]),
                 cap_nodes, cap_graphs)
    backend = dist.get_backend(group)
    dev = device or ("cuda" if backend == "nccl" else "cpu")
    send = torch.from_numpy(mine).to(dev)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)  # the ONE collective of the batch
    allb = recv.cpu().numpy().reshape(world, -1)
    state = np.empty(hb.num_nodes, np.uint8)
    totals = np.empty(hb.num_graphs, np.float64)
    rounds = np.empty(hb.num_graphs, np.int32)
    for r, (a, b) in enumerate(ranges):
        g = b - a
        n0, n1 = int(hb.graph_ptr[a]), int(hb.graph_ptr[b])
        totals[a:b] = allb[r, :g * 8].view(np.float64)
        rounds[a:b] = allb[r, cap_graphs * 8:cap_graphs * 8 + g * 4].view(np.int32)
        state[n0:n1] = allb[r, cap_graphs * 12:cap_graphs * 12 + (n1 - n0)]
    return {"state": state, "totals": totals, "rounds": rounds}


def _shape_probe(hb: HostBatch) -> HostBatch:
    """A two-graph stand-in with the full batch's extreme shapes (its largest and its densest graph): what
    ``dgcn_solve_supported`` looks at (max_nodes, max_graph_edges), identical on every rank."""
    sizes = np.diff(hb.graph_ptr)
    nnz = hb.row_ptr[hb.graph_ptr[1:]] - hb.row_ptr[hb.graph_ptr[:-1]]
    return hb.select(sorted({int(np.argmax(sizes)), int(np.argmax(nnz))}))


def solve_sharded_device(engine, model, hb: HostBatch, predict: str = "mwis", group=None) -> Dict[str, np.ndarray]:
    """``solve_sharded`` for the HIP engine without the host round trip: every rank uploads its shard, runs the ONE
    fused launch into a packed result buffer laid out for the largest shard (``Engine.solve_buffers(cap_*)``), the ranks
    exchange those buffers with ONE ``all_gather_into_tensor`` on the device (RCCL over xGMI), and a single
    device-to-host copy brings the whole batch's membership / totals / rounds home.  Every rank calls this with the
    SAME full host batch; shapes outside the fused kernel raise (use ``solve_sharded`` with the layered path)."""
    import torch
    import torch.distributed as dist
    from .engine import packed_layout, solve_buffer_specs, _NP
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ranges = shard_ranges(hb, world)
    lo, hi = ranges[rank]
    sub = hb.subset(lo, hi)
    cap_nodes = max(max(int(hb.graph_ptr[b] - hb.graph_ptr[a]) for a, b in ranges), 1)
    cap_graphs = max(max(b - a for a, b in ranges), 1)
    db = engine.upload(sub)
    dm = model if hasattr(model, "c") else model.device_model(engine)
    out = engine.solve_buffers(db, want_scores=False, cap_nodes=cap_nodes, cap_graphs=cap_graphs)
    # Every rank decides on the shape of the WHOLE batch (largest graph, densest graph), which they all hold: a decision
    # taken per shard could differ between ranks, and a rank that raised here would leave the others waiting in the
    # collective below.
    if hb.num_nodes and engine.solve_path(engine.upload(_shape_probe(hb)), dm) == 0:
        raise _lib.DgcnError("solve_sharded_device: this model / batch shape is outside the fused kernel")
    if sub.num_nodes:
        engine.solve_fused(db, dm, predict=predict, want_scores=False, out=out)
    flat = out["flat"]
    if world > 1:
        recv = torch.empty(world * flat.numel(), dtype=torch.uint8, device=flat.device)
        dist.all_gather_into_tensor(recv, flat, group=group)  # the ONE collective of the batch
    else:
        recv = flat
    allb = recv.cpu().numpy().reshape(world, -1)
    lay = out["layout"]
    state = np.empty(hb.num_nodes, np.uint8)
    totals = np.empty(hb.num_graphs, np.float64)
    rounds = np.empty(hb.num_graphs, np.int32)
    bits = 0
    view = lambda r, name: allb[r, lay[name][0]:lay[name][0] + lay[name][1]].view(_NP[lay[name][2]])
    for r, (a, b) in enumerate(ranges):
        n0, n1 = int(hb.graph_ptr[a]), int(hb.graph_ptr[b])
        state[n0:n1] = view(r, "state")[:n1 - n0]
        totals[a:b] = view(r, "totals")[:b - a]
        rounds[a:b] = view(r, "rounds")[:b - a]
        bits |= int(view(r, "status")[0])
    engine.check_status_bits(bits)
    return {"state": state, "totals": totals, "rounds": rounds}


# ------------------------------------------------------------------------------------------------
# Rank launcher and census.  One process per GPU; the launcher NEVER touches the GPU itself (a process that
# has initialised HIP must not be replaced or forked into ranks), it only starts fresh children.

def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched_by_a_launcher() -> bool:
    """True when RANK / WORLD_SIZE come from torchrun (or from ``spawn_local_ranks``)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_local_ranks(nproc: int, script: str, argv: Sequence[str], port: Optional[int] = None,
                      extra_env: Optional[Dict[str, str]] = None, timeout: Optional[float] = None) -> int:
    """Start ``nproc`` fresh interpreters running ``script argv`` with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set (what ``python -m torch.distributed.run --nnodes=1`` would set),
    wait for them, and return the worst exit code.  If one rank fails the others are ended (by their exact
    pids) so that nobody waits in a collective forever."""
    port = port or free_port()
    procs = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    import time
    t0 = time.monotonic()
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in live:  # a failed rank leaves the others stuck in the rendezvous / a collective
                    q.terminate()
        if timeout is not None and time.monotonic() - t0 > timeout:
            for q in live:
                q.kill()
            worst = worst or 124
        time.sleep(0.05)
    return worst


def init_rank_group(backend: str = "nccl"):
    """Join the job the launcher described in the environment.  -> (rank, world, local_rank).
    backend "nccl" is RCCL on ROCm (one device per rank, fails loudly when LOCAL_RANK has no device);
    "gloo" runs the same collectives on CPU tensors."""
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        have = torch.cuda.device_count()
        if local >= have:
            raise RuntimeError("rank %d of %d needs GPU %d but this node shows %d device(s): --gpus must not exceed "
                               "the GPUs present (no rank shares or fakes a device)" % (rank, world, local, have))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, world, local


def census(device) -> Dict[str, object]:
    """What the collective library itself saw: the group's world size, an all-reduced count of the ranks that
    took part, and every rank's (rank, device index) pair from an all_gather."""
    import torch
    import torch.distributed as dist
    one = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    dev_index = torch.device(device).index if torch.device(device).type == "cuda" else -1
    mine = torch.tensor([dist.get_rank(), -1 if dev_index is None else dev_index], dtype=torch.int32, device=device)
    everyone = torch.empty(2 * dist.get_world_size(), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(everyone, mine)
    pairs = everyone.cpu().numpy().reshape(-1, 2)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_counted_by_all_reduce": int(one.item()),
            "rank_device_pairs": [[int(a), int(b)] for a, b in pairs]}
