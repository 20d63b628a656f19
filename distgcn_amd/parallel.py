"""Multi-GPU: shard a batch of independent conflict graphs over ranks, one gather at the end.

The reference has no multi-process path; graphs are independent, so the batch shards by
graph with NO data-path collective.  Each rank solves its contiguous range of graphs; the only
exchange is ONE ``all_gather`` of a packed byte buffer per batch (membership bytes + float64 totals +
int32 rounds), padded to the largest shard.  Backend "nccl" is RCCL on ROCm (intra-node xGMI); "gloo"
runs the same code on CPU tensors (used by the tests).  C4-sized payloads are ~1 MB: latency-bound,
which is why it is a single call per batch and never per graph.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import numpy as np

from .batch import HostBatch


def shard_ranges(hb: HostBatch, world: int) -> List[Tuple[int, int]]:
    """Contiguous graph ranges [lo, hi) per rank, balanced on sum(nnz_g + N_g) (BA graphs vary 30x)."""
    B = hb.num_graphs
    if world <= 1:
        return [(0, B)]
    sizes = np.diff(hb.graph_ptr).astype(np.int64)
    nnz = (hb.row_ptr[hb.graph_ptr[1:]] - hb.row_ptr[hb.graph_ptr[:-1]]).astype(np.int64)
    cost = np.cumsum(sizes + nnz)
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
