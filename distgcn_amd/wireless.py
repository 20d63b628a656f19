"""The wireless link-scheduling simulation of the reference's ``wireless_dqn_test.py`` with
the scheduler on the GPU and every (conflict graph, load) instance advanced in lockstep.

The reference walks ``timeslots`` slots per instance (``wireless_dqn_test.py:219-293``): queues grow by the
slot's arrivals, per-link weights are built from queue lengths and link rates (``wt_sel``: ``qr`` / ``q`` /
``qor`` / ``qrm``, ``:222-233``), a solver picks an independent set of the conflict graph, and the scheduled
links drain ``min(queue, rate)`` packets (``:285-293``).  One solver call per slot per instance there; here one
launch per slot for ALL instances: queues, weights, schedule and departures stay on the device.

Solvers (``algo``): ``"DGCN-LGS-it"`` = ``solve_mwis_dit`` (``:251-254``) and ``"DGCN-RS"`` = ``solve_mwis_rollout_wrap``
(``:256-260``) of ``mwis_gdpg_call.DQNAgent``, every instance (connected component) advanced by the same launches;
``"Greedy"`` = ``local_greedy_search`` on the raw weights (``:236-238``); ``"DGCN-LGS"`` =
``mwis_dqn_call.DQNAgent.solve_mwis`` (``:271-283``), i.e. zero-weight links are dropped from the conflict graph
before the GCN runs (``mwis_dqn_call.py:202-207``) - a mask of the residual-graph kernel
(``dgcn_solve_residual_batch``) instead of a re-sliced matrix.  The Gurobi denominators (``mlp_gurobi``) and the
Poisson-disk topology generator (``graph_util``, absent from the reference) are out of scope: callers bring
their single-channel conflict graphs; ``total_wt`` per slot is returned so that any denominator can be applied.
``"CGCN-CGS"`` = ``solve_mwis_cgs_train(train=False)`` (``:262-266``; ``mwis_gdpg_call.py:778-839``).
The multi-channel expansion IS here: ``multichannel_conflict_simulate`` (per-channel copies of a conflict graph with
edges dropped at random, ``wireless_rollout_test_flood.py:70-95``) and ``multichannel_conflict_graph`` (the joint graph
on ``n_ch * nflows`` vertices: the channels' graphs on the diagonal blocks plus a clique over every flow's copies -
one radio per node, ``:98-133``), both checked against the reference's own functions (tests/golden/multichannel.npz).
Multi-channel graphs (vertex = channel * nflows + flow, ``order='F'`` at ``:234``): when several channels
schedule the same flow the reference keeps whichever its Python set iterates last; here the highest channel.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .api_common import as_csr, get_engine
from .batch import HostBatch


def make_traffic(nflows: int, timeslots: int, load: float, rate_lo: int = 0, rate_hi: int = 100, n_ch: int = 1,
                 seed: int = 0) -> Dict[str, np.ndarray]:
    """Arrivals and link rates of one instance, call for call what ``wireless_dqn_test.py:181-199`` draws after
    ``np.random.seed(seed)`` (a ``RandomState(seed)`` yields the same stream).
    -> {"arrival_pkts": [timeslots, nflows], "link_rates": int [timeslots, nflows, n_ch]}"""
    rs = np.random.RandomState(seed)
    arrival_rate = 0.5 * (rate_lo + rate_hi) * load
    interarrivals = rs.exponential(1.0 / arrival_rate, (nflows, int(2 * timeslots * arrival_rate)))
    arrival_time = np.cumsum(interarrivals, axis=1)
    acc_pkts = np.zeros(shape=(nflows, timeslots))
    for t in range(0, timeslots):
        acc_pkts[:, t] = np.count_nonzero(arrival_time < t, axis=1)
    arrival_pkts = np.diff(acc_pkts, prepend=0).transpose()
    link_rates = rs.normal(0.5 * (rate_lo + rate_hi), 0.25 * (rate_hi - rate_lo), size=[timeslots, nflows, n_ch])
    link_rates = link_rates.astype(int)
    link_rates[link_rates < rate_lo] = rate_lo
    link_rates[link_rates > rate_hi] = rate_hi
    return {"arrival_pkts": arrival_pkts, "link_rates": link_rates}


def slot_weights(queue, rates, wt_sel: str):
    """``wireless_dqn_test.py:222-233`` for one slot; works on NumPy arrays and torch tensors alike."""
    if wt_sel == "qr":
        return queue * rates
    if wt_sel == "q":
        return queue * 1.0
    if wt_sel == "qor":
        return queue / rates
    if wt_sel == "qrm":
        return np.minimum(queue, rates) if isinstance(queue, np.ndarray) else queue.minimum(rates)
    raise ValueError("wt_sel must be one of qr, q, qor, qrm (the reference's random weights are unseeded per slot)")


def multichannel_conflict_simulate(adj_i, k: int = 3, p: float = 0.8, rng=np.random) -> List:
    """``k`` per-channel copies of the conflict graph ``adj_i``, each edge of each copy kept with probability ``p``
    (``wireless_rollout_test_flood.py:83-95``; what ``wireless_dqn_test_mc.py:160`` calls
    ``multichannel_conflict_simulate``).  The reference draws ``np.random.rand()`` once per edge, channel by channel,
    edges in the order (u ascending, v < u ascending), and drops the edge when the draw exceeds ``p``: with
    ``rng=np.random`` and the same seed the same graphs come back.  -> list of symmetric CSR matrices (values 1.0)."""
    import scipy.sparse as sp
    a = sp.csr_matrix(adj_i)
    n = a.shape[0]
    low = sp.tril(a, k=-1).tocsr()  # row u holds its neighbours v < u, ascending after sort_indices
    low.sort_indices()
    us = np.repeat(np.arange(n), np.diff(low.indptr))
    vs = low.indices
    keep_val = low.data != 0
    us, vs = us[keep_val], vs[keep_val]
    out = []
    for _ in range(int(k)):
        # (one vectorised draw = the same stream as one rng.rand() per edge: the legacy generator fills arrays in order)
        draws = np.asarray(rng.rand(int(us.size))) if us.size else np.zeros(0)
        keep = ~(draws > p)
        u, v = us[keep], vs[keep]
        m = sp.csr_matrix((np.ones(2 * u.size), (np.concatenate([u, v]), np.concatenate([v, u]))), shape=(n, n))
        m.sort_indices()
        out.append(m)
    return out


def multichannel_conflict_graph(graphs: Sequence):
    """``(adj_list, adj_gK)`` of ``wireless_rollout_test_flood.py:98-133`` (used at ``wireless_dqn_test_mc.py:161``):
    ``adj_list[k]`` = channel k's conflict graph (for the channel-by-channel schedulers, ``simulate_seq``); ``adj_gK`` =
    the joint conflict graph on ``K * nn`` vertices, vertex ``k * nn + v`` = link v on channel k (``order='F'`` of the
    weight matrix, ``:240``): block k of the diagonal is channel k's graph, and the K copies of every link form a
    clique (a node has one radio: a link is active on at most one channel, ``:116-124``).  ``graphs``: K adjacency
    matrices (SciPy / dense / anything ``as_csr`` takes) of one size.  Values are 1.0; CSR with sorted rows."""
    import scipy.sparse as sp
    mats = [sp.csr_matrix(as_csr(g)) for g in graphs]
    if not mats:
        raise ValueError("no channel graphs")
    nn = mats[0].shape[0]
    if any(m.shape != (nn, nn) for m in mats):
        raise AssertionError("channel graphs differ in size")  # the reference's assert(len(set(no_nodes)) == 1)
    K = len(mats)
    adj_list = []
    for m in mats:
        c = sp.csr_matrix((np.ones(m.nnz), m.indices.copy(), m.indptr.copy()), shape=m.shape)
        c.sort_indices()
        adj_list.append(c)
    eye = sp.identity(nn, format="csr")
    blocks = [[adj_list[k] if k == j else eye for j in range(K)] for k in range(K)]
    adj_gK = sp.bmat(blocks, format="csr")
    adj_gK.data[:] = 1.0
    adj_gK.sort_indices()
    return adj_list, adj_gK


def simulate(adjs: Sequence, traffics: Sequence[Dict[str, np.ndarray]], algo: str = "DGCN-LGS", agent=None,
             wt_sel: str = "qr") -> List[Dict[str, np.ndarray]]:
    """Run all instances (``adjs[i]``: conflict graph on ``nflows_i * n_ch`` vertices, ``traffics[i]`` from
    ``make_traffic``) for their common number of slots.  -> per instance ``{"queue": [T, nflows], "depart":
    [T, nflows], "total_wt": [T], "scheduled": [T]}`` (queue lengths AFTER the slot's departures, as the
    reference's ``queue_mtx_dict``)."""
    import torch
    if algo not in ("Greedy", "DGCN-LGS", "DGCN-LGS-it", "DGCN-RS", "CGCN-CGS"):
        raise ValueError("algo must be 'Greedy', 'DGCN-LGS', 'DGCN-LGS-it', 'DGCN-RS' or 'CGCN-CGS'")
    if algo != "Greedy" and agent is None:
        raise ValueError("%s needs an agent (mwis_dqn_call.DQNAgent / mwis_gdpg_call.DQNAgent)" % algo)
    eng = get_engine()
    t = torch
    csrs = [as_csr(a) for a in adjs]
    if algo == "DGCN-RS":
        # solve_mwis_rollout_wrap (wireless_dqn_test.py:256-260) searches every connected component on its own
        # (mwis_gdpg_call.py:386-411).  The conflict graphs do not change from slot to slot, so the components are cut
        # once and become the graphs of the device batch; vertices are renumbered component by component.
        import scipy.sparse.csgraph as csg
        comp_csrs, order = [], []
        for c in csrs:
            ncomp, labels = csg.connected_components(c, directed=False)
            ids = np.argsort(labels, kind="stable")  # ascending vertex order inside every component
            order.append(ids)
            bounds = np.concatenate([[0], np.cumsum(np.bincount(labels, minlength=ncomp))])
            for k in range(ncomp):
                comp = ids[bounds[k]:bounds[k + 1]]
                comp_csrs.append(as_csr(c[comp][:, comp]))
    else:
        comp_csrs, order = csrs, [np.arange(c.shape[0]) for c in csrs]
    T = int(traffics[0]["arrival_pkts"].shape[0])
    chans = [int(tr["link_rates"].shape[2]) for tr in traffics]  # channels may differ between instances
    flows = [int(tr["arrival_pkts"].shape[1]) for tr in traffics]
    for c, f, k, tr in zip(csrs, flows, chans, traffics):
        if c.shape[0] != f * k or tr["arrival_pkts"].shape[0] != T or tr["link_rates"].shape[:2] != (T, f):
            raise ValueError("conflict graph / traffic shapes disagree")
    hb = HostBatch.from_csr_lists([c.indptr for c in comp_csrs], [c.indices for c in comp_csrs],
                                  [np.zeros(c.shape[0]) for c in comp_csrs])
    db = eng.upload(hb)
    foff = np.concatenate([[0], np.cumsum(flows)])
    # batch position -> (global flow, channel, instance); vertex v of instance i is channel v // F_i, flow v % F_i
    vflow = np.concatenate([foff[i] + order[i] % f for i, (f, k) in enumerate(zip(flows, chans))])
    vch = np.concatenate([order[i] // f for i, (f, k) in enumerate(zip(flows, chans))])
    vinst = np.concatenate([np.full(f * k, i) for i, (f, k) in enumerate(zip(flows, chans))])
    arr = t.from_numpy(np.concatenate([tr["arrival_pkts"] for tr in traffics], axis=1).astype(np.float64)).to(eng.device)
    rates = t.from_numpy(np.concatenate([tr["link_rates"].transpose(0, 2, 1).reshape(T, -1)[:, order[i]]
                                         for i, tr in enumerate(traffics)], axis=1).astype(np.float64)).to(eng.device)  # [T, batch positions]
    vflow_d = t.from_numpy(vflow).to(eng.device)
    vinst_d = t.from_numpy(vinst).to(eng.device)
    # per channel: batch positions of its vertices and their flows (a flow has at most one vertex per channel)
    ch_pos = [t.from_numpy(np.flatnonzero(vch == c)).to(eng.device) for c in range(max(chans))]
    ch_flow = [vflow_d[p] for p in ch_pos]
    F, I = int(foff[-1]), len(flows)
    q = t.zeros(F, dtype=t.float64, device=eng.device)
    queue_out = t.zeros((T, F), dtype=t.float64, device=eng.device)
    dep_out = t.zeros((T, F), dtype=t.float64, device=eng.device)
    tot_out = t.zeros((T, I), dtype=t.float64, device=eng.device)
    cnt_out = t.zeros((T, I), dtype=t.float64, device=eng.device)
    dm = agent.model.device_model(eng) if algo != "Greedy" else None
    if dm is not None and hb.num_nodes and eng.solve_path(db, dm) == 0:  # (1: fused kernels, 2: the any-size device path)
        raise NotImplementedError("conflict graphs / model outside the device solvers (graphs beyond 9 600 vertices, max_degree > 1)")
    out = eng.solve_buffers(db, False) if dm is not None else None
    state = t.zeros(max(hb.num_nodes, 1), dtype=t.uint8, device=eng.device)
    zero_w = t.zeros(hb.num_nodes, dtype=t.float64, device=eng.device)
    for ts in range(1, T):
        q += arr[ts]
        w = slot_weights(q[vflow_d], rates[ts], wt_sel)
        db.weights.copy_(w)
        if algo == "Greedy":
            res = eng.lgs(db, prio=db.weights, want_totals=False)
            st = res["state"]
            status = res["status"]
        elif algo == "DGCN-LGS":
            state.copy_((w <= 0).to(t.uint8) * 2)  # zero-weight links leave the graph (mwis_dqn_call.py:202-207)
            res = eng.solve_residual(db, dm, state, predict=agent.flags.predict, greedy=eng.GREEDY_ROUNDS, max_rounds=0,
                                     max_steps=1, out=out)
            st, status = state[:hb.num_nodes], res["status"]
        else:
            # 'DGCN-LGS-it' = solve_mwis_dit (wireless_dqn_test.py:251-254): the GCN is re-run on the residual graph
            # before every greedy round; 'DGCN-RS' = solve_mwis_rollout_wrap (:256-260): top-16 candidates, greedy
            # completions, per connected component; 'CGCN-CGS' = solve_mwis_cgs_train(train=False) (:262-266), the
            # centralised argmax step.  All: one launch per solver step for every instance / component.
            state.zero_()
            res = eng.solve_residual(db, dm, state, predict=agent.flags.predict,
                                     greedy={"DGCN-LGS-it": eng.GREEDY_ROUNDS, "CGCN-CGS": eng.GREEDY_CENTRAL}.get(algo, eng.GREEDY_ROLLOUT),
                                     max_rounds=1, beam=16, weight_features=agent.flags.predict != "mwis", out=out)
            st, status = state[:hb.num_nodes], res["status"]
        # (dense selects instead of boolean-mask indexing: nothing here waits for the device)
        sel = st == 1
        cap = t.zeros(F, dtype=t.float64, device=eng.device)
        for pos, fl in zip(ch_pos, ch_flow):  # ascending channel: the highest scheduled channel of a flow sets its capacity
            cap[fl] = t.where(sel[pos], rates[ts][pos], cap[fl])
        dep = t.minimum(q, cap)
        q -= dep
        queue_out[ts] = q
        dep_out[ts] = dep
        tot_out[ts].index_add_(0, vinst_d, t.where(sel, w, zero_w))
        cnt_out[ts].index_add_(0, vinst_d, sel.to(t.float64))
    eng.check_status(status)
    queue_h, dep_h, tot_h, cnt_h = (x.cpu().numpy() for x in (queue_out, dep_out, tot_out, cnt_out))
    return [{"queue": queue_h[:, foff[i]:foff[i + 1]], "depart": dep_h[:, foff[i]:foff[i + 1]], "total_wt": tot_h[:, i],
             "scheduled": cnt_h[:, i].astype(np.int64)} for i in range(I)]


def simulate_seq(adj_lists: Sequence[Sequence], traffics: Sequence[Dict[str, np.ndarray]], algo: str = "DGCN-LGS-Seq", agent=None
                 ) -> List[Dict[str, np.ndarray]]:
    """The channel-by-channel schedulers of ``wireless_dqn_test_mc.py:292-354`` (``--opt`` 5, 6, 7): in every slot the
    channels are served one after the other - channel ``ic`` is scheduled on ITS conflict graph ``adj_lists[i][ic]``
    (``nflows x nflows``) with the weights ``queue estimate x rate``, links without weight left out, and the queues the
    next channel sees are reduced by what the links just scheduled can send.

        "LGS-Seq"       ``local_greedy_search``                 (:292-311)
        "DGCN-LGS-Seq"  ``dqn_agent.solve_mwis``               (:312-332)
        "CGCN-RS-Seq"   ``dqn_agent.solve_mwis_rollout_wrap``  (:333-353)

    The reference loops over instances, slots and channels with one solver call each; here the calls of all instances for
    the same (slot, channel) are one batch (they are independent).  Weights are ``'qr'`` (the script asserts it).  The
    bookkeeping after the channel loop is the reference's expression on the same Python sets (:358-366): when a link ends
    up scheduled on several channels its capacity is the rate of whichever channel comes last in ``list(mwis)``.
    -> per instance ``{"queue", "depart": [T, nflows], "scheduled": [T]}``."""
    from . import heuristics
    if algo not in ("LGS-Seq", "DGCN-LGS-Seq", "CGCN-RS-Seq"):
        raise ValueError("algo must be 'LGS-Seq', 'DGCN-LGS-Seq' or 'CGCN-RS-Seq'")
    if algo != "LGS-Seq" and agent is None:
        raise ValueError("%s needs an agent" % algo)
    if algo == "CGCN-RS-Seq" and not hasattr(agent, "solve_mwis_rollout_wrap"):
        raise ValueError("CGCN-RS-Seq needs mwis_gdpg_call.DQNAgent (solve_mwis_rollout_wrap)")
    I = len(adj_lists)
    T = int(traffics[0]["arrival_pkts"].shape[0])
    flows = [int(tr["arrival_pkts"].shape[1]) for tr in traffics]
    chans = [int(tr["link_rates"].shape[2]) for tr in traffics]
    lists = [[as_csr(a) for a in al] for al in adj_lists]
    for al, f, k, tr in zip(lists, flows, chans, traffics):
        if len(al) != k or any(a.shape[0] != f for a in al) or tr["arrival_pkts"].shape[0] != T or tr["link_rates"].shape[:2] != (T, f):
            raise ValueError("conflict graphs / traffic shapes disagree")
    queue = [np.zeros((T, f)) for f in flows]
    dep = [np.zeros((T, f)) for f in flows]
    sched = np.zeros((T, I), dtype=np.int64)

    def solve_batch(jobs):
        """jobs: (adjacency, weights) with all weights non-zero -> list of sets"""
        if not jobs:
            return []
        if algo == "LGS-Seq":
            return [heuristics.local_greedy_search(a, w)[0] for a, w in jobs]
        if algo == "DGCN-LGS-Seq":
            if hasattr(agent, "solve_mwis_batch"):
                return [r[0] for r in agent.solve_mwis_batch([a for a, _ in jobs], [w for _, w in jobs])]
            return [agent.solve_mwis(a, w)[0] for a, w in jobs]
        return [agent.solve_mwis_rollout_wrap(a, w)[0] for a, w in jobs]

    for t in range(1, T):
        qa, mwis = [], []
        for i in range(I):
            queue[i][t] = queue[i][t - 1] + traffics[i]["arrival_pkts"][t]
            qa.append(np.multiply(np.expand_dims(queue[i][t], axis=1), np.ones((flows[i], chans[i]))))
            mwis.append(set())
        for ic in range(max(chans)):
            jobs, meta = [], []
            for i in range(I):
                if ic >= chans[i]:
                    continue
                wts_ic = qa[i][:, ic] * traffics[i]["link_rates"][t, :, ic]
                wts_idx, = np.nonzero(wts_ic)
                if wts_idx.size:
                    a = lists[i][ic]
                    jobs.append((as_csr(a[wts_idx, :][:, wts_idx]), wts_ic[wts_idx]))
                meta.append((i, wts_idx, bool(wts_idx.size)))
            sols = iter(solve_batch(jobs))
            for i, wts_idx, has in meta:
                mwis_c = next(sols) if has else set()
                mwis_ic = np.array(wts_idx[list(mwis_c)]) + ic * flows[i]
                mwis[i] = mwis[i].union(set(mwis_ic.flatten()))
                if ic + 1 < chans[i]:
                    mwis_ls = wts_idx[list(mwis_c)]
                    depart_est = np.minimum(qa[i][:, ic], traffics[i]["link_rates"][t, :, ic])
                    qa[i][:, ic + 1] = qa[i][:, ic]
                    qa[i][mwis_ls, ic + 1] -= depart_est[mwis_ls]
        for i in range(I):
            schedule_mv = np.array(list(mwis[i]), dtype=np.int64)
            link_rates_ts = np.reshape(traffics[i]["link_rates"][t, :, :], flows[i] * chans[i], order="F")
            capacity = np.zeros(flows[i])
            if schedule_mv.size:
                capacity[schedule_mv % flows[i]] = link_rates_ts[schedule_mv]
            dep[i][t] = np.minimum(qa[i][:, 0], capacity)
            queue[i][t] = queue[i][t] - dep[i][t]
            sched[t, i] = schedule_mv.size
    return [{"queue": queue[i], "depart": dep[i], "scheduled": sched[:, i]} for i in range(I)]


def summarize(result: Dict[str, np.ndarray]) -> Dict[str, float]:
    """The per-run metrics the reference writes to its CSV (``wireless_dqn_test.py:303-336``)."""
    qm = result["queue"]
    return {"avg_queue_len": float(np.mean(np.mean(qm, axis=1))), "50p_queue_len": float(np.mean(np.median(qm, axis=1))),
            "95p_queue_len": float(np.percentile(qm, 95)), "5p_queue_len": float(np.percentile(qm, 5))}
