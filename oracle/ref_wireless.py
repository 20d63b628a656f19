"""TEST INFRASTRUCTURE - CPU restatement of the slot loop of the reference's ``wireless_dqn_test.py:219-293``
for one instance ("parity unpinned": the script needs the absent ``graph_util`` module, Gurobi and TensorFlow).
``solve_fn(adj, wts) -> set`` plays ``local_greedy_search`` (``:238``) or ``dqn_agent.solve_mwis`` (``:281``)."""
import numpy as np
import scipy.sparse as sp


def simulate_one(adj_gK, arrival_pkts, link_rates, solve_fn, wt_sel="qr"):
    adj_gK = sp.csr_matrix(adj_gK)
    timeslots, nflows = arrival_pkts.shape
    n_ch = link_rates.shape[2]
    queue_mtx = np.zeros(shape=(timeslots, nflows))
    dep_pkts = np.zeros(shape=(timeslots, nflows))
    total = np.zeros(timeslots)
    for t in range(1, timeslots):
        queue_mtx[t, :] = queue_mtx[t - 1, :] + arrival_pkts[t, :]
        queue_mtx_algo = np.multiply(np.expand_dims(queue_mtx[t, :], axis=1), np.ones(shape=(nflows, n_ch)))
        if wt_sel == "qr":
            wts0 = queue_mtx_algo * link_rates[t, :, :]
        elif wt_sel == "q":
            wts0 = queue_mtx_algo
        elif wt_sel == "qor":
            wts0 = queue_mtx_algo / link_rates[t, :, :]
        elif wt_sel == "qrm":
            wts0 = np.minimum(queue_mtx_algo, link_rates[t, :, :])
        else:
            raise ValueError(wt_sel)
        wts1 = np.reshape(wts0, nflows * n_ch, order="F")
        mwis = solve_fn(adj_gK, wts1)
        total[t] = np.sum(wts1[sorted(mwis)]) if mwis else 0.0
        schedule_mv = np.array(sorted(mwis), dtype=np.int64)  # ascending: the highest channel of a flow wins
        link_rates_ts = np.reshape(link_rates[t, :, :], nflows * n_ch, order="F")
        capacity = np.zeros(shape=(nflows,))
        if schedule_mv.size:
            capacity[schedule_mv % nflows] = link_rates_ts[schedule_mv]
        dep_pkts[t, :] = np.minimum(queue_mtx_algo[:, 0], capacity)
        queue_mtx[t, :] = queue_mtx[t, :] - dep_pkts[t, :]
    return {"queue": queue_mtx, "depart": dep_pkts, "total_wt": total}


def simulate_seq_one(adj_list, arrival_pkts, link_rates, solve_fn):
    """``wireless_dqn_test_mc.py:292-354, 358-366`` for one instance: the channels of a slot are scheduled one after the
    other on their own conflict graphs (``adj_list[ic]``, nflows x nflows), links without weight left out, the queue
    estimate of the next channel reduced by what the scheduled links can send.  ``solve_fn(adj, wts) -> set`` plays
    ``local_greedy_search`` (:302), ``dqn_agent.solve_mwis`` (:322) or ``solve_mwis_rollout_wrap`` (:343)."""
    adj_list = [sp.csr_matrix(a) for a in adj_list]
    timeslots, nflows = arrival_pkts.shape
    n_ch = link_rates.shape[2]
    queue_mtx = np.zeros(shape=(timeslots, nflows))
    dep_pkts = np.zeros(shape=(timeslots, nflows))
    for t in range(1, timeslots):
        queue_mtx[t, :] = queue_mtx[t - 1, :] + arrival_pkts[t, :]
        queue_mtx_algo = np.multiply(np.expand_dims(queue_mtx[t, :], axis=1), np.ones(shape=(nflows, n_ch)))
        mwis = set()
        for ic in range(n_ch):
            wts_ic = queue_mtx_algo[:, ic] * link_rates[t, :, ic]
            wts_idx, = np.nonzero(wts_ic)
            adj_ic = adj_list[ic]
            adj_ii = adj_ic[wts_idx, :][:, wts_idx]
            mwis_c = solve_fn(adj_ii, wts_ic[wts_idx]) if wts_idx.size else set()
            mwis_ic = np.array(wts_idx[list(mwis_c)]) + ic * nflows
            mwis_ic = set(mwis_ic.flatten())
            mwis = mwis.union(mwis_ic)
            if ic + 1 < n_ch:
                mwis_ls = wts_idx[list(mwis_c)]
                depart_est = np.minimum(queue_mtx_algo[:, ic], link_rates[t, :, ic])
                queue_mtx_algo[:, ic + 1] = queue_mtx_algo[:, ic]
                queue_mtx_algo[mwis_ls, ic + 1] -= depart_est[mwis_ls]
        schedule_mv = np.array(list(mwis), dtype=np.int64)
        link_rates_ts = np.reshape(link_rates[t, :, :], nflows * n_ch, order="F")
        capacity = np.zeros(shape=(nflows,))
        if schedule_mv.size:
            capacity[schedule_mv % nflows] = link_rates_ts[schedule_mv]
        dep_pkts[t, :] = np.minimum(queue_mtx_algo[:, 0], capacity)
        queue_mtx[t, :] = queue_mtx[t, :] - dep_pkts[t, :]
    return {"queue": queue_mtx, "depart": dep_pkts}
