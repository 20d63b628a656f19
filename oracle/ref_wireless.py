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
