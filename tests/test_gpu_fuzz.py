"""Randomised parity: random batch shapes x model shapes through the fused path (ordinary launch, cluster variant,
largest-first dispatch, global-memory entry values - whatever the library picks for the shape) against the CPU twin,
bit for bit.  DGCN_FUZZ_CASES sets the number of cases (default 16: a few seconds)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_match_the_twin(engine):
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from oracle import ctwin
    cases = int(os.environ.get("DGCN_FUZZ_CASES", "16"))
    rng = np.random.default_rng(2026)
    for case in range(cases):
        layers_n = int(rng.choice([1, 2, 3, 8, 12, 20]))
        hidden = int(rng.choice([16, 32]))
        layers = datagen.random_model(layers_n, hidden, bias=bool(rng.integers(2)), seed=100 + case)
        kind = int(rng.integers(4))
        if kind == 0:  # a few graphs of one size: the cluster variant's ground
            n = int(rng.choice([60, 113, 150, 200, 257, 300, 400, 500, 512]))
            hb = datagen.er_batch(int(rng.integers(1, 12)), n, min(0.5, 12.0 / n), first_index=5000 + 50 * case)
        elif kind == 1:  # many graphs of one size
            n = int(rng.integers(1, 120))
            hb = datagen.er_batch(int(rng.integers(50, 400)), n, min(0.9, 6.0 / max(n, 2)), first_index=9000 + 500 * case)
        elif kind == 2:  # the BA mix: largest-first dispatch above 256 graphs
            hb = datagen.ba_test2_batch(int(rng.integers(20, 330)), first_index=13 * case)
        else:  # ragged, with empty and one-vertex graphs
            ps, cs, ws = [], [], []
            for _ in range(int(rng.integers(2, 40))):
                n = int(rng.choice([0, 1, 2, 17, 64, 129, 200, 333]))
                if n == 0:
                    ps.append(np.zeros(1, np.int32)); cs.append(np.zeros(0, np.int32)); ws.append(np.zeros(0))
                    continue
                g = datagen.er_batch(1, n, min(0.9, 8.0 / max(n, 2)), first_index=int(rng.integers(1 << 20)))
                ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
            hb = HostBatch.from_csr_lists(ps, cs, ws)
        if hb.num_nodes == 0:
            continue
        model = DeviceModel(layers, engine.device)
        db = engine.upload(hb)
        if not engine.solve_supported(db, model):
            continue
        out = engine.solve_buffers(db, True)
        engine.solve_fused(db, model, out=out, want_scores=True)
        got = engine.fetch_solve_buffers(out, hb.num_nodes, hb.num_graphs)
        assert got["status"] == 0, (case, got["status"])
        want = ctwin.solve(hb, layers)
        tag = (case, layers_n, hidden, kind, hb.num_graphs, hb.max_nodes)
        assert np.array_equal(got["scores"].ravel().view(np.uint32), np.asarray(want["scores"], np.float32).ravel().view(np.uint32)), tag
        assert np.array_equal(got["state"], want["state"]) and np.array_equal(got["rounds"], want["rounds"]), tag
        assert np.allclose(got["totals"], want["totals"], rtol=1e-12, atol=0), tag


def test_random_calls_through_the_host_solver_match_the_twin(engine):
    """The per-call path (one-slot HostSolver: packed batch read in place, completion word, cluster variant where it fits):
    random small batches, call after call on the same object, against the twin."""
    from distgcn_amd import datagen
    from distgcn_amd.batch import HostBatch
    from distgcn_amd.engine import DeviceModel
    from distgcn_amd.serving import HostSolver
    from oracle import ctwin
    calls = int(os.environ.get("DGCN_FUZZ_CASES", "16")) * 4
    rng = np.random.default_rng(77)
    solvers = {}
    for call in range(calls):
        layers_n = int(rng.choice([1, 3, 12, 20]))
        if layers_n not in solvers:
            layers = datagen.random_model(layers_n, 32, bias=True, seed=layers_n)
            solvers[layers_n] = (layers, HostSolver(engine, DeviceModel(layers, engine.device), depth=1, want_scores=True))
        layers, hs = solvers[layers_n]
        ps, cs, ws = [], [], []
        for _ in range(int(rng.integers(1, 9))):
            n = int(rng.choice([0, 1, 30, 128, 200, 200, 200, 300, 500]))
            if n == 0:
                ps.append(np.zeros(1, np.int32)); cs.append(np.zeros(0, np.int32)); ws.append(np.zeros(0))
                continue
            g = datagen.er_batch(1, n, min(0.9, 10.0 / max(n, 2)), first_index=int(rng.integers(1 << 20)))
            ps.append(g.row_ptr.astype(np.int32)); cs.append(g.col_idx.astype(np.int32)); ws.append(g.weights)
        hb = HostBatch.from_csr_lists(ps, cs, ws)
        got = hs.solve(ps, cs, ws)
        if hb.num_nodes == 0:
            assert got["state"].size == 0
            continue
        want = ctwin.solve(hb, layers)
        assert np.array_equal(got["state"], want["state"]) and np.array_equal(got["rounds"], want["rounds"]), (call, layers_n)
        assert np.array_equal(got["scores"].view(np.uint32), np.asarray(want["scores"], np.float32).ravel().view(np.uint32)), (call, layers_n)
        assert np.allclose(got["totals"], want["totals"], rtol=1e-12, atol=0), (call, layers_n)
    for _, hs in solvers.values():
        hs.close()
