"""GPU: the C ABI called from several host threads at once, each on a stream of its own ("stateless, re-entrant; all
launches asynchronous on the supplied stream; safe to call concurrently on different streams").  The caller owns every buffer
- outputs, scratch workspace - so a thread brings its own (an Engine per thread = a workspace per thread); what the threads share is
the library: the option table, the per-device attribute caches of the launch helpers, the cluster launch's nonce, the
thread-local error text.  Every thread repeats ITS call many times while the others run theirs - the fused kernel (two per CU),
the BA mix (1 024-thread launch + dispatch order), the one-layer kernel, the any-size path (k_big), a small batch on the
several-workgroups-per-graph launch, a complete rollout search - and every repetition must give the bits the same call gave
when it ran alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve_job(engine_cls, hb, layers, reps):
    """-> (prepare(), run() -> list of result dicts) for a plain solve on a private engine / stream."""
    import torch
    from distgcn_amd.engine import DeviceModel
    eng = engine_cls("cuda:0")
    db = eng.upload(hb)
    dm = DeviceModel(layers, eng.device)
    stream = torch.cuda.Stream()

    def once():
        out = eng.solve_buffers(db, True)
        res = eng.solve_fused(db, dm, out=out, want_scores=True)
        return res

    def fetch(res):
        return {k: res[k].cpu().numpy().copy() for k in ("state", "scores", "rounds", "totals", "status")}

    def run(results):
        with torch.cuda.stream(stream):
            for i in range(reps):
                res = once()
                stream.synchronize()
                if i % 25 == 0 or i == reps - 1:  # (every repetition runs under the others' load; every 25th is fetched and compared)
                    results.append(fetch(res))

    return once, fetch, run


def _search_job(engine_cls, hb, layers, reps):
    import torch
    from distgcn_amd.engine import DeviceModel
    eng = engine_cls("cuda:0")
    db = eng.upload(hb)
    dm = DeviceModel(layers, eng.device)
    stream = torch.cuda.Stream()

    def once():
        state = torch.zeros(hb.num_nodes, dtype=torch.uint8, device=eng.device)
        res = eng.solve_residual(db, dm, state, greedy=eng.GREEDY_ROLLOUT, max_rounds=1, beam=16)
        return {"state": state, "status": res["status"], "steps": res["steps"]}

    def fetch(res):
        return {"state": res["state"].cpu().numpy().copy(), "status": res["status"].cpu().numpy().copy(), "steps": np.array([res["steps"]])}

    def run(results):
        with torch.cuda.stream(stream):
            for _ in range(reps):
                res = once()
                stream.synchronize()
                results.append(fetch(res))

    return once, fetch, run


def test_concurrent_callers_on_their_own_streams(engine):
    import torch
    from distgcn_amd import _lib, datagen
    from distgcn_amd.engine import Engine
    jobs = [
        ("k_fused, two per CU", _solve_job(Engine, datagen.er_batch(300, 200, 0.1, first_index=11), datagen.random_model(20, 32), 1500)),
        ("BA mix, dispatch order", _solve_job(Engine, datagen.ba_test2_batch(330, first_index=5), datagen.random_model(8, 32, bias=True), 1500)),
        ("k_shallow", _solve_job(Engine, datagen.er_batch(500, 100, 0.1, first_index=900), datagen.random_model(1, 32), 3000)),
        ("k_big", _solve_job(Engine, datagen.er_batch(24, 700, 0.02, first_index=77), datagen.random_model(6, 32), 1500)),
        ("cluster launch", _solve_job(Engine, datagen.er_batch(6, 200, 0.1, first_index=333), datagen.random_model(12, 32), 300)),  # (few: a launch that loses its peers under this load waits 0.3 s before it says so)
        ("rollout search", _search_job(Engine, datagen.er_batch(8, 220, 0.04, first_index=4242), datagen.random_model(4, 32), 60)),
    ]
    # what each call gives when nothing else runs
    alone = {}
    for name, (once, fetch, _) in jobs:
        res = once()
        torch.cuda.synchronize()
        alone[name] = fetch(res)
        assert int(alone[name]["status"].ravel()[0]) == 0, name
    results = {name: [] for name, _ in jobs}
    errors = []

    def worker(name, run):
        try:
            run(results[name])
        except Exception as e:  # noqa: BLE001 - reported below, with the job's name
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=worker, args=(name, job[2])) for name, job in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert _lib.load().dgcn_get_cluster() in (-1, 0)  # (a placement fault of the cluster launch under load would have switched it off: allowed, reported below)
    for name, _ in jobs:
        assert len(results[name]) > 0
        for i, got in enumerate(results[name]):
            # The several-workgroups-per-graph launch needs all its workgroups resident at once; with five other callers
            # holding CUs it may give up and say so (DGCN_FAULT_CLUSTER, include/dgcn.h: the step left nothing half-written
            # and the caller repeats it - Engine.solve_residual and the host solver do by themselves).  Reported, not wrong.
            if name == "cluster launch" and int(got["status"].ravel()[0]) & 16:
                continue
            for k, want in alone[name].items():
                assert np.array_equal(got[k].view(np.uint8), want.view(np.uint8)), (name, i, k)
