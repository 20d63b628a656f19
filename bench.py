#!/usr/bin/env python3
"""Benchmark of the hot path: GCN forward + local-greedy MWIS on a batch of conflict graphs.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the whole hot path over one batch already resident in HBM:
support construction (L = I - D^-1/2 A D^-1/2) -> 20-layer c32 GCN forward -> priority product ->
local greedy search.  Workload at every N: BASELINE.json configs[2] (C3), 500 ER graphs
G(200, 0.1) per GPU (weak scaling: each rank owns its own 500 graphs; no data-path collective; the
one collective, the end-of-step all_gather of every rank's packed result buffer - membership bytes +
float64 totals + int32 rounds + status word - is inside the timed step).
``--scaling strong --family ba --graphs 4000`` is BASELINE.json configs[3] (C4): ONE 4 000-graph BA batch
sharded over the ranks by ``distgcn_amd.parallel.shard_ranges``.

Ranks: one process per GPU.  Under ``python -m torch.distributed.run`` the ranks already exist (RANK /
WORLD_SIZE in the environment).  Run plainly with ``--gpus N`` (N > 1) this process only LAUNCHES N fresh
rank processes - before anything here touches the GPU - and relays their exit code; a rank whose
LOCAL_RANK has no device fails loudly.  The JSON line carries what RCCL itself saw (``dist``).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MATRIX_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32 MFMA dense peak
LDS_PEAK_TBS = 256 * 128 * 2.4e9 / 1e12  # 256 CUs x 128 B/clk x 2.4 GHz = 78.6 TB/s


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 1500: ~0.3 s of timed region on C3; C5: 20 complete searches)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 20; C5: 2)")
    ap.add_argument("--graphs", type=int, default=None, help="graphs per GPU (weak scaling) or in the whole job (strong); default 500 (C5: 64, or any number given here: 256 puts one search on every CU)")
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--family", choices=["er", "ba", "mc"], default="er",
                    help="er: G(nodes, p) (C2 / C3); ba: the BA test2 mix of SURVEY 8d (C4), ignores --nodes/--p")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--layers", type=int, default=20)
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--mode", choices=["layered", "fused", "auto"], default="auto")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo: CPU ranks (tests only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-cpu-pool", action="store_true", help="skip the all-cores CPU figure (forked workers)")
    ap.add_argument("--no-gather", action="store_true", help="skip the end-of-step result gather at N>1")
    ap.add_argument("--no-spmm-probe", action="store_true", help="skip the stand-alone SpMM kernel measurement")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host (ingest + solve + fetch) measurement")
    ap.add_argument("--two-streams", action="store_true",
                    help="also time the step issued alternately on two HIP streams (side figure; its overlapping launches would "
                         "blur a kernel trace of the run, so it is not part of the default command)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even at N=1")
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C4-share", "C5", "ER500", "MC900", "MC900-l1", "MC1500", "MC900-rollout"], default=None,
                    help="BASELINE.json configuration shortcuts: C2 = 500 ER N=100 l=1; C3 = 500 ER N=200 l=20 (the default line); "
                         "C4 = the 4 000-graph BA batch over the ranks (--layers as given, default 20); C4-share = one GPU's 500 "
                         "graphs of it; C5 = GCN-guided rollout (b=16) on 64 ER N=500 graphs; beyond the fused kernel's 512 vertices / LDS "
                         "budget (the any-size path, csrc/general.hip + big.hip): ER500 = 256 ER N=500 p=0.1 (25 000 entries per graph), "
                         "MC900 = 256 joint 3-channel conflict graphs of 300 flows (900 vertices, wireless_rollout_test_flood.py:98-133); "
                         "MC900-l1 = the same graphs with the one-layer model the reference's multi-channel launcher runs "
                         "(bash/twc_major_wireless_mc_test.sh:3 --num_layer=1 --num_channels=3; csrc/wide.hip); "
                         "MC1500 = 256 joint graphs of 3 x 500 flows (1 500 vertices), l=20")
    ap.add_argument("--parity-seconds", type=float, default=25.0,
                    help="budget of the full-size parity report against the oracle restatement (0 = skip)")
    ap.add_argument("--beam", type=int, default=16, help="C5: rollout candidates per step")
    ap.add_argument("--any-size-path", action="store_true",
                    help="send the batch down dgcn_solve_batch's any-size path (csrc/general.hip + big.hip) even where the fused kernel "
                         "takes it: what a shape outside the fused kernel pays, measured on the benchmark's own batch")
    args = ap.parse_args(argv)
    if args.config == "C2":
        args.family, args.graphs, args.nodes, args.p, args.layers = "er", 500, 100, 0.1, 1
    elif args.config == "C3":
        args.family, args.graphs, args.nodes, args.p, args.layers = "er", 500, 200, 0.1, 20
    elif args.config == "C4":
        args.family, args.graphs, args.scaling = "ba", 4000, "strong"
    elif args.config == "C4-share":
        args.family, args.graphs = "ba", 500
    elif args.config == "C5":
        args.family, args.graphs, args.nodes, args.p, args.layers = "er", (args.graphs or 64), 500, 0.02, 20
    elif args.config == "ER500":
        args.family, args.graphs, args.nodes, args.p, args.layers = "er", (args.graphs or 256), 500, 0.1, 20
    elif args.config == "MC900":
        args.family, args.graphs, args.nodes, args.p, args.layers = "mc", (args.graphs or 256), 900, 0.03, 20
    elif args.config == "MC900-l1":
        args.family, args.graphs, args.nodes, args.p, args.layers = "mc", (args.graphs or 256), 900, 0.03, 1
    elif args.config == "MC1500":
        args.family, args.graphs, args.nodes, args.p, args.layers = "mc", (args.graphs or 256), 1500, 0.03, 20
    elif args.config == "MC900-rollout":  # C5's search on the multi-channel joint graphs: every step one launch of the any-size path
        args.family, args.graphs, args.nodes, args.p = "mc", (args.graphs or 64), 900, 0.03
    if args.config in ("ER500", "MC900", "MC1500") and args.steps is None:
        args.steps = 400
    if args.graphs is None:
        args.graphs = 500
    if args.steps is None:
        args.steps = 20 if args.config in ("C5", "MC900-rollout") else 1500
    if args.warmup is None:
        args.warmup = 2 if args.config in ("C5", "MC900-rollout") else 20
    return args


def load_layers(args):
    """Trained weights of the shipped IS4SAT l20/c32 checkpoint when the fixture copy is present
    (tests/golden/models.npz), random Glorot weights of the same architecture otherwise."""
    from distgcn_amd import datagen
    from distgcn_amd.gcn.models import layers_from_params
    path = os.path.join(ROOT, "tests", "golden", "models.npz")
    prefix = "DQNBA" if args.family == "ba" else "IS4SAT"
    name = "result_%s_deep_ld1_c%d_l%d_cheb1_diver1_mwis_dqn" % (prefix, args.hidden, args.layers)
    if os.path.isfile(path):
        z = np.load(path)
        pre = name + "|"
        params = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
        if params:
            return layers_from_params(params), "trained weights (fixture of the reference's %s)" % name
    return datagen.random_model(args.layers, args.hidden), "random-init weights"


def spmm_algorithmic_bytes(hb, layers, with_y0):
    """SURVEY 8d formula per SpMM launch, one entry per layer of the forward:
    sum_g[nnz_g*(4+4) + (N_g+1)*4] + 2*4*C*sum N_g  (+ 4*C*sum N_g for the fused '+ Z0' read)."""
    n, nnz_l = hb.num_nodes, hb.num_edges + hb.num_nodes
    csr = nnz_l * 8 + (n + hb.num_graphs) * 4
    per_launch = []
    for lyr in layers:
        c = lyr["weights"][0].shape[1]
        per_launch.append(csr + (3 if with_y0 else 2) * 4 * c * n)
    return per_launch


def workload_name(args):
    """BASELINE.json's config this run corresponds to (C3 is the bench line; the others are reference runs)."""
    if args.family == "ba":
        if args.scaling == "strong" and args.graphs == 4000:
            return "C4"
        return "C4 (one GPU's share)" if args.graphs == 500 else "custom"
    if args.family == "mc":
        if args.nodes == 900:
            return ("MC900-l1 (joint 3-channel conflict graphs, the multi-channel launcher's one-layer model)" if args.layers == 1
                    else "MC900 (joint 3-channel conflict graphs, beyond the fused kernel)")
        return "MC1500 (joint 3-channel conflict graphs of 500 flows)" if args.nodes == 1500 else "custom"
    key = (args.graphs, args.nodes, args.p, args.layers, args.hidden)
    return {(500, 200, 0.1, 20, 32): "C3", (500, 100, 0.1, 1, 32): "C2",
            (256, 500, 0.1, 20, 32): "ER500 (25 000 entries per graph: beyond the fused kernel)"}.get(key, "custom")


def cpu_baseline(hb, layers, budget_s):
    """The oracle restatement of the reference path (SciPy supports -> NumPy forward -> Python-set
    local greedy, oracle/ref_numpy.py), one process / one core like the reference, on a bounded
    sample of the same batch.  Reported beside the GPU number; never part of it."""
    from oracle import ref_numpy as orc
    done = 0
    slices = hb.graph_slices()
    t0 = time.perf_counter()
    while True:  # walk the batch (again, if it is exhausted) until the time budget is used
        g = done % hb.num_graphs
        n0, n1 = slices[g]
        orc.solve_mwis_gdpg(layers, hb.scipy_graph(g), hb.weights[n0:n1], feature_size=1)
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "graphs/s", "cores": 1, "kind": "port",
            "sample": "%d graph solves over rank 0's %d-graph batch, %.1f s, python oracle/ref_numpy.solve_mwis_gdpg"
                      % (done, hb.num_graphs, dt),
            "host_cpus": os.cpu_count()}


def cpu_baseline_all_cores(args, seconds):
    """SURVEY 8d's generous figure: the same restatement in one forked worker per host core
    (oracle/cpu_pool.py, a child process that never touches the GPU).  None if it cannot run."""
    import subprocess
    try:
        procs = len(os.sched_getaffinity(0))
    except AttributeError:
        procs = os.cpu_count() or 1
    procs = max(1, min(procs, 128))
    models = os.path.join(ROOT, "tests", "golden", "models.npz")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_pool.py"), str(min(args.graphs, 64)), str(args.nodes),
           str(args.p), str(args.layers), str(args.hidden), str(seconds), str(procs), models]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=seconds * 4 + 60)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": "graphs/s", "cores": d["cores"],
                "effective_cores": round(d.get("effective_cores", 0.0), 1),
                "sample": "%d graph solves in %d forked workers, %.1f s each; effective_cores = CPU-seconds per "
                          "second the workers were given" % (d["solves"], d["cores"], seconds)}
    except Exception as e:  # a reported extra; never fails the bench
        return {"value": None, "error": repr(e)[:200]}


def build_host_batch(args, rank, world):
    """-> (this rank's HostBatch, graphs in the whole job).  weak: every rank generates its own ``--graphs``
    graphs (disjoint seeds); strong: every rank generates the SAME ``--graphs``-graph batch and keeps the
    contiguous range ``parallel.shard_ranges`` gives it (balanced on sum(nnz + N): BA graphs vary 30x)."""
    from distgcn_amd import datagen, parallel
    def gen(count, first):
        if args.family == "ba":
            return datagen.ba_test2_batch(count, first_index=first)
        if args.family == "mc":
            return multichannel_batch(count, args.nodes // 3, args.p, first_index=first)
        return datagen.er_batch(count, args.nodes, args.p, first_index=first)
    if args.scaling == "weak":
        return gen(args.graphs, rank * args.graphs), world * args.graphs
    if world == 1:
        return gen(args.graphs, 0), args.graphs
    # strong: the shard boundaries need every graph's size (sum(nnz + N)), not the graphs: sizes are cheap to know
    # without generating (ER: drawn; BA: N and m fix the edge count), so each rank generates only its own range
    lo, hi = parallel.shard_ranges_from_sizes(graph_sizes(args), world)[rank]
    return gen(hi - lo, lo), args.graphs


def multichannel_batch(count, nflows, p, first_index=0, n_ch=3, keep=0.8):
    """(moved to distgcn_amd/datagen.py: the parity configurations use the same generator)"""
    from distgcn_amd import datagen
    return datagen.multichannel_batch(count, nflows, p, first_index=first_index, n_ch=n_ch, keep=keep)


def graph_sizes(args):
    """(vertices, directed entries) of every graph of the job's batch without building it.  BA(n, m) of
    datagen.ba_graph has exactly m * (n - m) undirected edges (star seed on m + 1 vertices, then m per new vertex);
    ER graphs are drawn (their edge count is random), which costs what generating them costs - only used for BA."""
    from distgcn_amd import datagen
    if args.family == "ba":
        cells = [(n, d) for n in datagen.TEST2_SIZES for d in datagen.TEST2_DEGREES]
        out = []
        for g in range(args.graphs):
            n, m = cells[g % len(cells)]
            m = max(1, min(int(m), n - 1))
            out.append((n, 2 * (m + m * (n - m - 1))))
        return out
    hb = datagen.er_batch(args.graphs, args.nodes, args.p)
    return [(int(n1 - n0), int(hb.row_ptr[n1] - hb.row_ptr[n0])) for n0, n1 in hb.graph_slices()]


class GpuWorkload:
    """This rank's share of the job on its MI355X: batch resident in HBM, one fused launch per step, results
    in a ring of packed buffers (``flat``: membership + totals + rounds + status) that the gather sends as is."""

    def __init__(self, args, rank, world, local):
        import torch
        from distgcn_amd.engine import Engine, DeviceModel, MODE_FUSED, MODE_LAYERED
        self.torch, self.args = torch, args
        self.dev = "cuda:%d" % local
        torch.cuda.set_device(local)
        self.hb, self.job_graphs = build_host_batch(args, rank, world)
        self.layers, self.weights_note = load_layers(args)
        self.eng = Engine(self.dev)
        if getattr(args, "any_size_path", False):
            self.eng.lib.dgcn_set_general(1)
        self.db = self.eng.upload(self.hb)
        self.model = DeviceModel(self.layers, self.dev)
        mode_name = args.mode
        if mode_name == "auto":
            mode_name = os.environ.get("DGCN_BENCH_MODE", "fused")
        self.mode_name = mode_name
        self.mode = MODE_FUSED if mode_name == "fused" else MODE_LAYERED
        self.ring = None
        self.counter = 0
        self.settle_s = 0.0

    def settle(self, seconds=0.3):
        """Untimed, before the warm-up steps: keep the GPU busy with the step itself until its clocks have ramped.
        The driver's default window (5 warm-up + 20 timed steps = 6 ms) otherwise measures the power-state ramp
        (248 us per launch) instead of the kernel (227 us once the clock is up); nothing inside the timed region changes."""
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(50):
                self.step()
            self.sync()
        self.settle_s = time.perf_counter() - t0

    def collective_device(self):
        return self.dev

    def make_buffers(self, cap_nodes, cap_graphs):
        from distgcn_amd.engine import MODE_FUSED
        if self.mode == MODE_FUSED:
            self.ring = [self.eng.solve_buffers(self.db, want_scores=False, cap_nodes=cap_nodes, cap_graphs=cap_graphs)
                         for _ in range(4)]
        elif (cap_nodes, cap_graphs) != (max(self.hb.num_nodes, 1), max(self.hb.num_graphs, 1)):
            raise RuntimeError("the layer-by-layer mode gathers only equally sized shards")

    def step(self):
        """-> result dict; ["flat"] is the packed byte tensor the gather sends."""
        if self.ring is not None:
            # steady-state serving loop: output buffers are re-used (4-deep ring: a buffer is rewritten
            # only after the gather that reads it has been waited for)
            self.counter += 1
            out = self.ring[self.counter % 4]
            res = self.eng.solve_fused(self.db, self.model, want_scores=False, out=out)
            res["flat"], res["layout"] = out["flat"], out["layout"]
            return res
        self.db.lap = None  # supports are part of the path: rebuild them every step
        return self.eng.solve(self.db, self.model, mode=self.mode)

    def sync(self):
        self.torch.cuda.synchronize()

    def check(self, res):
        self.eng.check_status(res["status"])

    def timing(self, on):
        self.eng.timing(on)

    def kernel_times(self):
        fam_ms = {}
        for fam in ("supports", "transform", "spmm", "layer", "lgs", "fused_forward", "fused_solve", "big_forward", "big_solve",
                    "general_prepare", "general_greedy", "wide_solve", "wide_residual"):
            ms, n = self.eng.timing_read(fam)
            if n:
                fam_ms[fam] = (ms, n)
        return fam_ms


class RolloutWorkload:
    """BASELINE config 5: GCN-guided rollout search (mwis_gdpg_call.py:596-659, b candidates per step) on N = 500 conflict
    graphs, every graph of the batch advanced by the same launches (dgcn_solve_residual_batch, greedy_mode 2).  One step
    of the benchmark = the COMPLETE search of the batch: ~ (set size) launches, each a 20-layer forward on the residual
    graphs + b greedy completions + the pick, all on the device; the host reads a progress word per group of launches."""

    def __init__(self, args, rank, world, local):
        import torch
        from distgcn_amd import datagen
        from distgcn_amd.engine import Engine
        from distgcn_amd.mwis_gdpg_call import DQNAgent
        from distgcn_amd.runtime_config import FLAGS
        self.torch, self.args = torch, args
        self.dev = "cuda:%d" % local
        torch.cuda.set_device(local)
        if args.family == "mc":  # joint K x nflows conflict graphs (wireless_dqn_test_mc.py:161, 244-289): the any-size path
            self.hb = datagen.multichannel_batch(args.graphs, args.nodes // 3, args.p, first_index=rank * args.graphs)
        else:
            self.hb = datagen.er_batch(args.graphs, args.nodes, args.p, first_index=rank * args.graphs)
        self.job_graphs = world * args.graphs
        self.layers, self.weights_note = load_layers(args)
        self.eng = Engine(self.dev)
        self.db = self.eng.upload(self.hb)
        flags = FLAGS.copy(feature_size=1, hidden1=args.hidden, num_layer=args.layers, diver_num=1, max_degree=1, predict="mwis")
        self.agent = DQNAgent(flags, seed=3)
        for lyr_dst, lyr_src in zip(self.agent.model.layers, self.layers):  # GCN2_DQN (activation on the last layer) with the trained weights
            lyr_dst["weights"] = lyr_src["weights"]
        for lyr in self.agent.model.layers:
            lyr["bias"] = None
        self.layers = self.agent.model.layers
        from distgcn_amd.engine import DeviceModel
        self.model = DeviceModel(self.layers, self.dev)
        self.ring = None
        self.mode_name = "residual-graph rollout (device-resident)"
        self.settle_s = 0.0
        self.launches = 0
        self.out = self.eng.solve_buffers(self.db, False)
        self.state = torch.zeros(self.hb.num_nodes, dtype=torch.uint8, device=self.dev)

    def collective_device(self):
        return self.dev

    def make_buffers(self, cap_nodes, cap_graphs):
        pass

    def step(self):
        self.state.zero_()
        res = self.eng.solve_residual(self.db, self.model, self.state, greedy=self.eng.GREEDY_ROLLOUT, max_rounds=1,
                                      beam=self.args.beam, out=self.out)
        self.launches = res["steps"]
        return {"status": res["status"], "state": res["state"], "flat": res["state"], "layout": None}

    def sync(self):
        self.torch.cuda.synchronize()

    def check(self, res):
        self.eng.check_status(res["status"])

    def timing(self, on):
        self.eng.timing(on)

    def kernel_times(self):
        out = {}
        for fam in ("fused_residual", "big_residual", "wide_residual", "tail_finish"):
            ms, n = self.eng.timing_read(fam)
            if n:
                out[fam] = (ms, n)
        return out

    def residual_bytes(self):
        """SURVEY 8d's layer-by-layer bytes of every launch of one search, from the residual graphs' actual sizes:
        the search replayed one launch at a time (untimed), the state read back after each."""
        hb, eng, t = self.hb, self.eng, self.torch
        rows = np.repeat(np.arange(hb.num_nodes), np.diff(hb.row_ptr))
        gid = np.repeat(np.arange(hb.num_graphs), np.diff(hb.graph_ptr))
        state = t.zeros(hb.num_nodes, dtype=t.uint8, device=self.dev)
        per_step, launches = [], 0
        while True:
            alive = state.cpu().numpy() == 0
            if not alive.any():
                break
            n_res = int(alive.sum())
            nnz_res = int((alive[rows] & alive[hb.col_idx]).sum())
            graphs_res = int(np.unique(gid[alive]).size)
            csr = (nnz_res + n_res) * 8 + (n_res + graphs_res) * 4
            per_step.append(sum(csr + 2 * 4 * lyr["weights"][0].shape[1] * n_res for lyr in self.layers))
            eng.solve_residual(self.db, self.model, state, greedy=eng.GREEDY_ROLLOUT, max_rounds=1, beam=self.args.beam,
                               out=self.out, max_steps=1)
            launches += 1
            if launches > hb.max_nodes + 2:
                break
        return per_step, launches


def c5_cpu_baseline(wl, budget_s):
    """The oracle's rollout (oracle/ref_numpy.solve_mwis_rollout: NumPy forward on the re-sliced residual graph + b
    Python greedy completions per step, like the reference) on the first graph(s) of the batch, one core."""
    from oracle import ref_numpy as orc
    hb = wl.hb
    fn = orc._default_scores_fn(wl.layers, 1, 1, "mwis")
    done, t0 = 0, time.perf_counter()
    while done < hb.num_graphs and (done == 0 or time.perf_counter() - t0 < budget_s):
        n0, n1 = hb.graph_slices()[done]
        orc.solve_mwis_rollout(fn, hb.scipy_graph(done), hb.weights[n0:n1], b=wl.args.beam, predict="mwis")
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "graphs/s", "cores": 1, "kind": "port",
            "sample": "%d complete rollout searches (b=%d) over the first graphs of rank 0's batch, %.1f s, python "
                      "oracle/ref_numpy.solve_mwis_rollout" % (done, wl.args.beam, dt), "host_cpus": os.cpu_count()}


def decode_flat(flat_host, layout, num_nodes, num_graphs):
    """One rank's packed result buffer (engine.Engine._packed layout) -> numpy views."""
    NP = {"uint8": np.uint8, "int32": np.int32, "int64": np.int64, "float32": np.float32, "float64": np.float64}
    out = {}
    for name, (o, nb, dt) in layout.items():
        out[name] = flat_host[o:o + nb].view(NP[dt])
    return {"state": out["state"][:num_nodes], "totals": out["totals"][:num_graphs], "rounds": out["rounds"][:num_graphs],
            "status": int(out["status"][0]) if "status" in out else 0}


def roofline_objects(args, wl, fam_ms):
    """roofline of the dominant kernel from the HIP events recorded around its launches in the timed region
    (on the launch stream), against SURVEY 8d's algorithmic bytes."""
    hb, layers = wl.hb, wl.layers
    roofline = None
    per_layer_bytes = spmm_algorithmic_bytes(hb, layers, with_y0=False)  # SURVEY 8d, layer by layer
    traffic_db = {}
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.isfile(tpath):
        traffic_db = json.load(open(tpath))
    traffic_note = "PMC FETCH_SIZE/WRITE_SIZE passes of an earlier run of this command, read from profiles/hbm_traffic.json (not measured in this run)"
    if fam_ms:
        dom = max(fam_ms, key=lambda k: fam_ms[k][0])
        ms, n = fam_ms[dom]
        avg_s = ms / n * 1e-3
        shape = ("ba%d" % args.graphs) if args.family == "ba" else "%dx%d" % (args.graphs, args.nodes)
        if args.family == "mc":
            shape = "mc%dx%d" % (args.graphs, args.nodes)
        tkey = ("spmm|%s|C%d" % (shape, args.hidden)) if dom == "spmm" else "%s|%s|l%d" % (dom, shape, args.layers)
        traffic = traffic_db.get(tkey, {}).get("hbm_bytes_per_launch")
        if dom in ("spmm", "layer"):
            per = spmm_algorithmic_bytes(hb, layers, with_y0=True)
            avg_bytes = sum(per) / len(per)
            ach = avg_bytes / avg_s / 1e9
            roofline = {"kernel": ("k_spmm_lds (all %d launches of a step)" % len(per)) if dom == "spmm" else
                                  "k_layer32 (aggregation + next layer's transform, %d launches of a step)" % int(round(n / args.steps)),
                        "bound": "hbm",
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_note, "avg_launch_us": avg_s * 1e6,
                        "algorithmic_bytes_per_launch": avg_bytes,
                        "formula": "SURVEY 8d B_spmm + 4*C*N for the fused '+Z0' read, averaged over the layers"}
        elif dom in ("fused_forward", "fused_solve", "big_forward", "big_solve", "wide_solve"):
            # one launch = every layer of every graph: SURVEY 8d counts the forward layer by layer
            # (1.658 MB per ER N=200 l=20 graph); the kernel keeps the graph in LDS, so its real HBM
            # traffic ('traffic', from PMC counters) is far BELOW this figure, not above it.
            algo = float(sum(per_layer_bytes))
            ach = algo / avg_s / 1e9
            flops = 0.0
            n_nodes, nnz_l = hb.num_nodes, hb.num_edges + hb.num_nodes
            for lyr in layers:
                cin, cout = lyr["weights"][0].shape
                flops += 2.0 * n_nodes * cin * 2 * cout + 2.0 * nnz_l * cout
            # the resources that actually pace these kernels are on the chip (the graph never leaves it): the LDS array (one
            # 128-byte row of Z1 per entry and hidden layer) and the fp32 MFMA pipe - printed beside the SURVEY 8d figure
            lds_bytes = float(sum(128.0 * nnz_l for lyr in layers[1:-1] if lyr["weights"][0].shape[1] == 32))
            kname = ("k_big2 (big_solve: whole path of graphs of 977 .. 1 920 vertices - Z1 in LDS a feature half at a time, two walks per aggregation - supports, every layer, priority, greedy search, one launch per step)"
                     if dom == "big_solve" and hb.max_nodes > 976 else
                     "k_big (big_solve: whole path of graphs beyond the fused kernel's LDS budget - supports, every layer, priority, greedy search - one launch per step)"
                     if dom == "big_solve" else
                     "k_big (whole forward of graphs beyond the fused kernel's LDS budget, one launch per step; supports and greedy search in launches of their own)"
                     if dom == "big_forward" else
                     "k_wide1 (one-layer model on graphs of any size: whole path, one launch per step)" if dom == "wide_solve" else
                     "k_shallow (one-layer model: whole path, one launch per step)" if len(layers) == 1 else
                     "k_fused (%s: whole path, one launch per step)" % dom)
            # what paces the kernel (the SURVEY 8d figure below is an EQUIVALENT bandwidth: layer-by-layer bytes / time): the
            # in-LDS deep-stack kernels sit on the LDS array + the fp32 MFMA pipe; the one-layer kernels are chains of dependent
            # round trips (latency); only the stand-alone SpMM lines are HBM-bound
            # (`bound` names the roofline `achieved` / `peak` / `frac` are computed against - the contract's HBM roofline, SURVEY 8d;
            # `paced_by` names what the kernel actually waits for, with its own peak and fraction in on_chip_view / fp32_matrix_view)
            roofline = {"kernel": kname, "bound": "hbm", "paced_by": "latency" if len(layers) == 1 else "lds+mfma",
                        "frac_is": "equivalent bandwidth: SURVEY 8d algorithmic bytes / launch time / HBM peak",
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_note, "avg_launch_us": avg_s * 1e6,
                        "algorithmic_bytes_per_launch": algo,
                        "formula": "SURVEY 8d: sum over layers of B_spmm (CSR + Z read + Y write), %d graphs" % hb.num_graphs,
                        "fp32_matrix_view": {"flops_per_launch": flops, "achieved_tflops": flops / avg_s / 1e12,
                                             "peak_tflops": F32_MATRIX_PEAK_TF,
                                             "frac": flops / avg_s / 1e12 / F32_MATRIX_PEAK_TF},
                        "on_chip_view": {"binding": "LDS array + fp32 MFMA pipe (HBM sees the input once: 'traffic')",
                                         "lds_gather_bytes_per_launch": lds_bytes,
                                         "lds_gather_tbs": lds_bytes / avg_s / 1e12, "lds_peak_tbs": LDS_PEAK_TBS,
                                         "lds_frac": lds_bytes / avg_s / 1e12 / LDS_PEAK_TBS,
                                         "note": "the SURVEY 8d 'achieved' above is an equivalent bandwidth (layer-by-layer bytes / time), "
                                                 "not bytes that crossed HBM; lds_peak = 256 CUs x 128 B/clk x 2.4 GHz"}}
            pmc = fused_pmc_reference(args, dom)
            if pmc:
                roofline["on_chip_view"]["pmc_reference"] = pmc
        else:
            roofline = {"kernel": dom, "bound": "hbm", "paced_by": "latency", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": None, "traffic": traffic, "avg_launch_us": avg_s * 1e6}
    kernel_us = {k: {"avg_us": v[0] / v[1] * 1e3, "launches_per_step": v[1] / args.steps} for k, v in fam_ms.items()}
    return roofline, kernel_us, traffic_db


def fused_pmc_reference(args, dom):
    """SQ counters of the C3 launch from a PMC pass of the same kernel build (profiles/r06_fused_pmc.txt: per-launch sums
    over the chip), as fractions: how long the LDS was active, how much of that was bank conflicts, how busy the MFMA pipes
    were.  Only for the configuration the pass was taken on; not measured in this run."""
    path = os.path.join(ROOT, "profiles", "r06_fused_pmc.txt")  # retaken on the round-6 build (tools/collect_profiles_r06.sh)
    if not os.path.isfile(path):
        return None
    if dom != "fused_solve" or args.family != "er" or (args.nodes, args.graphs, args.layers) != (200, 500, 20) or not os.path.isfile(path):
        return None
    c = {}
    for line in open(path):
        parts = line.split()
        if len(parts) == 2 and parts[0].startswith("SQ_") and parts[1].isdigit():
            c[parts[0]] = int(parts[1])
    need = ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA")
    if any(k not in c for k in need):
        return None
    return {"source": "profiles/r06_fused_pmc.txt (rocprofv3 --pmc passes of this kernel on this configuration, retaken on the round-6 build; not measured in this run)",
            "lds_active_frac_of_cu_busy_cycles": c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"],
            "lds_bank_conflict_frac_of_lds_active": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
            "mfma_busy_frac_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"]),
            "mfma_instructions_per_launch": c["SQ_INSTS_MFMA"]}


def spmm_probe(args, wl, traffic_db):
    """The stand-alone batched SpMM kernel of the north star (not part of the fused step): one hidden-layer
    aggregation with the GraphConvolution epilogue, timed with the same event hooks, outside the timed region.
    Two working sets: (a) this batch replayed (55 MB at C3: it stays in the 256 MiB Infinity Cache, so this is
    NOT an HBM measurement) and (b) ``--spmm-out-of-cache``: distinct batches visited round-robin so that
    every launch's inputs were last touched > 256 MiB of traffic ago (MI355X_MICROARCH.md, Infinity Cache
    residency rule) - the HBM figure."""
    torch, eng, hb, db = wl.torch, wl.eng, wl.hb, wl.db
    from distgcn_amd import datagen
    C = args.hidden
    dev = wl.dev

    def bytes_of(h, with_y0):
        return (h.num_edges + h.num_nodes) * 8 + (h.num_nodes + h.num_graphs) * 4 + (3 if with_y0 else 2) * 4 * C * h.num_nodes

    def make_set(h, d):
        lap = eng.supports(d)
        Zt = torch.randn(h.num_nodes, 2 * C, device=dev)
        Yt = torch.empty(h.num_nodes, C, device=dev)
        return (h, d, lap, Zt, Yt)

    plain = [False]  # True: K4 alone (gcn/layers.py:206: Y = S.Z, no Z0 / bias / activation) - exactly SURVEY 8d's B_spmm

    def launch(st):
        h, d, lap, Zt, Yt = st
        if plain[0]:
            eng.spmm(lap, Zt[:, C:], C, ldz=2 * C, graph_ptr=d.graph_ptr, num_graphs=h.num_graphs, max_nodes=h.max_nodes, out=Yt)
        else:
            eng.spmm(lap, Zt[:, C:], C, ldz=2 * C, graph_ptr=d.graph_ptr, num_graphs=h.num_graphs,
                     max_nodes=h.max_nodes, Y0=Zt, ldy0=2 * C, act="leaky_relu", out=Yt)

    def timed(sets, reps, plain_spmm=False):
        plain[0] = plain_spmm
        for st in sets:
            launch(st)
        torch.cuda.synchronize()
        eng.timing(True)
        for _ in range(reps):
            for st in sets:
                launch(st)
        torch.cuda.synchronize()
        eng.timing(False)
        ms, n = eng.timing_read("spmm")
        return ms / n * 1e-3

    first = make_set(hb, db)
    avg_s = timed([first], 50)
    nb, nb_plain = bytes_of(hb, True), bytes_of(hb, False)
    tkey = "spmm|%dx%d|C%d" % (args.graphs, args.nodes, C)
    line = {"kernel": "k_spmm_lds C=%d with the GraphConvolution epilogue" % C, "bound": "hbm",
            "working_set": "one batch replayed: %.1f MB, resident in the 256 MiB Infinity Cache (not an HBM figure)" % (nb / 1e6),
            "achieved": nb / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nb / avg_s / 1e9 / HBM_PEAK_GBS,
            "frac_plain_B_spmm": nb_plain / avg_s / 1e9 / HBM_PEAK_GBS,
            "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": nb, "plain_B_spmm_bytes_per_launch": nb_plain,
            "traffic": traffic_db.get(tkey, {}).get("hbm_bytes_per_launch")}
    # (b) out of the Infinity Cache: as many distinct batches as it takes to put > 320 MiB between two uses of a line
    per_set = nb  # bytes one launch touches
    nsets = int(np.ceil(320 * 2 ** 20 / per_set)) + 1
    if nsets <= 64 and args.family == "er":
        sets = [first]
        for i in range(1, nsets):
            h = datagen.er_batch(args.graphs, args.nodes, args.p, first_index=1_000_000 + i * args.graphs)
            sets.append(make_set(h, eng.upload(h)))
        avg2 = timed(sets, 4)
        nb_avg = float(np.mean([bytes_of(s[0], True) for s in sets]))
        nbp_avg = float(np.mean([bytes_of(s[0], False) for s in sets]))
        # (c) the same working set as ONE launch: a 4 000-graph batch (8 x this one), steady state instead of the
        # fill / drain of a 500-workgroup launch
        bigh = datagen.er_batch(8 * args.graphs, args.nodes, args.p, first_index=3_000_000)
        big = make_set(bigh, eng.upload(bigh))
        avg3 = timed([big], 6)
        avg3p = timed([big], 6, plain_spmm=True)
        avg2p = timed(sets, 4, plain_spmm=True)
        # K4 ALONE (no '+ Z0', bias, activation): the kernel and the byte count of SURVEY 8d's B_spmm, nothing else in the launch
        line["plain_spmm"] = {
            "kernel": "k_spmm_lds C=%d, Y = S.Z only (gcn/layers.py:206)" % C, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "one_launch_%d_graphs" % bigh.num_graphs: {"avg_launch_us": avg3p * 1e6, "achieved": bytes_of(bigh, False) / avg3p / 1e9,
                                                      "frac": bytes_of(bigh, False) / avg3p / 1e9 / HBM_PEAK_GBS},
            "rotating_%d_graph_launches" % args.graphs: {"avg_launch_us": avg2p * 1e6, "achieved": nbp_avg / avg2p / 1e9,
                                                        "frac": nbp_avg / avg2p / 1e9 / HBM_PEAK_GBS}}
        line["out_of_cache_one_launch"] = {
            "working_set": "one launch over %d graphs: %.0f MB algorithmic per launch (> 256 MiB Infinity Cache)"
                           % (bigh.num_graphs, bytes_of(bigh, True) / 1e6),
            "avg_launch_us": avg3 * 1e6, "achieved": bytes_of(bigh, True) / avg3 / 1e9,
            "frac": bytes_of(bigh, True) / avg3 / 1e9 / HBM_PEAK_GBS,
            "frac_plain_B_spmm": bytes_of(bigh, False) / avg3 / 1e9 / HBM_PEAK_GBS, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "traffic": traffic_db.get(tkey + "|one4000", {}).get("hbm_bytes_per_launch")}
        del big
        line["out_of_cache"] = {
            "working_set": "%d distinct batches visited round-robin: %.0f MB between two uses of a line (> 256 MiB Infinity Cache)"
                           % (nsets, nsets * nb_avg / 1e6),
            "avg_launch_us": avg2 * 1e6, "achieved": nb_avg / avg2 / 1e9, "frac": nb_avg / avg2 / 1e9 / HBM_PEAK_GBS,
            "frac_plain_B_spmm": nbp_avg / avg2 / 1e9 / HBM_PEAK_GBS, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "traffic": traffic_db.get(tkey + "|out_of_cache", {}).get("hbm_bytes_per_launch")}
    return line


def e2e_probe(args, wl, reference_state):
    """Host to host: per-graph CSR arrays (what a .mat parser hands over, mwis_dqn_test.py:304-310) in host memory
    -> membership bytes + totals + rounds in host memory, through distgcn_amd.serving.HostSolver (dgcn_host_solver_*:
    native packing into pinned memory, one H2D copy, one fused launch, one D2H copy; three batches in flight).
    PCIe-inclusive, so it is reported BESIDE ``value``, never as it."""
    from distgcn_amd.serving import HostSolver
    from distgcn_amd.batch import pack_csr_lists
    hb = wl.hb
    ps, cs, ws = [], [], []
    for n0, n1 in hb.graph_slices():
        e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
        ps.append(np.ascontiguousarray(hb.row_ptr[n0:n1 + 1] - e0, dtype=np.int32))
        cs.append(np.ascontiguousarray(hb.col_idx[e0:e1] - n0, dtype=np.int32))
        ws.append(np.ascontiguousarray(hb.weights[n0:n1]))
    pipe = HostSolver(wl.eng, wl.model, depth=3, pack_threads=16)
    batches = 300
    last = None
    for r in pipe.solve_many(((ps, cs, ws) for _ in range(60)), copy=False):  # (the packing threads and the pinned buffers warm)
        last = r
    same = bool(np.array_equal(last["state"], reference_state))
    runs = []
    for _ in range(3):  # three timed runs, the median reported (a 65 ms host-paced measurement: one run alone wanders by 10 %)
        wl.sync()
        t0 = time.perf_counter()
        for r in pipe.solve_many(((ps, cs, ws) for _ in range(batches)), copy=False):
            last = r
        runs.append(time.perf_counter() - t0)
    dt = sorted(runs)[1]
    # the host stage alone (packing into memory that is already mapped)
    staging = np.empty(32 * 1024 * 1024 + hb.num_edges * 4 + hb.num_nodes * 16, dtype=np.uint8)
    pack_csr_lists(ps, cs, ws, staging=staging)
    t1 = time.perf_counter()
    for _ in range(20):
        _, info = pack_csr_lists(ps, cs, ws, staging=staging)
    pack_ms = (time.perf_counter() - t1) / 20 * 1e3
    import ctypes
    from distgcn_amd import _lib
    ci = _lib.DgcnCompactInfo()
    compact = _lib.get_option("host_compact") != 0 and _lib.load().dgcn_pack_compact_layout(ctypes.byref(info), ctypes.byref(ci)) == 0
    h2d = int(ci.total_bytes) if compact else int(info.total_bytes)
    return {"value": hb.num_graphs * batches / dt, "unit": "graphs/s", "ms_per_batch": dt / batches * 1e3, "batches": batches,
            "path": "dgcn_host_solver_submit / _result: per-graph CSR arrays in host memory -> native packing into pinned memory (%s) -> "
                    "1 H2D copy (%.1f MB) -> %sdgcn_solve_batch -> 1 D2H copy (%.0f KB) -> membership + totals + rounds in host memory; "
                    "3 batches in flight"
                    % ("compact transfer format: 16-bit local column ids + degrees" if compact else "block-diagonal int32 CSR", h2d / 1e6,
                       ("(one-layer models: k_expand_compact -> ) " if args.layers == 1 or _lib.get_option("host_compact_direct") == 0
                        else "(read as it is by the fused kernel's image build) ") if compact else "", (hb.num_nodes + 12 * hb.num_graphs) / 1e3),
            "h2d_bytes_per_batch": h2d, "pack_ms_per_batch_ordinary_format": pack_ms, "results_equal_resident_step": same,
            "runs_graphs_per_s": [round(hb.num_graphs * batches / t) for t in runs]}


def two_stream_probe(args, wl, reference_state):
    """The same step issued alternately on two HIP streams (a second engine = a second scratch workspace, the same resident
    batch): the next launch's first workgroups take the CUs the previous launch's last ones have left - what a serving loop
    with two batches in flight gets.  Launch durations overlap, so this is a throughput figure BESIDE ``value`` (whose
    timed region stays on one stream and keeps the per-launch roofline meaningful)."""
    import torch
    from distgcn_amd.engine import Engine
    engs = [wl.eng, Engine(wl.dev)]
    outs = [[e.solve_buffers(wl.db, want_scores=False) for _ in range(2)] for e in engs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i & 1
            with torch.cuda.stream(streams[k]):
                engs[k].solve_fused(wl.db, wl.model, want_scores=False, out=outs[k][(i >> 1) & 1])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    run(200)
    steps = max(200, args.steps)
    dt = run(steps)
    same = all(bool(np.array_equal(o["state"].cpu().numpy()[:wl.hb.num_nodes], reference_state)) for pair in outs for o in pair)
    return {"value": wl.hb.num_graphs / dt, "unit": "graphs/s", "ms_per_step": dt * 1e3, "steps": steps, "streams": 2,
            "results_equal_resident_step": same}


def single_graph_probe(args, wl):
    """The reference's own call pattern - ONE graph per call (mwis_dqn_call.py:140-143, once per slot in the wireless
    loop): host CSR arrays of one graph of this batch -> its set, total and rounds in host memory through a one-slot
    HostSolver (one interpreter call around dgcn_host_solver_submit / _result), and the kernel alone on a resident
    one-graph batch.  Latency, reported beside ``value``."""
    import torch
    from distgcn_amd.serving import HostSolver
    hb = wl.hb
    graphs = []
    for n0, n1 in list(hb.graph_slices())[:8]:
        e0, e1 = int(hb.row_ptr[n0]), int(hb.row_ptr[n1])
        graphs.append(([np.ascontiguousarray(hb.row_ptr[n0:n1 + 1] - e0, dtype=np.int32)],
                       [np.ascontiguousarray(hb.col_idx[e0:e1] - n0, dtype=np.int32)], [np.ascontiguousarray(hb.weights[n0:n1])]))
    hs = HostSolver(wl.eng, wl.model, depth=1)
    calls = 400
    for i in range(40):
        hs.solve(*graphs[i % len(graphs)])
    t0 = time.perf_counter()
    for i in range(calls):
        hs.solve(*graphs[i % len(graphs)])
    call_us = (time.perf_counter() - t0) / calls * 1e6
    hs.close()
    db = wl.eng.upload(hb.subset(0, 1))
    out = wl.eng.solve_buffers(db, False)
    for _ in range(40):
        wl.eng.solve_fused(db, wl.model, out=out)
    torch.cuda.synchronize()
    wl.eng.timing(True)
    for _ in range(200):
        wl.eng.solve_fused(db, wl.model, out=out)
    torch.cuda.synchronize()
    wl.eng.timing(False)
    ms, n = wl.eng.timing_read("fused_solve")
    return {"call_us": call_us, "kernel_us": ms / max(n, 1) * 1e3, "calls": calls,
            "path": "HostSolver(depth=1).solve on one graph of the batch: native pack into pinned memory, the solve kernel (k_fused; k_big beyond its LDS budget) reading it in "
                    "place (several workgroups per graph on one XCD when the stack is deep enough), results written to pinned memory"}


def parity_probe(wl, budget_s):
    """Scores and sets of THIS run's batch, fetched from the GPU, against the oracle restatement graph by graph
    (oracle/parity.py, the checker: float32 and float64 evaluations of the reference's formula, the reference's greedy
    search on the restatement's priorities), for as many graphs as fit the time budget (all of C2 / C3)."""
    from oracle import parity
    eng, db, hb = wl.eng, wl.db, wl.hb
    res = eng.solve_fused(db, wl.model, want_scores=True)
    wl.sync()
    scores = res["scores"].reshape(-1).cpu().numpy()
    state = res["state"].cpu().numpy()
    reports, t0, g = [], time.perf_counter(), 0
    stride = 1
    order = list(range(hb.num_graphs))
    while g < len(order) and (time.perf_counter() - t0 < budget_s or g < 8):
        reports += parity.batch_report(hb, wl.layers, scores, state, graphs=[order[g]])
        g += stride
    summ = parity.summarize(reports)
    out = {"graphs_checked": len(reports), "of": hb.num_graphs, "seconds": round(time.perf_counter() - t0, 1),
           "max_err": summ["max_err_vs_f32_restatement"], "graphs_over_1e-5": summ["graphs_over_1e-5_vs_f32_restatement"],
           "of_those_restatement_further_from_f64": summ["of_those_restatement_further_from_f64"],
           "max_err_vs_f64": summ["max_err_vs_f64"], "graphs_over_1e-5_vs_f64": summ["graphs_over_1e-5_vs_f64"],
           "restatement_max_err_vs_f64": summ["restatement_max_err_vs_f64"], "sets_differing": summ["sets_differing"],
           "against": "oracle/ref_numpy float32 / float64 restatements, one graph per call; sets: reference local_greedy_search on the "
                      "float32 restatement's priorities; errors in units of max(1, |score|)",
           "whole_configurations": "profiles/r03_parity_full_size.json (tests/test_full_size_parity.py: every graph of C2, C3, C4 l=1/l=20, C5-size)"}
    return out


def margin_probe(wl, measured_err=None):
    """how many of this batch's selected sets could a score error flip?  For delta = 2 x the score
    tolerance (1e-5) and 2 x the error parity_probe just measured against the float32 restatement on this very batch,
    the number of graphs with at least one excluded vertex whose exclusion
    does not survive a per-score error of delta (dgcn_margin_risk_batch); the others provably keep their set."""
    eng, db = wl.eng, wl.db
    res = eng.solve_fused(db, wl.model, want_scores=True)
    scores = res["scores"].reshape(-1)
    out = {"criterion": "excluded vertex v is safe iff a member neighbour u has p_u - p_v > delta*(|w_u|+|w_v|)", "graphs": wl.hb.num_graphs}
    deltas = [("delta_2x_tolerance_2e-5", 2e-5)]
    if measured_err is not None:  # twice what parity_probe measured against the float32 restatement on THIS batch
        deltas.append(("delta_2x_measured_error", 2.0 * measured_err))
    for name, delta in deltas:
        r = eng.margin_risk(db, res["state"], delta, scores=scores, weights=db.weights).cpu().numpy()
        out[name] = {"delta": delta, "graphs_at_risk": int((r > 0).sum()), "vertices_at_risk": int(r.sum())}
    return out


def run_ranks_if_asked(args, argv):
    """``--gpus N`` without a launcher: start N rank processes (this process has not touched the GPU and
    never will) and exit with their code."""
    from distgcn_amd import parallel
    if args.gpus <= 1 or parallel.launched_by_a_launcher():
        return
    script = os.path.abspath(sys.argv[0])
    rc = parallel.spawn_local_ranks(args.gpus, script, list(sys.argv[1:] if argv is None else argv))
    sys.stdout.flush()
    sys.exit(rc)


def main(argv=None, workload_factory=None):
    """``workload_factory(args, rank, world, local)`` builds the per-rank compute (default: ``GpuWorkload``);
    tests/_bench_gloo_driver.py passes a CPU stand-in so that the launcher, rendezvous, sharding, gather and
    report code below - the very same code - runs on two gloo ranks without a GPU."""
    args = parse(argv)
    run_ranks_if_asked(args, argv)

    import torch
    import torch.distributed as dist
    from distgcn_amd import parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if rank != 0:  # only rank 0 reports; keep other ranks' library banners out of the launcher's stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    use_dist = world > 1 or args.force_dist or os.environ.get("DGCN_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        parallel.init_rank_group(args.backend)

    if args.config in ("C5", "MC900-rollout"):
        args.no_gather = True  # (the searches' sets stay on their ranks: nothing of C5 is gathered per launch)
    wl = (workload_factory or (RolloutWorkload if args.config in ("C5", "MC900-rollout") else GpuWorkload))(args, rank, world, local)
    hb = wl.hb
    cdev = wl.collective_device()

    # one buffer layout for every rank: sizes of the largest shard
    caps = torch.tensor([max(hb.num_nodes, 1), max(hb.num_graphs, 1)], dtype=torch.int64, device=cdev)
    sizes_all = None
    if use_dist:
        mine = torch.tensor([hb.num_nodes, hb.num_graphs], dtype=torch.int64, device=cdev)
        sizes_all = torch.empty(2 * world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(sizes_all, mine)
        sizes_all = sizes_all.cpu().numpy().reshape(world, 2)
        dist.all_reduce(caps, op=dist.ReduceOp.MAX)
    cap_nodes, cap_graphs = (int(x) for x in caps.cpu().numpy())
    wl.make_buffers(cap_nodes, cap_graphs)

    gather_buf = None
    pending = []  # (work handle, tensor it reads) of gathers still in flight

    def step():
        res = wl.step()
        nonlocal gather_buf
        if use_dist and not args.no_gather:
            flat = res["flat"]
            if gather_buf is None:
                gather_buf = torch.empty(world * flat.numel(), dtype=torch.uint8, device=flat.device)
            # the batch gather of SURVEY 8e: membership + totals + rounds (+ status) of every rank, ONE collective.
            # Issued async: RCCL runs it on its own stream behind this step's kernel, so it overlaps the NEXT
            # step's compute instead of stalling the compute stream for a latency-bound ~100 KB collective.
            work = dist.all_gather_into_tensor(gather_buf, flat, async_op=True)
            pending.append((work, flat))
            if len(pending) > 2:
                pending.pop(0)[0].wait()
        return res

    def drain():
        while pending:
            pending.pop(0)[0].wait()

    # (the event pairs of the timed launches exist before anything runs: creating them is a millisecond the GPU would sit idle
    # right in front of the timed region - and with the driver's 20-step window the clock dip behind such a gap is visible)
    wl.timing(True)
    wl.timing(False)
    if hasattr(wl, "settle") and args.config not in ("C5", "MC900-rollout"):
        wl.settle()
    # W untimed warm-up steps; the result is validated after the FIRST of them (a device-to-host copy and host work), so that
    # nothing but launches lies between the rest of the warm-up and the barrier that opens the timed region
    res = None
    for i in range(args.warmup):
        res = step()
        if i == 0:
            drain()
            wl.sync()
            wl.check(res)
    drain()
    wl.sync()

    wl.timing(True)
    if use_dist:
        dist.barrier()
    wl.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    drain()  # every gather of the timed steps has completed before the clock stops
    wl.sync()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    wl.timing(False)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    wl.check(res)

    # ---- what the collective library saw, and what the last gather delivered (decoded on rank 0)
    dist_report = None
    if use_dist:
        dist_report = parallel.census(cdev)
        if gather_buf is not None:
            got = gather_buf.cpu().numpy().reshape(world, -1)
            own = res["flat"].cpu().numpy()
            graphs = members = 0
            weight = 0.0
            faults = 0
            for r in range(world):
                d = decode_flat(got[r], res["layout"], int(sizes_all[r, 0]), int(sizes_all[r, 1]))
                graphs += int((d["rounds"] >= 0).sum())
                members += int((d["state"] == 1).sum())
                weight += float(d["totals"].sum())
                faults |= d["status"]
            dist_report["gathered_last_step"] = {
                "bytes_per_rank": int(got.shape[1]), "graphs": graphs, "set_members": members, "total_weight": weight,
                "status_bits": faults, "own_slot_matches_own_result": bool(np.array_equal(got[rank], own)),
                "content": "state[uint8 per vertex] + totals[f64 per graph] + rounds[i32 per graph] + status[i32]"}

    fam_ms = wl.kernel_times()
    uninstrumented = None
    if not isinstance(wl, RolloutWorkload) and not use_dist and world == 1:
        # (the same K steps once more with plain launches: what the HIP event pair round every launch of the timed region costs)
        wl.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            res_u = step()
        drain()
        wl.sync()
        dt_u = time.perf_counter() - t1
        wl.check(res_u)
        uninstrumented = {"ms_per_step": 1e3 * dt_u / max(args.steps, 1), "graphs_per_s": wl.job_graphs * args.steps / dt_u, "steps": args.steps,
                          "note": "the K steps of the timed region again, WITHOUT the per-launch HIP event pairs (hipExtLaunchKernelGGL) that "
                                  "`value` / `ms_per_step` include; not `value`: the contract's figure is the instrumented one"}
    if isinstance(wl, RolloutWorkload) and not use_dist:
        # The timed region above brackets EVERY launch with a HIP event pair (the contract's live kernel times).  A search is ~140
        # launches, half of them a few microseconds long (the tail's probes, the empty launches behind the end of a search), and an
        # event pair costs a launch several microseconds of queue time: the same K searches once more without the events.
        wl.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            res_u = wl.step()
        wl.sync()
        dt_u = time.perf_counter() - t1
        wl.check(res_u)
        uninstrumented = {"ms_per_search": 1e3 * dt_u / max(args.steps, 1), "graphs_per_s": args.graphs * args.steps / dt_u, "searches": args.steps,
                          "note": "the K searches of the timed region again, WITHOUT the per-launch HIP event pairs (hipExtLaunchKernelGGL) that "
                                  "`value` / `ms_per_step` include; not `value`: the contract's figure is the instrumented one"}
    if isinstance(wl, RolloutWorkload):
        step_family = next((f for f in ("fused_residual", "big_residual", "wide_residual") if fam_ms.get(f, (0.0, 0))[1]), "fused_residual")
        ms, n = fam_ms.get(step_family, (0.0, 0))
        per_step, search_steps = wl.residual_bytes() if rank == 0 else ([0.0], 1)
        # A search of `search_steps` solver steps (the untimed replay's count) is `calls` calls of dgcn_solve_residual_batch:
        # one step each, and - once every graph has at most 64 undecided vertices - the rest inside ONE launch of the tail
        # kernel (csrc/tail.hip).  The timed milliseconds of k_fused belong to its PRODUCTIVE launches: the host loop also
        # issues a few behind the end of a search (groups of up to 32 between progress read-backs) that find nothing left and
        # return at once - counting them would understate the launch time and overstate the rate.
        calls = max(min(wl.launches, search_steps), 1)
        productive = max(min(n, calls * max(args.steps, 1)), 1)
        avg_s = ms / productive * 1e-3
        algo = float(sum(per_step[:calls]))
        ach = (algo / calls) / avg_s / 1e9 if avg_s > 0 else None
        tms, tn = fam_ms.get("tail_finish", (0.0, 0))
        c5_traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.isfile(tpath):  # FETCH_SIZE / WRITE_SIZE passes of a whole search, averaged over its k_fused launches (the empty ones included)
            c5_traffic = json.load(open(tpath)).get("%s|%dx%d|l%d" % (step_family, args.graphs, args.nodes, args.layers), {}).get("hbm_bytes_per_launch")
        step_kernel = {"fused_residual": "k_fused<residual graph>", "big_residual": "k_big / k_big2<residual graph> (any-size path)",
                       "wide_residual": "k_wide1 residual mode (any-size path, one- and two-layer models)"}[step_family]
        roofline = {"kernel": "%s (one launch = forward on every residual graph + %d greedy completions + pick)" % (step_kernel, args.beam),
                    "bound": "hbm", "paced_by": "latency" if step_family == "wide_residual" else "lds+mfma",
                    "frac_is": "equivalent bandwidth: SURVEY 8d algorithmic bytes / launch time / HBM peak", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if ach else None,
                    "traffic": c5_traffic,
                    "traffic_source": "PMC FETCH_SIZE/WRITE_SIZE passes of tools/run_iterative.py --only rollout on this configuration (profiles/hbm_traffic.json): average over ALL launches of the step kernel in a search, the empty ones behind its end included (null: no pass taken for this configuration)", "avg_launch_us": avg_s * 1e6, "steps_per_search": search_steps, "launches_per_search": calls,
                    "empty_launches_per_search": n / max(args.steps, 1) - calls,
                    "algorithmic_bytes_per_launch": algo / calls,
                    "formula": "SURVEY 8d: sum over layers of B_spmm on the RESIDUAL graphs of each launch (sizes from an untimed replay), averaged over a search's launches of the step kernel",
                    "tail": {"kernel": "k_tail (the last steps of every graph in one launch, from <= 64 undecided vertices on)",
                             "steps_inside": search_steps - calls, "ms_per_search": tms / max(args.steps, 1),
                             "launches_per_search": tn / max(args.steps, 1),
                             "algorithmic_bytes": float(sum(per_step[calls:]))}}
        kernel_us = {step_family: {"avg_us": avg_s * 1e6, "launches_per_step": n / max(args.steps, 1)},
                     "tail_finish": {"ms_per_step": tms / max(args.steps, 1), "launches_per_step": tn / max(args.steps, 1)}}
        traffic_db = {}
    else:
        roofline, kernel_us, traffic_db = roofline_objects(args, wl, fam_ms)

    spmm_line = None
    if rank == 0 and world == 1 and not args.no_spmm_probe and isinstance(wl, GpuWorkload):  # N = 1 line only: ranks must not wait on it
        spmm_line = spmm_probe(args, wl, traffic_db)

    margin = parity_rep = None
    if rank == 0 and world == 1 and isinstance(wl, GpuWorkload) and wl.ring is not None:
        if args.parity_seconds > 0:
            parity_rep = parity_probe(wl, args.parity_seconds)
        margin = margin_probe(wl, parity_rep["max_err"] if parity_rep else None)
    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e and isinstance(wl, GpuWorkload) and wl.ring is not None:
        e2e = e2e_probe(args, wl, res["state"].cpu().numpy())

    single = two = None
    if rank == 0 and world == 1 and not args.no_e2e and isinstance(wl, GpuWorkload) and wl.ring is not None:
        single = single_graph_probe(args, wl)
        if args.two_streams:
            two = two_stream_probe(args, wl, res["state"].cpu().numpy()[:wl.hb.num_nodes])

    if rank == 0:
        per_gpu = args.graphs if args.scaling == "weak" else None
        out = {
            "metric": ("graphs/sec (GCN fwd + greedy MWIS) on ER N=%d p=%g" % (args.nodes, args.p)) if args.family == "er"
                      else "graphs/sec (GCN fwd + greedy MWIS) on joint 3-channel conflict graphs of %d flows" % (args.nodes // 3) if args.family == "mc"
                      else "graphs/sec (GCN fwd + greedy MWIS) on the BA test2 mix",
            "value": wl.job_graphs * args.steps / dt,
            "unit": "graphs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic %s graphs (seeded), uniform(0,1) weights; " % {"er": "ER", "ba": "BA", "mc": "multi-channel joint conflict"}[args.family] + wl.weights_note,
            "config": {"workload": "%s: %d %s %s, l=%d c=%d GCN forward + local greedy, supports rebuilt every step"
                                   % (workload_name(args), args.graphs,
                                      ("ER graphs N=%d p=%g" % (args.nodes, args.p)) if args.family == "er"
                                      else ("joint conflict graphs of 3 channels x %d flows (base ER p=%g, 80 %% of the edges per channel)" % (args.nodes // 3, args.p)) if args.family == "mc"
                                      else "BA test2-mix graphs (N 100..300)",
                                      "per GPU" if args.scaling == "weak" else "in the job, sharded by graph over the ranks",
                                      args.layers, args.hidden),
                       "settle_s_before_warmup": round(getattr(wl, "settle_s", 0.0), 2),
                       "forward_mode": getattr(wl, "mode_name", "?") + (" (forced down the any-size path)" if args.any_size_path else ""),
                       "graphs_per_gpu": per_gpu, "job_graphs": wl.job_graphs,
                       "parallelism": "graph-sharded x%d" % world},
            "e2e": e2e,
            "single_graph": single,
            "two_streams": two,
            "parity_full_size": parity_rep,
            "margin_risk": margin,
            "dist": dist_report,
            "roofline": roofline,
            "without_launch_events": uninstrumented,
            "spmm_kernel_roofline": spmm_line,
            "kernels": kernel_us,
        }
        if isinstance(wl, RolloutWorkload):
            if args.family == "mc":
                out["metric"] = "graphs/sec (GCN-guided rollout search, b=%d, to completion) on joint 3-channel conflict graphs of %d vertices" % (args.beam, args.nodes)
                out["config"]["workload"] = ("MC900-rollout: %d joint 3 x %d-flow conflict graphs per GPU, l=%d c=%d GCN2_DQN forward + %d-candidate rollout per "
                                             "step of the search, every step ONE launch of the any-size path, ~%d calls per search"
                                             % (args.graphs, args.nodes // 3, args.layers, args.hidden, args.beam, wl.launches))
            else:
                out["metric"] = "graphs/sec (GCN-guided rollout search, b=%d, to completion) on ER N=%d p=%g" % (args.beam, args.nodes, args.p)
                out["config"]["workload"] = ("C5: %d ER graphs N=%d p=%g per GPU, l=%d c=%d GCN2_DQN forward + %d-candidate rollout per step of the search, "
                                             "~%d calls per search" % (args.graphs, args.nodes, args.p, args.layers, args.hidden, args.beam, wl.launches))
        if world == 1 and args.cpu_seconds > 0 and isinstance(wl, RolloutWorkload):
            out["cpu_baseline"] = c5_cpu_baseline(wl, args.cpu_seconds)
        elif world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(hb, wl.layers, args.cpu_seconds)
            if not args.no_cpu_pool and args.family == "er":
                out["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(args, min(args.cpu_seconds, 6.0))
        else:
            out["cpu_baseline"] = None
        # RCCL writes a version banner to the C stdout buffer; push it out first so that the JSON
        # line is the LAST line this process prints
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:  # anything RCCL still holds in its C stdio buffer must not follow the JSON line
        try:
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 1)
        except OSError:
            pass


if __name__ == "__main__":
    main()
