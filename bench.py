#!/usr/bin/env python3
"""Benchmark of the hot path: GCN forward + local-greedy MWIS on a batch of conflict graphs.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the whole hot path over one batch already resident in HBM:
support construction (L = I - D^-1/2 A D^-1/2) -> 20-layer c32 GCN forward -> priority product ->
local greedy search.  Workload at every N: BASELINE.json configs[2] (C3), 500 ER graphs
G(200, 0.1) per GPU (weak scaling: each rank owns its own 500 graphs; no data-path collective; the
one collective, the end-of-step all_gather of the membership bytes, is inside the timed step).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graphs", type=int, default=500, help="graphs per GPU")
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--family", choices=["er", "ba"], default="er",
                    help="er: G(nodes, p) (C2 / C3); ba: the BA test2 mix of SURVEY 8d (C4: one GPU's share), ignores --nodes/--p")
    ap.add_argument("--layers", type=int, default=20)
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--mode", choices=["layered", "fused", "auto"], default="auto")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-cpu-pool", action="store_true", help="skip the all-cores CPU figure (forked workers)")
    ap.add_argument("--no-gather", action="store_true", help="skip the end-of-step membership gather at N>1")
    ap.add_argument("--no-spmm-probe", action="store_true", help="skip the stand-alone SpMM kernel measurement")
    return ap.parse_args()


def load_layers(args):
    """Trained weights of the shipped IS4SAT l20/c32 checkpoint when the fixture copy is present
    (tests/golden/models.npz), random Glorot weights of the same architecture otherwise."""
    from distgcn_amd import datagen
    from distgcn_amd.gcn.models import layers_from_params
    path = os.path.join(ROOT, "tests", "golden", "models.npz")
    name = "result_IS4SAT_deep_ld1_c%d_l%d_cheb1_diver1_mwis_dqn" % (args.hidden, args.layers)
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
        return "C4 (one GPU's share)" if args.graphs == 500 else "custom"
    key = (args.graphs, args.nodes, args.p, args.layers, args.hidden)
    return {(500, 200, 0.1, 20, 32): "C3", (500, 100, 0.1, 1, 32): "C2"}.get(key, "custom")


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


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from distgcn_amd import datagen
    from distgcn_amd.engine import Engine, DeviceModel, MODE_LAYERED, MODE_FUSED

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:  # only rank 0 reports; keep other ranks' library banners out of the launcher's stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    use_dist = world > 1 or os.environ.get("DGCN_BENCH_FORCE_DIST") == "1"  # the latter: exercise the RCCL path on 1 GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)

    if args.family == "ba":
        hb = datagen.ba_test2_batch(args.graphs, first_index=rank * args.graphs)
    else:
        hb = datagen.er_batch(args.graphs, args.nodes, args.p, first_index=rank * args.graphs)
    layers, weights_note = load_layers(args)
    eng = Engine(dev)
    db = eng.upload(hb)
    model = DeviceModel(layers, dev)
    mode_name = args.mode
    if mode_name == "auto":
        mode_name = os.environ.get("DGCN_BENCH_MODE", "fused")
    mode = MODE_FUSED if mode_name == "fused" else MODE_LAYERED

    gather_buf = None
    if use_dist and not args.no_gather:
        gather_buf = torch.empty(world * hb.num_nodes, dtype=torch.uint8, device=dev)

    pending = []  # (work handle, tensors it reads) of gathers still in flight
    ring = [eng.solve_buffers(db, want_scores=False) for _ in range(4)] if mode == MODE_FUSED else None
    counter = [0]

    def step():
        if ring is not None:
            # steady-state serving loop: output buffers are re-used (4-deep ring: a buffer is rewritten
            # only after the gather that reads it has been waited for)
            counter[0] += 1
            res = eng.solve_fused(db, model, want_scores=False, out=ring[counter[0] % 4])
        else:
            db.lap = None  # supports are part of the path: rebuild them every step
            res = eng.solve(db, model, mode=mode)
        if gather_buf is not None:
            # the batch gather of SURVEY 8e (membership only; every rank has equal N).  Issued async: RCCL
            # runs it on its own stream behind this step's kernel, so it overlaps the NEXT step's compute
            # instead of stalling the compute stream for a latency-bound ~100 KB collective.
            work = dist.all_gather_into_tensor(gather_buf, res["state"], async_op=True)
            pending.append((work, res["state"]))
            if len(pending) > 2:
                pending.pop(0)[0].wait()
        return res

    def drain():
        while pending:
            pending.pop(0)[0].wait()

    for _ in range(args.warmup):
        res = step()
    drain()
    torch.cuda.synchronize()
    eng.check_status(res["status"])

    eng.timing(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    drain()  # every gather of the timed steps has completed before the clock stops
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    eng.timing(False)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel, from HIP events recorded around its launches in the
    # timed region (on the launch stream), against SURVEY 8d's algorithmic bytes
    fam_ms = {}
    for fam in ("supports", "transform", "spmm", "lgs", "fused_forward", "fused_solve"):
        ms, n = eng.timing_read(fam)
        if n:
            fam_ms[fam] = (ms, n)
    roofline = None
    per_layer_bytes = spmm_algorithmic_bytes(hb, layers, with_y0=False)  # SURVEY 8d, layer by layer
    traffic_db = {}
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.isfile(tpath):
        traffic_db = json.load(open(tpath))
    if fam_ms:
        dom = max(fam_ms, key=lambda k: fam_ms[k][0])
        ms, n = fam_ms[dom]
        avg_s = ms / n * 1e-3
        tkey = ("spmm|%dx%d|C%d" % (args.graphs, args.nodes, args.hidden)) if dom == "spmm" else \
            "%s|%dx%d|l%d" % (dom, args.graphs, args.nodes, args.layers)
        traffic = traffic_db.get(tkey, {}).get("hbm_bytes_per_launch")
        if dom == "spmm":
            per = spmm_algorithmic_bytes(hb, layers, with_y0=True)
            avg_bytes = sum(per) / len(per)
            ach = avg_bytes / avg_s / 1e9
            roofline = {"kernel": "k_spmm_lds (all %d launches of a step)" % len(per), "bound": "hbm",
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": traffic, "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": avg_bytes,
                        "formula": "SURVEY 8d B_spmm + 4*C*N for the fused '+Z0' read, averaged over the layers"}
        elif dom in ("fused_forward", "fused_solve"):
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
            roofline = {"kernel": "k_fused (%s: whole path, one launch per step)" % dom, "bound": "hbm",
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": traffic, "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": algo,
                        "formula": "SURVEY 8d: sum over layers of B_spmm (CSR + Z read + Y write), %d graphs" % hb.num_graphs,
                        "fp32_matrix_view": {"flops_per_launch": flops, "achieved_tflops": flops / avg_s / 1e12,
                                             "peak_tflops": F32_MATRIX_PEAK_TF,
                                             "frac": flops / avg_s / 1e12 / F32_MATRIX_PEAK_TF}}
        else:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": None, "traffic": traffic, "avg_launch_us": avg_s * 1e6}
    kernel_us = {k: {"avg_us": v[0] / v[1] * 1e3, "launches_per_step": v[1] / args.steps} for k, v in fam_ms.items()}

    # ---- the stand-alone batched SpMM kernel of the north star (not part of the fused step): one
    # hidden-layer aggregation over the same batch, timed with the same event hooks, outside the timed region
    spmm_line = None
    if rank == 0 and not args.no_spmm_probe:
        lap = eng.supports(db)
        C = args.hidden
        Zt = torch.randn(hb.num_nodes, 2 * C, device=dev)
        Yt = torch.empty(hb.num_nodes, C, device=dev)
        def one():
            eng.spmm(lap, Zt[:, C:], C, ldz=2 * C, graph_ptr=db.graph_ptr, num_graphs=hb.num_graphs,
                     max_nodes=hb.max_nodes, Y0=Zt, ldy0=2 * C, act="leaky_relu", out=Yt)
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        eng.timing(True)
        for _ in range(50):
            one()
        torch.cuda.synchronize()
        eng.timing(False)
        ms, n = eng.timing_read("spmm")
        nb = (hb.num_edges + hb.num_nodes) * 8 + (hb.num_nodes + hb.num_graphs) * 4 + 3 * 4 * C * hb.num_nodes
        ach = nb / (ms / n * 1e-3) / 1e9
        spmm_line = {"kernel": "k_spmm_lds C=%d with the GraphConvolution epilogue" % C, "bound": "hbm", "achieved": ach,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_us": ms / n * 1e3,
                     "algorithmic_bytes_per_launch": nb,
                     "traffic": traffic_db.get("spmm|%dx%d|C%d" % (args.graphs, args.nodes, C), {}).get("hbm_bytes_per_launch")}

    if rank == 0:
        out = {
            "metric": ("graphs/sec (GCN fwd + greedy MWIS) on ER N=%d p=%g" % (args.nodes, args.p)) if args.family == "er"
                      else "graphs/sec (GCN fwd + greedy MWIS) on the BA test2 mix",
            "value": world * args.graphs * args.steps / dt,
            "unit": "graphs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic %s graphs (seeded), uniform(0,1) weights; " % ("ER" if args.family == "er" else "BA") + weights_note,
            "config": {"workload": "%s: %d %s per GPU, l=%d c=%d GCN forward + local greedy, supports rebuilt every step"
                                   % (workload_name(args), args.graphs,
                                      ("ER graphs N=%d p=%g" % (args.nodes, args.p)) if args.family == "er"
                                      else "BA test2-mix graphs (N 100..300)", args.layers, args.hidden),
                       "forward_mode": mode_name, "graphs_per_gpu": args.graphs, "parallelism": "graph-sharded x%d" % world},
            "roofline": roofline,
            "spmm_kernel_roofline": spmm_line,
            "kernels": kernel_us,
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(hb, layers, args.cpu_seconds)
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
