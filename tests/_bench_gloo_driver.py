"""Runs bench.py's OWN main() - launcher, rendezvous, sharding, result gather, census, JSON report - on CPU
ranks over gloo.  Only the per-rank compute is replaced: the CPU twin (oracle/, allowed in tests) stands in
for the MI355X engine, which cannot run here.  Started by tests/test_dist_gloo.py as
``python tests/_bench_gloo_driver.py --gpus 2 --backend gloo ...``; bench.py re-launches THIS script per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench  # noqa: E402


class CpuTwinWorkload:
    def __init__(self, args, rank, world, local):
        import torch
        from distgcn_amd import datagen
        self.torch, self.args = torch, args
        self.hb, self.job_graphs = bench.build_host_batch(args, rank, world)
        self.layers, self.weights_note = datagen.random_model(args.layers, args.hidden), "random-init weights (CPU twin stand-in)"
        self.mode_name = "cpu-twin"
        self.calls = 0

    def collective_device(self):
        return "cpu"

    def make_buffers(self, cap_nodes, cap_graphs):
        from distgcn_amd.engine import packed_layout, solve_buffer_specs
        self.size, self.layout = packed_layout(solve_buffer_specs(cap_nodes, cap_graphs, False))

    def step(self):
        from oracle import ctwin
        self.calls += 1
        flat = np.zeros(self.size, dtype=np.uint8)
        if self.hb.num_graphs:
            r = ctwin.solve(self.hb, self.layers)
            for name, arr in (("state", r["state"].astype(np.uint8)), ("totals", r["totals"].astype(np.float64)),
                              ("rounds", r["rounds"].astype(np.int32))):
                o, nb, dt = self.layout[name]
                flat[o:o + arr.nbytes] = arr.view(np.uint8)
        t = self.torch.from_numpy(flat)
        return {"flat": t, "layout": self.layout, "status": self.torch.zeros(1, dtype=self.torch.int32)}

    def sync(self):
        pass

    def check(self, res):
        assert int(res["status"].item()) == 0

    def timing(self, on):
        pass

    def kernel_times(self):
        return {}


if __name__ == "__main__":
    bench.main(workload_factory=CpuTwinWorkload)
