#!/usr/bin/env python
"""Where the reference-arithmetic mode spends its time: one synthetic workload (default BASELINE configs[2]: 1M x 50, K = 100, 10
batches), `ref_arith = 1`, to convergence; prints wall time per run, GPU time per phase (HIP events) and the host timers.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table (tools/final_profiles.sh, part b)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_data import synth  # noqa: E402
from harmony_amd import Harmony, prepare_setup_args  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=1000000)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--passes", type=int, default=0, help="seq_passes (0: the library's default, 2 since round 5)")
ap.add_argument("--mode", default="ref_arith")
ap.add_argument("--c5", action="store_true", help="BASELINE configs[4]'s shape: K = 200, three nested covariates 8 > 64 > 128")
a = ap.parse_args()
if a.c5:
    Z, meta, _ = synth(a.cells, d=50, levels=(8, 64, 128), seed=7, nested=True)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=200)
else:
    Z, meta, _ = synth(a.cells, d=50, levels=(10,), seed=7)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
g = Harmony(seed=1, **({a.mode: 1} if a.mode != "default" else {}))
if a.passes:
    g._set("seq_passes", a.passes)
g.setup(**skw)


def run():
    g.restart()
    g.init_cluster_cpp()
    it = 0
    for it in range(1, 11):
        g.cluster_cpp()
        g.moe_correct_ridge_cpp()
        if g.check_convergence(1):
            break
    return it


run()
g.set_profile(2)
t0 = time.perf_counter()
its = [run() for _ in range(a.steps)]
g.getZcorr()
dt = (time.perf_counter() - t0)
phases = {k: round(g._scalar("gputimer:" + k) / a.steps, 3) for k in
          ("kmeans_centers", "cluster_head", "randomize", "EO_update", "Rcells_update", "objective", "correct_ridge_loop", "ridge_statistics",
           "arma_inv", "update_Zcorr")}
host = {k: round(g.timer(k) / (a.steps + 1), 3) for k in ("init_cluster", "cluster", "update_R", "moe_correct_ridge")}
print(json.dumps({"mode": a.mode, "shape": "configs[4]" if a.c5 else "configs[2]", "cells": a.cells, "passes": a.passes, "iterations": its, "ms_per_run_incl_egress": 1e3 * dt / a.steps,
                  "gpu_phase_ms_per_run": phases, "host_wall_ms_per_run": host, "seq_residual": g._scalar("seq:residual"),
                  "seq_runs": g._scalar("seq:runs"), "passes_per_group": [float(x) for x in g._get("seq:group_passes")], "evaluations_per_group": [float(x) for x in g._get("seq:group_runs")]}))
