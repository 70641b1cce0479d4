#!/usr/bin/env python
"""Round 5: what the reference-arithmetic mode's distance to the faithful oracle depends on.  BASELINE configs[2] (1M x 50, K = 100,
10 batches, the gap table's seeds) with `ref_arith = 1` under several settings of the restarted sums -- default pass counts,
`seq_strict` (every group iterated to its fixed point), a fixed larger pass count -- each against the oracle's faithful run: Z_corr,
max |dR|, hard-assignment flips by margin, where the largest |dR| sits (cell, cluster, both memberships, the cluster's O / E rows),
passes per group and wall time.  Writes gpurun_out/r5_strict_probe_<N>.json."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harmony_amd import Harmony, prepare_setup_args  # noqa: E402
from helpers import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.oracle import OracleHarmony  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=1000000)
ap.add_argument("--settings", default="default,strict,passes6")
a = ap.parse_args()
N, K, B, seed = a.cells, 100, 10, 3
Z, meta, _ = synth(N, d=50, levels=(B,), seed=7)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
g0 = Harmony(seed=seed)
g0.setup(**skw)
Y0 = g0.kmeans_centers()
del g0


def iterate(o):
    it = 0
    for it in range(1, 11):
        assert o.cluster_cpp() == 0
        o.moe_correct_ridge_cpp()
        if o.check_convergence(1):
            break
    return it


ref = {}


def cpu():
    o = OracleHarmony(mask=0, seed=seed)
    o.setup(**skw)
    t0 = time.time()
    o.init_cluster_cpp(Y0)
    it = iterate(o)
    ref.update(Z=o.getZcorr(), R=o.R, O=o.O, E=o.E, it=it, obj=o.objective_kmeans.copy(), s=time.time() - t0)


orc.use_openblas(4)
th = threading.Thread(target=cpu)
th.start()
runs = {}
for name in a.settings.split(","):
    o = Harmony(seed=seed, ref_arith=1)
    if name == "strict":
        o._set("seq_strict", 1)
    else:               # e.g. passes2, passes2tol10000 (seq_passes, seq_tol in parts per billion)
        import re
        m = re.match(r"passes(\d+)(?:tol(\d+))?$", name)
        if m:
            o._set("seq_passes", int(m.group(1)))
            if m.group(2):
                o._set("seq_tol_ppb", int(m.group(2)))
    o.setup(**skw)
    for rep in range(2):          # second run: warm allocations, the time that counts
        o.restart()
        t0 = time.time()
        o.init_cluster_cpp(Y0)
        it = iterate(o)
        Zc = o.getZcorr()
        dt = time.time() - t0
    runs[name] = dict(Z=Zc, R=o.R, O=o.O, E=o.E, it=it, obj=o.objective_kmeans.copy(), s=dt,
                      passes=o._get("seq:group_passes").tolist(), group_runs=o._get("seq:group_runs").tolist(),
                      residual=float(o._scalar("seq:residual")), mismatch=float(o._scalar("seq:mismatch")), unsettled=float(o._scalar("seq:unsettled")))
    del o
th.join()


def relfro(x, y):
    return float(np.linalg.norm(x - y) / np.linalg.norm(y))


out = {"cells": N, "oracle_faithful_s": ref["s"], "oracle_iterations": ref["it"], "settings": {}}
for name, r in runs.items():
    dR = np.abs(r["R"] - ref["R"])
    aa, ab = r["R"].argmax(axis=0), ref["R"].argmax(axis=0)
    bad = np.where(aa != ab)[0]
    srt = np.sort(ref["R"][:, bad], axis=0) if bad.size else np.zeros((2, 0))
    marg = srt[-1] - srt[-2] if bad.size else np.zeros(0)
    k, i = np.unravel_index(int(dR.argmax()), dR.shape)
    percell = dR.max(axis=0)
    n = min(len(r["obj"]), len(ref["obj"]))
    out["settings"][name] = {
        "seconds": r["s"], "iterations": r["it"], "passes_per_group_oe_obj_ridge_pairs": r["passes"], "runs_per_group": r["group_runs"],
        "seq_residual": r["residual"], "seq_mismatch": r["mismatch"], "seq_unsettled": r["unsettled"],
        "Z_rel": relfro(r["Z"], ref["Z"]), "R_maxabs": float(dR.max()), "O_rel": relfro(r["O"], ref["O"]), "E_rel": relfro(r["E"], ref["E"]),
        "O_maxabs": float(np.abs(r["O"] - ref["O"]).max()),
        "objective_rel_max": float(np.max(np.abs(r["obj"][:n] - ref["obj"][:n]) / np.abs(ref["obj"][:n]))),
        "argmax_diff": int(bad.size), "argmax_diff_margin_ge_1e-5": int((marg >= 1e-5).sum()), "argmax_diff_margin_ge_1e-4": int((marg >= 1e-4).sum()),
        "cells_with_dR_above": {t: int((percell > float(t)).sum()) for t in ("5e-5", "2e-5", "1e-5", "1e-6")},
        "dR_by_cluster_top5": [[int(c), float(v)] for c, v in sorted(enumerate(dR.max(axis=1)), key=lambda t: -t[1])[:5]],
        "largest": {"cell": int(i), "cluster": int(k), "level": int(meta["cov0"][i]), "R_gpu": float(r["R"][k, i]), "R_oracle": float(ref["R"][k, i]),
                    "top2_oracle": [float(v) for v in np.sort(ref["R"][:, i])[-2:]],
                    "O_row_gpu": r["O"][k].tolist(), "O_row_oracle": ref["O"][k].tolist(), "E_row_gpu": r["E"][k].tolist(), "E_row_oracle": ref["E"][k].tolist()},
    }
    # pairwise: how far apart the settings are from each other (is the distance to the oracle the settings' or the inputs'?)
for x in runs:
    for y in runs:
        if x < y:
            out["settings"]["%s_vs_%s" % (x, y)] = {"Z_rel": relfro(runs[x]["Z"], runs[y]["Z"]), "R_maxabs": float(np.abs(runs[x]["R"] - runs[y]["R"]).max())}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r5_strict_probe_%d.json" % N), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out))
