#!/usr/bin/env python
"""GPU: how far is the product's reference-arithmetic mode from the faithful oracle at its DEFAULT point (the orders the oracle header fixes) and
at the "OpenBLAS point" -- every Armadillo / BLAS liberty set to what a build on OpenBLAS 0.3.28 runs (oracle liberty bits 7, and 2 + 6 with several
covariates; distance GEMM through sgemm) --, same inputs, same centres, same shuffles.  Both oracle points equal the reference's own sources bit
for bit (tests/test_oracle_ref.py); the product is tuned to the default point, so the second distance should be of the size of the distance
BETWEEN the two points (profiles/r5_oracle_liberties.json).  Output: gpurun_out/r5z/gpu_vs_openblas_point.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harmony_amd import Harmony, prepare_setup_args  # noqa: E402
from helpers import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.oracle import OracleHarmony  # noqa: E402


def iterate(o, n=10):
    it = 0
    for it in range(1, n + 1):
        assert o.cluster_cpp() == 0
        o.moe_correct_ridge_cpp()
        if o.check_convergence(1):
            break
    return it


def dist(a, b):
    Za, Zb = a.getZcorr(), b.getZcorr()
    bad = np.where(a.R.argmax(axis=0) != b.R.argmax(axis=0))[0]
    srt = np.sort(b.R[:, bad], axis=0) if bad.size else np.zeros((2, 0))
    return {"Z_rel": float(np.linalg.norm(Za - Zb) / np.linalg.norm(Zb)), "R_maxabs": float(np.abs(a.R - b.R).max()), "flips": int(bad.size),
            "flips_margin_ge_1e-5": int(((srt[-1] - srt[-2]) >= 1e-5).sum()) if bad.size else 0}


out = {}
assert orc.use_lapack() and orc.use_openblas(1)
fx = np.load(os.path.join(ROOT, "tests", "golden", "cell_lines.npz"))
cases = [("100k_one_covariate_K100", synth(100000, d=50, levels=(10,), seed=7)[:2], 100, 10),
         ("cell_lines_two_crossed_covariates_K20", (fx["pcs"], {"dataset": fx["dataset_levels"][fx["dataset"]], "cell_type": fx["cell_type_levels"][fx["cell_type"]]}), 20, 4),
         ("40k_three_nested_covariates_K60", synth(40000, d=50, levels=(4, 12, 24), seed=5, nested=True)[:2], 60, 4)]
for name, (Z, meta), K, max_iter in cases:
    vu = list(meta)
    skw, _ = prepare_setup_args(Z, meta, vu, nclust=K)
    g = Harmony(seed=3, ref_arith=1)
    g.setup(**skw)
    Y0 = g.kmeans_centers()
    g.init_cluster_cpp(Y0)
    ig = iterate(g, max_iter)
    row = {"iterations_gpu": ig}
    objs = {}
    for tag, lib in (("oracle_default_point", 0), ("oracle_openblas_point", 128 | (4 | 64 if len(vu) > 1 else 0))):
        c = OracleHarmony(mask=0, seed=3, liberty=lib)
        c.setup(**skw)
        c.init_cluster_cpp(Y0)
        ic = iterate(c, max_iter)
        objs[tag] = c
        row["gpu_ref_arith_vs_" + tag] = dict(dist(g, c), iterations=[ig, ic])
    row["openblas_point_vs_default_point"] = dist(objs["oracle_openblas_point"], objs["oracle_default_point"])
    out[name] = row
    print(name, json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5z"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5z", "gpu_vs_openblas_point.json"), "w"), indent=1)
