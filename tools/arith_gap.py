#!/usr/bin/env python
"""Attribution of the gap between the reference's fp32 arithmetic (oracle *faithful* mode) and the fp64-accumulator
restatement (oracle *accurate* mode, the GPU parity target).  CPU only (no GPU, no product code).

For one synthetic workload (bench_data.synth, reference defaults, shared initial centroids and shared block partitions)
it runs the oracle to convergence in several arithmetic variants and prints / saves one row per variant:

  F          faithful: every accumulator in fp32, in the reference's order (mask 0), naive k-ordered GEMM
  F_blas     F with the K x d x N GEMM through OpenBLAS sgemm (blocked / FMA order)  -- the reference on another BLAS
  F_perm     F on the SAME problem with the cells stored in a different order (inputs permuted, identical block
             membership injected) -- the reference on the same data listed in another order
  F+oe, F+obj, F+stat, F+solve   F with ONE accumulator group switched to fp64 (masks 1, 2, 4, 8)
  A          accurate: all four groups in fp64 (mask 15)
  A-oe, ...  A with ONE group switched back to fp32 (masks 14, 13, 11, 7)

Columns: rel. Frobenius distance of Z_corr to F and to A, hard-assignment flips vs F / vs A (raw count and the count
with a top-2 margin >= 1e-5 in the comparison target), harmony iterations, final objective.

    python tools/arith_gap.py --cells 100000 [--out profiles/r2_arith_gap_100k.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench_data import synth  # noqa: E402
from harmony_amd.ui import prepare_setup_args  # noqa: E402  (argument prep only: pure numpy, no device code)
from oracle import oracle as orc  # noqa: E402


def run(skw, Y0, orders, mask, max_iter=10):
    o = orc.OracleHarmony(mask=mask, seed=1)
    o.setup(**skw)
    o.init_cluster_cpp(Y0)
    it = 0
    k = 0
    for it in range(1, max_iter + 1):
        for _ in range(skw["max_iter_kmeans"]):
            o.push_update_order(orders[k % len(orders)] if orders is not None else None)
            k += 1
        o.cluster_cpp()
        o.moe_correct_ridge_cpp()
        if o.check_convergence(1):
            break
    return {"Z": o.getZcorr(), "R": o.R, "it": it, "obj": o.objective_harmony.copy(), "rounds": o.kmeans_rounds.copy()}


def relfro(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def flips(Ra, Rb, margin=1e-5):
    aa, ab = Ra.argmax(axis=0), Rb.argmax(axis=0)
    bad = np.where(aa != ab)[0]
    if bad.size == 0:
        return 0, 0
    srt = np.sort(Rb[:, bad], axis=0)
    return int(bad.size), int(((srt[-1] - srt[-2]) >= margin).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=100000)
    ap.add_argument("--pcs", type=int, default=50)
    ap.add_argument("--clusters", type=int, default=100)
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--variants", default="F,F_blas,F_perm,F+oe,F+obj,F+stat,F+solve,A,A-oe,A-obj,A-stat,A-solve")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    N, d, K, B = a.cells, a.pcs, a.clusters, a.batches
    Z, meta, _ = synth(N, d=d, levels=(B,), seed=a.seed)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    rng = np.random.default_rng(12345)
    # shared random choices: initial centroids = K distinct cells (normalised by init_cluster), 64 block partitions
    Zn = np.asarray(skw["Z"], dtype=np.float64)
    Zn = Zn / np.linalg.norm(Zn, axis=0, keepdims=True)
    Y0 = Zn[:, rng.choice(N, size=K, replace=False)].copy()
    n_orders = 40
    orders = [rng.permutation(N).astype(np.int64) for _ in range(n_orders)]
    masks = {"F": 0, "F+oe": 1, "F+obj": 2, "F+stat": 4, "F+solve": 8, "A": 15, "A-oe": 14, "A-obj": 13, "A-stat": 11,
             "A-solve": 7}
    res = {}
    for v in a.variants.split(","):
        t0 = time.time()
        if v == "F_blas":
            assert orc.use_openblas(1)
            res[v] = run(skw, Y0, orders, 0)
            orc.load().orc_set_sgemm(None)
        elif v == "F_perm":
            # the same cells listed in another order: column p of the permuted problem is cell pi[p]
            pi = rng.permutation(N)
            inv = np.empty(N, dtype=np.int64)
            inv[pi] = np.arange(N)
            meta_p = {k_: np.asarray(x)[pi] for k_, x in meta.items()}
            skw_p, _ = prepare_setup_args(Z[pi], meta_p, list(meta), nclust=K)
            orders_p = [inv[o] for o in orders]           # position p of a shuffle holds the same CELL as before
            r = run(skw_p, Y0, orders_p, 0)
            r["Z"] = r["Z"][:, inv]
            r["R"] = r["R"][:, inv]
            res[v] = r
        else:
            res[v] = run(skw, Y0, orders, masks[v])
        res[v]["sec"] = time.time() - t0
        print("ran %-8s %6.1f s  iterations %d" % (v, res[v]["sec"], res[v]["it"]), flush=True)
    rows = []
    for v, r in res.items():
        row = {"variant": v, "iterations": int(r["it"]), "objective": float(r["obj"][-1]), "seconds": round(r["sec"], 1)}
        for ref in ("F", "A"):
            if ref in res:
                row["Z_rel_vs_" + ref] = relfro(r["Z"], res[ref]["Z"])
                f, fc = flips(r["R"], res[ref]["R"])
                row["flips_vs_" + ref] = f
                row["flips_clear_vs_" + ref] = fc
        rows.append(row)
    hdr = ["variant", "iterations", "objective", "Z_rel_vs_F", "flips_vs_F", "flips_clear_vs_F", "Z_rel_vs_A", "flips_vs_A",
           "flips_clear_vs_A", "seconds"]
    print(" | ".join(hdr))
    for row in rows:
        print(" | ".join(("%.3g" % row[h]) if isinstance(row.get(h), float) else str(row.get(h, "-")) for h in hdr))
    out = {"workload": {"cells": N, "pcs": d, "clusters": K, "batches": B, "seed": a.seed}, "rows": rows}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
