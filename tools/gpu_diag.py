"""One-shot GPU diagnostic: stage-by-stage parity numbers (no asserts) + phase timings.
Usage on the GPU box:  python tools/gpu_diag.py [--big N]  > gpurun_out/diag.log"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from harmony_amd import Harmony, harmony_options, prepare_setup_args  # noqa: E402
from helpers import synth  # noqa: E402
from parity import compare_state, make_pair, relfro, run_both  # noqa: E402


def stage(name, fn):
    t = time.time()
    try:
        r = fn()
        print("[%s] %.2fs %s" % (name, time.time() - t, r), flush=True)
    except Exception:
        print("[%s] EXCEPTION\n%s" % (name, traceback.format_exc()), flush=True)


def fx(name):
    f = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    meta = {"dataset": f["dataset_levels"][f["dataset"]], "cell_type": f["cell_type_levels"][f["cell_type"]]}
    return f["pcs"], meta


def staged(Z, meta, vars_use, K, seed=1, **kw):
    g, c = make_pair(Z, meta, vars_use, nclust=K, seed=seed, **kw)
    out = {}
    out["Zorig_eq"] = bool(np.array_equal(g.getZorig(), c.getZorig()))
    out["Zcos_rel"] = relfro(g.getZcorr(), c.getZcorr())
    Y0 = np.asfortranarray(np.asarray(Z)[:K].T)
    g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
    out["init"] = compare_state(g, c, ("R", "O", "E", "Y", "obj"))
    g.cluster_cpp(); c.cluster_cpp()
    out["cluster1"] = compare_state(g, c, ("R", "O", "E", "obj"))
    g.moe_correct_ridge_cpp(); c.moe_correct_ridge_cpp()
    out["moe1"] = compare_state(g, c, ("Y", "Z"))
    out["W_rel"] = relfro(g.W, c.W) if g.W.shape == c.W.shape else (g.W.shape, c.W.shape)
    g.cluster_cpp(); c.cluster_cpp()
    out["cluster2"] = compare_state(g, c, ("R", "O", "E", "obj"))
    g.moe_correct_ridge_cpp(); c.moe_correct_ridge_cpp()
    out["moe2"] = compare_state(g, c, ("Y", "Z"))
    return "\n   " + "\n   ".join("%s: %s" % kv for kv in out.items())


def kmeans_check(Z, meta, K):
    g, c = make_pair(Z, meta, "dataset", nclust=K, seed=11)
    Yg = g.kmeans_centers()
    c.init_cluster_cpp()
    Yg_n = Yg / np.linalg.norm(Yg, axis=0, keepdims=True)
    return "Y_rel=%g" % relfro(Yg_n, c.Y)


def timing(N, K=100, B=10, iters=3):
    Z, meta, _ = synth(N, d=50, levels=(B,), seed=7)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
    g = Harmony(seed=3)
    t = time.time(); g.setup(**skw); t_setup = time.time() - t
    res = []
    for rep in range(2):
        g.restart()
        g.set_profile(2)
        t = time.time(); g.init_cluster_cpp(); t_init = time.time() - t
        tc = tm = 0.0
        nit = 0
        for it in range(iters):
            t = time.time(); g.cluster_cpp(); tc += time.time() - t
            t = time.time(); g.moe_correct_ridge_cpp(); tm += time.time() - t
            nit += 1
            if g.check_convergence(1):
                break
        ums, ul, uc = g._scalar("prof:update_ms"), g._scalar("prof:update_launches"), g._scalar("prof:update_cells")
        res.append(dict(init_s=round(t_init, 4), cluster_s=round(tc, 4), moe_s=round(tm, 4), iters=nit,
                        upd_ms_total=round(ums, 3), upd_launches=int(ul), upd_us_per_launch=round(1e3 * ums / max(ul, 1), 2),
                        upd_ns_per_cell=round(1e6 * ums / max(uc, 1), 3), moe_host_ms=round(g.timer("moe_solve_host"), 2),
                        obj=[round(float(x), 4) for x in g.objective_harmony]))
    return "setup=%.2fs %s" % (t_setup, res)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=1000000)
    a = ap.parse_args()
    Zs, ms = fx("cell_lines_small")
    Zl, ml = fx("cell_lines")
    stage("small K=10 1cov", lambda: staged(Zs, ms, "dataset", 10))
    stage("cell_lines K=50 1cov", lambda: staged(Zl, ml, "dataset", 50, theta=2))
    stage("cell_lines K=50 2cov", lambda: staged(Zl, ml, ["cell_type", "dataset"], 50, theta=[1, 1]))
    stage("kmeans small", lambda: kmeans_check(Zs, ms, 10))
    stage("kmeans cell_lines", lambda: kmeans_check(Zl, ml, 20))
    Zb, mb, _ = synth(20000, d=50, levels=(10,), seed=3)
    stage("synth 20k K=100", lambda: staged(Zb, mb, "cov0", 100))
    Zn, mn, _ = synth(6000, d=20, levels=(4, 8, 16), seed=5, nested=True)
    stage("nested 3cov K=40", lambda: staged(Zn, mn, list(mn), 40))
    Zd, md, _ = synth(4000, d=70, levels=(4,), seed=9)
    stage("d=70 K=130", lambda: staged(Zd, md, "cov0", 130))
    stage("timing %d" % a.big, lambda: timing(a.big))
