import sys, time, json, os
sys.path.insert(0, "/root/repo")
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
for n, levels, K in ((100000, (10,), 100), (1000000, (10,), 100), (2000000, (10,), 100), (300000, (8, 64, 128), 200)):
    Z, meta, _ = synth(n, d=50, levels=levels, seed=7, nested=len(levels) > 1)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    res = {}
    for name, kw in (("full", dict(ref_arith=1)), ("no_oe", dict(obj_arith=1, ridge_arith=1, solve_arith=1)), ("obj_only", dict(obj_arith=1)), ("exact", dict())):
        o = Harmony(seed=1, **kw)
        o.setup(**skw)
        run_to_convergence(o)
        o._scalar("sync"); t0 = time.perf_counter()
        it = run_to_convergence(o)
        o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0)
        res[name] = (ms, it, o.getZcorr().copy(), o.getR().copy() if n <= 300000 else None, list(o.kmeans_rounds), np.array(o.objective_kmeans))
        del o
    f = res["full"]
    for name in ("no_oe", "obj_only", "exact"):
        r = res[name]
        zr = np.linalg.norm(r[2] - f[2]) / np.linalg.norm(f[2])
        m = min(len(r[5]), len(f[5]))
        msg = "n=%d K=%d %-8s %.1f ms (full %.1f) it %d/%d rounds_equal %s Z_rel_vs_full %.2e obj_rel %.2e" % (n, K, name, r[0], f[0], r[1], f[1], r[4] == f[4], zr, float(np.max(np.abs(r[5][:m] - f[5][:m]) / np.abs(f[5][:m]))))
        if r[3] is not None:
            fl = int(np.sum(np.argmax(r[3], 0) != np.argmax(f[3], 0)))
            msg += " argmax_flips %d" % fl
        print(msg, flush=True)
