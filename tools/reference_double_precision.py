#!/usr/bin/env python
"""Is the parity target of the product's default mode (the oracle's ACCURATE mode: fp32 state, exact accumulators) the reference's algorithm?
CPU only.  The reference has its own precision switch (src/types.h:5-9, -DHARMONY_SCALAR_DOUBLE): oracle/_ref/libharmony_ref_f64.so is the
reference's own harmony.cpp / utils.cpp built that way over oracle/shim/, libharmony_ref.so the build as it ships.  Same inputs, same initial
centroids, same shuffles (injected into the reference's arma::shuffle), defaults, to convergence:

    accurate oracle   vs reference(double)     how far the parity target is from the reference's algorithm without fp32 rounding
    faithful oracle   vs reference(single)     (bit-identical on R's stream without shared centroids: tests/test_oracle_ref.py; here an ulp apart)
    reference(single) vs reference(double)     the reference's own fp32 bias -- what a "1e-4 against the reference" claim has to live with

    python tools/reference_double_precision.py CELLS CLUSTERS LEVELS[,LEVELS...] [nested]     -> merges into profiles/r5_reference_double_precision.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harmony_amd import prepare_setup_args  # noqa: E402
from helpers import synth  # noqa: E402
from oracle.oracle import OracleHarmony, feistel_order  # noqa: E402
from oracle.ref import RefHarmony  # noqa: E402

OUT = os.path.join(ROOT, "profiles", "r5_reference_double_precision.json")


def relfro(x, y):
    return float(np.linalg.norm(x - y) / np.linalg.norm(y))


def run(N, K, levels, seed=3, iters=10, nested=False):
    Z, meta, _ = synth(N, d=50, levels=levels, seed=7, nested=nested)
    skw, _ = prepare_setup_args(Z, meta, list(meta) if len(levels) > 1 else "cov0", nclust=K)
    o0 = OracleHarmony(mask=15, seed=seed)          # shared initial centroids: the oracle's own k-means (documented generator)
    o0.setup(**skw)
    o0.init_cluster_cpp()
    Y0 = o0.Y.copy()
    del o0
    make = {"accurate": lambda: OracleHarmony(mask=15, seed=seed), "faithful": lambda: OracleHarmony(mask=0, seed=seed),
            "reference_double": lambda: RefHarmony(seed=seed, double=True), "reference_single": lambda: RefHarmony(seed=seed)}
    res = {}
    for name, mk in make.items():
        t0 = time.time()
        o = mk()
        o.setup(**skw)
        o.init_cluster_cpp(Y0)
        done, it = 0, 0
        for it in range(1, iters + 1):
            if name.startswith("reference"):
                o.clear_update_orders()
                for r in range(4):
                    o.push_update_order(feistel_order(seed, done + r, N))
            assert o.cluster_cpp() == 0
            done = int(np.sum(o.kmeans_rounds))
            o.moe_correct_ridge_cpp()
            if o.check_convergence(1):
                break
        res[name] = dict(Z=o.getZcorr(), R=o.R, it=it, obj=o.objective_kmeans, s=time.time() - t0)
        del o

    def cmp(a, b):
        A, B = res[a], res[b]
        bad = np.where(A["R"].argmax(0) != B["R"].argmax(0))[0]
        srt = np.sort(B["R"][:, bad], axis=0) if bad.size else np.zeros((2, 0))
        n = min(len(A["obj"]), len(B["obj"]))
        return dict(Z_rel=relfro(A["Z"], B["Z"]), R_maxabs=float(np.abs(A["R"] - B["R"]).max()), argmax_diff=int(bad.size),
                    argmax_diff_margin_ge_1e_5=int(((srt[-1] - srt[-2]) >= 1e-5).sum()) if bad.size else 0, iterations=[A["it"], B["it"]],
                    objective_rel_max=float(np.max(np.abs(A["obj"][:n] - B["obj"][:n]) / np.abs(B["obj"][:n]))))
    out = {"workload": dict(cells=N, pcs=50, clusters=K, levels=list(levels), nested=nested, seed=seed), "seconds": {k: v["s"] for k, v in res.items()},
           "pairs": {"%s_vs_%s" % (a, b): cmp(a, b) for a, b in (("accurate", "reference_double"), ("faithful", "reference_single"),
                                                                 ("reference_single", "reference_double"), ("faithful", "accurate"))},
           "note": "the double-precision build keeps my_accu's float accumulator and the float objective series (src/utils.cpp:67-75, src/harmony.h:54): its "
                   "objective carries the fp32 summation error, hence objective_rel_max of accurate_vs_reference_double ~ that of faithful_vs_accurate"}
    return out


if __name__ == "__main__":
    N, K = int(sys.argv[1]), int(sys.argv[2])
    levels = tuple(int(x) for x in sys.argv[3].split(","))
    r = run(N, K, levels, nested=len(sys.argv) > 4)
    allr = json.load(open(OUT)) if os.path.exists(OUT) else {}
    allr["%dk_K%d_levels_%s" % (N // 1000, K, "_".join(str(x) for x in levels))] = r
    json.dump(allr, open(OUT, "w"), indent=1)
    print(json.dumps(r["pairs"], indent=1))
