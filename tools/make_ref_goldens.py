"""Golden OUTPUT vectors from the reference's own engine sources -> tests/golden/ref_sources_*.npz.

Run in the build container only (it needs /root/reference to build oracle/_ref):   python tools/make_ref_goldens.py

What produces them: oracle/_ref/libharmony_ref.so -- /root/reference/src/harmony.cpp, utils.cpp, timer.cpp, compiled where they lie and
unmodified, over oracle/shim/ (a from-scratch stand-in for the Armadillo / Rcpp headers; oracle/shim/arma_min.hpp states what it restates).
So these are the reference's control flow and expression order with Armadillo's kernels restated -- not outputs of the R package as CRAN
builds it (no R, no Armadillo, no BLAS in this image).  Inputs: the reference's bundled fixtures (tests/golden/cell_lines*.npz, dumped by
tools/make_golden.py) and RunHarmony's defaults; randomness: R's stream after set.seed(seed) (k-means++ race, Lloyd, one arma::shuffle per round).

Each file holds the parameters of the run and, after convergence (or max_iter): Z_corr, R, Y, O, E (fp32: the reference's precision, exact),
the four objective series, kmeans_rounds, the number of harmony iterations.  tests/test_oracle_ref.py::test_golden_vectors_* checks the
restated oracle against them bit for bit wherever the tree is (the library itself only exists where /root/reference does or travelled to).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harmony_amd import harmony_options, prepare_setup_args  # noqa: E402
from oracle import ref as oref  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = {   # name -> (fixture, vars_use, nclust, seed, max_iter, kwargs of RunHarmony)
    # (these fixtures converge after one iteration under the default epsilon.harmony: early_stop = FALSE walks max_iter of them)
    "ref_sources_cell_lines_small": ("cell_lines_small", ["dataset"], 10, 1, 3, dict(early_stop=False)),
    "ref_sources_cell_lines_small_test_integration": ("cell_lines_small", ["dataset"], 50, 1, 3, dict(theta=1, max_iter_cluster=10, early_stop=False)),   # tests/testthat/test_integration.R:5-7
    "ref_sources_cell_lines_two_covariates": ("cell_lines", ["cell_type", "dataset"], 20, 1, 3, dict(theta=[1, 1], max_iter_cluster=10, early_stop=False)),   # test_two_variable.R:5-11 with 20 clusters
}


def setup_kwargs(case):
    fixture, vars_use, nclust, seed, max_iter, kw = CASES[case]
    fx = np.load(os.path.join(GOLDEN, fixture + ".npz"), allow_pickle=False)
    meta = {"dataset": fx["dataset_levels"][fx["dataset"]], "cell_type": fx["cell_type_levels"][fx["cell_type"]]}
    kw = dict(kw)
    opts = harmony_options(**({"max_iter_cluster": kw.pop("max_iter_cluster")} if "max_iter_cluster" in kw else {}))
    skw, _ = prepare_setup_args(fx["pcs"], meta, vars_use if len(vars_use) > 1 else vars_use[0], nclust=nclust, options=opts, **kw)
    return skw, seed, max_iter


def walk(obj, max_iter):
    """harmonize (R/utils.R:15-46) on an object that is set up and initialised"""
    it = 0
    for it in range(1, max_iter + 1):
        assert obj.cluster_cpp() == 0
        obj.moe_correct_ridge_cpp()
        if obj.check_convergence(1):
            break
    return it


def main():
    for case in CASES:
        skw, seed, max_iter = setup_kwargs(case)
        r = oref.RefHarmony(seed=seed)
        r.setup(**skw)
        r.init_cluster_cpp()
        it = walk(r, max_iter)
        f32 = lambda a: np.asarray(a, dtype=np.float32)   # noqa: E731  (every value is an fp32 of the reference: exact)
        for name in ("R", "Y", "O", "E"):
            assert np.array_equal(f32(getattr(r, name)).astype(np.float64), getattr(r, name))
        np.savez_compressed(os.path.join(GOLDEN, case + ".npz"), case=json.dumps(dict(zip(("fixture", "vars_use", "nclust", "seed", "max_iter", "kwargs"), CASES[case]))),
                            Z_corr=f32(r.getZcorr()), R=f32(r.R), Y=f32(r.Y), O=f32(r.O), E=f32(r.E), Lambda=f32(r.getLambda()),
                            objective_kmeans=f32(r.objective_kmeans), objective_kmeans_dist=f32(r.objective_kmeans_dist),
                            objective_kmeans_entropy=f32(r.objective_kmeans_entropy), objective_kmeans_cross=f32(r.objective_kmeans_cross),
                            objective_harmony=f32(r.objective_harmony), kmeans_rounds=r.kmeans_rounds.astype(np.int32), iterations=np.int32(it))
        print(case, "iterations", it, "rounds", r.kmeans_rounds.tolist(), "%.1f kB" % (os.path.getsize(os.path.join(GOLDEN, case + ".npz")) / 1e3))


if __name__ == "__main__":
    main()
