"""Stage-by-stage comparison of the HIP path (through the C ABI) with the CPU oracle on identical
inputs and identical random choices (shared initial centroids, shared block partitions)."""
import numpy as np

from harmony_amd import Harmony, prepare_setup_args
from oracle.oracle import OracleHarmony

# tolerances (written here once; quoted in DESIGN.md)
TOL_Z = 1e-4        # north_star: relative Frobenius norm of the corrected embedding
TOL_R = 5e-5        # max |R_gpu - R_cpu|
TOL_TAB = 1e-4      # O, E, Y relative (Frobenius)
TOL_OBJ = 1e-4      # objective series, relative
MARGIN = 1e-5       # a hard assignment may differ only where the oracle's top-2 margin is below this (the raw
                    # number of differing assignments is part of every report: "argmax_diff")


def relfro(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def argmax_mismatch(Rg, Rc):
    ag, ac = Rg.argmax(axis=0), Rc.argmax(axis=0)
    bad = np.where(ag != ac)[0]
    if bad.size == 0:
        return 0, 0
    srt = np.sort(Rc[:, bad], axis=0)
    margin = srt[-1] - srt[-2]
    return int(bad.size), int((margin >= MARGIN).sum())


def make_pair(Z, meta, vars_use, seed=1, accurate=True, **kw):
    skw, _ = prepare_setup_args(Z, meta, vars_use, **kw)
    g = Harmony(seed=seed)
    g.setup(**skw)
    c = OracleHarmony(accurate=accurate, seed=seed)
    c.setup(**skw)
    return g, c


def compare_state(g, c, what=("R", "O", "E", "Y")):
    out = {}
    if "R" in what:
        Rg, Rc = g.R, c.R
        out["R_maxabs"] = float(np.abs(Rg - Rc).max())
        out["R_colsum_err"] = float(np.abs(Rg.sum(axis=0) - 1).max())
        out["argmax_diff"], out["argmax_diff_clear"] = argmax_mismatch(Rg, Rc)
    if "O" in what:
        out["O_rel"] = relfro(g.O, c.O)
    if "E" in what:
        out["E_rel"] = relfro(g.E, c.E)
    if "Y" in what:
        out["Y_rel"] = relfro(g.Y, c.Y)
    if "L" in what:     # getLambda (src/harmony.cpp:657-669): K x (B + 1), column 0 = the intercept's 0
        out["Lambda_rel"] = relfro(g.getLambda(), c.getLambda())
    if "Z" in what:
        out["Z_rel"] = relfro(g.getZcorr(), c.getZcorr())
    if "obj" in what:
        a, b = g.objective_kmeans, c.objective_kmeans
        out["obj_len"] = (len(a), len(b))
        n = min(len(a), len(b))
        out["obj_rel"] = float(np.max(np.abs(a[:n] - b[:n]) / np.abs(b[:n]))) if n else 0.0
        for nm in ("objective_kmeans_dist", "objective_kmeans_entropy", "objective_kmeans_cross"):
            x, y = getattr(g, nm), getattr(c, nm)
            out[nm + "_rel"] = float(np.max(np.abs(x[:n] - y[:n]) / np.maximum(np.abs(y[:n]), 1e-6))) if n else 0.0
    return out


def run_both(Z, meta, vars_use, max_iter=10, seed=1, gpu_init=True, **kw):
    """Full RunHarmony on both backends with shared random choices; returns (gpu, cpu, n_iter_gpu, n_iter_cpu)."""
    g, c = make_pair(Z, meta, vars_use, seed=seed, **kw)
    if gpu_init:
        Y0 = g.kmeans_centers()
        g.init_cluster_cpp(Y0)
        c.init_cluster_cpp(Y0)
    else:
        c.init_cluster_cpp()
        Y0 = None
        g.init_cluster_cpp(c.Y)  # already normalised; normalising again is the identity up to 1 ulp
    iters = []
    for obj in (g, c):
        it = 0
        for it in range(1, max_iter + 1):
            assert obj.cluster_cpp() == 0
            obj.moe_correct_ridge_cpp()
            if obj.check_convergence(1):
                break
        iters.append(it)
    return g, c, iters[0], iters[1]


def assert_parity(g, c, ig=None, ic=None, tol_z=TOL_Z):
    s = compare_state(g, c, ("R", "O", "E", "Y", "Z", "obj", "L"))
    msg = repr(s)
    assert s["Z_rel"] <= tol_z, msg
    assert s["R_maxabs"] <= TOL_R, msg
    assert s["argmax_diff_clear"] == 0, msg
    assert s["O_rel"] <= TOL_TAB and s["E_rel"] <= TOL_TAB and s["Y_rel"] <= TOL_TAB, msg
    assert s["Lambda_rel"] <= TOL_TAB, msg
    assert s["obj_len"][0] == s["obj_len"][1], msg
    assert s["obj_rel"] <= TOL_OBJ, msg
    assert np.array_equal(g.kmeans_rounds, c.kmeans_rounds), (g.kmeans_rounds, c.kmeans_rounds)
    if ig is not None:
        assert ig == ic, (ig, ic)
    return s
