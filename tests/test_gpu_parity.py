"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded inputs -- the reference's fixtures, multi-covariate / subset-path /
ragged-shape synthetic cases -- plus the reference's own test invariants on the GPU backend and
size-independent properties at BASELINE's full single-GPU size."""
import os
import subprocess
import sys

import numpy as np
import pytest

from harmony_amd import Harmony, RunHarmony, harmony_options, prepare_setup_args
from helpers import chi2, run_backend, synth
from oracle.oracle import OracleHarmony, feistel_order
from parity import TOL_R, TOL_TAB, TOL_Z, assert_parity, compare_state, make_pair, relfro, run_both

pytestmark = pytest.mark.gpu


def _meta(fx):
    return {"dataset": fx["dataset_levels"][fx["dataset"]], "cell_type": fx["cell_type_levels"][fx["cell_type"]]}


# ---------------------------------------------------------------- stage-by-stage
def test_ingest_roundtrip_and_normalise(cell_lines_small):
    g, c = make_pair(cell_lines_small["pcs"], _meta(cell_lines_small), "dataset", nclust=10)
    np.testing.assert_array_equal(g.getZorig(), c.getZorig())           # double -> fp32 -> double, un-permuted
    assert relfro(g.getZcorr(), c.getZcorr()) < 1e-6                    # Z_corr = normalise(Z_orig)
    np.testing.assert_allclose(g.Pr_b, c.Pr_b, rtol=1e-7)


def test_init_cluster_stage(cell_lines):
    g, c = make_pair(cell_lines["pcs"], _meta(cell_lines), "dataset", nclust=50, theta=2)
    Y0 = np.asfortranarray(cell_lines["pcs"][:50].T)
    g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
    s = compare_state(g, c, ("R", "O", "E", "Y", "obj"))
    assert s["R_maxabs"] < TOL_R and s["argmax_diff_clear"] == 0, s
    assert s["O_rel"] < 1e-5 and s["E_rel"] < 1e-5 and s["obj_rel"] < 1e-5, s
    assert s["objective_kmeans_cross_rel"] < 1e-4, s


def test_one_cluster_call_then_correction(cell_lines):
    g, c = make_pair(cell_lines["pcs"], _meta(cell_lines), "dataset", nclust=30, theta=2, seed=5)
    Y0 = np.asfortranarray(cell_lines["pcs"][100:130].T)
    g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
    assert g.cluster_cpp() == 0 and c.cluster_cpp() == 0
    s = compare_state(g, c, ("R", "O", "E", "obj"))
    assert s["R_maxabs"] < TOL_R and s["argmax_diff_clear"] == 0 and s["O_rel"] < TOL_TAB and s["obj_rel"] < 1e-4, s
    g.moe_correct_ridge_cpp(); c.moe_correct_ridge_cpp()
    s = compare_state(g, c, ("Y", "Z"))
    assert s["Z_rel"] < TOL_Z and s["Y_rel"] < TOL_TAB, s
    assert relfro(g.W, c.W) < 1e-3
    # second iteration exercises the cold-start head of cluster_cpp (src/harmony.cpp:214-228)
    assert g.cluster_cpp() == 0 and c.cluster_cpp() == 0
    s = compare_state(g, c, ("R", "O", "E", "obj"))
    assert s["R_maxabs"] < TOL_R and s["argmax_diff_clear"] == 0 and s["obj_rel"] < 1e-4, s


def test_injected_update_order(cell_lines_small):
    g, c = make_pair(cell_lines_small["pcs"], _meta(cell_lines_small), "dataset", nclust=10, seed=3)
    Y0 = np.asfortranarray(cell_lines_small["pcs"][:10].T)
    g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
    rng = np.random.default_rng(0)
    for _ in range(4):
        o = rng.permutation(300)
        g.push_update_order(o); c.push_update_order(o)
    assert g.cluster_cpp() == 0 and c.cluster_cpp() == 0
    s = compare_state(g, c, ("R", "O", "E"))
    assert s["R_maxabs"] < TOL_R and s["argmax_diff_clear"] == 0 and s["O_rel"] < TOL_TAB, s


def test_kmeans_centers_vs_oracle(cell_lines):
    g, c = make_pair(cell_lines["pcs"], _meta(cell_lines), "dataset", nclust=20, seed=11)
    Yg = g.kmeans_centers()
    c.init_cluster_cpp()   # oracle: kmeans_centers + normalise
    Yg_n = Yg / np.linalg.norm(Yg, axis=0, keepdims=True)
    assert relfro(Yg_n, c.Y) < 1e-4, relfro(Yg_n, c.Y)


# ---------------------------------------------------------------- end-to-end on the reference's fixtures
def test_cell_lines_small_full_run(cell_lines_small):
    # tests/testthat/test_integration.R:5-7
    g, c, ig, ic = run_both(cell_lines_small["pcs"], _meta(cell_lines_small), "dataset", max_iter=5, theta=1, nclust=50,
                            options=harmony_options(max_iter_cluster=10))
    assert_parity(g, c, ig, ic)


def test_cell_lines_two_covariates_full_run(cell_lines):
    # tests/testthat/test_two_variable.R:5-11 -> multi-covariate ridge (arma::inv branch)
    g, c, ig, ic = run_both(cell_lines["pcs"], _meta(cell_lines), ["cell_type", "dataset"], max_iter=10, theta=[1, 1],
                            nclust=50, options=harmony_options(max_iter_cluster=10))
    assert_parity(g, c, ig, ic)


def test_pbmc_default_run(pbmc):
    meta = {"stim": pbmc["stim_levels"][pbmc["stim"]]}
    g, c, ig, ic = run_both(pbmc["pcs"].astype(np.float64), meta, "stim", nclust=50)
    assert_parity(g, c, ig, ic)


def test_pbmc_30k_default_run():
    """BASELINE configs[1] at its stated size (~30k cells x 50 PCs, K = 50, stim / ctrl): the shipped 2 000-cell sample resampled to 30k
    cells (bench_data.pbmc30k), reference defaults, to convergence, against the oracle; bench.py times the same workload (`also.pbmc30k`)"""
    from bench_data import pbmc30k
    Z, meta = pbmc30k()
    g, c, ig, ic = run_both(Z, meta, "stim", nclust=50)
    s = assert_parity(g, c, ig, ic)
    print("pbmc 30k:", s, "iterations", ig)


def test_fixed_lambda_tau_vector_sigma(cell_lines):
    K = 12
    g, c, ig, ic = run_both(cell_lines["pcs"], _meta(cell_lines), ["dataset", "cell_type"], max_iter=3, nclust=K,
                            lambda_=[0.5, 2.0], theta=[2, 1], sigma=np.linspace(0.08, 0.15, K),
                            options=harmony_options(tau=5, block_size=0.1, max_iter_cluster=6))
    assert_parity(g, c, ig, ic)


def test_getLambda_against_oracle(cell_lines):
    """getLambda (src/harmony.cpp:657-669): K x (B + 1); estimated lambda = find_lambda_cpp(alpha, E[k, :]) = [0, alpha * E[k, :]]
    (src/utils.cpp:159-163) on the CURRENT E -- compared after the head and after every correction --, fixed lambda = the caller's vector in
    every row (R/ui.R:235-248: c(0, rep(lambda_c, B_c)...))."""
    K = 15
    g, c = make_pair(cell_lines["pcs"], _meta(cell_lines), ["dataset", "cell_type"], nclust=K)
    Y0 = g.kmeans_centers()
    g.init_cluster_cpp(Y0)
    c.init_cluster_cpp(Y0)
    for it in range(3):
        Lg, Lc = g.getLambda(), c.getLambda()
        assert Lg.shape == (K, g.B + 1) == Lc.shape
        assert np.all(Lg[:, 0] == 0) and np.all(Lc[:, 0] == 0)
        np.testing.assert_allclose(Lg, Lc, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(Lg[:, 1:], np.float32(g.alpha) * g.E.astype(np.float32), rtol=1e-6)      # the definition, on the GPU's own E
        for o in (g, c):
            assert o.cluster_cpp() == 0
            o.moe_correct_ridge_cpp()
    lam = [0.5, 2.0]
    g, c, ig, ic = run_both(cell_lines["pcs"], _meta(cell_lines), ["dataset", "cell_type"], max_iter=1, nclust=K, lambda_=lam)
    Lg, Lc = g.getLambda(), c.getLambda()
    Bv = g.B_vec
    want = np.concatenate([[0.0], np.repeat(lam, Bv)])
    assert np.array_equal(Lg, np.tile(want, (K, 1))) and np.array_equal(Lc, Lg)


@pytest.mark.parametrize("N,d,K,levels,nested", [
    (5000, 50, 100, (10,), False),      # BASELINE shape, two clusters per lane
    (3000, 17, 7, (3,), False),         # ragged small shapes
    (4000, 70, 130, (4,), False),       # d > 64 and three clusters per lane
    (6000, 30, 200, (5, 7), False),     # crossed covariates, four clusters per lane
    (6000, 20, 40, (4, 8, 16), True),   # nested 3-covariate: batch-subset path (src/harmony.cpp:440-547)
    (2011, 3, 5, (2,), False),          # d < 4 (one partial MFMA k-step), K < 16 (one partial cluster tile), ragged N
    (4099, 64, 16, (3,), False),        # d and K exact multiples of the tile sizes, prime N
])
def test_synthetic_shapes(N, d, K, levels, nested):
    Z, meta, _ = synth(N, d=d, levels=levels, seed=N, nested=nested)
    g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=3, nclust=K, seed=N)
    assert_parity(g, c, ig, ic)
    if nested:
        assert c.subset_clusters > 0 and int(g._scalar("subset_clusters")) == c.subset_clusters


def test_single_cluster(cell_lines_small):
    """nclust = 1 (R/ui.R:192-194 allows it): R == 1 everywhere, one ridge system"""
    g, c, ig, ic = run_both(cell_lines_small["pcs"], _meta(cell_lines_small), "dataset", max_iter=3, nclust=1, seed=2)
    assert_parity(g, c, ig, ic)
    assert np.allclose(g.R, 1.0)


def test_stand_alone_compute_objective_after_a_correction(cell_lines):
    """harmony::compute_objective called between moe_correct_ridge_cpp and the next cluster_cpp reads the reference's STORED dist_mat
    (src/harmony.cpp:160): with "stale_dist" the library reproduces that value; without, it documents a deviation (recomputed distances)"""
    skw, _ = prepare_setup_args(cell_lines["pcs"], _meta(cell_lines), "dataset", nclust=20)
    Y0 = np.asfortranarray(cell_lines["pcs"][:20].T)
    vals = {}
    for name, obj in (("stale", Harmony(seed=1, stale_dist=1)), ("fresh", Harmony(seed=1)), ("oracle", OracleHarmony(accurate=True, seed=1))):
        obj.setup(**skw)
        obj.init_cluster_cpp(Y0)
        assert obj.cluster_cpp() == 0
        obj.moe_correct_ridge_cpp()
        obj.compute_objective()
        vals[name] = np.array(obj.objective_kmeans)
    assert len(vals["stale"]) == len(vals["oracle"])
    np.testing.assert_allclose(vals["stale"], vals["oracle"], rtol=1e-5)
    assert abs(vals["fresh"][-1] - vals["oracle"][-1]) > 1e-3 * abs(vals["oracle"][-1])      # (the documented deviation is real)


def test_stand_alone_compute_objective_in_reference_arithmetic(cell_lines):
    """the same stand-alone call with ref_arith = 1: obj_arith re-derives the three term matrices -- they must come from
    the stale snapshot too, not from the corrected Z_corr / new Y -- and is compared with the FAITHFUL oracle"""
    skw, _ = prepare_setup_args(cell_lines["pcs"], _meta(cell_lines), "dataset", nclust=20)
    Y0 = np.asfortranarray(cell_lines["pcs"][:20].T)
    vals = {}
    for name, obj in (("stale", Harmony(seed=1, stale_dist=1, ref_arith=1)), ("oracle", OracleHarmony(accurate=False, seed=1))):
        obj.setup(**skw)
        obj.init_cluster_cpp(Y0)
        assert obj.cluster_cpp() == 0
        obj.moe_correct_ridge_cpp()
        obj.compute_objective()
        vals[name] = [np.array(getattr(obj, nm)) for nm in ("objective_kmeans", "objective_kmeans_dist", "objective_kmeans_entropy", "objective_kmeans_cross")]
    for a, b in zip(vals["stale"], vals["oracle"]):
        assert len(a) == len(b)
        np.testing.assert_allclose(a, b, rtol=2e-5)


def test_tiny_N_block_size_warning():
    rng = np.random.default_rng(1)
    Z = rng.normal(size=(30, 8)); meta = {"b": np.arange(30) % 2}
    with pytest.warns(UserWarning, match="Too few cells"):
        g, c, ig, ic = run_both(Z, meta, "b", max_iter=2, nclust=3)
    assert_parity(g, c, ig, ic)
    with pytest.raises(Exception, match="less than 6 cells"):
        RunHarmony(Z[:5], meta["b"][:5], verbose=False, nclust=2)


def test_100k_cells_against_oracle():
    Z, meta, _ = synth(100000, d=50, levels=(10,), seed=42)
    g, c, ig, ic = run_both(Z, meta, "cov0", max_iter=2, nclust=100, seed=42)
    s = assert_parity(g, c, ig, ic)
    print("100k parity:", s)


# ---------------------------------------------------------------- the reference's own invariants, GPU backend
def test_reference_invariants_on_gpu(cell_lines_small):
    m = _meta(cell_lines_small)
    obj = RunHarmony(cell_lines_small["pcs"], m, "dataset", theta=1, nclust=50, max_iter=5, return_object=True,
                     verbose=False, options=harmony_options(max_iter_cluster=10), seed=1)
    assert obj.Y.shape == (obj.d, obj.K) and obj.getZcorr().shape == (obj.d, obj.N) and obj.R.shape == (obj.K, obj.N)
    R = obj.R
    assert R.min() >= 0 and R.max() <= 1
    np.testing.assert_allclose(R.sum(axis=0), 1.0, atol=1e-5)
    assert np.all(np.isfinite(obj.getZcorr()))
    o0 = RunHarmony(cell_lines_small["pcs"], m, "dataset", theta=0, nclust=20, max_iter=2, return_object=True, verbose=False, seed=1)
    o1 = RunHarmony(cell_lines_small["pcs"], m, "dataset", theta=1, nclust=5, max_iter=2, return_object=True, verbose=False, seed=1)
    assert chi2(o0) > chi2(o1)
    out = RunHarmony(cell_lines_small["pcs"], m, "dataset", verbose=False)
    assert out.shape == (300, 20)      # t(Z_corr): cells x PCs, like the input


def test_reentrancy_and_max_iter_kmeans_field(cell_lines_small):
    m = _meta(cell_lines_small)
    obj = RunHarmony(cell_lines_small["pcs"], m, "dataset", nclust=10, max_iter=1, return_object=True, verbose=False, seed=2)
    obj.max_iter_kmeans = 2                      # vignettes/detailedWalkthrough.Rmd:364
    n0 = len(obj.objective_kmeans)
    assert obj.cluster_cpp() == 0
    assert len(obj.objective_kmeans) == n0 + 2 and obj.kmeans_rounds[-1] == 2
    obj.compute_objective()
    ok = obj.objective_kmeans
    assert abs(ok[-1] - ok[-2]) / abs(ok[-2]) < 1e-5   # recomputed objective == fused objective of the last round
    polled = []
    obj.set_abort_poll(lambda: polled.append(1) or True)
    assert obj.cluster_cpp() == -1 and polled          # Progress::check_abort -> -1 (src/harmony.cpp:233-234)


# ---------------------------------------------------------------- size-independent properties at full size
def test_full_size_properties():
    N, K, B = 1000000, 100, 10
    Z, meta, _ = synth(N, d=50, levels=(B,), seed=7)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
    g = Harmony(seed=3)
    g.setup(**skw)
    g.init_cluster_cpp()
    assert g.cluster_cpp() == 0
    O, E = g.O, g.E
    N_b = np.bincount(meta["cov0"], minlength=B)
    np.testing.assert_allclose(O.sum(axis=0), N_b, rtol=1e-6)            # sum_k O[k,b] = N_b  (R columns sum to 1)
    np.testing.assert_allclose(E.sum(axis=0), N_b, rtol=1e-6)
    np.testing.assert_allclose(E, O.sum(axis=1, keepdims=True) * (N_b / N)[None, :], rtol=1e-5)
    R = g.R
    assert R.min() >= 0 and np.abs(R.sum(axis=0) - 1).max() < 1e-5
    # O is the exact sum of R over each level's cells (fixed-point accumulation == direct fp64 sum)
    Odirect = np.stack([R[:, meta["cov0"] == b].sum(axis=1) for b in range(B)], axis=1)
    np.testing.assert_allclose(O, Odirect, rtol=2e-6, atol=1e-3)
    g.moe_correct_ridge_cpp()
    Zc = g.getZcorr()
    assert np.all(np.isfinite(Zc)) and Zc.shape == (50, N)
    # determinism: restart + identical seed => bit-identical corrected embedding and objective series
    obj1 = g.objective_kmeans.copy()
    g.restart(); g.init_cluster_cpp(); assert g.cluster_cpp() == 0; g.moe_correct_ridge_cpp()
    np.testing.assert_array_equal(g.objective_kmeans, obj1)
    np.testing.assert_array_equal(g.getZcorr(), Zc)     # ridge statistics are reduced in a fixed order: no fp atomics anywhere


# ---------------------------------------------------------------- multi-GPU path on one GPU: virtual shards
def _run_sharded(Z, meta, G, K, seed, max_iter):
    """G handles (one per thread) each holding a contiguous shard; the all-reduce hook sums the shards' device
    buffers.  This is the production sharded code path with the collective replaced by a thread rendezvous
    (device buffers are summed on the host through the same HIP runtime the library links)."""
    import ctypes
    import threading
    from harmony_amd.dist import shard_bounds
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    N = Z.shape[0]
    bounds = shard_bounds(N, G)
    barrier = threading.Barrier(G)
    slots, results, errors = [None] * G, [None] * G, []
    N_b = np.bincount(meta["cov0"]).astype(float)

    def hook_for(rank):
        def hook(user, buf, count, dtype, stream):
            assert hip.hipDeviceSynchronize() == 0
            host = np.empty(count, dtype=np.float64 if dtype == 1 else np.int64)
            assert hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) == 0   # D2H
            slots[rank] = host
            barrier.wait()
            st = np.stack(slots)
            red = st.min(axis=0) if dtype == 2 else st.sum(axis=0)
            barrier.wait()
            assert hip.hipMemcpy(buf, red.ctypes.data, red.nbytes, 1) == 0     # H2D
            barrier.wait()
            return 0
        return hook

    def work(rank):
        try:
            lo, hi = bounds[rank]
            m = {k: v[lo:hi] for k, v in meta.items()}
            # factor levels must be global: build Phi from the global level set
            skw, _ = prepare_setup_args(Z[lo:hi], m, "cov0", nclust=K, N_b=N_b, levels={"cov0": np.arange(len(N_b))})
            g = Harmony(seed=seed)
            g.set_shard(rank, G, lo, N, hook_for(rank))
            g.setup(**skw)
            g.init_cluster_cpp()
            it = 0
            for it in range(1, max_iter + 1):
                assert g.cluster_cpp() == 0
                g.moe_correct_ridge_cpp()
                if g.check_convergence(1):
                    break
            results[rank] = (g.getZcorr(), g.O, g.objective_kmeans, it, g.R.argmax(axis=0))
        except Exception as e:  # pragma: no cover
            errors.append(e)
            barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errors:
        raise errors[0]
    return results


@pytest.mark.parametrize("G,fused,carry", [(2, "1", "0"), (3, "1", "0"), (2, "0", "0"), (2, "1", "1"), (3, "0", "1")])
def test_virtual_shards_equal_single_shard(G, fused, carry, monkeypatch):
    """fused = "1": block steps = update kernel (fold in its prologue) + in-place all-reduce of the replica set (default);
    "0": the step-by-step sharded path (k_fold, all-reduce of one table, k_foldpen, update).  carry = "1": the old contributions
    of a round's blocks come from the previous round's tile kernels (every shard files its cells' sums, the table is all-reduced
    at the start of the round) instead of a pass over R -- forced on at this size, for the sharded runs and the single-shard one."""
    monkeypatch.setenv("HMX_FUSED_FOLD", fused)
    monkeypatch.setenv("HMX_SOLD_CARRY", carry)
    Z, meta, _ = synth(30000, d=50, levels=(10,), seed=21)
    K, seed = 100, 4
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
    one = Harmony(seed=seed)
    one.setup(**skw)
    one.init_cluster_cpp()
    it1 = 0
    for it1 in range(1, 4):
        assert one.cluster_cpp() == 0
        one.moe_correct_ridge_cpp()
        if one.check_convergence(1):
            break
    res = _run_sharded(Z, meta, G, K, seed, 3)
    Zs = np.concatenate([r[0] for r in res], axis=1)
    assert all(r[3] == it1 for r in res)
    for r in res:                                              # replicated tables are identical on every shard
        np.testing.assert_array_equal(r[1], res[0][1])
        np.testing.assert_allclose(r[2], res[0][2], rtol=0, atol=0)
    np.testing.assert_allclose(res[0][1], one.O, rtol=1e-6, atol=1e-4)   # exact integer sums => shard-count independent
    np.testing.assert_allclose(res[0][2], one.objective_kmeans, rtol=1e-6)
    assert relfro(Zs, one.getZcorr()) < 1e-6
    assert np.array_equal(np.concatenate([r[4] for r in res]), one.R.argmax(axis=0))


@pytest.mark.parametrize("mode", ["chain", "steps", "sharded"])
def test_elided_R_stores_change_nothing(mode, monkeypatch):
    """R rows that nobody reads are not stored (the head inside cluster_cpp and every round that cannot be a call's last keep them in
    registers; the next round takes its old contributions from the sums the tile kernels filed).  Default against HMX_R_STORE=1 (every
    pass stores): R, O, the objective series and Z_corr must be BIT-identical -- on the persistent chain, on the launch-per-step kernels
    and on two virtual shards -- and the default must really have elided rounds ("rounds_without_R")."""
    monkeypatch.setenv("HMX_SOLD_CARRY", "1")          # (the carry -- and with it the elision -- starts at ~800k cells by default)
    Z, meta, _ = synth(30000, d=50, levels=(10,), seed=23)
    K, seed = 100, 9
    out = []
    for store in ("0", "1"):
        monkeypatch.setenv("HMX_R_STORE", store)
        if mode == "sharded":
            res = _run_sharded(Z, meta, 2, K, seed, 3)
            out.append((np.concatenate([r[0] for r in res], axis=1), res[0][1], res[0][2], np.concatenate([r[4] for r in res]), None))
            continue
        monkeypatch.setenv("HMX_CHAIN", "1" if mode == "chain" else "0")
        skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
        g = Harmony(seed=seed)
        g.setup(**skw)
        g.init_cluster_cpp()
        for it in range(3):
            assert g.cluster_cpp() == 0
            g.moe_correct_ridge_cpp()
        assert g.cluster_cpp() == 0
        assert int(g._scalar("chain")) == (1 if mode == "chain" else 0)
        assert (g._scalar("rounds_without_R") > 0) == (store == "0")
        out.append((g.getZcorr(), g.O, g.objective_kmeans, g.R, int(g._scalar("rounds_without_R"))))
    a, b = out
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])
    np.testing.assert_array_equal(a[0], b[0])


def test_objective_in_reference_arithmetic_never_reads_elided_rows(monkeypatch):
    """obj_arith alone (exact O / E tables, so the round-to-round carry stays on): the round's objective is summed from the R rows
    themselves (src/utils.cpp:67-75 over R % dist, R % log R, ...), so a round must store them even where the next round would not read
    them (the elision of hmx_api_update.inc ignored obj_arith and the objective of a call's first rounds came from stale rows).
    Default against HMX_R_STORE=1: the objective series must be bit-identical, and no round may have elided its stores."""
    monkeypatch.setenv("HMX_SOLD_CARRY", "1")
    Z, meta, _ = synth(30000, d=50, levels=(10,), seed=23)
    K, seed = 100, 9
    out = []
    for store in ("0", "1"):
        monkeypatch.setenv("HMX_R_STORE", store)
        skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
        g = Harmony(seed=seed, obj_arith=1)
        g.setup(**skw)
        g.init_cluster_cpp()
        for it in range(3):
            assert g.cluster_cpp() == 0
            g.moe_correct_ridge_cpp()
        assert int(g._scalar("sold_carry")) == 1
        assert int(g._scalar("rounds_without_R")) == 0
        out.append((np.array(g.objective_kmeans), np.array(g.objective_kmeans_dist), np.array(g.objective_kmeans_entropy), g.getZcorr()))
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)


def test_builtin_rccl_communicator_single_rank():
    """The built-in communicator (hmx_comm_init -> ncclCommInitRank, ncclAllReduce issued by the C library on its own
    stream) with a 1-rank communicator and forced collectives reproduces the plain run.  torch-first subprocess: the
    library binds to the RCCL/HIP runtime torch has loaded (tools/rccl_probe2.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_probe2.py")], capture_output=True, text=True,
                       timeout=200, stdin=subprocess.DEVNULL)
    assert "RCCL_PROBE2_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])


def test_torch_nccl_hook_single_rank():
    """torch.distributed (backend nccl == RCCL) all-reduce hook on the library's device buffers: a 1-rank process
    group with forced collectives must reproduce the plain run.  Separate process because torch has to initialise
    its HIP runtime BEFORE the library is loaded (they then share one runtime)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "hook_probe.py")], capture_output=True, text=True,
                       timeout=400, stdin=subprocess.DEVNULL)
    assert "HOOK_PROBE_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])


def test_theta_zero_and_strong_theta():
    """theta = 0 switches the diversity penalty off (pen^0 = 1, src/harmony.cpp:319-321); theta = 6 makes it dominant:
    both ends of the range the fast penalty power exp2(theta * log2(x)) has to cover."""
    Z, meta, _ = synth(6000, d=20, levels=(5,), seed=21)
    for th in (0.0, 6.0):
        g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=3, nclust=30, seed=2, theta=th)
        assert_parity(g, c, ig, ic)


def test_per_cluster_sigma_vector():
    """sigma as a length-K vector (R/ui.R:219-221 allows it): the general-sigma kernel variants (per-lane ce / cl arrays)
    instead of the scalar-constant ones the default uniform sigma selects."""
    Z, meta, _ = synth(6000, d=30, levels=(4,), seed=11)
    K = 40
    sig = np.linspace(0.08, 0.16, K)
    g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=3, nclust=K, seed=3, sigma=sig)
    assert int(g._scalar("usig")) == 0
    assert_parity(g, c, ig, ic)


def test_small_K_uses_four_waves_per_simd():
    """K <= 64 with the default uniform sigma: the 1024-thread (4 waves per SIMD) variants; same parity bar."""
    Z, meta, _ = synth(8000, d=50, levels=(6,), seed=5)
    g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=3, nclust=48, seed=4)
    assert int(g._scalar("usig")) == 1 and int(g._scalar("upd_wps")) == 4
    assert_parity(g, c, ig, ic)


@pytest.mark.parametrize("N,d,K,levels,nested", [
    (20000, 50, 200, (8, 64, 128), True),   # BASELINE configs[4] scaled down: 3 nested covariates, 200 levels, K=200
    (3000, 128, 256, (3,), False),          # the envelope's corner: d = 128, K = 256
    (2500, 64, 128, (2, 3), False),         # exact multiples of the MFMA tile sizes
])
def test_envelope_shapes(N, d, K, levels, nested):
    Z, meta, _ = synth(N, d=d, levels=levels, seed=N + d, nested=nested)
    g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=2, nclust=K, seed=N)
    s = assert_parity(g, c, ig, ic)
    if nested:
        assert c.subset_clusters > 0 and int(g._scalar("subset_clusters")) == c.subset_clusters


@pytest.mark.parametrize("env", [
    {"HMX_UPDATE_IMPL": "v1", "HMX_TILE_IMPL": "v1", "HMX_MOE_IMPL": "v1"},   # first-generation cluster-lane VALU kernels
    {"HMX_FOLD_IMPL": "split"},                                               # k_fold + k_penalty instead of k_foldpen
    {"HMX_FUSED_FOLD": "0"},                                                  # separate k_foldpen launch per block step
    {"HMX_NREP": "1", "HMX_UPD_THREADS": "256", "HMX_UPD_MAXBLOCKS": "64", "HMX_UPD_TPW": "3"},   # launch geometry knobs
    {"HMX_OLDSUM_IMPL": "gather", "HMX_USIG": "0", "HMX_UPD_WPS": "2"},       # gather k_oldsum, general-sigma variants
])
def test_fallback_paths_parity(cell_lines, monkeypatch, env):
    """Every fallback / tuning path keeps the parity bar (they serve shapes outside the MFMA envelope)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g, c, ig, ic = run_both(cell_lines["pcs"], _meta(cell_lines), ["cell_type", "dataset"], max_iter=3, theta=[1, 1], nclust=50)
    assert_parity(g, c, ig, ic)
    Z, meta, _ = synth(20000, d=50, levels=(10,), seed=5)
    g, c, ig, ic = run_both(Z, meta, "cov0", max_iter=2, nclust=100, seed=8)
    assert_parity(g, c, ig, ic)


def test_plain_c_host_example_runs(tmp_path):
    """The C ABI driven from a plain C program (examples/host_example.c): no Python in the loop."""
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "host_example")
    libdir = os.path.join(root, "harmony_amd", "lib")
    p = subprocess.run([gcc, "-std=c11", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "host_example.c"),
                        "-L" + libdir, "-lharmony_mi355x", "-Wl,-rpath," + libdir, "-lm", "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert r.returncode == 0 and "converged after" in r.stdout and "nan" not in r.stdout.lower(), (r.stdout, r.stderr[-2000:])
