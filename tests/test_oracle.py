"""CPU tests: pin the oracle against the reference's own test invariants on the reference's
bundled fixtures (tests/testthat/test_integration.R, test_two_variable.R), and check the
oracle's two arithmetic modes against each other.  No GPU needed."""
import numpy as np
import pytest

from harmony_amd import harmony_options
from helpers import chi2, run_backend, synth
from oracle.oracle import OracleHarmony


def _meta(fx):
    return {"dataset": fx["dataset_levels"][fx["dataset"]], "cell_type": fx["cell_type_levels"][fx["cell_type"]]}


@pytest.fixture(scope="module")
def obj_small(cell_lines_small):
    # test_integration.R:5-7
    return run_backend(OracleHarmony(accurate=True, seed=1), cell_lines_small["pcs"], _meta(cell_lines_small), "dataset",
                       theta=1, nclust=50, max_iter=5, options=harmony_options(max_iter_cluster=10))


def test_dimensions(obj_small):  # test_integration.R:9-14
    o = obj_small
    assert o.Y.shape == (o.d, o.K) and o.getZcorr().shape == (o.d, o.N)
    assert o.getZorig().shape == (o.d, o.N) and o.R.shape == (o.K, o.N)
    assert (o.d, o.K, o.N) == (20, 50, 300)


def test_R_is_distribution(obj_small):  # :16-20
    R = obj_small.R
    assert R.min() >= 0 and R.max() <= 1
    np.testing.assert_allclose(R.sum(axis=0), 1.0, atol=1e-5)


def test_Zcorr_finite(obj_small):  # :22-26
    assert np.all(np.isfinite(obj_small.getZcorr()))


def test_theta_decreases_chi2(cell_lines_small):  # :29-41
    m = _meta(cell_lines_small)
    o0 = run_backend(OracleHarmony(seed=1), cell_lines_small["pcs"], m, "dataset", theta=0, nclust=20, max_iter=2)
    o1 = run_backend(OracleHarmony(seed=1), cell_lines_small["pcs"], m, "dataset", theta=1, nclust=5, max_iter=2)
    assert chi2(o0) > chi2(o1)


def test_error_messages(cell_lines_small):  # :43-56
    m = _meta(cell_lines_small)
    with pytest.raises(ValueError):
        run_backend(OracleHarmony(), cell_lines_small["pcs"], m, "fake_variable")
    with pytest.raises(ValueError):
        run_backend(OracleHarmony(), cell_lines_small["pcs"], m, "dataset", lambda_=[1, 2])
    with pytest.raises(ValueError):
        run_backend(OracleHarmony(), cell_lines_small["pcs"], {k: v[:-1] for k, v in m.items()}, "dataset")


@pytest.fixture(scope="module")
def obj_two(cell_lines):
    # test_two_variable.R:5-11 (multi-covariate arma::inv branch)
    return run_backend(OracleHarmony(accurate=True, seed=1), cell_lines["pcs"], _meta(cell_lines), ["cell_type", "dataset"],
                       theta=[1, 1], nclust=50, max_iter=10, options=harmony_options(max_iter_cluster=10))


def test_two_variable_invariants(obj_two):  # test_two_variable.R:13-37
    o = obj_two
    assert o.Y.shape == (o.d, o.K) and o.R.shape == (o.K, o.N) and o.getZcorr().shape == (o.d, o.N)
    assert o.O.shape[1] == 5 and o.E.shape[1] == 5
    R = o.R
    assert R.min() >= 0 and R.max() <= 1
    np.testing.assert_allclose(R.sum(axis=0), 1.0, atol=1e-5)
    assert np.all(np.isfinite(o.getZcorr()))


def test_two_variable_theta(cell_lines):  # :39-55
    m = _meta(cell_lines)
    lo = run_backend(OracleHarmony(seed=1), cell_lines["pcs"], m, ["cell_type", "dataset"], theta=[0, 0], nclust=20, max_iter=2)
    hi = run_backend(OracleHarmony(seed=1), cell_lines["pcs"], m, ["cell_type", "dataset"], theta=[2, 2], nclust=20, max_iter=2)
    assert chi2(lo) > chi2(hi)


def test_walkthrough_plausibility(cell_lines):
    """doc/detailedWalkthrough.html (harmony 1.2.4, set.seed(1), nclust=5): initial clusters are nearly pure per
    cell line and E rows are proportional to the batch sizes 846/824/700.  RNG-dependent, so only the structure
    is checked ('weak golden numbers')."""
    m = _meta(cell_lines)
    o = OracleHarmony(seed=1)
    from harmony_amd import prepare_setup_args
    skw, _ = prepare_setup_args(cell_lines["pcs"], m, "dataset", theta=1, nclust=5)
    o.setup(**skw)
    o.init_cluster_cpp()
    O, E = o.O, o.E
    np.testing.assert_allclose(O.sum(), 2370, rtol=1e-4)
    np.testing.assert_allclose(O.sum(axis=0), [846, 824, 700], rtol=1e-4)
    np.testing.assert_allclose(E / E.sum(axis=1, keepdims=True), np.tile(np.array([846, 824, 700]) / 2370.0, (5, 1)), rtol=1e-4)
    # jurkat-only and 293t-only datasets never share a cluster at initialisation: some O entries ~ 0
    assert (np.round(O) == 0).sum() >= 3


def test_faithful_vs_accurate_noise_floor(cell_lines_small):
    """The two arithmetic modes agree to fp32 round-off on a small problem: this gap is the noise floor
    quoted in DESIGN.md for the 1e-4 parity bar."""
    m = _meta(cell_lines_small)
    kw = dict(theta=2, nclust=10, max_iter=3)
    a = run_backend(OracleHarmony(accurate=True, seed=3), cell_lines_small["pcs"], m, "dataset", **kw)
    f = run_backend(OracleHarmony(accurate=False, seed=3), cell_lines_small["pcs"], m, "dataset", **kw)
    za, zf = a.getZcorr(), f.getZcorr()
    assert np.linalg.norm(za - zf) / np.linalg.norm(za) < 1e-4
    assert np.array_equal(a.R.argmax(axis=0), f.R.argmax(axis=0))


def test_subset_path_is_exercised():
    """Nested 3-covariate synthetic data sends most clusters down the batch-subset path (src/harmony.cpp:440-547)."""
    Z, meta, _ = synth(4000, d=20, levels=(4, 8, 16), seed=5, nested=True)
    o = run_backend(OracleHarmony(seed=2), Z, meta, ["cov0", "cov1", "cov2"], nclust=40, max_iter=2)
    assert o.subset_clusters > 0
    assert np.all(np.isfinite(o.getZcorr()))
    np.testing.assert_allclose(o.R.sum(axis=0), 1.0, atol=1e-5)


def test_injected_order_equals_generated(cell_lines_small):
    """Pushing the generator's own permutation through the injection hook reproduces the generated run exactly."""
    from oracle.oracle import feistel_order
    m = _meta(cell_lines_small)
    kw = dict(theta=2, nclust=10, max_iter=1)
    a = run_backend(OracleHarmony(seed=7), cell_lines_small["pcs"], m, "dataset", **kw)
    b = OracleHarmony(seed=7)
    from harmony_amd import prepare_setup_args
    skw, _ = prepare_setup_args(cell_lines_small["pcs"], m, "dataset", theta=2, nclust=10)
    b.setup(**skw)
    b.init_cluster_cpp()
    for r in range(4):
        b.push_update_order(feistel_order(7, r, 300))
    b.cluster_cpp()
    b.moe_correct_ridge_cpp()
    np.testing.assert_array_equal(a.getZcorr(), b.getZcorr())


def test_getLambda_definition(cell_lines_small, cell_lines):
    """getLambda (src/harmony.cpp:657-669): K x (B + 1); estimated lambda = [0, alpha * E[k, :]] (find_lambda_cpp, src/utils.cpp:159-163),
    fixed lambda = the caller's vector in every row"""
    o = run_backend(OracleHarmony(accurate=False, seed=1), cell_lines_small["pcs"], _meta(cell_lines_small), "dataset", nclust=10, max_iter=2)
    L = o.getLambda()
    assert L.shape == (o.K, o.B + 1) and np.all(L[:, 0] == 0)
    np.testing.assert_allclose(L[:, 1:], np.float32(0.2) * o.E.astype(np.float32), rtol=1e-6)
    o = run_backend(OracleHarmony(accurate=False, seed=1), cell_lines["pcs"], _meta(cell_lines), ["dataset", "cell_type"], nclust=8, max_iter=1,
                    lambda_=[0.5, 2.0])
    assert np.array_equal(o.getLambda(), np.tile(np.concatenate([[0.0], np.repeat([0.5, 2.0], [3, 2])]), (8, 1)))


def test_liberties_of_the_faithful_mode_are_ulp_level(cell_lines):
    """The places where Armadillo / BLAS -- not /root/reference -- fix the operation order (oracle header, LIBERTIES): each one flipped on its
    own changes the faithful run (it IS a different rounding sequence) but only at the level of fp32 noise amplified by the iteration --
    nowhere near the reference's accumulation bias the mode exists to reproduce."""
    m = _meta(cell_lines)
    kw = dict(theta=[1, 1], nclust=30, max_iter=4, options=harmony_options(max_iter_cluster=6))
    base = run_backend(OracleHarmony(accurate=False, seed=2), cell_lines["pcs"], m, ["cell_type", "dataset"], **kw).getZcorr()
    moved = 0
    for bit in (1, 2, 4, 8, 16):
        z = run_backend(OracleHarmony(accurate=False, seed=2, liberty=bit), cell_lines["pcs"], m, ["cell_type", "dataset"], **kw).getZcorr()
        rel = float(np.linalg.norm(z - base) / np.linalg.norm(base))
        assert rel < 1e-4, (bit, rel)
        moved += rel > 0
    assert moved >= 3      # (the switches are really wired: the L1 / L2 sum orders and the per-non-zero apply change bits)
