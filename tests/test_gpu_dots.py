"""The two builds of the tile kernels' distance GEMM: fp32 MFMA (an fmaf chain, bit for bit) and the split-bf16 form
(six v_mfma_f32_16x16x32_bf16 on three exact bf16 parts per operand).  Both must be fp32-accurate; the second is the default wherever
its LDS image fits, the first the fallback (HMX_DOT=f32) -- both are checked against the oracle and against fp64 here."""
import numpy as np
import pytest

from harmony_amd import Harmony, prepare_setup_args
from helpers import synth
from parity import assert_parity, relfro, run_both

pytestmark = pytest.mark.gpu


def _head(Z, meta, K, monkeypatch, dot):
    """normalise + centroids Y0 + the head's soft assignments (src/harmony.cpp:141-150), on the build `dot`"""
    if dot:
        monkeypatch.setenv("HMX_DOT", dot)
    else:
        monkeypatch.delenv("HMX_DOT", raising=False)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    g = Harmony(seed=3)
    g.setup(**skw)
    Zn = Z / np.linalg.norm(Z, axis=1, keepdims=True)
    Y0 = np.asfortranarray(Zn[np.random.default_rng(K).choice(len(Z), K, replace=False)].T)        # d x K
    g.init_cluster_cpp(Y0)
    return g, Zn, Y0


@pytest.mark.parametrize("d,K", [(50, 100), (20, 12), (32, 64), (64, 128), (7, 30), (50, 200), (100, 60), (128, 256)])
def test_split_bf16_distances_are_fp32_accurate(monkeypatch, d, K):
    """R of the head = softmax_k(-(2 - 2 z.y_k) / sigma) from the same Y on both builds against fp64: the split-bf16 distances carry
    no more error than the fp32 fmaf chain (both ~1e-7 absolute on |z.y| <= 1)"""
    Z, meta, _ = synth(5000, d=d, levels=(3,), seed=d + K)
    gb, Zn, Y0 = _head(Z, meta, K, monkeypatch, None)
    used_bf = int(gb._scalar("dot_bf"))
    Rb = gb.R.copy()
    gf, _, _ = _head(Z, meta, K, monkeypatch, "f32")
    assert int(gf._scalar("dot_bf")) == 0
    Rf = gf.R.copy()
    Yn = Y0 / np.linalg.norm(Y0, axis=0, keepdims=True)
    t = -(2.0 - 2.0 * (Zn.astype(np.float32).astype(np.float64) @ Yn.astype(np.float32).astype(np.float64))) / 0.1
    t -= t.max(axis=1, keepdims=True)
    R64 = np.exp(t); R64 /= R64.sum(axis=1, keepdims=True)
    eb, ef = float(np.abs(Rb - R64.T).max()), float(np.abs(Rf - R64.T).max())
    print("d=%d K=%d  split-bf16 offered=%d  max|R - fp64|: split-bf16 %.2e, fp32 chain %.2e, between the builds %.2e"
          % (d, K, used_bf, eb, ef, float(np.abs(Rb - Rf).max())))
    assert ef < 2e-5 and eb < 2e-5 and eb < 3 * ef + 1e-6, (eb, ef)
    # the objective (dist + entropy sums over all cells) agrees to fp32 summation noise
    assert abs(gb.objective_kmeans[-1] - gf.objective_kmeans[-1]) <= 2e-6 * abs(gf.objective_kmeans[-1])


@pytest.mark.parametrize("K,d,levels", [(100, 50, (10,)), (40, 20, (3, 4)), (200, 50, (6, 5)), (60, 100, (4,))])
def test_fp32_build_still_matches_the_oracle(monkeypatch, K, d, levels):
    """the fallback build (HMX_DOT=f32: shapes whose split-bf16 image does not fit the LDS take it by themselves) end to end"""
    monkeypatch.setenv("HMX_DOT", "f32")
    Z, meta, _ = synth(20000, d=d, levels=levels, seed=K)
    g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=3, nclust=K, seed=K + 1)
    assert int(g._scalar("dot_bf")) == 0
    assert_parity(g, c, ig, ic)


def test_both_builds_agree_end_to_end(monkeypatch):
    """the same run on the two builds: same iteration counts, corrected embeddings equal to fp32 noise"""
    Z, meta, _ = synth(60000, d=50, levels=(10,), seed=2)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
    out = []
    for dot in (None, "f32"):
        if dot:
            monkeypatch.setenv("HMX_DOT", dot)
        g = Harmony(seed=5)
        g.setup(**skw)
        g.init_cluster_cpp()
        it = 0
        for it in range(1, 11):
            assert g.cluster_cpp() == 0
            g.moe_correct_ridge_cpp()
            if g.check_convergence(1):
                break
        out.append((it, g.getZcorr().copy(), np.array(g.objective_kmeans), int(g._scalar("chain")), int(g._scalar("dot_bf"))))
    (ib, Zb, ob, chb, bfb), (i_f, Zf, of, chf, bff) = out
    print("iterations %d / %d, chain %d / %d, Z_corr rel. diff %.2e, objective rel. diff %.2e" % (ib, i_f, chb, chf, relfro(Zb, Zf),
          float(np.max(np.abs(ob[:len(of)] - of[:len(ob)]) / np.abs(of[:len(ob)])))))
    assert bfb == 1 and bff == 0
    assert ib == i_f and len(ob) == len(of)
    assert relfro(Zb, Zf) < 2e-5
