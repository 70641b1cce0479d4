"""Reference arithmetic on the GPU (-m gpu): the restarted sequential sums against a one-after-the-other fp32 loop (bit equality,
certified by the library's own consistency check), and the reference-arithmetic mode end to end against the oracle in FAITHFUL mode
(every accumulator fp32 in the reference's operation order, src/harmony.cpp:149-150,312-330,567-609, src/utils.cpp:67-75)."""
import ctypes as C

import numpy as np
import pytest

from harmony_amd import Harmony, _lib, harmony_options, prepare_setup_args
from helpers import synth
from oracle import oracle as orc
from oracle.oracle import OracleHarmony
from parity import relfro

pytestmark = pytest.mark.gpu


def _seq32(x, axis=0):
    """s = 0; for t in x: s = fl32(s + t) -- numpy's accumulate adds one element after the other in the array's dtype"""
    return np.add.accumulate(np.asarray(x, dtype=np.float32), axis=axis, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _oe_case(n=120000, K=100, B=7, seed=5):
    rng = np.random.default_rng(seed)
    # soft assignments: a few large entries per row, thousands of tiny ones per cluster -- the regime in which fp32 accumulators drop terms
    logits = rng.normal(size=(n, K)).astype(np.float32) * 6.0
    R = np.exp(logits - logits.max(axis=1, keepdims=True)).astype(np.float32)
    R /= R.sum(axis=1, keepdims=True)
    R = np.ascontiguousarray(R, dtype=np.float32)
    lst = rng.permutation(n).astype(np.int32)                     # a shuffled order
    lev = rng.integers(0, B, size=n).astype(np.int32)
    cuts = [0, n // 20, n // 20, n // 3, n]                        # four blocks, one of them empty
    off = np.array(cuts[:-1], np.int32)
    cnt = np.diff(np.array(cuts)).astype(np.int32)
    want = np.zeros((len(off), 1 + B, K), np.float32)
    for c in range(len(off)):
        cells = lst[off[c]:off[c] + cnt[c]]
        if cnt[c]:
            want[c, 0] = _seq32(R[cells])[-1]
            for b in range(B):
                sub = cells[lev[cells] == b]
                if sub.size:
                    want[c, 1 + b] = _seq32(R[sub])[-1]
    return R, lev, lst, off, cnt, want


def _run_oe(case, seg, passes):
    R, lev, lst, off, cnt, want = case
    n, K = R.shape
    B = want.shape[1] - 1
    tot = np.empty(want.shape, np.float32)
    mm, res = C.c_int64(-1), C.c_double(-1)
    st = _lib.load().hmx_debug_seq_oe(_fp(R), n, K, _ip(lev), B, _ip(lst), len(lst), _ip(off), _ip(cnt), len(off), seg, passes, _fp(tot),
                                      C.byref(mm), C.byref(res))
    assert st == 0
    rel = float(np.max(np.abs(tot.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30)))
    return tot, rel, mm.value, res.value


@pytest.mark.parametrize("B", [7, 20, 40])
def test_restarted_sums_reach_the_sequential_loop(B):
    """O / E block sums (a shuffled cell list, all cells + per level, src/harmony.cpp:312-313): every pass moves the segment starts closer
    to the fixed point, and at the fixed point the result IS the one-after-the-other fp32 loop, bit for bit.  B = 7 / 20: the level rows in
    registers (16 / 32 of them, round 5), B = 40: in LDS."""
    case = _oe_case(B=B, n=120000 if B == 7 else 60000)
    want = case[-1]
    log = []
    for seg, passes in ((64, 2), (64, 3), (256, 3), (64, 5), (64, 8), (64, 24), (256, 24)):
        tot, rel, mm, res = _run_oe(case, seg, passes)
        log.append((seg, passes, rel, mm, res))
        if passes == 2:
            assert rel <= 2e-5, log                          # the default since round 5 (cold: two passes)
        if passes == 3:
            assert rel <= 2e-6 and res <= 1e-5, log          # the default: already inside fp32 noise of the terms themselves
        if passes == 24:
            assert np.array_equal(tot.view(np.uint32), want.view(np.uint32)) and mm == 0 and res == 0.0, log
    print("restarted O/E sums (segment cells, passes, max rel. error vs the sequential loop, segment starts still moving, largest relative move):", log)
    # the sums really are in the regime the mode exists for: the sequential fp32 total differs from the exact one
    R, lev, lst, off, cnt, _ = case
    exact = R[lst[off[3]:off[3] + cnt[3]]].astype(np.float64).sum(axis=0)
    assert np.max(np.abs(want[3, 0] - exact) / exact) > 1e-6


def test_restarted_sums_over_term_arrays_reach_the_sequential_loop():
    rng = np.random.default_rng(9)
    n = 3_000_000
    a = rng.random(n, dtype=np.float32) * rng.choice(np.array([1e-6, 1e-3, 1.0], np.float32), size=n)       # monotone, crosses ~20 binades
    b = -(rng.random(n, dtype=np.float32) ** 8)                                                               # negative terms (the entropy sum)
    c = (rng.normal(size=n) * np.exp(rng.normal(size=n) * 3)).astype(np.float32)                              # mixed signs, heavy tails
    T = np.ascontiguousarray(np.stack([a, b, c]), dtype=np.float32)
    want = np.array([_seq32(T[i])[-1] for i in range(3)], np.float32)
    log = []
    for passes in (2, 3, 5, 8, 32):
        tot = np.empty(3, np.float32)
        mm, res = C.c_int64(-1), C.c_double(-1)
        st = _lib.load().hmx_debug_seq_arr(_fp(T), n, 3, 4096, passes, _fp(tot), C.byref(mm), C.byref(res))
        assert st == 0
        rel = np.abs(tot.astype(np.float64) - want) / np.abs(want)
        log.append((passes, rel.tolist(), mm.value, res.value))
        if passes == 2:
            assert rel[0] <= 2e-5 and rel[1] <= 2e-5, log
        if passes == 3:
            assert rel[0] <= 1e-6 and rel[1] <= 1e-6, log       # (the mixed-sign chain is reported only: its total is a small difference of large sums)
        if passes == 32:
            assert np.array_equal(tot[:2].view(np.uint32), want[:2].view(np.uint32)), log
    print("restarted sums over term arrays (passes, rel. error per chain, segment starts still moving, largest relative move):", log)
    # round 6: the one-launch form (k_seq_obj_fused: segments of 32 terms in registers, the scans between the passes through the in-launch exchange of
    # workgroup aggregates) -- the same fixed point, reached the same way; several launches when more than 8 passes are asked for
    log = []
    for nn in (n, 1000, 36):
        Tn = np.ascontiguousarray(T[:, :nn])
        wn = np.array([_seq32(Tn[i])[-1] for i in range(3)], np.float32)
        for passes in (2, 3, 5, 8, 32):
            tot = np.empty(3, np.float32)
            mm, res = C.c_int64(-1), C.c_double(-1)
            st = _lib.load().hmx_debug_seq_arr(_fp(Tn), nn, 3, 0, passes, _fp(tot), C.byref(mm), C.byref(res))
            assert st == 0
            rel = np.abs(tot.astype(np.float64) - wn) / np.abs(wn)
            log.append((nn, passes, rel.tolist(), mm.value, res.value))
            if passes == 2:
                assert rel[0] <= 2e-5 and rel[1] <= 2e-5, log
            if passes == 3:
                assert rel[0] <= 1e-6 and rel[1] <= 1e-6, log
            if passes == 32:
                assert np.array_equal(tot[:2].view(np.uint32), wn[:2].view(np.uint32)), log
    print("the same in one launch (terms, passes, rel. error per chain, starts still moving, largest relative move):", log)


def test_one_launch_sums_across_super_groups():
    """k_seq_obj_fused's exchange has three levels (groups of 64 workgroups, super-groups of 64 groups): 40M terms per chain are 4883 workgroups, i.e. two
    super-groups -- the totals must still follow the one-after-the-other loop (the 10M-cell runs have 30 super-groups)."""
    rng = np.random.default_rng(11)
    n = 40_000_000
    a = rng.random(n, dtype=np.float32) * rng.choice(np.array([1e-6, 1e-3, 1.0], np.float32), size=n)
    b = -(rng.random(n, dtype=np.float32) ** 8)
    c = (rng.random(n, dtype=np.float32) * np.float32(1e-4)).astype(np.float32)
    T = np.ascontiguousarray(np.stack([a, b, c]), dtype=np.float32)
    want = np.array([_seq32(T[i])[-1] for i in range(3)], np.float32)
    log = []
    for passes in (3, 8):
        tot = np.empty(3, np.float32)
        mm, res = C.c_int64(-1), C.c_double(-1)
        assert _lib.load().hmx_debug_seq_arr(_fp(T), n, 3, 0, passes, _fp(tot), C.byref(mm), C.byref(res)) == 0
        rel = np.abs(tot.astype(np.float64) - want) / np.abs(want)
        log.append((passes, rel.tolist(), mm.value, res.value))
    print("one-launch sums over 40M terms per chain (passes, rel. error per chain, workgroups still moving, largest relative move):", log)
    assert max(log[0][1]) <= 1e-3 and max(log[1][1]) <= 2e-6, log          # (three passes: the negative chain of 40M terms saturates and is still 2e-4 away; eight: on the loop)


def _iterate(obj, max_iter=10):
    it = 0
    for it in range(1, max_iter + 1):
        assert obj.cluster_cpp() == 0
        obj.moe_correct_ridge_cpp()
        if obj.check_convergence(1):
            break
    return it


def _pair(Z, meta, vars_use, seed, max_iter=10, **kw):
    skw, _ = prepare_setup_args(Z, meta, vars_use, **kw)
    g = Harmony(seed=seed, ref_arith=1)
    g._set("seq_stats", 1)
    g.setup(**skw)
    Y0 = g.kmeans_centers()
    c = OracleHarmony(mask=0, seed=seed)          # faithful: all four accumulator groups in the reference's fp32
    c.setup(**skw)
    g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
    ig, ic = _iterate(g, max_iter), _iterate(c, max_iter)
    g.Y0_shared = Y0
    return g, c, ig, ic


def _report(g, c):
    n = min(len(g.objective_kmeans), len(c.objective_kmeans))
    ag, ac = g.R.argmax(axis=0), c.R.argmax(axis=0)
    bad = np.where(ag != ac)[0]
    srt = np.sort(c.R[:, bad], axis=0) if bad.size else np.zeros((2, 0))
    return dict(Z_rel=relfro(g.getZcorr(), c.getZcorr()), O_rel=relfro(g.O, c.O), E_rel=relfro(g.E, c.E), Y_rel=relfro(g.Y, c.Y),
                R_maxabs=float(np.abs(g.R - c.R).max()), flips=int(bad.size), clear_flips=int(((srt[-1] - srt[-2]) >= 1e-5).sum()) if bad.size else 0,
                obj_rel=float(np.max(np.abs(g.objective_kmeans[:n] - c.objective_kmeans[:n]) / np.abs(c.objective_kmeans[:n]))),
                obj_len=(len(g.objective_kmeans), len(c.objective_kmeans)), mismatch=int(g._scalar("seq:mismatch")),
                residual=float(g._scalar("seq:residual")))


def test_reference_arithmetic_fixture(cell_lines):
    """the reference's bundled 2370-cell fixture, one covariate, test-suite parameters"""
    orc.use_openblas(1)
    meta = {"dataset": cell_lines["dataset_levels"][cell_lines["dataset"]]}
    g, c, ig, ic = _pair(cell_lines["pcs"], meta, "dataset", seed=2, max_iter=5, nclust=50, theta=1,
                        options=harmony_options(max_iter_cluster=10))
    s = _report(g, c)
    print("ref_arith cell_lines:", s)
    assert ig == ic and s["obj_len"][0] == s["obj_len"][1], (ig, ic, s)
    assert np.array_equal(g.kmeans_rounds, c.kmeans_rounds)
    assert s["Z_rel"] <= 1e-5 and s["clear_flips"] == 0 and s["obj_rel"] <= 2e-5 and s["R_maxabs"] <= 5e-5, s
    assert s["O_rel"] <= 1e-5 and s["E_rel"] <= 1e-5 and s["Y_rel"] <= 1e-5, s


@pytest.mark.timeout(900, method="thread")
def test_reference_arithmetic_100k():
    """100k x 50, K = 100, 10 batches, defaults, to convergence: at this size the reference's fp32 accumulators are 2e-4 away from exact
    accumulation (profiles/r2_arith_gap_100k.json); the reference-arithmetic mode has to follow them, not the exact result"""
    Z, meta, _ = synth(100000, d=50, levels=(10,), seed=7)
    orc.use_openblas(4)
    g, c, ig, ic = _pair(Z, meta, "cov0", seed=3, nclust=100)
    s = _report(g, c)
    print("ref_arith 100k:", s)
    assert ig == ic and s["obj_len"][0] == s["obj_len"][1], (ig, ic, s)
    assert s["Z_rel"] <= 1e-5 and s["clear_flips"] == 0 and s["obj_rel"] <= 1e-4, s
    # and it is NOT the exact-accumulator result: the default mode differs from the faithful oracle by an order of magnitude more
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
    d = Harmony(seed=3)
    d.setup(**skw)
    d.init_cluster_cpp(g.Y0_shared)
    _iterate(d)
    assert relfro(d.getZcorr(), c.getZcorr()) > 5 * s["Z_rel"]


@pytest.mark.parametrize("case", ["cell_lines", "synth20k"])
def test_reference_arithmetic_against_the_reference_sources(case, cell_lines):
    """The product's reference-arithmetic mode against THE REFERENCE'S OWN harmony.cpp / utils.cpp (oracle/_ref: compiled where they lie,
    unmodified, over oracle/shim/ -- tests/test_oracle_ref.py), on R's stream after the same set.seed and with NO shared random choice:
    each side runs its own k-means++ race, Lloyd iterations and one arma::shuffle per round.  The restated oracle (faithful, same stream)
    must equal the reference's sources bit for bit here too (the library travelled to this machine, it was not rebuilt)."""
    from oracle import ref as oref
    if not oref.available():
        pytest.skip("oracle/_ref/libharmony_ref.so did not travel with the tree")
    if case == "cell_lines":
        Z, meta, vu, K = cell_lines["pcs"], {"dataset": cell_lines["dataset_levels"][cell_lines["dataset"]]}, "dataset", 20
    else:
        (Z, meta, _), vu, K = synth(20000, d=50, levels=(5,), seed=3), "cov0", 50
    skw, _ = prepare_setup_args(Z, meta, vu, nclust=K)
    orc.load().orc_set_sgemm(None)               # the oracle's own sequential dot products, as the shim's dense product
    g = Harmony(seed=42, rng="R", ref_arith=1)
    g.setup(**skw)
    r = oref.RefHarmony(seed=42)
    r.setup(**skw)
    c = OracleHarmony(mask=0, seed=42, rng=1)
    c.setup(**skw)
    g.init_cluster_cpp(); r.init_cluster_cpp(); c.init_cluster_cpp()
    ig, ir, ic = _iterate(g, 4), _iterate(r, 4), _iterate(c, 4)
    assert ic == ir and np.array_equal(c.getZcorr(), r.getZcorr()) and np.array_equal(c.R, r.R) and np.array_equal(c.objective_kmeans, r.objective_kmeans)
    s = _report(g, r)
    print("ref_arith vs the reference's sources (%s):" % case, s)
    assert ig == ir and s["obj_len"][0] == s["obj_len"][1] and np.array_equal(g.kmeans_rounds, r.kmeans_rounds), (ig, ir, s)
    assert s["Z_rel"] <= 1e-5 and s["clear_flips"] == 0 and s["obj_rel"] <= 1e-4 and s["R_maxabs"] <= 1e-3, s
    assert s["O_rel"] <= 1e-4 and s["E_rel"] <= 1e-4 and s["Y_rel"] <= 1e-4, s


@pytest.mark.parametrize("case", ["ref_sources_cell_lines_small", "ref_sources_cell_lines_small_test_integration", "ref_sources_cell_lines_two_covariates"])
def test_reference_arithmetic_against_the_golden_vectors_of_the_reference_sources(case):
    """tests/golden/ref_sources_*.npz: what the reference's own sources (over oracle/shim/) leave after three harmony iterations on the
    reference's bundled fixtures, R's stream after set.seed(1) (tools/make_ref_goldens.py) -- committed, so this comparison needs neither
    /root/reference nor the built library.  The product (ref_arith, rng = R) draws its own race, Lloyd iterations and shuffles."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_ref_goldens as mrg
    from conftest import load_fixture
    gold = load_fixture(case)
    skw, seed, max_iter = mrg.setup_kwargs(case)
    g = Harmony(seed=seed, rng="R", ref_arith=1)
    g.setup(**skw)
    g.init_cluster_cpp()
    it = mrg.walk(g, max_iter)
    Zg, Zr, Rg, Rr = g.getZcorr(), gold["Z_corr"].astype(np.float64), g.R, gold["R"].astype(np.float64)
    bad = np.where(Rg.argmax(axis=0) != Rr.argmax(axis=0))[0]
    srt = np.sort(Rr[:, bad], axis=0) if bad.size else np.zeros((2, 0))
    og, orf = g.objective_kmeans, gold["objective_kmeans"].astype(np.float64)
    s = dict(Z_rel=relfro(Zg, Zr), R_maxabs=float(np.abs(Rg - Rr).max()), flips=int(bad.size), clear_flips=int(((srt[-1] - srt[-2]) >= 1e-5).sum()) if bad.size else 0,
             Y_rel=relfro(g.Y, gold["Y"].astype(np.float64)), O_rel=relfro(g.O, gold["O"].astype(np.float64)), obj_len=(len(og), len(orf)),
             obj_rel=float(np.max(np.abs(og[:len(orf)] - orf[:len(og)]) / np.abs(orf[:len(og)]))), iterations=(it, int(gold["iterations"])))
    print("ref_arith vs golden vectors of the reference's sources (%s):" % case, s)
    assert it == int(gold["iterations"]) and s["obj_len"][0] == s["obj_len"][1] and np.array_equal(g.kmeans_rounds, gold["kmeans_rounds"]), s
    tol = 1e-4 if "two_covariates" in case else 1e-5
    assert s["Z_rel"] <= tol and s["clear_flips"] == 0 and s["obj_rel"] <= 1e-4 and s["Y_rel"] <= 1e-4 and s["O_rel"] <= 1e-4, s


@pytest.mark.parametrize("shape", [dict(N=60000, K=100, levels=(10,)), dict(N=30000, K=40, levels=(3, 4))])
def test_kept_distances_give_the_same_objective_bits(shape):
    """Round 6: inside a cluster_cpp call dist_mat is computed by the first objective evaluation and kept for the call's other rounds (Y and Z_corr do not
    change between them); R % dist is then formed inside the one-launch sums with the single rounding of the terms kernel.  Same terms, same chains: the
    objective series and everything that follows from it -- round counts, Z_corr -- are BIT-identical to the run that recomputes the distances every round
    ("seq_fused" bit 3 off)."""
    Z, meta, _ = synth(shape["N"], d=50, levels=shape["levels"], seed=5)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=shape["K"])
    out, Y0 = [], None
    for fused in (13, 5):
        g = Harmony(seed=2, ref_arith=1)
        g._set("seq_fused", fused)
        g.setup(**skw)
        if Y0 is None:
            Y0 = g.kmeans_centers()
        g.init_cluster_cpp(Y0)
        it = _iterate(g, 3)
        out.append((it, np.array(g.objective_kmeans), list(g.kmeans_rounds), g.getZcorr().copy()))
        del g
    (i1, o1, r1, z1), (i0, o0, r0, z0) = out
    assert i1 == i0 and r1 == r0 and np.array_equal(o1, o0) and np.array_equal(z1, z0)


def test_reference_arithmetic_groups_can_be_switched_one_by_one(cell_lines_small):
    """each switch alone against the oracle with the matching arithmetic mask (oracle bit set = fp64: mask = 15 minus the group)"""
    meta = {"dataset": cell_lines_small["dataset_levels"][cell_lines_small["dataset"]]}
    skw, _ = prepare_setup_args(cell_lines_small["pcs"], meta, "dataset", nclust=10)
    for kw, mask in ((dict(oe_arith=1), 14), (dict(obj_arith=1), 13), (dict(ridge_arith=1), 11), (dict(solve_arith=1), 7)):
        g = Harmony(seed=1, **kw); g.setup(**skw)
        c = OracleHarmony(mask=mask, seed=1); c.setup(**skw)
        Y0 = g.kmeans_centers()
        g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
        ig, ic = _iterate(g, 4), _iterate(c, 4)
        s = _report(g, c)
        assert ig == ic and s["Z_rel"] <= 1e-5 and s["obj_rel"] <= 2e-5 and s["clear_flips"] == 0, (kw, s)


def test_reference_arithmetic_several_covariates(cell_lines):
    """two crossed covariates (dataset x cell_type) on the reference's 2370-cell fixture and three nested ones on a synthetic: the ridge
    statistics are the level and level-PAIR chains of Phi_Rk * Phi_moe_t (src/harmony.cpp:561-568), the inverse the oracle's unblocked
    fp32 LU (arma::inv, :573) -- against the faithful oracle (mask 0)"""
    orc.use_openblas(1)
    meta = {"dataset": cell_lines["dataset_levels"][cell_lines["dataset"]], "cell_type": cell_lines["cell_type_levels"][cell_lines["cell_type"]]}
    g, c, ig, ic = _pair(cell_lines["pcs"], meta, ["dataset", "cell_type"], seed=1, max_iter=4, nclust=20)
    s = _report(g, c)
    print("ref_arith, two crossed covariates:", s)
    assert ig == ic and s["obj_len"][0] == s["obj_len"][1] and np.array_equal(g.kmeans_rounds, c.kmeans_rounds), (ig, ic, s)
    assert s["Z_rel"] <= 1e-4 and s["clear_flips"] == 0 and s["obj_rel"] <= 1e-4, s
    Z, meta3, _ = synth(40000, d=50, levels=(4, 12, 24), seed=5, nested=True)
    orc.use_openblas(4)
    g, c, ig, ic = _pair(Z, meta3, list(meta3), seed=2, max_iter=4, nclust=60)
    s = _report(g, c)
    print("ref_arith, three nested covariates:", s, "subset clusters", int(g._scalar("subset_clusters")), c.subset_clusters)
    assert ig == ic and s["obj_len"][0] == s["obj_len"][1], (ig, ic, s)
    assert int(g._scalar("subset_clusters")) == c.subset_clusters
    assert s["Z_rel"] <= 1e-4 and s["clear_flips"] == 0 and s["obj_rel"] <= 1e-4, s
    # each group alone still follows its own oracle mask (the statistics alone: fp64 solve on both sides)
    skw, _ = prepare_setup_args(cell_lines["pcs"], meta, ["dataset", "cell_type"], nclust=20)
    for kw, mask in ((dict(ridge_arith=1), 11), (dict(oe_arith=1, obj_arith=1), 12)):
        g = Harmony(seed=1, **kw); g.setup(**skw)
        c = OracleHarmony(mask=mask, seed=1); c.setup(**skw)
        Y0 = g.kmeans_centers()
        g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
        ig, ic = _iterate(g, 3), _iterate(c, 3)
        s = _report(g, c)
        assert ig == ic and s["Z_rel"] <= 1e-5 and s["obj_rel"] <= 2e-5 and s["clear_flips"] == 0, (kw, s)


def test_reference_arithmetic_needs_one_gpu(cell_lines_small):
    """a sharded handle refuses the reference-arithmetic switches: a block's shuffled order interleaves the shards cell by cell, the
    reference's sequential sums cannot be followed from contiguous shards"""
    meta = {"dataset": cell_lines_small["dataset_levels"][cell_lines_small["dataset"]]}
    skw, _ = prepare_setup_args(cell_lines_small["pcs"], meta, "dataset", nclust=10)
    g = Harmony(seed=1, ref_arith=1)
    g._set("comm_force", 1)
    with pytest.raises(Exception, match="one GPU"):
        g.setup(**skw)


def test_reference_arithmetic_with_the_hosts_shuffles(cell_lines):
    """the sequential sums run in the round's shuffled order, whoever drew it: with the R-compatible stream (arma::shuffle restated on the
    host) and with injected update orders the library follows the faithful oracle just as it does with its own generator"""
    orc.use_openblas(1)
    meta = {"dataset": cell_lines["dataset_levels"][cell_lines["dataset"]]}
    skw, _ = prepare_setup_args(cell_lines["pcs"], meta, "dataset", nclust=20)
    N = cell_lines["pcs"].shape[0]
    # (a) rng = R on both sides, nothing shared: seeds and shuffles from MT19937
    g = Harmony(seed=42, rng="R", ref_arith=1); g.setup(**skw)
    c = OracleHarmony(mask=0, seed=42, rng=1); c.setup(**skw)
    g.init_cluster_cpp(); c.init_cluster_cpp()
    assert relfro(g.Y, c.Y) < 1e-4
    ig, ic = _iterate(g, 4), _iterate(c, 4)
    s = _report(g, c)
    print("ref_arith, R stream:", s)
    assert ig == ic and s["Z_rel"] <= 2e-5 and s["clear_flips"] == 0 and s["obj_rel"] <= 2e-5, (ig, ic, s)
    # (b) injected orders, one clustering pass + one correction
    rng = np.random.default_rng(0)
    g = Harmony(seed=1, ref_arith=1); g.setup(**skw)
    c = OracleHarmony(mask=0, seed=1); c.setup(**skw)
    Y0 = g.kmeans_centers()
    g.init_cluster_cpp(Y0); c.init_cluster_cpp(Y0)
    for _ in range(3):
        o = rng.permutation(N).astype(np.int64)
        g.push_update_order(o); c.push_update_order(o)
    assert g.cluster_cpp() == 0 and c.cluster_cpp() == 0
    g.moe_correct_ridge_cpp(); c.moe_correct_ridge_cpp()
    s = _report(g, c)
    print("ref_arith, injected orders:", s)
    assert s["Z_rel"] <= 1e-5 and s["O_rel"] <= 1e-5 and s["clear_flips"] == 0 and s["obj_rel"] <= 2e-5, s


@pytest.mark.timeout(600, method="thread")
def test_reference_arithmetic_against_the_openblas_point(cell_lines):
    """The product is tuned to the oracle's DEFAULT point of the faithful interval (the operation orders the oracle header fixes).  The other
    point that matters is what the reference's binary runs when RcppArmadillo sits on OpenBLAS: norms and column sums as Armadillo's
    op_norm / op_sum form them (two accumulators below 32 elements, sasum / snrm2 above), arma::inv as spotrf + spotri, one rounded product per
    non-zero in the several-covariate apply, the distance GEMM through sgemm -- all from the real OpenBLAS 0.3.28 inside scipy (oracle
    liberty bits 2 + 6 + 7; the reference's own sources take the same routes and equal the oracle bit for bit: tests/test_oracle_ref.py).
    The GPU's reference-arithmetic mode has to be as close to THAT point as the two points are to each other: same centres, same
    shuffles, to convergence; one covariate at 100k cells, two crossed covariates (the reference's fixture), three nested covariates.
    Measured on the final tree (profiles/r5_gpu_vs_openblas_point.json): 2.1e-6 / 2.0e-6 / 5.9e-6 against 2.0e-6 / 2.3e-6 / 5.9e-6 between the
    points, no assignment flip at a margin of 1e-5."""
    if not (orc.use_lapack() and orc.use_openblas(1)):
        pytest.skip("scipy's bundled OpenBLAS not found")
    cases = [(synth(100000, d=50, levels=(10,), seed=7)[:2], 100, 10, 1e-5),
             ((cell_lines["pcs"], {"dataset": cell_lines["dataset_levels"][cell_lines["dataset"]],
                                   "cell_type": cell_lines["cell_type_levels"][cell_lines["cell_type"]]}), 20, 4, 1e-5),
             (synth(40000, d=50, levels=(4, 12, 24), seed=5, nested=True)[:2], 60, 4, 3e-5)]
    try:
        for (Z, meta), K, max_iter, tol in cases:
            vu = list(meta)
            skw, _ = prepare_setup_args(Z, meta, vu, nclust=K)
            g = Harmony(seed=3, ref_arith=1)
            g.setup(**skw)
            Y0 = g.kmeans_centers()
            g.init_cluster_cpp(Y0)
            ig = _iterate(g, max_iter)
            c = OracleHarmony(mask=0, seed=3, liberty=128 | (4 | 64 if len(vu) > 1 else 0))
            c.setup(**skw)
            c.init_cluster_cpp(Y0)
            ic = _iterate(c, max_iter)
            bad = np.where(g.R.argmax(axis=0) != c.R.argmax(axis=0))[0]
            srt = np.sort(c.R[:, bad], axis=0) if bad.size else np.zeros((2, 0))
            og, oc = np.asarray(g.objective_kmeans), np.asarray(c.objective_kmeans)
            s = dict(Z_rel=relfro(g.getZcorr(), c.getZcorr()), R_maxabs=float(np.abs(g.R - c.R).max()), flips=int(bad.size),
                     clear_flips=int(((srt[-1] - srt[-2]) >= 1e-5).sum()) if bad.size else 0, obj_len=(len(og), len(oc)))
            print("ref_arith vs the OpenBLAS point, %d covariate(s):" % len(vu), s)
            assert ig == ic and len(og) == len(oc), (ig, ic, s)
            assert float(np.max(np.abs(og - oc) / np.abs(oc))) <= 1e-4, s
            assert s["Z_rel"] <= tol and s["clear_flips"] == 0, s
    finally:
        orc.use_openblas(4)
