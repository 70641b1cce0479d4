"""Round-2 GPU parity tests (-m gpu): the arithmetic-gap table (GPU vs the oracle in BOTH arithmetic modes at 20k / 100k /
1M cells, incl. BASELINE configs[2] exactly), the configs[4] shape at 200k cells, k-means initialisation at the headline
shape and on the fallback kernels, the R-compatible random stream end to end, reference-arithmetic ridge statistics,
single-precision / device-pointer ingest and egress."""
import json
import os
import threading
import time

import numpy as np
import pytest

from harmony_amd import Harmony, prepare_setup_args
from helpers import synth
from oracle import oracle as orc
from oracle.oracle import OracleHarmony
from parity import assert_parity, compare_state, relfro, run_both

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _iterate(obj, max_iter=10):
    it = 0
    for it in range(1, max_iter + 1):
        assert obj.cluster_cpp() == 0
        obj.moe_correct_ridge_cpp()
        if obj.check_convergence(1):
            break
    return it


def _flips(Ra, Rb, margin):
    aa, ab = Ra.argmax(axis=0), Rb.argmax(axis=0)
    bad = np.where(aa != ab)[0]
    if bad.size == 0:
        return 0, 0
    srt = np.sort(Rb[:, bad], axis=0)
    return int(bad.size), int(((srt[-1] - srt[-2]) >= margin).sum())


# ---------------------------------------------------------------- VERDICT r1 item 1 + 2a: the arithmetic-gap table
@pytest.mark.timeout(1500, method="thread")
@pytest.mark.parametrize("N", [20000, 100000, 1000000, 2000000])
def test_arithmetic_gap_table(N):
    """GPU (default: exact accumulators) and GPU (ref_arith = 1: every accumulator group in the reference's fp32 operation order --
    ridge statistics, O / E tables, objective sums, closed-form inverse) against the oracle in accurate (fp64 accumulators) AND
    faithful (the reference's fp32 arithmetic) mode: N x 50, K = 100, 10 batches, reference defaults, to convergence -- N = 1M is
    BASELINE configs[2] exactly, N = 2M pins a faithful comparison beyond it in the regular suite.  Shared random choices: the GPU's k-means centres, the documented Feistel block partitions (same
    seed).  The numbers go to gpurun_out/r6_parity_table_<N>.json (copied to profiles/ and quoted in DESIGN.md section 2)."""
    K, B, seed = 100, 10, 3
    Z, meta, _ = synth(N, d=50, levels=(B,), seed=7)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
    g = Harmony(seed=seed)
    g.setup(**skw)
    Y0 = g.kmeans_centers()
    res, timing = {}, {}

    def gpu(name, ref_arith):
        o = g if ref_arith == 0 else Harmony(seed=seed, ref_arith=ref_arith)
        if ref_arith:
            o._set("seq_stats", 1)
            o.setup(**skw)
        t0 = time.time()
        o.init_cluster_cpp(Y0)
        it = _iterate(o)
        timing[name] = time.time() - t0
        res[name] = dict(Z=o.getZcorr(), R=o.R, it=it, obj=o.objective_kmeans.copy(), rounds=o.kmeans_rounds.copy())
        if ref_arith:
            res[name]["seq_residual"] = float(o._scalar("seq:residual"))

    def cpu(name, mask, liberty=0):
        o = OracleHarmony(mask=mask, seed=seed, liberty=liberty)
        o.setup(**skw)
        t0 = time.time()
        o.init_cluster_cpp(Y0)
        it = _iterate(o)
        timing[name] = time.time() - t0
        res[name] = dict(Z=o.getZcorr(), R=o.R, it=it, obj=o.objective_kmeans.copy(), rounds=o.kmeans_rounds.copy())

    orc.use_openblas(4)
    th = [threading.Thread(target=cpu, args=("oracle_accurate", 15)), threading.Thread(target=cpu, args=("oracle_faithful", 0))]
    [t.start() for t in th]
    gpu("gpu", 0)
    gpu("gpu_ref_arith", 1)
    gpu("gpu_ref_arith2", 2)       # (round 6) "ref_arith" = 2: the reference's accumulators for the objective, the ridge statistics and the inverse, exact O / E tables
    [t.join() for t in th]
    if N == 1000000:       # BASELINE configs[2]: the width of "faithful" itself, measured next to the GPU -- the same oracle with ONE of its liberties flipped
        # (L1 sums with Armadillo's two accumulators, oracle header; on its own AFTER the other two: a third concurrent caller of the
        #  bundled OpenBLAS's sgemm came back with a different trajectory at this size -- 3 iterations instead of 5)
        cpu("oracle_faithful_liberty1", 0, 1)
    assert {"gpu", "gpu_ref_arith", "oracle_accurate", "oracle_faithful"} <= set(res)
    rows = {}
    pairs = [("gpu", "oracle_accurate"), ("gpu", "oracle_faithful"), ("gpu_ref_arith", "oracle_faithful"),
             ("gpu_ref_arith", "oracle_accurate"), ("oracle_faithful", "oracle_accurate"), ("gpu_ref_arith2", "oracle_faithful"), ("gpu_ref_arith2", "gpu_ref_arith")]
    if "oracle_faithful_liberty1" in res:
        pairs.append(("oracle_faithful_liberty1", "oracle_faithful"))
    for a, b in pairs:
        ra, rb = res[a], res[b]
        n = min(len(ra["obj"]), len(rb["obj"]))
        f, f5 = _flips(ra["R"], rb["R"], 1e-5)
        _, f4 = _flips(ra["R"], rb["R"], 1e-4)
        rows["%s_vs_%s" % (a, b)] = {
            "Z_rel": relfro(ra["Z"], rb["Z"]), "R_maxabs": float(np.abs(ra["R"] - rb["R"]).max()),
            "argmax_diff": f, "argmax_diff_margin_ge_1e-5": f5, "argmax_diff_margin_ge_1e-4": f4, "iterations": [int(ra["it"]), int(rb["it"])],
            "objective_rel_max": float(np.max(np.abs(ra["obj"][:n] - rb["obj"][:n]) / np.abs(rb["obj"][:n]))),
            "final_objective": [float(ra["obj"][-1]), float(rb["obj"][-1])]}
    out = {"workload": {"cells": N, "pcs": 50, "clusters": K, "batches": B}, "seconds": timing, "pairs": rows,
           "seq_residual": res["gpu_ref_arith"]["seq_residual"]}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "r6_parity_table_%d.json" % N), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
    ga, gf, rf = rows["gpu_vs_oracle_accurate"], rows["gpu_vs_oracle_faithful"], rows["gpu_ref_arith_vs_oracle_faithful"]
    # (1) The parity target proper -- same algorithm, exact accumulators: the STRICT bar at every size of the suite (round 4 had loosened it at
    # exactly BASELINE's 1M cells): Z_corr 2e-5, max |dR| within tests/parity.py's TOL_R, NO hard assignment differs where the oracle's top-2
    # margin is 1e-5 or more, objective series 1e-4, same iteration count.
    # (max |dR|: TOL_R = 5e-5 below 1M cells; from 1M on 1e-4 -- the largest difference always sits in the same cell of one small, ill-conditioned
    #  cluster and moved from 3.1e-5 to 6.1e-5 when the tables' fixed-point quantum changed in round 5, with Z_corr at 2.3e-7 both times)
    assert ga["Z_rel"] <= 2e-5 and ga["R_maxabs"] <= (5e-5 if N < 1000000 else 1e-4) and ga["argmax_diff_margin_ge_1e-5"] == 0 and ga["iterations"][0] == ga["iterations"][1], ga
    assert ga["objective_rel_max"] <= 1e-4, ga
    # (2) The reference's own arithmetic: with its operation order reproduced the GPU follows the reference's fp32 results, not the exact ones:
    # Z_corr 1e-5 (north_star: 1e-4), objective 1e-4, same iterations.  Hard assignments and max |dR|: a faithful fp32 run is a CHAOTIC
    # trajectory at the 1e-4 level of R -- the oracle moves its own R by 1.5e-4 .. 2.8e-4 (Z_corr by 1.7 .. 2.4e-6, hundreds of raw flips at 1M)
    # when ONE of its sums is taken in another order that Armadillo / BLAS are free to choose (profiles/r5_oracle_liberties.json; at 1M the row
    # oracle_faithful_liberty1_vs_oracle_faithful of this very table), and iterating the GPU's restarted sums to their bit-exact fixed point leaves
    # its distance to the oracle where it is (tools/strict_probe.py, profiles/r5_strict_probe_1M_*.json: 1.9e-6 / 3.3e-4 / 1 with seq_strict, the
    # largest |dR| always the same cell of one small cluster).  So below 1M cells: no flip at margin 1e-5; from 1M on: none at margin 1e-4, at most
    # N / 10^5 at 1e-5 (what the liberties themselves produce), max |dR| inside the width of "faithful" (1e-3).
    def flips_ok(row):
        if N < 1000000:
            return row["argmax_diff_margin_ge_1e-5"] == 0
        return row["argmax_diff_margin_ge_1e-4"] == 0 and row["argmax_diff_margin_ge_1e-5"] <= N // 100000
    assert rf["Z_rel"] <= 1e-5 and rf["R_maxabs"] <= 1e-3 and flips_ok(rf) and rf["iterations"][0] == rf["iterations"][1], rf
    assert rf["objective_rel_max"] <= 1e-4, rf
    if "oracle_faithful_liberty1_vs_oracle_faithful" in rows:      # the GPU is as close to the oracle as the oracle is to itself
        lf = rows["oracle_faithful_liberty1_vs_oracle_faithful"]
        assert rf["Z_rel"] <= 3 * lf["Z_rel"], (rf, lf)       # (max |dR| is one cell of one small cluster on both sides: reported, 5e-5 .. 1.4e-3 across liberties and centres)
    # (3) "ref_arith" = 2 -- every group but the O / E tables: north_star's 1e-4 against the faithful oracle with the same iteration counts; what the exact tables
    # leave (the reference's -= / += drift of O and E) is 4e-6 at 100k cells, 4e-5 at 1M, 6e-5 at 2M.  Hard assignments: reported (the drift it does not follow moves
    # R by a few 1e-4: clear flips in the tens at 1M), bounded at N / 10^4.
    r2 = rows["gpu_ref_arith2_vs_oracle_faithful"]
    assert r2["Z_rel"] <= 1e-4 and r2["iterations"][0] == r2["iterations"][1] and r2["objective_rel_max"] <= 1e-3, r2
    assert r2["argmax_diff_margin_ge_1e-4"] <= max(2, N // 10000), r2
    # (gf -- default GPU vs the reference's fp32 drift -- is REPORTED, not asserted: it is the reference's N-dependent bias)
    assert gf["iterations"][0] == gf["iterations"][1], gf


# ---------------------------------------------------------------- configs[4] shape at 200k cells
@pytest.mark.timeout(1500, method="thread")
def test_config5_shape_200k():
    """BASELINE configs[4] scaled to one test: 200k x 50, K = 200, three nested covariates 8 > 64 > 128 = 200 levels;
    batch-subset ridge path asserted (src/harmony.cpp:440-547)."""
    Z, meta, _ = synth(200000, d=50, levels=(8, 64, 128), seed=11, nested=True)
    orc.use_openblas(4)
    g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=2, nclust=200, seed=5)
    s = assert_parity(g, c, ig, ic)
    assert c.subset_clusters > 0 and int(g._scalar("subset_clusters")) == c.subset_clusters
    print("config5 200k:", s, "subset clusters", c.subset_clusters)


def test_chain_with_three_and_more_tiles_per_wave_equals_launch_per_step(monkeypatch):
    """3M cells: the persistent chain's waves own three and more tiles of a block -- the tile loop in front of the deferred epilogues, rows
    of the LAST tile stored behind the arrival (the suite's other chain cases stop at two tiles per wave) -- against the launch-per-step
    kernels on the same data: same kernels' arithmetic, integer O sums => bit-identical."""
    Z, meta, _ = synth(3000000, d=50, levels=(10,), seed=11)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
    out, Y0 = [], None
    for chain in ("1", "0"):
        monkeypatch.setenv("HMX_CHAIN", chain)
        g = Harmony(seed=3)
        g.setup(**skw)
        if Y0 is None:
            Y0 = g.kmeans_centers()
        g.init_cluster_cpp(Y0)
        for _ in range(2):
            assert g.cluster_cpp() == 0
            g.moe_correct_ridge_cpp()
        out.append((int(g._scalar("chain")), g.getZcorr().copy(), np.array(g.objective_kmeans), np.array(g.O)))
        del g
    (c1, Z1, o1, O1), (c0, Z0, o0, O0) = out
    assert c1 == 1 and c0 == 0
    assert np.array_equal(O1, O0) and len(o1) == len(o0)
    assert relfro(Z1, Z0) < 1e-6 and float(np.max(np.abs(o1 - o0) / np.abs(o0))) < 1e-6


@pytest.mark.parametrize("N,levels,K", [(60000, (8, 64, 128), 200), (1300000, (8, 64, 128), 200), (300000, (10,), 200), (250000, (6, 5), 116), (400000, (10,), 152), (500000, (4, 30), 180)])
def test_wave_pair_chain_equals_launch_per_step(monkeypatch, N, levels, K):
    """BASELINE configs[4]'s shape (K = 200, 8 > 64 > 128 nested levels): the block chain by wave pairs (k_tile MODE 6, round 6: two halves of the clusters on the
    two waves of a pair, a row's normalisation sum exchanged through LDS, several folder workgroups, the penalty rows from memory) against the
    launch-per-step kernels on the same data.  60k cells: at most one tile per pair and block, streams without a tile; 1.3M cells: four and five tiles
    per pair -- the tile loop, the deferred last epilogue, and (fifth tile) the penalty rows fetched outside the wave's LDS cache.  The two paths add the halves
    of a row sum in a different order: R agrees to the last bits, the integer tables to 1e-6, nothing is bit-identical.  300k cells with ONE covariate of 10
    levels: the round-to-round carry of the old contributions is on (tiles keyed by the next block, rounds without R stores) inside the wave-pair chain."""
    Z, meta, _ = synth(N, d=50, levels=levels, seed=11, nested=len(levels) == 3)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)      # (K = 116 / 152 / 180: halves of 4 / 5 / 6 cluster tiles)
    out, Y0 = [], None
    for pair in ("1", "0"):
        monkeypatch.setenv("HMX_CHAIN_PAIR", pair)
        g = Harmony(seed=3)
        g.setup(**skw)
        if Y0 is None:
            Y0 = g.kmeans_centers()
        g.init_cluster_cpp(Y0)
        for _ in range(2):
            assert g.cluster_cpp() == 0
            g.moe_correct_ridge_cpp()
        out.append((int(g._scalar("chain_pair")), int(g._scalar("chain")), g.getZcorr().copy(), np.array(g.objective_kmeans), np.array(g.O), g.getR().copy() if N <= 100000 else None))
        if levels == (10,) and K == 200:
            assert g._scalar("sold_carry") == 1
        del g
    (p1, c1, Z1, o1, O1, R1), (p0, c0, Z0, o0, O0, R0) = out
    assert p1 == 1 and c1 == 1 and p0 == 0 and c0 == 0
    assert len(o1) == len(o0) and float(np.max(np.abs(o1 - o0) / np.abs(o0))) < 1e-6
    assert relfro(Z1, Z0) < 1e-6 and np.allclose(O1, O0, rtol=1e-5, atol=1e-3)
    if R1 is not None:
        assert float(np.max(np.abs(R1 - R0))) < 1e-5


@pytest.mark.parametrize("shape", [dict(N=30000, K=100, levels=(10,)), dict(N=20000, K=60, levels=(3, 4)), dict(N=9000, K=24, levels=(5,))])
@pytest.mark.parametrize("chain", ["1", "0"])
def test_carried_old_contributions_equal_a_fresh_pass(monkeypatch, shape, chain):
    """update_R removes a block's cells from O before it updates them (src/harmony.cpp:312-313).  With the library's own shuffle
    every tile is keyed by its cells' block of the NEXT round, and the tile kernels file a tile's new R sums as that block's old
    contribution -- the next round then needs no pass over R.  Forced on (HMX_SOLD_CARRY=1; by default it starts at ~800k cells)
    against the per-round pass (=0), persistent chain and launch-per-step kernels: bit-identical O and R (integer sums), same
    iteration count, Z_corr to 1e-6; and O equals the direct sum over R."""
    Z, meta, _ = synth(shape["N"], d=30, levels=shape["levels"], seed=31)
    vars_use = list(meta)
    skw, _ = prepare_setup_args(Z, meta, vars_use, nclust=shape["K"])
    monkeypatch.setenv("HMX_CHAIN", chain)
    out = []
    for carry in ("1", "0"):
        monkeypatch.setenv("HMX_SOLD_CARRY", carry)
        g = Harmony(seed=6)
        g.setup(**skw)
        g.init_cluster_cpp()
        it = 0
        for it in range(1, 4):
            assert g.cluster_cpp() == 0
            g.moe_correct_ridge_cpp()
            if g.check_convergence(1):
                break
        assert g.cluster_cpp() == 0
        assert int(g._scalar("sold_carry")) == int(carry)
        assert (g._scalar("carried_rounds") > 0) == (carry == "1")
        out.append((it, g.O.copy(), g.R.copy(), g.getZcorr().copy()))
    (ia, Oa, Ra, Za), (ib, Ob, Rb, Zb) = out
    assert ia == ib
    np.testing.assert_array_equal(Oa, Ob)
    np.testing.assert_array_equal(Ra, Rb)
    assert relfro(Za, Zb) < 1e-6
    lab = meta[vars_use[0]]
    L0 = shape["levels"][0]
    Odirect = np.stack([Ra[:, lab == b].sum(axis=1) for b in range(L0)], axis=1)
    np.testing.assert_allclose(Oa[:, :L0], Odirect, rtol=2e-6, atol=1e-3)


@pytest.mark.parametrize("shape", [dict(N=30000, K=100, levels=(10,)), dict(N=20000, K=60, levels=(3, 4)), dict(N=250000, K=100, levels=(3,)), dict(N=70001, K=40, levels=(2,))])
def test_sort_free_shuffle_equals_the_counting_sort(monkeypatch, shape):
    """The padded block order of a round (update_R's shuffle, src/harmony.cpp:272-300) is built without a sort on one GPU: the cells of
    a block are the inverse images of its positions under the shuffle's bijection (k_shuf_count / scan / place).  Against the counting
    sort of the same (seed, round) permutation: the same cells in the same (block, combination, next block) bins, so every
    order-independent result -- O, R, Z_corr -- is bit-identical, with the same iteration count.  Sizes that are not multiples of the part
    size, one and several combinations, a domain with heavy cycle walking (70001 cells in 2^18)."""
    Z, meta, _ = synth(shape["N"], d=30, levels=shape["levels"], seed=33)
    vars_use = list(meta)
    skw, _ = prepare_setup_args(Z, meta, vars_use, nclust=shape["K"])
    monkeypatch.setenv("HMX_SOLD_CARRY", "1")
    out = []
    for inv in ("1", "0"):
        monkeypatch.setenv("HMX_SHUFFLE_INV", inv)
        g = Harmony(seed=9)
        g.setup(**skw)
        assert int(g._scalar("shuffle_inv")) == int(inv)
        g.init_cluster_cpp()
        it = 0
        for it in range(1, 4):
            assert g.cluster_cpp() == 0
            g.moe_correct_ridge_cpp()
            if g.check_convergence(1):
                break
        assert g.cluster_cpp() == 0
        assert g._scalar("carried_rounds") > 0
        out.append((it, g.O.copy(), g.R.copy(), g.getZcorr().copy(), list(g.objective_kmeans)))
    (ia, Oa, Ra, Za, ja), (ib, Ob, Rb, Zb, jb) = out
    assert ia == ib and len(ja) == len(jb)
    np.testing.assert_array_equal(Oa, Ob)
    np.testing.assert_array_equal(Ra, Rb)
    np.testing.assert_array_equal(Za, Zb)
    # (the objective's two per-cell sums are fp64 sums of per-TILE fp32 partials: which cells share a tile depends on the order inside a bin, which neither
    #  shuffle form specifies (slots come from LDS atomics) -- differences of ~1e-9 relative are its rounding; 1e-9 itself failed once in round 5)
    np.testing.assert_allclose(ja, jb, rtol=1e-7)


@pytest.mark.parametrize("K", [12, 60, 64, 104, 116, 128, 148, 192, 196, 200, 208, 224, 256])      # (196 .. 224: the wave-pair chain, halves of 100 + 96, 100 + 100, 104 + 104, 112 + 112 clusters)
def test_every_cluster_tile_shape(K):
    """The tile kernels deal clusters to MFMA columns in quads (a lane's columns are consecutive clusters: 16 / 12 / 8 / 4-byte R
    stores, 16-byte penalty reads, one contribution atomic per 64 clusters).  Cluster counts that end inside a full quad, inside
    the 1-, 2- and 3-tile remainder, on a tile boundary, with partially valid lanes, for 1..16 cluster tiles: whole run vs oracle."""
    Z, meta, _ = synth(6000, d=20, levels=(3, 4), seed=K)
    g, c, ig, ic = run_both(Z, meta, list(meta), max_iter=2, nclust=K, seed=K + 1)
    assert_parity(g, c, ig, ic)


# ---------------------------------------------------------------- k-means initialisation
@pytest.mark.parametrize("env", [{}, {"HMX_TILE_IMPL": "v1"}])
def test_kmeans_centers_headline_shape(monkeypatch, env):
    """kmeans_centers (src/utils.cpp:10-64: seeding race + 10 Lloyd iterations) at d = 50, K = 100, 100k cells against the
    oracle, on the MFMA tile kernels and on the cluster-lane fallbacks (k_seed_probe, k_lloyd)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    Z, meta, _ = synth(100000, d=50, levels=(10,), seed=9)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
    g = Harmony(seed=17)
    g.setup(**skw)
    Yg = g.kmeans_centers()
    c = OracleHarmony(accurate=True, seed=17)
    c.setup(**skw)
    c.init_cluster_cpp()                     # kmeans_centers + normalise
    sg, sc = g._get("seed_cells").astype(np.int64), c._get("seed_cells").astype(np.int64)
    assert np.array_equal(sg, sc), np.where(sg != sc)[0]
    Yg_n = Yg / np.linalg.norm(Yg, axis=0, keepdims=True)
    assert relfro(Yg_n, c.Y) < 1e-4, relfro(Yg_n, c.Y)


def test_kmeans_duplicate_winner_resample():
    """More anchors than the data can serve distinctly: several anchors win the same cell and must be re-sampled among the
    cells not yet chosen, in cluster order (src/utils.cpp:38-43)."""
    rng = np.random.default_rng(5)
    base = rng.normal(size=(6, 5))
    Z = np.repeat(base, 10, axis=0) + 1e-3 * rng.normal(size=(60, 5))      # 6 tight groups of 10 cells
    meta = {"b": np.arange(60) % 2}
    skw, _ = prepare_setup_args(Z, meta, "b", nclust=40)
    for seed in (1, 2, 3):
        g = Harmony(seed=seed)
        g.setup(**skw)
        Yg = g.kmeans_centers()
        c = OracleHarmony(accurate=True, seed=seed)
        c.setup(**skw)
        c.init_cluster_cpp()
        sg, sc = g._get("seed_cells").astype(np.int64), c._get("seed_cells").astype(np.int64)
        assert np.array_equal(sg, sc), (seed, np.where(sg != sc)[0])       # the race and the re-sampling pick the same cells
        assert len(set(sg.tolist())) == 40
        # the centres themselves are tie-sensitive here BY CONSTRUCTION (40 centres in 6 tight groups): loose bar
        Yg_n = Yg / np.linalg.norm(Yg, axis=0, keepdims=True)
        assert relfro(Yg_n, c.Y) < 2e-3, (seed, relfro(Yg_n, c.Y))


# ---------------------------------------------------------------- R-compatible randomness, end to end
@pytest.mark.parametrize("case", ["cell_lines", "synth20k"])
def test_r_compatible_stream_end_to_end(case, cell_lines):
    """rng = R on both sides, NO shared random choices: seeds (K + K*N uniforms) and every round's arma::shuffle come from
    MT19937 seeded like set.seed(seed) -- two independent implementations (product / oracle) must walk the same stream and
    reach the same corrected embedding."""
    if case == "cell_lines":
        Z = cell_lines["pcs"]
        meta = {"dataset": cell_lines["dataset_levels"][cell_lines["dataset"]]}
        K, vu = 20, "dataset"
    else:
        Z, meta, _ = synth(20000, d=50, levels=(5,), seed=3)
        K, vu = 50, "cov0"
    skw, _ = prepare_setup_args(Z, meta, vu, nclust=K)
    g = Harmony(seed=42, rng="R")
    g.setup(**skw)
    c = OracleHarmony(accurate=True, seed=42, rng=1)
    c.setup(**skw)
    g.init_cluster_cpp()
    c.init_cluster_cpp()
    assert relfro(g.Y, c.Y) < 1e-4, relfro(g.Y, c.Y)
    ig, ic = _iterate(g, 4), _iterate(c, 4)
    s = assert_parity(g, c, ig, ic)
    print("R-stream parity:", s)
    # and the stream matters: another seed gives another partition history
    g2 = Harmony(seed=43, rng="R")
    g2.setup(**skw)
    g2.init_cluster_cpp()
    assert relfro(g2.Y, g.Y) > 1e-3


def _run_shards_as_threads(Z, meta, vars_use, G, K, seed, Y0, max_iter=10):
    """G handles (one thread each) holding contiguous cell shards of ONE job on this GPU: the production sharded code path, the all-reduce hook a thread
    rendezvous (as tests/test_gpu_parity.py::_run_sharded; any covariate structure, shared initial centroids).  Returns Z_corr (d x N), R (K x N),
    kmeans_rounds, harmony iterations."""
    import ctypes
    import threading
    from harmony_amd.dist import shard_bounds
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    N = Z.shape[0]
    vars_use = [vars_use] if isinstance(vars_use, str) else list(vars_use)
    bounds = shard_bounds(N, G)
    barrier = threading.Barrier(G)
    slots, results, errors = [None] * G, [None] * G, []
    nlev = {v: int(np.max(meta[v])) + 1 for v in vars_use}
    N_b = np.concatenate([np.bincount(meta[v], minlength=nlev[v]).astype(float) for v in vars_use])

    def hook_for(rank):
        def hook(user, buf, count, dtype, stream):
            assert hip.hipDeviceSynchronize() == 0
            host = np.empty(count, dtype=np.float64 if dtype == 1 else np.int64)
            assert hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) == 0
            slots[rank] = host
            barrier.wait()
            st = np.stack(slots)
            red = st.min(axis=0) if dtype == 2 else st.sum(axis=0)
            barrier.wait()
            assert hip.hipMemcpy(buf, red.ctypes.data, red.nbytes, 1) == 0
            barrier.wait()
            return 0
        return hook

    def work(rank):
        try:
            lo, hi = bounds[rank]
            m = {k: v[lo:hi] for k, v in meta.items()}
            skw, _ = prepare_setup_args(Z[lo:hi], m, vars_use, nclust=K, N_b=N_b, levels={v: np.arange(nlev[v]) for v in vars_use})
            g = Harmony(seed=seed)
            g.set_shard(rank, G, lo, N, hook_for(rank))
            g.setup(**skw)
            g.init_cluster_cpp(Y0)
            it = 0
            for it in range(1, max_iter + 1):
                assert g.cluster_cpp() == 0
                g.moe_correct_ridge_cpp()
                if g.check_convergence(1):
                    break
            results[rank] = (g.getZcorr(), g.R, np.array(g.kmeans_rounds), it)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]
    return (np.concatenate([r[0] for r in results], axis=1), np.concatenate([r[1] for r in results], axis=1), results[0][2], results[0][3])


@pytest.mark.timeout(1500, method="thread")
@pytest.mark.parametrize("case", ["configs3_share", "configs4_shape"])
def test_sharded_runs_return_the_reference_in_double_precision(case):
    """WHICH result a multi-GPU run returns ("Which result a sharded run returns").  The reference's fp32 accumulators are
    sequential sums over ALL cells in orders that interleave the shards cell by cell (a block's shuffled order, src/harmony.cpp:285-313), so their rounding
    cannot be formed shard-locally; a sharded run therefore targets what the reference's own sources compute when built with the reference's own precision
    switch (-DHARMONY_SCALAR_DOUBLE, src/types.h:5-9): exact accumulators, identical for every shard count.  Two shards of one job on this GPU (the
    production sharded path) against oracle/_ref/libharmony_ref_f64.so, same centroids, same shuffles, to convergence, at the two 8-GPU configs' shapes:
    configs[3]'s per-rank share scaled by ten (125k cells x 50 PCs, K = 100, 20 batches; the full 1.25M share with HMX_SLOW=1) and configs[4]'s shape
    (K = 200, three nested covariates 8 > 64 > 128)."""
    from oracle import ref as oref
    if not oref.available():
        pytest.skip("oracle/_ref/libharmony_ref_f64.so did not travel with the tree")
    from oracle.oracle import feistel_order
    seed = 3
    if case == "configs3_share":
        N, K = (1250000 if os.environ.get("HMX_SLOW") == "1" else 125000), 100
        Z, meta, _ = synth(N, d=50, levels=(20,), seed=7)
    else:
        N, K = 30000, 200
        Z, meta, _ = synth(N, d=50, levels=(8, 64, 128), seed=7, nested=True)
    vars_use = list(meta)
    skw, _ = prepare_setup_args(Z, meta, vars_use, nclust=K)
    one = Harmony(seed=seed)
    one.setup(**skw)
    Y0 = one.kmeans_centers()
    one.init_cluster_cpp(Y0)
    i1 = _iterate(one)
    Zs, Rs, rounds_s, i_s = _run_shards_as_threads(Z, meta, vars_use, 2, K, seed, Y0)
    r = oref.RefHarmony(seed=seed, double=True)
    r.setup(**skw)
    r.init_cluster_cpp(Y0)
    done, i64 = 0, 0
    for i64 in range(1, 11):
        r.clear_update_orders()
        for k in range(4):
            r.push_update_order(feistel_order(seed, done + k, N))
        assert r.cluster_cpp() == 0
        done = int(np.sum(r.kmeans_rounds))
        r.moe_correct_ridge_cpp()
        if r.check_convergence(1):
            break
    Z64, R64 = r.getZcorr(), r.R
    s = dict(case=case, cells=N, sharded_vs_reference_double=relfro(Zs, Z64), one_gpu_vs_reference_double=relfro(one.getZcorr(), Z64), sharded_vs_one_gpu=relfro(Zs, one.getZcorr()),
             R_maxabs=float(np.abs(Rs - R64).max()), clear_flips=_flips(Rs, R64, 1e-5)[1], iterations=(i_s, i1, i64))
    print("two shards vs the reference's sources in double precision:", s)
    assert i_s == i64 == i1 and np.array_equal(rounds_s, r.kmeans_rounds), s
    assert s["sharded_vs_reference_double"] <= 5e-6 and s["clear_flips"] == 0 and s["R_maxabs"] <= 1e-4, s
    assert s["sharded_vs_one_gpu"] <= 1e-6, s


# ---------------------------------------------------------------- the default mode against the reference's own sources in double precision
def test_default_mode_against_the_reference_in_double_precision():
    """oracle/_ref/libharmony_ref_f64.so: the reference's own harmony.cpp / utils.cpp built with the reference's own precision switch
    (-DHARMONY_SCALAR_DOUBLE, src/types.h:5-9) over oracle/shim/ -- what the reference's algorithm computes without its fp32 rounding
    (tests/test_oracle_ref.py).  The product's DEFAULT mode (fp32 state, exact accumulators), same centroids, same shuffles (injected into the
    reference's arma::shuffle), to convergence: a few 1e-7 from it, no assignment flipped, same iteration count -- while the reference's own
    single-precision build is two orders of magnitude further from its double-precision one."""
    from oracle import ref as oref
    if not oref.available():
        pytest.skip("oracle/_ref/libharmony_ref_f64.so did not travel with the tree")
    from oracle.oracle import feistel_order
    N, K, seed = 20000, 100, 3
    Z, meta, _ = synth(N, d=50, levels=(10,), seed=7)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
    g = Harmony(seed=seed)
    g.setup(**skw)
    Y0 = g.kmeans_centers()
    g.init_cluster_cpp(Y0)
    ig = _iterate(g)

    def walk_ref(double):
        r = oref.RefHarmony(seed=seed, double=double)
        r.setup(**skw)
        r.init_cluster_cpp(Y0)
        done, it = 0, 0
        for it in range(1, 11):
            r.clear_update_orders()
            for k in range(4):
                r.push_update_order(feistel_order(seed, done + k, N))
            assert r.cluster_cpp() == 0
            done = int(np.sum(r.kmeans_rounds))
            r.moe_correct_ridge_cpp()
            if r.check_convergence(1):
                break
        return r, it
    r64, i64 = walk_ref(True)
    r32, i32 = walk_ref(False)
    Zg, Z64, Z32 = g.getZcorr(), r64.getZcorr(), r32.getZcorr()
    fl = _flips(g.R, r64.R, 1e-5)[1]
    s = dict(gpu_default_vs_reference_double=relfro(Zg, Z64), reference_single_vs_reference_double=relfro(Z32, Z64), gpu_default_vs_reference_single=relfro(Zg, Z32),
             R_maxabs=float(np.abs(g.R - r64.R).max()), clear_flips=fl, iterations=(ig, i64, i32))
    print("default mode vs the reference's sources in double precision:", s)
    assert ig == i64 and np.array_equal(g.kmeans_rounds, r64.kmeans_rounds), s
    assert s["gpu_default_vs_reference_double"] <= 2e-6 and fl == 0 and s["R_maxabs"] <= 5e-5, s
    assert s["reference_single_vs_reference_double"] >= 10 * s["gpu_default_vs_reference_double"], s


# ---------------------------------------------------------------- the .Call glue, executed (against an emulated R API)
def test_r_glue_executes_the_whole_sequence():
    """r/harmony_mi355x_glue.c linked with tests/stubs/r_emul.c (R is not installed: an emulation of the R C API calls the glue makes) and
    driven exactly like r/harmony_mi355x.R drives it (tests/r_emul.py: .Call names through the registered table, R-typed arguments,
    results rebuilt from numeric vectors + dims): setup -> init -> cluster / correct until convergence -> getters must equal the ctypes
    path bit for bit; the fp32 seam both ways; R's own stream through unif_rand between GetRNGstate / PutRNGstate; a user interrupt;
    the N < 40 warning through Rf_warning; the finalizer."""
    import r_emul
    Z, meta, _ = synth(20000, d=30, levels=(4,), seed=11)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=40)
    a = Harmony(seed=7)
    a.setup(**skw)
    a.init_cluster_cpp()
    ia = _iterate(a, 5)
    g = r_emul.GlueHarmony(seed=7)
    g.setup(**skw)
    assert (g.get("N")[0], g.get("d")[0], g.get("K")[0], g.get("B")[0]) == (20000, 30, 40, 4)
    g.init_cluster_cpp()
    ig = _iterate(g, 5)
    assert ig == ia and np.array_equal(g.kmeans_rounds, a.kmeans_rounds)
    for name in ("R", "O", "E", "Y", "objective_kmeans"):
        np.testing.assert_array_equal(getattr(g, name), getattr(a, name), err_msg=name)
    np.testing.assert_array_equal(g.getZcorr(), a.getZcorr())
    np.testing.assert_array_equal(g.getLambda(), a.getLambda())
    # fp32 seam: a float32 object's bit matrix in, fp32 bits out
    np.testing.assert_array_equal(g.getZcorr(single=True), a.getZcorr().astype(np.float32))
    g32 = r_emul.GlueHarmony(seed=7)
    g32.setup(single=True, **skw)
    g32.init_cluster_cpp()
    assert _iterate(g32, 5) == ia
    np.testing.assert_array_equal(g32.getZcorr(), a.getZcorr())          # (the library rounds the doubles to fp32 first thing, as the reference does)
    # vignette: max_iter_kmeans is writable
    g32.set_max_iter_kmeans(2)
    assert g32.get("max_iter_kmeans")[0] == 2
    # R's own stream: set.seed(42) in the "interpreter", the library draws through unif_rand -- same run as the built-in R-compatible generator
    b = Harmony(seed=42, rng="R")
    b.setup(**skw)
    b.init_cluster_cpp()
    ib = _iterate(b, 3)
    r_emul.load().emul_set_seed(42)
    h = r_emul.GlueHarmony(r_rng=True)
    h.setup(**skw)
    h.init_cluster_cpp()
    assert r_emul.load().emul_rng_brackets() == 1                        # (counted per .Call: emul_clear resets it)
    assert _iterate(h, 3) == ib
    np.testing.assert_array_equal(h.getZcorr(), b.getZcorr())
    np.testing.assert_array_equal(h.R, b.R)
    # Progress::check_abort(): a pending user interrupt ends cluster_cpp with -1 (R/utils.R:26-32 turns that into a message)
    r_emul.load().emul_set_interrupt(1)
    assert h.cluster_cpp() == -1
    # Rcpp::warning -> Rf_warning
    rng = np.random.default_rng(1)
    skw30, _ = prepare_setup_args(rng.normal(size=(30, 8)), {"b": np.arange(30) % 2}, "b", nclust=3)
    w = r_emul.GlueHarmony(seed=1)
    w.setup(**skw30)
    assert "Too few cells" in w.warnings()
    with pytest.raises(r_emul.RError, match="less than 6 cells"):
        skw5, _ = prepare_setup_args(rng.normal(size=(5, 3)), {"b": np.array([0, 1, 0, 1, 0])}, "b", nclust=2)
        r_emul.GlueHarmony(seed=1).setup(**skw5)
    for o in (g, g32, h, w):
        o.release()
    with pytest.raises(r_emul.RError, match="has been destroyed"):
        g.get("N")


# ---------------------------------------------------------------- single precision / device pointers
def test_f32_and_device_pointer_ingest_egress():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    N, d, K = 30011, 50, 40
    Z, meta, _ = synth(N, d=d, levels=(4,), seed=2)
    skw, dm = prepare_setup_args(Z, meta, "cov0", nclust=K)
    ref = Harmony(seed=1)
    ref.setup(**skw)                                        # float64 host = the R seam
    Z32 = np.asfortranarray(skw["Z"], dtype=np.float32)
    a = Harmony(seed=1)
    a.setup(**dict(skw, Z=Z32))                             # float32 host
    np.testing.assert_array_equal(a.getZorig(), ref.getZorig())
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), Z32.nbytes) == 0
    assert hip.hipMemcpy(dptr, Z32.ctypes.data, Z32.nbytes, 1) == 0
    b = Harmony(seed=1)
    b.setup(**dict(skw, Z=(d, N, np.float32, dptr.value)))  # float32 already in HBM
    hip.hipFree(dptr)
    np.testing.assert_array_equal(b.getZorig(), ref.getZorig())
    for o in (ref, b):
        o.init_cluster_cpp(np.asfortranarray(skw["Z"][:, :K]))
        assert o.cluster_cpp() == 0
        o.moe_correct_ridge_cpp()
    Zc = ref.getZcorr()
    np.testing.assert_array_equal(b.getZcorr(), Zc)
    np.testing.assert_array_equal(b.get_matrix("Z_corr", np.float32), Zc.astype(np.float32))
    np.testing.assert_array_equal(b.get_matrix("R", np.float64), ref.R)
    out = C.c_void_p()
    assert hip.hipMalloc(C.byref(out), N * d * 4) == 0
    b.get_matrix("Z_corr", np.float32, device_ptr=out.value)
    back = np.empty((d, N), dtype=np.float32, order="F")
    assert hip.hipMemcpy(back.ctypes.data, out, back.nbytes, 2) == 0
    hip.hipFree(out)
    np.testing.assert_array_equal(back, Zc.astype(np.float32))
    assert ref.timer("ingest_Z") > 0 and ref.timer("egress_Z_corr") > 0


def test_update_order_must_be_a_permutation(cell_lines_small):
    from harmony_amd import HarmonyError
    meta = {"dataset": cell_lines_small["dataset_levels"][cell_lines_small["dataset"]]}
    skw, _ = prepare_setup_args(cell_lines_small["pcs"], meta, "dataset", nclust=5)
    g = Harmony(seed=1)
    g.setup(**skw)
    bad = np.arange(300)
    bad[7] = 8
    with pytest.raises(HarmonyError, match="permutation"):
        g.push_update_order(bad)
    bad[7] = 300
    with pytest.raises(HarmonyError, match="permutation"):
        g.push_update_order(bad)
    g.push_update_order(np.arange(300)[::-1].copy())


def _run_pair_to_convergence(Z, meta, K, seed, gpu_kw, masks, max_iter=10, blas_threads=4):
    """one GPU handle and one oracle per arithmetic mask on the same problem (shared k-means centres, shared documented shuffles), each to
    convergence; the first two oracles run in threads next to the GPU, further ones after them.  Returns {name: dict(Z, R, it, obj, rounds, subset)}."""
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    res, timing = {}, {}
    g0 = Harmony(seed=seed)
    g0.setup(**skw)
    Y0 = g0.kmeans_centers()
    del g0

    def drive(name, o):
        t0 = time.time()
        o.init_cluster_cpp(Y0)
        subset, it = [], 0
        for it in range(1, max_iter + 1):
            assert o.cluster_cpp() == 0
            o.moe_correct_ridge_cpp()
            subset.append(int(o._scalar("subset_clusters")) if hasattr(o, "_scalar") else int(o.subset_clusters))
            if o.check_convergence(1):
                break
        timing[name] = time.time() - t0
        res[name] = dict(Z=o.getZcorr(), R=o.R, it=it, obj=np.array(o.objective_kmeans), rounds=np.array(o.kmeans_rounds), subset=subset)

    def cpu(name, mask, liberty=0):
        o = OracleHarmony(mask=mask, seed=seed, liberty=liberty)
        o.setup(**skw)
        drive(name, o)

    orc.use_openblas(blas_threads)
    # AT MOST TWO oracles call the bundled OpenBLAS at a time.  With three concurrent callers one or two of them came back with a different trajectory -- on some
    # boxes only: the 5M table's second builder run (two faithful oracles off by 1.4e-3 .. 1.8e-3, the accurate one fine) and a run of the 1M-shape test in which
    # the fp64-accumulator oracle itself moved by 3.6e-3 against every GPU variant and against its own earlier runs (round 6; test_arithmetic_gap_table had met
    # the same with its third oracle in round 5 and runs it on its own).  The others follow one by one.
    th = [threading.Thread(target=cpu, args=((nm,) + (mk if isinstance(mk, tuple) else (mk, 0)))) for nm, mk in masks.items()]      # (mask, or (mask, liberty bits))
    [t.start() for t in th[:2]]
    for nm, kw in gpu_kw.items():
        o = Harmony(seed=seed, **kw)
        o.setup(**skw)
        drive(nm, o)
        del o
    [t.join() for t in th[:2]]
    for t in th[2:]:
        t.start(); t.join()
    return res, timing


def _pair_row(ra, rb):
    n = min(len(ra["obj"]), len(rb["obj"]))
    f, f5 = _flips(ra["R"], rb["R"], 1e-5)
    return {"Z_rel": relfro(ra["Z"], rb["Z"]), "R_maxabs": float(np.abs(ra["R"] - rb["R"]).max()), "argmax_diff": f, "argmax_diff_margin_ge_1e-5": f5,
            "iterations": [int(ra["it"]), int(rb["it"])], "objective_rel_max": float(np.max(np.abs(ra["obj"][:n] - rb["obj"][:n]) / np.abs(rb["obj"][:n]))),
            "subset_clusters_per_iteration": [ra["subset"], rb["subset"]], "kmeans_rounds_equal": bool(np.array_equal(ra["rounds"], rb["rounds"]))}


# ---------------------------------------------------------------- configs[4] shape at 1M cells TO CONVERGENCE against the oracle
@pytest.mark.timeout(1500, method="thread")
def test_config5_shape_1M_to_convergence():
    """BASELINE configs[4]'s shape at 1M cells: K = 200, three nested covariates 8 > 64 > 128 = 200 levels, reference defaults, to convergence
    (7 harmony iterations): GPU (default arithmetic) against the oracle with exact accumulators, and GPU REFERENCE ARITHMETIC (round 4:
    ridge statistics of several covariates as sequential fp32 chains incl. the level-pair sums of Phi_Rk * Phi_moe_t, arma::inv as the
    oracle's unblocked fp32 LU, src/harmony.cpp:561-574) against the faithful oracle; the batch-subset ridge path counted per iteration
    (:440-547).  Table -> gpurun_out/r6_parity_c5_1M.json (profiles/)."""
    Z, meta, _ = synth(1_000_000, d=50, levels=(8, 64, 128), seed=11, nested=True)
    # (round 6: + the faithful oracle with ONE of its own sums taken in another order Armadillo is free to choose -- L1 sums with two accumulators,
    #  liberty bit 0: how wide "faithful" is at THIS shape, the yardstick of the reference-arithmetic pair's flips)
    res, timing = _run_pair_to_convergence(Z, meta, 200, 5, {"gpu": {}, "gpu_ref_arith": {"ref_arith": 1}, "gpu_ref_arith2": {"ref_arith": 2}},
                                           {"oracle_accurate": 15, "oracle_faithful": 0, "oracle_faithful_liberty1": (0, 1)})
    rows = {"gpu_vs_oracle_accurate": _pair_row(res["gpu"], res["oracle_accurate"]), "gpu_vs_oracle_faithful": _pair_row(res["gpu"], res["oracle_faithful"]),
            "gpu_ref_arith_vs_oracle_faithful": _pair_row(res["gpu_ref_arith"], res["oracle_faithful"]),
            "gpu_ref_arith2_vs_oracle_faithful": _pair_row(res["gpu_ref_arith2"], res["oracle_faithful"]),
            "oracle_faithful_liberty1_vs_oracle_faithful": _pair_row(res["oracle_faithful_liberty1"], res["oracle_faithful"]),
            "oracle_faithful_vs_oracle_accurate": _pair_row(res["oracle_faithful"], res["oracle_accurate"])}
    out = {"workload": {"cells": 1000000, "pcs": 50, "clusters": 200, "levels": [8, 64, 128], "nested": True}, "seconds": timing, "pairs": rows}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "r6_parity_c5_1M.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
    ga = rows["gpu_vs_oracle_accurate"]
    assert ga["Z_rel"] <= 2e-5 and ga["argmax_diff_margin_ge_1e-5"] == 0 and ga["iterations"][0] == ga["iterations"][1] and ga["kmeans_rounds_equal"], ga
    assert ga["objective_rel_max"] <= 1e-4, ga
    assert ga["subset_clusters_per_iteration"][0] == ga["subset_clusters_per_iteration"][1] and max(ga["subset_clusters_per_iteration"][0]) > 0, ga
    # the reference's own arithmetic at this shape: north_star's 1e-4 (the fp32 LU amplifies the ~1e-7 differences between the two
    # implementations' R by the systems' condition number, 5e3 .. 1.6e4: the bar here is the contract's, not the one-covariate 1e-5)
    rf = rows["gpu_ref_arith_vs_oracle_faithful"]
    assert rf["Z_rel"] <= 1e-4 and rf["iterations"][0] == rf["iterations"][1] and rf["kmeans_rounds_equal"], rf
    assert rf["objective_rel_max"] <= 1e-4, rf
    assert rf["subset_clusters_per_iteration"][0] == rf["subset_clusters_per_iteration"][1], rf
    # (hard assignments / max |dR| of this pair: inside the width of "faithful" itself -- test_arithmetic_gap_table's comment; here the fp32 LU of
    #  condition 5e3 .. 1.6e4 widens it further: bounded, not zero)
    assert rf["R_maxabs"] <= 2e-3 and rf["argmax_diff_margin_ge_1e-5"] <= 100, rf
    # ... and bounded by what the faithful oracle does to ITSELF at this shape (VERDICT r5 #1b): distance and clear flips of the pair within 3 x the liberty row
    lib = rows["oracle_faithful_liberty1_vs_oracle_faithful"]
    assert lib["iterations"][0] == lib["iterations"][1], lib
    assert rf["Z_rel"] <= 3 * lib["Z_rel"] + 2e-6 and rf["argmax_diff_margin_ge_1e-5"] <= 3 * lib["argmax_diff_margin_ge_1e-5"] + 5, (rf, lib)
    gf = rows["gpu_vs_oracle_faithful"]      # reported: the default mode against the reference's fp32 drift at this shape
    assert gf["iterations"][0] == gf["iterations"][1], gf
    # "ref_arith" = 2 at this shape (round 6): every group but the O / E tables -- north_star's 1e-4 against the faithful oracle, same iterations and subset clusters
    r2 = rows["gpu_ref_arith2_vs_oracle_faithful"]
    assert r2["Z_rel"] <= 1e-4 and r2["iterations"][0] == r2["iterations"][1] and r2["subset_clusters_per_iteration"][0] == r2["subset_clusters_per_iteration"][1], r2


@pytest.mark.timeout(3000, method="thread")
@pytest.mark.skipif(os.environ.get("HMX_SLOW", "0") != "1", reason="builder run (HMX_SLOW=1): ~15 minutes of CPU for the oracle at 10M cells; table in profiles/")
def test_config4_10M_against_the_oracle():
    """BASELINE configs[3] at FULL size on one GPU -- 10M x 50, K = 100, 20 batches, to convergence: GPU default vs the oracle with exact
    accumulators, and GPU reference arithmetic vs the faithful oracle.  Table -> gpurun_out/r6_parity_c4_10M.json (profiles/)."""
    with open("/proc/meminfo") as fh:
        avail_gb = [int(l.split()[1]) for l in fh if l.startswith("MemAvailable")][0] / 1048576.0
    if avail_gb < 120:
        pytest.skip("two 10M-cell oracles need ~70 GB of host memory (%.0f GB available)" % avail_gb)
    Z, meta, _ = synth(10_000_000, d=50, levels=(20,), seed=7)
    res, timing = _run_pair_to_convergence(Z, meta, 100, 3, {"gpu": {}, "gpu_ref_arith": {"ref_arith": 1}}, {"oracle_accurate": 15, "oracle_faithful": 0},
                                           blas_threads=8)
    rows = {"gpu_vs_oracle_accurate": _pair_row(res["gpu"], res["oracle_accurate"]), "gpu_ref_arith_vs_oracle_faithful": _pair_row(res["gpu_ref_arith"], res["oracle_faithful"]),
            "gpu_vs_oracle_faithful": _pair_row(res["gpu"], res["oracle_faithful"]), "oracle_faithful_vs_oracle_accurate": _pair_row(res["oracle_faithful"], res["oracle_accurate"])}
    out = {"workload": {"cells": 10000000, "pcs": 50, "clusters": 100, "batches": 20}, "seconds": timing, "pairs": rows}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "r6_parity_c4_10M.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
    ga, rf = rows["gpu_vs_oracle_accurate"], rows["gpu_ref_arith_vs_oracle_faithful"]
    assert ga["Z_rel"] <= 2e-5 and ga["argmax_diff_margin_ge_1e-5"] == 0 and ga["iterations"][0] == ga["iterations"][1], ga
    assert rf["Z_rel"] <= 1e-4 and rf["iterations"][0] == rf["iterations"][1], rf


@pytest.mark.timeout(4000, method="thread")
@pytest.mark.skipif(os.environ.get("HMX_SLOW", "0") != "1", reason="builder run (HMX_SLOW=1): ~20 minutes of CPU for the oracle at 5M cells x 200 clusters; table in profiles/")
def test_config5_5M_against_the_oracle():
    """BASELINE configs[4] at its FULL stated size on one GPU -- 5M x 50, K = 200, three nested covariates 8 > 64 > 128 = 200 levels, to
    convergence: GPU default vs the oracle with exact accumulators (subset-cluster counts per iteration included), and GPU reference arithmetic
    vs the faithful oracle.  Table -> gpurun_out/r6_parity_c5_5M.json (profiles/)."""
    with open("/proc/meminfo") as fh:
        avail_gb = [int(l.split()[1]) for l in fh if l.startswith("MemAvailable")][0] / 1048576.0
    if avail_gb < 150:
        pytest.skip("two 5M-cell x 200-cluster oracles need ~100 GB of host memory (%.0f GB available)" % avail_gb)
    Z, meta, _ = synth(5_000_000, d=50, levels=(8, 64, 128), seed=11, nested=True)
    res, timing = _run_pair_to_convergence(Z, meta, 200, 5, {"gpu": {}, "gpu_ref_arith": {"ref_arith": 1}},
                                           {"oracle_accurate": 15, "oracle_faithful": 0, "oracle_faithful_liberty1": (0, 1)}, blas_threads=8)
    rows = {"gpu_vs_oracle_accurate": _pair_row(res["gpu"], res["oracle_accurate"]), "gpu_ref_arith_vs_oracle_faithful": _pair_row(res["gpu_ref_arith"], res["oracle_faithful"]),
            "oracle_faithful_liberty1_vs_oracle_faithful": _pair_row(res["oracle_faithful_liberty1"], res["oracle_faithful"]),
            "gpu_vs_oracle_faithful": _pair_row(res["gpu"], res["oracle_faithful"]), "oracle_faithful_vs_oracle_accurate": _pair_row(res["oracle_faithful"], res["oracle_accurate"])}
    out = {"workload": {"cells": 5000000, "pcs": 50, "clusters": 200, "levels": [8, 64, 128], "nested": True}, "seconds": timing, "pairs": rows}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "r6_parity_c5_5M.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
    ga, rf = rows["gpu_vs_oracle_accurate"], rows["gpu_ref_arith_vs_oracle_faithful"]
    assert ga["Z_rel"] <= 2e-5 and ga["argmax_diff_margin_ge_1e-5"] == 0 and ga["iterations"][0] == ga["iterations"][1] and ga["kmeans_rounds_equal"], ga
    assert ga["objective_rel_max"] <= 1e-4, ga
    assert ga["subset_clusters_per_iteration"][0] == ga["subset_clusters_per_iteration"][1], ga
    assert rf["iterations"][0] == rf["iterations"][1], rf
    # the pair's distance and clear flips against the oracle's own width at this size (VERDICT r5 #1b: 309 clear flips at 5M were reported, not bounded)
    lib = rows["oracle_faithful_liberty1_vs_oracle_faithful"]
    assert lib["iterations"][0] == lib["iterations"][1], lib
    assert rf["Z_rel"] <= 3 * lib["Z_rel"] + 2e-6 and rf["argmax_diff_margin_ge_1e-5"] <= 3 * lib["argmax_diff_margin_ge_1e-5"] + 5 * 5, (rf, lib)
    if lib["subset_clusters_per_iteration"][0] == lib["subset_clusters_per_iteration"][1]:
        # where the oracle agrees with itself on which clusters take the batch-subset ridge path: north_star's 1e-4, and the same clusters on the GPU
        assert rf["Z_rel"] <= 1e-4 and rf["subset_clusters_per_iteration"][0] == rf["subset_clusters_per_iteration"][1], rf
    else:
        # (Round 6, second builder run: the faithful oracle kept 27 clusters on the subset path in the first correction, its liberty variant 29, the GPU 31, ALL pairs
        #  1.4e-3 .. 1.8e-3 apart -- profiles/r6_parity_c5_5M_second_run.json.  Read as a knife edge of the reference's arithmetic at first; it was three oracle threads
        #  in the bundled OpenBLAS at once, see _run_pair_to_convergence.  The branch stays as the bound for an oracle that disagrees with itself.)
        spread = abs(lib["subset_clusters_per_iteration"][0][0] - lib["subset_clusters_per_iteration"][1][0])
        assert abs(rf["subset_clusters_per_iteration"][0][0] - rf["subset_clusters_per_iteration"][1][0]) <= 2 * spread + 2, (rf, lib)


# ---------------------------------------------------------------- the sharded path through two PROCESSES
@pytest.mark.timeout(600, method="thread")
@pytest.mark.parametrize("cfg", ["C4", "C5"])
def test_full_size_named_configs(cfg):
    """BASELINE configs[3] (10M x 50, K=100, 20 batches) and configs[4] (5M x 50, K=200, 8/64/128 nested levels) at FULL size on one
    GPU, to convergence, through the size-independent properties of the path (the oracle needs ~10 min per run there):
    R columns are distributions; O is exactly the per-level sum of R (fixed-point accumulation carried through every block
    update of every round == a direct sum over the final R); sum_k O[k,b] = N_b; E = rowsum(R) Pr_b^T; the objective went down;
    Z_corr finite and moved."""
    if cfg == "C4":
        N, K, levels, nested = 10_000_000, 100, (20,), False
    else:
        N, K, levels, nested = 5_000_000, 200, (8, 64, 128), True
    Z, meta, _ = synth(N, d=50, levels=levels, seed=5, nested=nested)
    vars_use = list(meta)
    skw, _ = prepare_setup_args(Z, meta, vars_use, nclust=K)
    g = Harmony(seed=2)
    g.setup(**skw)
    znorm = float(np.linalg.norm(Z[::1000]))
    g.init_cluster_cpp()
    it = 0
    for it in range(1, 11):
        assert g.cluster_cpp() == 0
        g.moe_correct_ridge_cpp()
        if g.check_convergence(1):
            break
    assert 2 <= it <= 10
    assert g.cluster_cpp() == 0              # O, E and R of one more clustering pass: consistent with each other
    oh = np.asarray(g.objective_harmony)
    assert np.all(np.isfinite(oh)) and oh[-1] < oh[0]
    R = g.get_matrix("R", np.float32)         # [K, N] column-major == cells x K row-major
    assert R.shape == (K, N)
    Rc = R.T                                  # contiguous rows, one per cell
    assert Rc.flags["C_CONTIGUOUS"] and float(Rc.min()) >= 0.0
    assert np.abs(Rc.sum(axis=1, dtype=np.float64) - 1).max() < 1e-5
    O, E = g.O, g.E
    off = 0
    for v, L in zip(vars_use, levels):
        lab = meta[v]
        N_b = np.bincount(lab, minlength=L).astype(np.float64)
        Ob = O[:, off:off + L]
        np.testing.assert_allclose(Ob.sum(axis=0), N_b, rtol=1e-6, atol=1e-2)
        if v == vars_use[0] or cfg == "C4":
            Odirect = np.stack([Rc[lab == b].sum(axis=0, dtype=np.float64) for b in range(L)], axis=1)
            np.testing.assert_allclose(Ob, Odirect, rtol=5e-6, atol=5e-3)
        np.testing.assert_allclose(E[:, off:off + L], O[:, :levels[0]].sum(axis=1, keepdims=True) * (N_b / N)[None, :], rtol=2e-5, atol=1e-3)
        off += L
    del R, Rc
    Zc = g.get_matrix("Z_corr", np.float32)
    assert Zc.shape == (50, N) and np.all(np.isfinite(Zc))
    moved = float(np.linalg.norm(Zc[:, ::1000].T.astype(np.float64) - Z[::1000])) / znorm
    assert 0.01 < moved < 1.0, moved


def test_two_processes_sharded_run():
    """Two processes (torch.distributed.run, world 2), one shard each, the whole RunHarmony with every accumulator all-reduced
    through the hook; both ranks share the box's single GPU, so the transport is gloo (RCCL refuses two ranks per device) --
    the same script runs with --backend nccl on a multi-GPU node.  Rank 0 checks against the unsharded run."""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_two_proc.py")],
                       capture_output=True, text=True, timeout=500, stdin=subprocess.DEVNULL)
    assert "DIST2_OK world=2" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_two_processes_unequal_shards_agree_on_the_protocol():
    """Shards of 90 % / 10 % of the cells with the chain threshold moved between them (HMX_CHAIN_MAX_TPW): left alone, the small rank
    would run the persistent chain with the in-launch exchange while the large one waits in a per-block all-reduce -- a deadlock.
    hmx_setup agrees on the minimum over the ranks: both take the per-block path, the run matches the unsharded one."""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_two_proc.py"), "--p2p", "--split", "0.9",
                        "--chain-max-tpw", "0.02"], capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert "DIST2_OK world=2" in p.stdout and "chain=0" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_two_processes_peer_to_peer_chain():
    """The same two-process run with the per-block all-reduce replaced by the in-launch exchange of the persistent chain: each
    rank's folder writes its K x B contribution table into the other's inbox (fine-grained device memory shared through HIP IPC)
    and sums what arrived in its own.  Both ranks share this box's one GPU (120 workgroups each), so the transport under test is
    the protocol (tags, parities, IPC mapping), not xGMI.  Rank 0 checks against the unsharded run; ~480 fewer collectives."""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_two_proc.py"), "--p2p"],
                       capture_output=True, text=True, timeout=500, stdin=subprocess.DEVNULL)
    assert "DIST2_OK world=2" in p.stdout and "p2p=1" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    print(p.stdout[-400:])
    # (round 4) with the inboxes on, the small collectives travel through them too: what is left on the hook is the setup's handful
    left = int(p.stdout.split("collectives/rank=")[1].split()[0])
    assert left <= 12 and int(p.stdout.split("inbox_allreduces/rank=")[1].split()[0]) > 20, p.stdout[-400:]
    # the same with the old contributions carried from round to round inside the chain (what a 1M-cells-per-GPU job runs by default)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_two_proc.py"), "--p2p", "--carry"],
                       capture_output=True, text=True, timeout=500, stdin=subprocess.DEVNULL)
    assert "DIST2_OK world=2" in p.stdout and "p2p=1" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_two_processes_configs4_shape_through_the_inboxes():
    """BASELINE configs[4]'s shape on two ranks (K = 200, 8 > 64 > 128 nested levels, 40k cells): round 6 -- the block chain by wave pairs (k_tile MODE 6), the
    folders of the two ranks exchange every block step's slice of the K x B table through the inboxes INSIDE the launch (rounds 2-5: one launch and one inbox
    all-reduce per block step); the ridge statistics (Q K (d + 1) doubles, far above the 65536 entries
    of the one-shot form) take the inboxes' reduce-scatter + all-gather windows (round 5) -- what is left on the hook is the setup's handful.
    Rank 0 checks against the unsharded run."""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_two_proc.py"), "--p2p", "--workload", "c5", "--cells", "40000"],
                       capture_output=True, text=True, timeout=600, stdin=subprocess.DEVNULL)
    assert "DIST2_OK world=2" in p.stdout and "p2p=1" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    print(p.stdout[-400:])
    assert int(p.stdout.split("collectives/rank=")[1].split()[0]) <= 12, p.stdout[-400:]
    assert int(p.stdout.split("big_windows/rank=")[1].split()[0]) > 0, p.stdout[-400:]
    assert "chain=1" in p.stdout, p.stdout[-400:]      # (round 6: the block steps run inside the wave-pair chain, their tables exchanged through the inboxes in the launch)


def test_bench_bootstraps_without_torch():
    """bench.py --bootstrap file: the bench's step without torch in the process (the unique-id file / hmx_comm_init path of a plain C or R
    host; N = 1 here -- RCCL refuses two ranks on this box's single device -- so the line is produced by the library alone)"""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, "-c", "import sys, runpy; sys.argv = ['bench.py', '--bootstrap', 'file', '--gpus', '1', '--steps', '2', '--warmup', '1', "
                        "'--cells-per-gpu', '200000']; runpy.run_path(%r, run_name='__main__'); assert 'torch' not in sys.modules, 'torch was imported'"
                        % os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=500, stdin=subprocess.DEVNULL, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "no torch" in line["config"]["bootstrap"] and line["roofline"]["frac"] > 0, line


@pytest.mark.timeout(1500, method="thread")
def test_bench_two_ranks_default_is_configs3_strong_scaling():
    """`bench.py --gpus 2` with no size on the command line IS BASELINE configs[3] (10M cells x 50 PCs, K = 100, 20 batches in total, cell-sharded,
    "scaling": "strong"), carries DESIGN 5.2's prediction for that line and the weak-scaling companion under `also`.  Two ranks
    share this box's one GPU over gloo: a protocol check of the line, not a scaling number."""
    import subprocess
    import sys
    env = dict(os.environ, HMX_BENCH_PREROLL="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--no-e2e"],
                       capture_output=True, text=True, timeout=1400, stdin=subprocess.DEVNULL, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0, line
    assert line["config"]["baseline_config"].startswith("configs[3]") and "10000000 cells" in line["config"]["workload"], line["config"]
    assert line["config"]["shard_check"]["O_identical_on_all_ranks"], line["config"]
    assert line["predicted"]["ms_per_step"][0] > 0 and line["config"]["comm"] is not None, line
    weak = line["also"]["weak_scaling_1M_per_gpu"]
    assert weak.get("scaling") == "weak" and weak["cells_per_s"] > 0, weak
    print("bench --gpus 2 (two ranks sharing one GPU, gloo): %.1f ms per step at configs[3], predicted on two GPUs %s; weak leg %.1f ms"
          % (line["ms_per_step"], line["predicted"]["ms_per_step"], weak["ms_per_step"]))


def test_torch_free_c_host_with_builtin_rccl(tmp_path):
    """examples/comm_example.c: a process WITHOUT torch (the R / plain-C host) brings up the built-in RCCL communicator from the
    system librccl with the unique id shipped through a file, and runs the sharded code path with forced collectives
    (world 1 on this single-GPU box; `comm_example <rank> <world> <file>` per GPU on a node).  Round 1 reported a hang here:
    ncclCommInitRank simply takes ~5 s in a fresh process (kernel loading)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "comm_example")
    libdir = os.path.join(ROOT, "harmony_amd", "lib")
    p = subprocess.run([gcc, "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "comm_example.c"),
                        "-L" + libdir, "-lharmony_mi355x", "-Wl,-rpath," + libdir, "-lm", "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    r = subprocess.run([exe, "0", "1", str(tmp_path / "uid")], capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert r.returncode == 0 and "COMM_EXAMPLE_OK rank 0/1" in r.stdout, (r.stdout, r.stderr[-2000:])
    assert int(r.stdout.split("collectives")[1].split()[0]) > 100


def test_torch_free_two_ranks_on_two_gpus(tmp_path):
    """examples/comm_example.c with world 2: two torch-free processes, one GPU each, hmx_comm_init alone bootstraps the RCCL communicator
    (unique id through a file) AND the peer inboxes of the block chain; the checksum of O must equal the one-rank run's.  Needs two
    GPUs in the box (RCCL refuses two ranks on one device): skipped on the single-GPU boxes this repository is developed on."""
    import shutil
    import subprocess
    try:
        import torch
        ndev = torch.cuda.device_count()
    except Exception:
        ndev = 0
    if ndev < 2:
        pytest.skip("needs two GPUs")
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "comm_example")
    libdir = os.path.join(ROOT, "harmony_amd", "lib")
    p = subprocess.run([gcc, "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "comm_example.c"),
                        "-L" + libdir, "-lharmony_mi355x", "-Wl,-rpath," + libdir, "-lm", "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    one = subprocess.run([exe, "0", "1", str(tmp_path / "uid1")], capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert "COMM_EXAMPLE_OK" in one.stdout, (one.stdout, one.stderr[-2000:])
    procs = [subprocess.Popen([exe, str(r), "2", str(tmp_path / "uid2")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, stdin=subprocess.DEVNULL)
             for r in range(2)]
    outs = [pr.communicate(timeout=400) for pr in procs]
    chk = lambda txt: [w for w in txt.split("COMM_EXAMPLE_OK")[1].split() if w][txt.split("COMM_EXAMPLE_OK")[1].split().index("O-checksum") + 1]
    for (so, se) in outs:
        assert "COMM_EXAMPLE_OK" in so, (so, se[-2000:])
        assert chk(so) == chk(one.stdout), (so, one.stdout)

