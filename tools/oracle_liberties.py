#!/usr/bin/env python
"""CPU only: how wide is "faithful"?  The oracle's faithful mode fixes an operation order in a few places where the reference leaves it
to Armadillo / BLAS (oracle/harmony_oracle.cpp header, LIBERTIES).  Each liberty is flipped on its own and the run -- synthetic N x 50,
K = 100 (one covariate, 10 batches) and K = 60 with two crossed covariates, reference defaults, to convergence, shared centres and block
partitions -- is compared with the default faithful run: rel. Frobenius distance of Z_corr, max |dR|, hard-assignment flips (raw / at a
top-2 margin >= 1e-5), objective series, iterations.  Output: profiles/r5_oracle_liberties.json (quoted in DESIGN 2.1)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harmony_amd.ui import prepare_setup_args  # noqa: E402  (host-side argument preparation only: no device)
from helpers import synth  # noqa: E402
from oracle.oracle import OracleHarmony  # noqa: E402

BITS = {1: "L1 sums: two accumulators", 2: "L1 sums: eight strided accumulators", 4: "apply: one rounded product per non-zero",
        16: "L2 norms accumulated in double", 31 - 8: "all of them"}
# several covariates only (src/harmony.cpp:573): arma::inv through OpenBLAS 0.3.28's LAPACK (scipy's copy), Armadillo's two call sequences
BITS_INV = {32: "arma::inv = LAPACK sgetrf + sgetri (OpenBLAS 0.3.28)", 64: "arma::inv = LAPACK spotrf + spotri (OpenBLAS 0.3.28)",
            4 | 64: "per-non-zero apply + LAPACK spotrf + spotri (what a current RcppArmadillo on OpenBLAS would run)"}


def run(skw, Y0, liberty, seed=3):
    o = OracleHarmony(mask=0, seed=seed, liberty=liberty)
    o.setup(**skw)
    o.init_cluster_cpp(Y0)
    it = 0
    for it in range(1, 11):
        assert o.cluster_cpp() == 0
        o.moe_correct_ridge_cpp()
        if o.check_convergence(1):
            break
    return dict(Z=o.getZcorr(), R=o.R, it=it, obj=np.array(o.objective_kmeans))


out = {}


def flips(Ra, Rb, margin):
    aa, ab = Ra.argmax(axis=0), Rb.argmax(axis=0)
    bad = np.where(aa != ab)[0]
    if not bad.size:
        return 0, 0
    srt = np.sort(Rb[:, bad], axis=0)
    return int(bad.size), int(((srt[-1] - srt[-2]) >= margin).sum())


cases = [("%dk_one_covariate_K100" % (n // 1000), n, (10,), 100) for n in (20000, 100000)] + [("20k_two_covariates_K60", 20000, (3, 4), 60)]
if len(sys.argv) > 1:       # e.g. `1000k_one_covariate_K100` (BASELINE configs[2]; ~1 minute of CPU per run); a second argument `new`: only the missing rows
    cases = [c for c in cases + [("1000k_one_covariate_K100", 1000000, (10,), 100), ("100k_three_nested_covariates_K200", 100000, (8, 64, 128), 200)] if c[0] in sys.argv[1:2]]
PATH = os.path.join(ROOT, "profiles", "r5_oracle_liberties.json")
if os.path.exists(PATH):
    out = json.load(open(PATH))
for name, N, levels, K in cases:
    Z, meta, _ = synth(N, d=50, levels=levels, seed=7, nested=(len(levels) == 3))
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    rng = np.random.default_rng(1)
    Y0 = np.asfortranarray(Z[rng.choice(N, K, replace=False)].T)
    base = run(skw, Y0, 0)
    rows = {}
    bits = dict(BITS)
    if len(levels) > 1:
        from oracle import oracle as _orc
        if _orc.use_lapack():
            bits.update(BITS_INV)
    from oracle import oracle as _orc
    if _orc.use_lapack():
        bits[128] = "norms / column sums as Armadillo's op_norm / op_sum form them on OpenBLAS 0.3.28 (sasum, snrm2)"
    bits["sgemm"] = "distance GEMM Y.t() * Z through OpenBLAS 0.3.28 sgemm (one thread)"
    bits["openblas"] = "EVERYTHING the reference's binary takes from OpenBLAS 0.3.28 at once: sgemm, sasum / snrm2%s" % (
        ", spotrf + spotri, per-non-zero apply" if len(levels) > 1 else "")
    if len(sys.argv) > 2 and sys.argv[2] == "new":      # only the rows the committed table does not hold yet
        bits = {b: w for b, w in bits.items() if w not in out.get(name, {})}
    for bit, what in bits.items():
        if bit in ("sgemm", "openblas"):          # (:144,222 dense x dense: BLAS sgemm in the reference's binary; the oracle's default is the sequential dot product)
            if not _orc.use_openblas(1) or (bit == "openblas" and not _orc.use_lapack()):
                continue
            r = run(skw, Y0, 0 if bit == "sgemm" else (128 | (4 | 64 if len(levels) > 1 else 0)))
            _orc.load().orc_set_sgemm(None)
        else:
            r = run(skw, Y0, bit)
        n = min(len(r["obj"]), len(base["obj"]))
        f, fc = flips(r["R"], base["R"], 1e-5)
        rows[what] = {"liberty": bit, "Z_rel": float(np.linalg.norm(r["Z"] - base["Z"]) / np.linalg.norm(base["Z"])),
                      "R_maxabs": float(np.abs(r["R"] - base["R"]).max()), "argmax_diff": f, "argmax_diff_margin_ge_1e-5": fc,
                      "objective_rel_max": float(np.max(np.abs(r["obj"][:n] - base["obj"][:n]) / np.abs(base["obj"][:n]))),
                      "iterations": [r["it"], base["it"]]}
        print(name, what, rows[what], flush=True)
    out[name] = dict(out.get(name, {}), **rows)
with open(PATH, "w") as fh:
    json.dump(out, fh, indent=1)
