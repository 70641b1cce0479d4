#!/bin/bash
# round 3: the persistent chain where waves own THREE and more tiles per block (3M cells; the suite's chain cases stop at two) against the
# launch-per-step path on the same data
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 200 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args
Z, meta, _ = synth(3000000, d=50, levels=(10,), seed=11)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
out = []
for chain in ("1", "0"):
    os.environ["HMX_CHAIN"] = chain
    g = Harmony(seed=3); g.setup(**skw)
    if not out: Y0 = g.kmeans_centers()
    g.init_cluster_cpp(Y0)
    for it in range(2):
        assert g.cluster_cpp() == 0
        g.moe_correct_ridge_cpp()
    out.append((int(g._scalar("chain")), g.getZcorr().copy(), np.array(g.objective_kmeans), np.array(g.O)))
    del g
(c1, Z1, o1, O1), (c0, Z0, o0, O0) = out
rel = np.linalg.norm(Z1 - Z0) / np.linalg.norm(Z0)
print("chain flags", c1, c0, "Z_corr rel diff %.2e" % rel, "objective rel diff %.2e" % float(np.max(np.abs(o1 - o0) / np.abs(o0))), "O rel diff %.2e" % (np.linalg.norm(O1 - O0) / np.linalg.norm(O0)), "rounds", len(o1), len(o0))
assert c1 == 1 and c0 == 0 and rel < 1e-6 and len(o1) == len(o0)
print("CHAIN3M_OK")
PY
