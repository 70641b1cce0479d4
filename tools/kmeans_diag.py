"""Diagnostic: where does the GPU's kmeans_centers leave the oracle's (seeds vs Lloyd)?  GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args
from oracle.oracle import OracleHarmony
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 17
Z, meta, _ = synth(N, d=50, levels=(10,), seed=9)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
g = Harmony(seed=seed); g.setup(**skw); Yg = g.kmeans_centers()
c = OracleHarmony(accurate=True, seed=seed); c.setup(**skw); c.init_cluster_cpp()
sg, sc = g._get("seed_cells").astype(np.int64), c._get("seed_cells").astype(np.int64)
print("seeds equal:", np.array_equal(sg, sc), "differing anchors:", np.where(sg != sc)[0])
Zn = skw["Z"] / np.linalg.norm(skw["Z"], axis=0, keepdims=True)
Yg_n = Yg / np.linalg.norm(Yg, axis=0, keepdims=True)
dcol = np.linalg.norm(Yg_n - c.Y, axis=0)
print("columns off by > 1e-5:", np.where(dcol > 1e-5)[0], dcol[dcol > 1e-5])
for i in np.where(sg != sc)[0][:5]:
    print("anchor", i, "gpu cell", sg[i], "oracle cell", sc[i], "cos between:", float(Zn[:, sg[i]] @ Zn[:, sc[i]]))
# emulate Lloyd in numpy (fp64) from the oracle's seeds to see which side deviates
Y = Zn[:, sc].astype(np.float32).copy()
X = Zn.astype(np.float32)
for it in range(10):
    sc_ = (Y * Y).sum(axis=0)[:, None] - 2 * (Y.T @ X)
    a = sc_.argmin(axis=0)
    for k in range(Y.shape[1]):
        m = a == k
        if m.any():
            Y[:, k] = X[:, m].astype(np.float64).mean(axis=1)
Yn = Y / np.linalg.norm(Y, axis=0, keepdims=True)
print("numpy Lloyd vs oracle:", np.linalg.norm(Yn - c.Y) / np.linalg.norm(c.Y), " vs gpu:", np.linalg.norm(Yn - Yg_n) / np.linalg.norm(Yg_n))
