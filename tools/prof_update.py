"""Small driver for rocprofv3 counter passes: setup + init + a few clustering rounds at N cells."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 10
Z, meta, _ = synth(N, d=50, levels=(B,), seed=7)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
g = Harmony(seed=3)
g.setup(**skw)
g.init_cluster_cpp()
g.cluster_cpp()
g.moe_correct_ridge_cpp()
g.cluster_cpp()
print("done", g.objective_harmony)
