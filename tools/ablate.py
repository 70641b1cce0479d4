"""Timing-only ablation of the dominant kernel (results are garbage while a mask is set)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
masks = [0]
Z, meta, _ = synth(N, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
g = Harmony(seed=3)
g.setup(**skw)
g.init_cluster_cpp()
g.cluster_cpp()
for tpw in (1, 2):
    g._set("upd_tpw", tpw)
    for m in masks:
        g._set("ablate", m)
        g.set_profile(True)
        g.cluster_cpp()
        ms, n = g._scalar("prof:update_ms"), g._scalar("prof:update_launches")
        print("tpw=%d mask=%2d  %.1f us/launch" % (tpw, m, 1e3 * ms / n), flush=True)
    g._set("ablate", 0)
