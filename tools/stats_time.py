"""Time the ridge statistics pass alone (HIP events through torch) under the kernel's diagnostic switches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args
Z, meta, _ = synth(1000000, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
g = Harmony(seed=1); g.set_stream(torch.cuda.current_stream().cuda_stream); g.setup(**skw); g.init_cluster_cpp(); g.cluster_cpp()
for dbg in (0,):
    g._set("upd_debug", dbg)
    g.moe_correct_ridge_cpp(); torch.cuda.synchronize()
    g.set_profile(2)
    for _ in range(5): g.moe_correct_ridge_cpp()
    print("dbg", dbg, "ridge_statistics ms per call", g._scalar("gputimer:ridge_statistics") / 5)
    g.set_profile(False)
