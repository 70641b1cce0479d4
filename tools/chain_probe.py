"""Where a block step of the persistent chain goes: the headline workload (1M x 50, K = 100, 10 batches) on the diagnostics build
(-DHMX_TRACE, harmony_amd/lib/libharmony_mi355x_trace.so) with parts of the step switched OFF -- WRONG RESULTS, timing only:
    upd_debug  4  no R stores      8  no contribution atomics at all      16  no carry atomics (Sold_next)
Prints the chain's phase clocks (us per block step) for every combination asked for.

    HMX_LIB_PATH=harmony_amd/lib/libharmony_mi355x_trace.so python tools/chain_probe.py 0 16 8 4 12"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bench_data import synth  # noqa: E402
from harmony_amd import Harmony, prepare_setup_args  # noqa: E402

modes = [int(x) for x in sys.argv[1:]] or [0]
n = int(os.environ.get("PROBE_CELLS", "1000000"))
Z, meta, _ = synth(n, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
names = ["folder_wait_arrivals", "folder_fold", "folder_publish", None, "worker_wait_flag", "worker_copy_table", "worker_wait_stores_retired",
         "worker_barrier_arrive", "worker_next_mfma", "worker_tiles_but_last", "worker_last_epilogue", "worker_flush_and_store_issue"]
g = Harmony(seed=7)
g.setup(**skw)
g.init_cluster_cpp()
Y0 = g.Y.copy()
for m in modes:
    g.restart()
    g.init_cluster_cpp(Y0)
    g._set("upd_debug", m)
    g._get("chain_dbg")                  # (reading resets the clocks)
    t0 = time.perf_counter()
    for _ in range(2):
        g.cluster_cpp()
        g.moe_correct_ridge_cpp()
    g.getZcorr()
    wall = time.perf_counter() - t0
    dbg = np.array(g._get("chain_dbg"), dtype=np.float64)
    steps = max(dbg[3] * g._scalar("n_blocks"), 1.0)
    out = {nm: round(dbg[i] / 100.0 / steps, 2) for i, nm in enumerate(names) if nm}
    out["wg0_wave_busy_us"] = [round(dbg[16 + w] / 100.0 / steps, 2) for w in range(8)]
    print(json.dumps({"upd_debug": m, "chain_launches": int(dbg[3]), "wall_s": round(wall, 3), "us_per_block_step": out}), flush=True)
    g._set("upd_debug", 0)
