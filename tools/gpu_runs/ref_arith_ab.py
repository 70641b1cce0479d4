"""A/B (python tools/gpu_runs/ref_arith_ab.py on the GPU box; AB_SET="0,5" settings of "seq_fused", AB_CELLS, AB_C5=1 for configs[4]'s shape): the reference-arithmetic mode with the round-6 kernel forms switched off / on ("seq_fused" bits): time per run, phases, and how far the results
of the two settings are from each other (the lane = cluster ridge pass must be BIT-identical to the round-5 kernel: same chains, same roundings)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
n = int(os.environ.get("AB_CELLS", "1000000")); c5 = os.environ.get("AB_C5", "0") == "1"
settings = [int(x) for x in os.environ.get("AB_SET", "1,5").split(",")]
if c5:
    Z, meta, _ = synth(n, d=50, levels=(8, 64, 128), seed=7, nested=True); K = 200
else:
    Z, meta, _ = synth(n, d=50, levels=(10,), seed=7); K = 100
skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
res = []
for fused in settings:
    o = Harmony(seed=1, ref_arith=1)
    o._set("seq_fused", fused)
    o.setup(**skw)
    run_to_convergence(o)
    o._scalar("sync"); t0 = time.perf_counter()
    its = [run_to_convergence(o) for _ in range(2)]
    o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0) / 2
    o.set_profile(2); run_to_convergence(o); o._scalar("sync")
    ph = {k: round(o._scalar("gputimer:" + k), 3) for k in ("cluster_head", "EO_update", "Rcells_update", "objective", "ridge_statistics", "arma_inv")}
    res.append((o.getZcorr(), np.asarray(o.objective_kmeans)))
    print(json.dumps({"seq_fused": fused, "ms": ms, "its": its, "phases": ph, "passes": o._get("seq:group_passes").tolist(), "runs": o._get("seq:group_runs").tolist()}), flush=True)
    del o
for i in range(1, len(res)):
    a, b = res[0], res[i]
    print("setting %d vs %d: Z_corr rel %.3e, bit-identical %s, objective rel %.3e" % (settings[i], settings[0], np.linalg.norm(a[0] - b[0]) / np.linalg.norm(a[0]), np.array_equal(a[0], b[0]),
          np.max(np.abs(a[1][:min(len(a[1]), len(b[1]))] - b[1][:min(len(a[1]), len(b[1]))]) / np.abs(a[1][:min(len(a[1]), len(b[1]))]))))
