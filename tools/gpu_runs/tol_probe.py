"""Reference arithmetic: what the settle tolerance of the long restarted sums buys (python tools/gpu_runs/tol_probe.py on the GPU box): "seq_tol_ppb" 10000 (default, 1e-5)
against looser settings, both reference-arithmetic modes, 1M cells: ms per run, passes per group, distance of the results from the default's."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
n = int(os.environ.get("TP_CELLS", "1000000"))
Z, meta, _ = synth(n, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=100)
for mode in (1, 2):
    base = None
    for ppb in (10000, 30000, 100000, 1000000):
        o = Harmony(seed=1, ref_arith=mode)
        o._set("seq_tol_ppb", ppb)
        o.setup(**skw)
        run_to_convergence(o)
        o._scalar("sync"); t0 = time.perf_counter()
        it = run_to_convergence(o)
        o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0)
        zc = o.getZcorr().copy(); ob = np.array(o.objective_kmeans)
        if base is None: base = (zc, ob)
        m = min(len(ob), len(base[1]))
        print(json.dumps({"mode": mode, "seq_tol": ppb * 1e-9, "ms": round(ms, 2), "it": it, "passes": o._get("seq:group_passes").tolist(), "runs": o._get("seq:group_runs").tolist(),
                          "Z_rel_vs_default_tol": float(np.linalg.norm(zc - base[0]) / np.linalg.norm(base[0])), "obj_rel": float(np.max(np.abs(ob[:m] - base[1][:m]) / np.abs(base[1][:m])))}), flush=True)
        del o
