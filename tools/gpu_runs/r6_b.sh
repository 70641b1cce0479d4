#!/bin/bash
# round 6: the objective's fused launch -- probe against the sequential loop, the reference-arithmetic tests, the ref leg of the bench with and without it
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_seq.py -q -x -s -k "term_arrays" 2>&1 | tail -12 > gpurun_out/r6_b_probe.txt
timeout 1200 python -m pytest tests/test_gpu_seq.py -q -x -k "not term_arrays and not openblas_point" 2>&1 | tail -8 > gpurun_out/r6_b_tests.txt
cat > /tmp/leg.py <<'PY'
import sys, time, json, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
Z, meta, _ = synth(1000000, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=100)
for fused in (0, 1):
    o = Harmony(seed=1, ref_arith=1)
    o._set("seq_fused", fused)
    o.setup(**skw)
    run_to_convergence(o)
    o._scalar("sync"); t0 = time.perf_counter()
    its = [run_to_convergence(o) for _ in range(3)]
    o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0) / 3
    o.set_profile(2); run_to_convergence(o); o._scalar("sync")
    ph = {k: round(o._scalar("gputimer:" + k), 3) for k in ("cluster_head", "EO_update", "objective", "ridge_statistics")}
    print(json.dumps({"seq_fused": fused, "ms": ms, "its": its, "phases": ph, "obj": np.asarray(o.objective_kmeans)[-3:].tolist(),
                      "passes": o._get("seq:group_passes").tolist(), "runs": o._get("seq:group_runs").tolist(), "resid": o._scalar("seq:residual")}))
    del o
PY
HMX_OBJF_TRACE=1 timeout 600 python /tmp/leg.py > gpurun_out/r6_b_leg.txt 2>&1
cat gpurun_out/r6_b_probe.txt | cut -c1-600; cat gpurun_out/r6_b_tests.txt; grep -c settled gpurun_out/r6_b_leg.txt; grep -c continue gpurun_out/r6_b_leg.txt; grep objf gpurun_out/r6_b_leg.txt | tail -30; grep seq_fused gpurun_out/r6_b_leg.txt
