#!/bin/bash
# round 6: O/E segment length probe
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r6_e.txt
for seg in 128 64 32; do
  echo "== HMX_SEQ_ROUND_SEG=$seg" >> gpurun_out/r6_e.txt
  HMX_SEQ_ROUND_SEG=$seg timeout 300 python /dev/stdin >> gpurun_out/r6_e.txt 2>&1 <<'PY'
import sys, time, json, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
Z, meta, _ = synth(1000000, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=100)
o = Harmony(seed=1, ref_arith=1)
o.setup(**skw)
run_to_convergence(o)
o._scalar("sync"); t0 = time.perf_counter()
its = [run_to_convergence(o) for _ in range(3)]
o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0) / 3
o.set_profile(2); run_to_convergence(o); o._scalar("sync")
ph = {k: round(o._scalar("gputimer:" + k), 3) for k in ("cluster_head", "EO_update", "objective", "ridge_statistics")}
print(json.dumps({"ms": ms, "its": its, "phases": ph, "obj": np.asarray(o.objective_kmeans)[-2:].tolist()}))
PY
done
cat gpurun_out/r6_e.txt
