#!/bin/bash
# round 6: reference arithmetic at the shapes of the two 8-GPU configs (configs[4] shape at 1M; configs[3]'s per-rank share 1.25M x 20 batches) + seq tests
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_seq.py -q -x 2>&1 | tail -2 > gpurun_out/r6_h.txt
AB_CELLS=1000000 AB_C5=1 AB_SET=7 python tools/gpu_runs/r6_ab.py 2>&1 | grep "seq_fused" | cut -c1-300 >> gpurun_out/r6_h.txt
python - >> gpurun_out/r6_h.txt 2>&1 <<'PY'
import sys, time, json, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
Z, meta, _ = synth(1250000, d=50, levels=(20,), seed=7)
skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=100)
o = Harmony(seed=1, ref_arith=1)
o.setup(**skw)
run_to_convergence(o)
o._scalar("sync"); t0 = time.perf_counter()
its = [run_to_convergence(o) for _ in range(2)]
o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0) / 2
o.set_profile(2); run_to_convergence(o); o._scalar("sync")
ph = {k: round(o._scalar("gputimer:" + k), 3) for k in ("cluster_head", "EO_update", "Rcells_update", "objective", "ridge_statistics")}
print(json.dumps({"workload": "1.25M x 20 batches, ref_arith", "ms": ms, "its": its, "phases": ph}))
PY
cat gpurun_out/r6_h.txt
