#!/bin/bash
# round 3: Schur-complement solve + split fold/penalty (configs[4] shape), lazy solve-result sync; parity first, then the numbers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "config5 or every_cluster or unequal_shards or carried_old" 2>&1 | tail -30 | cut -c1-400
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-e2e --also c5 > $O/bench_c5.json 2> $O/bench_c5.err; echo rc=$?
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3/bench_c5.json").read().strip().splitlines()[-1])
print("default ms_per_step", j["ms_per_step"], j["config"]["harmony_iterations"], j["roofline"]["avg_block_step_us"])
print(j["config"]["gpu_phase_ms_per_step"])
for k, v in (j.get("also") or {}).items():
    print("also", k, {a: v.get(a) for a in ("ms_per_step", "cells_per_s", "harmony_iterations", "avg_block_step_us", "roofline_run_frac", "error")})
    print("     ", v.get("gpu_phase_ms_per_step"))
PY
