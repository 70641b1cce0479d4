#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python bench.py --workload c5 --cells-per-gpu 5000000 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e > gpurun_out/c21_c5_5M.json 2> gpurun_out/c21_c5_5M.err; echo rc=$?
python - <<'PY'
import json
j = json.loads(open("gpurun_out/c21_c5_5M.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["config"]["harmony_iterations"], j["roofline"]["avg_block_step_us"], j["roofline"]["frac"], j["roofline"]["run"]["frac"])
print(j["config"]["gpu_phase_ms_per_step"])
PY
tail -3 gpurun_out/c21_c5_5M.err
