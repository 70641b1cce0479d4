#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c20_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e > gpurun_out/c20_bench.json 2> gpurun_out/c20_bench.err; echo rc=$?
python - <<'PY'
import json
j = json.loads(open("gpurun_out/c20_bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["config"]["harmony_iterations"], j["roofline"]["avg_block_step_us"])
print(j["config"]["gpu_phase_ms_per_step"])
PY
