#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "virtual_shards" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "two_processes or full_size_named" 2>&1 | tail -8
timeout 400 python bench.py --cpu-sample 0 --cells-per-gpu 10000000 --batches 20 --steps 3 --no-e2e > gpurun_out/c26_10M.json 2> gpurun_out/c26_10M.err; echo rc=$?
python - <<'PY'
import json
j = json.loads(open("gpurun_out/c26_10M.json").read().strip().splitlines()[-1])
print("10M:", j["ms_per_step"], j["config"]["harmony_iterations"], j["roofline"]["avg_block_step_us"])
print(j["config"]["gpu_phase_ms_per_step"])
PY
