#!/bin/bash
# parity suite + the carry tests + one default bench line (about 40 s of box time)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "carried_old or every_cluster or full_size_named" 2>&1 | tail -2
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err; echo rc=$?
python - <<'PY'
import json
j = json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["config"]["harmony_iterations"], j["roofline"]["avg_block_step_us"])
print(j["config"]["gpu_phase_ms_per_step"])
PY
