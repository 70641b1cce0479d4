#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e > gpurun_out/c17_bench.json 2> gpurun_out/c17_bench.err; echo rc=$?
python - <<'PY'
import json
j = json.loads(open("gpurun_out/c17_bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["config"]["harmony_iterations"], j["roofline"]["avg_block_step_us"])
for k, v in j["config"]["chain_us_per_block_step"].items(): print("  ", k, v)
PY
