#!/bin/bash
# round 3: quick look at one build -- the chain tests, then the bench line's chain clocks
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3bf; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dots.py -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_q$i.json 2> $O/bench_q$i.err; echo "bench rc=$?"
python - <<PY
import json
j = json.loads(open("gpurun_out/r3bf/bench_q$i.json").read().strip().splitlines()[-1])
print("ms_per_step", j["ms_per_step"], "iters", j["config"]["harmony_iterations"], "step_us", j["roofline"].get("avg_block_step_us"), "frac", j["roofline"]["frac"])
print("   phases", j["config"]["gpu_phase_ms_per_step"])
print("   chain", j["config"].get("chain_us_per_block_step"))
PY
done
