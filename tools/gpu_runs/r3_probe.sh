#!/bin/bash
# round 3: chain step after hoisting the pair loads; then the probe (diagnostics build) with parts of the step switched off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3bf; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dots.py tests/test_gpu_parity2.py -x -q -k "not 1M and not 10M and not config5 and not gap_table and not full_size" 2>&1 | tail -4 | tee $O/parity2.log
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_hoist.json 2> $O/bench_hoist.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3bf/bench_hoist.json").read().strip().splitlines()[-1])
print("ms_per_step", j["ms_per_step"], "iters", j["config"]["harmony_iterations"], "step_us", j["roofline"].get("avg_block_step_us"), "frac", j["roofline"]["frac"])
print("   phases", j["config"]["gpu_phase_ms_per_step"])
print("   chain", j["config"].get("chain_us_per_block_step"))
PY
HMX_LIB_PATH=harmony_amd/lib/libharmony_mi355x_trace.so timeout 300 python tools/chain_probe.py 0 8 4 12 2>&1 | tee $O/probe.log
