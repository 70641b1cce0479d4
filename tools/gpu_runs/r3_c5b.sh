#!/bin/bash
# round 3: configs[4] shape (K = 200 -> 13 cluster tiles) after the NCT = 13 instantiation
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_dots.py -x -q -k "tile_shape or config5_shape_200k or two_covariates or fixed_lambda or synthetic_shapes or envelope_shapes or virtual_shards or fallback_paths or pbmc or fp32_build" 2>&1 | tail -3
timeout 600 python bench.py --workload c5 --cells-per-gpu 1000000 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > $O/bench_c5_1M.json 2> $O/bench_c5_1M.err; echo rc=$?
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3c5/bench_c5_1M.json").read().strip().splitlines()[-1])
print("ms_per_step", j["ms_per_step"], "iters", j["config"]["harmony_iterations"], "step_us", j["roofline"].get("avg_block_step_us"), j["roofline"]["kernel"][:40], "frac", j["roofline"]["frac"], j["roofline"]["run"]["frac"])
print("   phases", j["config"]["gpu_phase_ms_per_step"])
PY
