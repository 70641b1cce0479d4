#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "carried_old or every_cluster or unequal_shards or two_processes or arithmetic_gap_table and not 1000000" 2>&1 | tail -3
for w in 256 252 248 244 240; do
  HMX_CHAIN_WGS=$w timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_tw$w.json 2> $O/bench_tw$w.err
  python - $w <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r3/bench_tw%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("chain_wgs", sys.argv[1], "ms_per_step %.3f" % j["ms_per_step"], "block step %.2f" % j["roofline"]["avg_block_step_us"], j["config"]["harmony_iterations"], j["config"]["gpu_phase_ms_per_step"])
PY
done
