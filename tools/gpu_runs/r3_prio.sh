#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
for pr in low normal high; do for w in 256 248; do
  HMX_SIDE_PRIO=$pr HMX_CHAIN_WGS=$w timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_p.json 2> $O/bench_p.err
  python - $pr $w <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r3/bench_p.json").read().strip().splitlines()[-1])
p = j["config"]["gpu_phase_ms_per_step"]
print("side prio", sys.argv[1], "chain_wgs", sys.argv[2], "ms_per_step %.3f" % j["ms_per_step"], "block step %.2f" % j["roofline"]["avg_block_step_us"], "randomize", p["randomize"], "chain", p["Rcells_update"])
PY
done; done
