#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "carried_old or every_cluster or unequal_shards or two_processes or arithmetic_gap_table and not 1000000" 2>&1 | tail -3
for off in 0 1; do
  if [ $off = 1 ]; then export HMX_CHAIN_TAIL_OFF=1; else unset HMX_CHAIN_TAIL_OFF; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_tail$off.json 2> $O/bench_tail$off.err
  python - $off <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r3/bench_tail%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("tail_off", sys.argv[1], "ms_per_step %.3f" % j["ms_per_step"], "block step %.2f" % j["roofline"]["avg_block_step_us"], j["config"]["harmony_iterations"], j["config"]["gpu_phase_ms_per_step"])
PY
done
