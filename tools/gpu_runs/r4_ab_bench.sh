#!/bin/bash
# round 4: bench-only A/B of an environment switch on one box (no tests; usage: r4_ab_bench.sh VAR "v1 v2 ..." [reps])
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4ab; mkdir -p $O
VAR=${1:-HMX_CHAIN_PRIO}; VALS=${2:-"0 1"}; REPS=${3:-2}
for i in $(seq 1 $REPS); do for v in $VALS; do
env $VAR=$v timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/b_${v}_$i.json 2> $O/b_${v}_$i.err
python - <<PY
import json
j = json.loads(open("$O/b_${v}_$i.json").read().strip().splitlines()[-1])
c = j["config"].get("chain_us_per_block_step") or {}
print("$VAR=$v ms_per_step", round(j["ms_per_step"], 3), "step_us", round(j["roofline"]["avg_block_step_us"], 2), "frac", round(j["roofline"]["frac"], 4), "wg0 busy", c.get("wg0_wave_busy_us"), "barrier", c.get("worker_barrier_arrive"), "mfma", c.get("worker_next_mfma"))
print("   wave0", {k[7:]: c[k] for k in c if k.startswith("worker_")})
print("   wave4", c.get("wg0_wave4_phases_us")); print("   wave5", c.get("wg0_wave5_phases_us"))
PY
done; done
