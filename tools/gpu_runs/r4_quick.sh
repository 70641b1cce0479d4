#!/bin/bash
# round 4: A/B driver -- parity subset + default bench lines with an environment switch (usage: r4_quick.sh VAR "v1 v2 ..." ["extra bench.py arguments"])
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4q; mkdir -p $O
VAR=${1:-HMX_R_STORE}; VALS=${2:-"0 1"}; XARGS=${3:-}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "carried_old or every_cluster or full_size_named or chain_with_three or kmeans" 2>&1 | tail -3
for v in $VALS; do
for i in 1 2; do
env $VAR=$v timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none $XARGS > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err; echo "bench $VAR=$v rc=$?"
python - <<PY
import json
j = json.loads(open("$O/bench_${v}_$i.json").read().strip().splitlines()[-1])
print("$VAR=$v ms_per_step", round(j["ms_per_step"], 3), "iters", j["config"]["harmony_iterations"][:3], "step_us", round(j["roofline"].get("avg_block_step_us"), 2), "frac", round(j["roofline"]["frac"], 4))
print("   phases", j["config"]["gpu_phase_ms_per_step"])
c = j["config"].get("chain_us_per_block_step") or {}
print("   chain", {k: c[k] for k in c if not k.startswith("wg")})
PY
done
done
