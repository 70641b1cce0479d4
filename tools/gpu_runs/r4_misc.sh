#!/bin/bash
# round 4: configs[4] shape with / without the round-to-round carry (dead R stores), e2e first call, current round timeline
exec </dev/null
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seq.py -x -q -m gpu 2>&1 | tail -3
for c in 0 1; do
HMX_SOLD_CARRY=$c timeout 300 python bench.py --workload c5 --steps 3 --warmup 1 --cpu-sample 0 --no-e2e --also none > $O/c5_carry$c.json 2> $O/c5_carry$c.err
python - <<PY
import json
j = json.loads(open("$O/c5_carry$c.json").read().strip().splitlines()[-1])
print("c5 1M carry=$c ms_per_step", round(j["ms_per_step"], 2), "iters", j["config"]["harmony_iterations"], "step_us", round(j["roofline"]["avg_block_step_us"], 1))
print("   phases", j["config"]["gpu_phase_ms_per_step"])
PY
done
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --also none > $O/e2e.json 2> $O/e2e.err
python - <<PY
import json
j = json.loads(open("$O/e2e.json").read().strip().splitlines()[-1])
print("default ms_per_step", round(j["ms_per_step"], 3), "e2e", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in j["config"]["e2e"].items() if k != "note"})
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > /dev/null 2> $R/$O/trace.err
python $R/tools/trace_gaps.py $R/$O/trace/t_kernel_trace.csv > $R/$O/round_timeline.txt 2>&1; rm -rf $R/$O/trace
head -10 $R/$O/round_timeline.txt; tail -8 $R/$O/round_timeline.txt
