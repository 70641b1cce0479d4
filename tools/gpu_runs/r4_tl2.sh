#!/bin/bash
exec </dev/null
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_run or 100k or virtual or synthetic" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/b$i.json 2> $O/b$i.err
python - <<PY
import json
j = json.loads(open("$O/b$i.json").read().strip().splitlines()[-1])
print("ms_per_step", round(j["ms_per_step"], 3), "step_us", round(j["roofline"]["avg_block_step_us"], 2), j["config"]["gpu_phase_ms_per_step"])
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > /dev/null 2> $R/$O/trace.err
python $R/tools/trace_gaps.py $R/$O/trace/t_kernel_trace.csv > $R/$O/round_timeline.txt 2>&1; rm -rf $R/$O/trace
head -3 $R/$O/round_timeline.txt; tail -7 $R/$O/round_timeline.txt
