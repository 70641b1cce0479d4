#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
for i in 1 2 3; do
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_o.json 2> $O/bench_o.err
  python - <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r3/bench_o.json").read().strip().splitlines()[-1])
p = j["config"]["gpu_phase_ms_per_step"]
print("ms_per_step %.3f" % j["ms_per_step"], "block step %.2f" % j["roofline"]["avg_block_step_us"], "randomize", p["randomize"], "chain", p["Rcells_update"], "head", p["cluster_head"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace_bench.err
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $O/trace/t_kernel_trace.csv > $O/trace_gaps.txt 2>&1; head -3 $O/trace_gaps.txt; tail -9 $O/trace_gaps.txt
rm -rf $O/trace
