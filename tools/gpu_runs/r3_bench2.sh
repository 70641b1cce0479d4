#!/bin/bash
# round 3 (split-bf16 build): the default bench line with its extra legs, then the whole -m gpu suite + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b2; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?; tail -3 $O/bench_default.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3b2/bench_default.json").read().strip().splitlines()[-1])
print("ms_per_step", j["ms_per_step"], "value", j["value"], "iters", j["config"]["harmony_iterations"])
print("roofline", {k: j["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_us", "avg_block_step_us")}, j["roofline"]["run"]["frac"])
print("phases", j["config"]["gpu_phase_ms_per_step"])
print("e2e", j["config"].get("e2e"))
for k, v in (j.get("also") or {}).items():
    print("also", k, {a: v.get(a) for a in ("ms_per_step", "cells_per_s", "harmony_iterations", "avg_block_step_us", "roofline_run_frac", "error")})
    print("     ", v.get("gpu_phase_ms_per_step"))
cb = j.get("cpu_baseline") or {}
print("cpu", cb.get("value"), cb.get("full_size"), cb.get("gpu_reference_arith_vs_this_run"))
PY
timeout 1800 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -14 | tee $O/full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
