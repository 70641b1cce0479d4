#!/bin/bash
# round 5: the whole GPU suite + smoke + the default bench line (with its legs)
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=gpurun_out/${1:-r5e}; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -40 > $O/tests.log; tail -25 $O/tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - $O/bench_default.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "first_run", d.get("first_run_of_this_process_ms"), "ref", d.get("ms_per_step_reference_arith"))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_block_step_us", "avg_launch_us")}, d["roofline"]["achieved_weighted_by_variant"])
print("phases", d["config"]["gpu_phase_ms_per_step"])
for k, v in d.get("also", {}).items():
    print(k, {a: v.get(a) for a in ("ms_per_step", "harmony_iterations", "avg_block_step_us")}, v.get("gpu_phase_ms_per_step"), v.get("error"))
P
