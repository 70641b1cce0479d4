#!/bin/bash
# round 5, closing check of the final tree: the 1M gap table (both arithmetic modes against both oracles + the liberty row), the default line with its reference-arithmetic leg
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=$R/gpurun_out/r5fin; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity2.py -q -m gpu -k "gap_table and 1000000" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --also ref --cpu-sample 0 --no-e2e > $O/bench_final_ref.json 2>/dev/null
python - $O/bench_final_ref.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "step_us", d["roofline"]["avg_block_step_us"], "frac", d["roofline"]["frac"], "ref ms", d.get("ms_per_step_reference_arith"), d.get("value_reference_arith"))
print(d["also"]["reference_arith"]["gpu_phase_ms_per_step"])
P
