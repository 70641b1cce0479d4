#!/bin/bash
# round 5: the shuffle-equivalence test a few times over (it failed once on its 1e-9 objective tolerance) and the default line with the reference-arithmetic leg on the final code
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=$R/gpurun_out/r5fin; mkdir -p $O
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity2.py -q -m gpu -k "sort_free_shuffle" 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --also ref --cpu-sample 0 --no-e2e > $O/bench_final_ref.json 2>/dev/null
python - $O/bench_final_ref.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "step_us", d["roofline"]["avg_block_step_us"], "frac", d["roofline"]["frac"], "ref ms", d.get("ms_per_step_reference_arith"), d.get("value_reference_arith"))
print(d["also"]["reference_arith"]["gpu_phase_ms_per_step"])
P
