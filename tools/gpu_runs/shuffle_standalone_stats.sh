#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HMX_SORT_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --cpu-sample 0 --no-e2e --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
grep -E "k_sort|fillBuffer|k_round_tail|k_oldsum" $O/stats/b_kernel_stats.csv | cut -c1-160
python - <<PY
import json
j = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["config"]["gpu_phase_ms_per_step"])
PY
rm -rf $O/stats
