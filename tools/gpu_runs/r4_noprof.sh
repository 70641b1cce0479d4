#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4np; mkdir -p $O
for sched in 1 2; do for np in 0 1; do
HMX_SORT_SCHED=$sched HMX_BENCH_NOPROF=$np timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/b_${sched}_$np.json 2> $O/b_${sched}_$np.err
python - <<PY
import json
j = json.loads(open("$O/b_${sched}_$np.json").read().strip().splitlines()[-1])
print("sched=$sched noprof=$np ms_per_step", round(j["ms_per_step"], 3))
PY
done; done
