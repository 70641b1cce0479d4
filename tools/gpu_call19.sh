#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for nrep in 8 4 2; do
HMX_NREP=$nrep timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e > gpurun_out/c19_bench_$nrep.json 2> gpurun_out/c19_bench.err; echo rc=$?
python - <<PY
import json
j = json.loads(open("gpurun_out/c19_bench_$nrep.json").read().strip().splitlines()[-1])
c = j["config"]["chain_us_per_block_step"]
print("nrep", $nrep, j["ms_per_step"], j["roofline"]["avg_block_step_us"], {k: c[k] for k in ("folder_wait_arrivals", "folder_fold", "folder_publish", "worker_flush_and_store_issue", "worker_barrier_arrive")}, c["wg0_wave_busy_us"])
PY
done
