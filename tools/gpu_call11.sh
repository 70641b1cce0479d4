#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp



( timeout 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e ) > gpurun_out/c11_bench.json 2> gpurun_out/c11_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/c11_bench.json'))
print(d['ms_per_step'], d['config']['harmony_iterations'][:2]); print(d['config']['gpu_phase_ms_per_step']); print(d['roofline']['avg_block_step_us'])
PY
