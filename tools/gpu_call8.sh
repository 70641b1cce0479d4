#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -k "not 1000000 and not config5" 2>&1 | tail -8 ) > gpurun_out/c8_tests.log 2>&1
tail -5 gpurun_out/c8_tests.log

python - <<PY
import sys; sys.exit(0)
import json
d=json.load(open('gpurun_out/c8_bench.json'))
print(d['ms_per_step'], d['value'], d['config']['harmony_iterations'][:2]); print(d['config']['gpu_phase_ms_per_step']); print(d['roofline']['avg_block_step_us'], d['roofline']['frac'], d['roofline']['run']['frac']); print(d['config']['e2e']); print(d['cpu_baseline'])
PY
