#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -k "kmeans or synthetic_shapes or cell_lines_small_full or full_size or fallback or envelope" 2>&1 | tail -6 ) > gpurun_out/c13_tests.log 2>&1
tail -3 gpurun_out/c13_tests.log
( timeout 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e ) > gpurun_out/c13_bench.json 2> gpurun_out/c13_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/c13_bench.json'))
print(d['ms_per_step'], d['config']['harmony_iterations'][:2]); print(d['config']['gpu_phase_ms_per_step']); print(d['roofline']['avg_block_step_us'])
PY
