#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synthetic_shapes or 100k or full_size or cell_lines_small_full or two_cov or pbmc or envelope or fallback" 2>&1 | tail -6 ) > gpurun_out/c5_tests.log 2>&1
tail -3 gpurun_out/c5_tests.log
for o in 0; do
( HMX_CHAIN_OLD=$o timeout 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e ) > gpurun_out/c5_bench_o$o.json 2> gpurun_out/c5_bench_o$o.err
python - <<PY
import json
d=json.load(open('gpurun_out/c5_bench_o$o.json'))
print("chain_old $o:", d['ms_per_step'], d['config']['harmony_iterations'][:2]); print(d['config']['gpu_phase_ms_per_step']); print(d['config']['chain_us_per_block_step']); print(d['roofline']['avg_block_step_us'])
PY
done
