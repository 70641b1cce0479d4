#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -k "two_cov or synthetic_shapes or envelope or fixed_lambda or cell_lines_small_full or fallback or config5 or one_cluster" 2>&1 | tail -6 ) > gpurun_out/c10_tests.log 2>&1
tail -3 gpurun_out/c10_tests.log
( timeout 400 python bench.py --cpu-sample 0 --workload c5 --cells-per-gpu 1000000 --steps 3 --no-e2e ) > gpurun_out/c10_bench_c5.json 2> gpurun_out/c10_bench_c5.err
( HMX_CHAIN=0 timeout 400 python bench.py --cpu-sample 0 --cells-per-gpu 10000000 --batches 20 --steps 3 --no-e2e ) > gpurun_out/c10_bench_10M_nochain.json 2> gpurun_out/c10_bench_10M_nochain.err
python - <<PY
import json
for f in ('c10_bench_c5','c10_bench_10M_nochain'):
    d=json.load(open('gpurun_out/%s.json'%f))
    print(f, d['ms_per_step'], d['config']['harmony_iterations']); print(d['config']['gpu_phase_ms_per_step']); print(d['roofline']['avg_block_step_us'], d['roofline']['frac'])
PY
