#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synthetic_shapes or 100k or full_size or cell_lines_small_full or two_cov or pbmc" 2>&1 | tail -15 ) > gpurun_out/c3_tests.log 2>&1
( timeout 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e ) > gpurun_out/c3_bench_chain.json 2> gpurun_out/c3_bench_chain.err
( HMX_CHAIN=0 timeout 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e ) > gpurun_out/c3_bench_nochain.json 2> gpurun_out/c3_bench_nochain.err
tail -8 gpurun_out/c3_tests.log; cat gpurun_out/c3_bench_chain.json | head -c 1500; echo; cat gpurun_out/c3_bench_nochain.json | head -c 700; tail -3 gpurun_out/c3_bench_chain.err
