#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/c12_tests.log 2>&1
tail -4 gpurun_out/c12_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/c12_smoke.log 2>&1; tail -2 gpurun_out/c12_smoke.log
( timeout 600 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --cells-per-gpu 200000 --cpu-sample 0 ) > gpurun_out/c12_bench_g2.json 2> gpurun_out/c12_bench_g2.err; echo "rc=$?"; head -c 700 gpurun_out/c12_bench_g2.json; tail -5 gpurun_out/c12_bench_g2.err
