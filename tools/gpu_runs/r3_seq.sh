#!/bin/bash
# round 3: the reference-arithmetic tests first (fail fast), then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_gpu_seq.py -q -m gpu -s --durations=8 2>&1 | tail -60 | tee gpurun_out/r3/seq_tests.log
timeout 1800 python -m pytest tests -q -m gpu --durations=8 --deselect tests/test_gpu_seq.py 2>&1 | tail -40 | tee gpurun_out/r3/full_suite.log
