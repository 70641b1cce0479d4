#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "carried_old" 2>&1 | tail -25 | tee gpurun_out/c24_tests.log
