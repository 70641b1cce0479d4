#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_parity2.py -m gpu -q -k "two_processes or torch_free or 1000000 or config5" 2>&1 | tail -30 ) > gpurun_out/c9_tests.log 2>&1
tail -12 gpurun_out/c9_tests.log
