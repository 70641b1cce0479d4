#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python tools/kmeans_diag.py 100000 17 ) > gpurun_out/c2_kdiag.log 2>&1
( HMX_TILE_IMPL=v1 timeout 300 python tools/kmeans_diag.py 100000 17 ) > gpurun_out/c2_kdiag_v1.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_parity2.py -m gpu -q -k "not 1000000 and not config5 and not headline" 2>&1 | tail -40 ) > gpurun_out/c2_tests.log 2>&1
cat gpurun_out/c2_kdiag.log gpurun_out/c2_kdiag_v1.log; tail -30 gpurun_out/c2_tests.log
