#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity2.py -q -m gpu -k "unequal_shards" 2>&1 | tail -3
timeout 1700 python -m pytest tests/test_gpu_parity2.py -q -m gpu -s -k "config5_shape_1M" 2>&1 | tail -12 | cut -c1-3000 | tee $O/c5_parity.log
