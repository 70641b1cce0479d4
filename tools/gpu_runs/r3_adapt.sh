#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seq.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -12 | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_parity2.py -q -m gpu -k "arithmetic_gap_table" 2>&1 | tail -3
timeout 300 python tools/ref_arith_profile.py > $O/ref_profile.json 2> $O/ref_profile.err; tail -1 $O/ref_profile.json | cut -c1-900
