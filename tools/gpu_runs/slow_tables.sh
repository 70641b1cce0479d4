#!/bin/bash
# the two 8-GPU configs at FULL size on one GPU against the CPU oracle (HMX_SLOW=1: ~15-20 minutes of oracle each), tables -> gpurun_out/r6_parity_c4_10M.json, r6_parity_c5_5M.json
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
HMX_SLOW=1 timeout 5400 python -m pytest tests/test_gpu_parity2.py -q -s -k "config4_10M_against or config5_5M_against" 2>&1 | tail -12 | cut -c1-3000 > gpurun_out/slow_tables.txt
tail -5 gpurun_out/slow_tables.txt | cut -c1-600
