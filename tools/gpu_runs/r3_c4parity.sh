#!/bin/bash
# builder run: BASELINE configs[3] at FULL size (10M cells) against the oracle in both arithmetic modes (~15 minutes, mostly CPU)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
free -g | head -2
HMX_SLOW=1 timeout 2700 python -m pytest tests/test_gpu_parity2.py -q -m gpu -s -k "config4_10M" 2>&1 | tail -8 | cut -c1-3500 | tee $O/c4_parity.log
