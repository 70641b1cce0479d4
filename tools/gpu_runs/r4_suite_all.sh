#!/bin/bash
# round 4: the whole GPU suite (no -x), then the slow-marked builder run (configs[3] at its full 10M cells against the oracle), then smoke
exec </dev/null
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 | tee gpurun_out/full_suite.log
if [ "$1" = "slow" ]; then HMX_SLOW=1 timeout 1700 python -m pytest tests/test_gpu_parity2.py -q -m gpu -k "config4_10M" -s 2>&1 | tail -6 | tee gpurun_out/slow_10M.log; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
