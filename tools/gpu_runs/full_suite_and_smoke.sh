#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -22 | tee gpurun_out/full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
