#!/bin/bash
# round 6: the whole GPU suite (the driver's command) + smoke, output kept
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > gpurun_out/r6_suite.txt 2>&1
python __graft_entry__.py smoke >> gpurun_out/r6_suite.txt 2>&1
tail -25 gpurun_out/r6_suite.txt
