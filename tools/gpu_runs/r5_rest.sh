#!/bin/bash
# round 5: the second half of the GPU suite (gap tables, two-process runs, reference-arithmetic machinery), no -x
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=gpurun_out/${1:-r5f}; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_seq.py -q -m gpu --durations=10 2>&1 | tail -60 > $O/tests.log; tail -45 $O/tests.log | cut -c1-1500
timeout 300 python tools/ref_arith_profile.py --passes 2 > $O/ref_profile.json 2>/dev/null; cat $O/ref_profile.json | cut -c1-600
