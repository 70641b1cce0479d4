#!/bin/bash
# round 3: unit tests of the restarted sums, the reference-arithmetic end-to-end tests, then where the mode spends its time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seq.py -q -m gpu -s --durations=5 2>&1 | grep -v "^$" | tail -45 | tee $O/seq_tests.log
timeout 300 python tools/ref_arith_profile.py > $O/ref_profile.json 2> $O/ref_profile.err; tail -1 $O/ref_profile.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/refstats -o r -- python $GRAFT_REPO_ROOT/tools/ref_arith_profile.py --steps 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/ref_rocprof.err
cd $GRAFT_REPO_ROOT
cp $O/refstats/r_kernel_stats.csv $O/ref_kernel_stats.csv 2>/dev/null; rm -rf $O/refstats
head -25 $O/ref_kernel_stats.csv | cut -c1-160
