#!/bin/bash
# round 4: reference-arithmetic tests, then where the mode spends its time (phase table + rocprof kernel table)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4ref; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seq.py -q -m gpu -s --durations=5 -x 2>&1 | grep -v "^$" | tail -30 | tee $O/seq_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stand_alone or pbmc or cell_lines_small_full or two_covariates or synthetic_shapes or 100k" 2>&1 | tail -5
timeout 300 python tools/ref_arith_profile.py > $O/ref_profile.json 2> $O/ref_profile.err; tail -1 $O/ref_profile.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/refstats -o r -- python $GRAFT_REPO_ROOT/tools/ref_arith_profile.py --steps 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/ref_rocprof.err
cd $GRAFT_REPO_ROOT
cp $O/refstats/r_kernel_stats.csv $O/ref_kernel_stats.csv 2>/dev/null; rm -rf $O/refstats
head -22 $O/ref_kernel_stats.csv | cut -c1-70,150-260
