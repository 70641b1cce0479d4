#!/bin/bash
# round 6, first call: the new tests of the ADVICE / bench items, the reference-arithmetic baseline on this box, an ATT attempt
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -k "never_reads_elided or elided_R_stores or restarted_sums or reference_arithmetic_fixture" 2>&1 | tail -5 > gpurun_out/r6_a_tests.txt
HMX_BENCH_PREROLL=4 timeout 600 python bench.py --also ref --cpu-sample 0 --no-e2e --steps 5 > gpurun_out/r6_a_bench.json 2> gpurun_out/r6_a_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --att --kernel-include-regex "k_tile<7, 5" -d $R/gpurun_out/r6_att -- python $R/tools/prof_update.py 1000000 100 10 > $R/gpurun_out/r6_att_attempt.txt 2>&1
ls -R $R/gpurun_out/r6_att 2>/dev/null | head -30 >> $R/gpurun_out/r6_att_attempt.txt
cd $R
timeout 1500 python -m pytest tests -q -m gpu -k "two_ranks_default_is_configs3" -s 2>&1 | tail -8 > gpurun_out/r6_a_bench2.txt
cat gpurun_out/r6_a_tests.txt gpurun_out/r6_a_bench2.txt; tail -c 600 gpurun_out/r6_att_attempt.txt
