#!/bin/bash
# Round-3 evidence for profiles/ on the final code (split-bf16 build of the tile kernels): the default bench line with its extra legs,
# the same under rocprofv3 kernel stats, the round timeline, PMC passes (separate, --kernel-trace only).  Writes gpurun_out/r3fin/;
# tools/pmc_summary_r3.py turns it into profiles/r3_*.  (tools/final_profiles_r3.sh is the longer mid-round version: reference-arithmetic
# profile, configs[4] at 5M, strong-scaling line, two-rank protocol runs.)
exec </dev/null
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3fin; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --cpu-sample 0 --no-e2e --also none > $O/bench_rocprof.json 2> $O/bench_rocprof.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > /dev/null 2> $O/trace.err
python $R/tools/trace_gaps.py $O/trace/t_kernel_trace.csv > $O/round_timeline.txt 2>&1; rm -rf $O/trace
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/tools/prof_update.py 1000000 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/tools/prof_update.py 1000000 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/tools/prof_update.py 1000000 > $O/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python $R/tools/prof_update.py 1000000 > $O/pmc_sq2.log 2>&1
cd $R
for p in fetch write sq sq2; do f=$O/pmc_$p/p_counter_collection.csv; test -f $f && python tools/pmc_report.py $f "k_tile" "k_copy" "k_sort" "k_moe" > $O/pmc_$p.txt; rm -f $O/pmc_$p/p_kernel_trace.csv; rm -f $O/pmc_$p/*.db; done
cp $O/stats/b_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats
tail -1 $O/bench_default.json | cut -c1-300; head -14 $O/kernel_stats.csv | cut -c1-130; cat $O/pmc_fetch.txt $O/pmc_write.txt | cut -c1-300; cat $O/round_timeline.txt | tail -8
