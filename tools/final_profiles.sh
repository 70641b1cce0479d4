# Final evidence for profiles/: default bench under rocprofv3 kernel stats, PMC passes (separate, --kernel-trace only), 10M bench.
exec </dev/null
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py > $O/bench_rocprof.json 2> $O/bench_rocprof.err
timeout 200 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python $R/bench.py --cpu-sample 0 --cells-per-gpu 10000000 --steps 3 > $O/bench_10M.json 2> $O/bench_10M.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/tools/prof_update.py 1000000 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/tools/prof_update.py 1000000 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/tools/prof_update.py 1000000 > /dev/null 2>&1
cd $R
for p in fetch write sq; do f=$O/pmc_$p/p_counter_collection.csv; test -f $f && python tools/pmc_report.py $f "k_tile" "k_copy" "k_oldsum" > $O/pmc_$p.txt; rm -f $O/pmc_$p/p_kernel_trace.csv; done
rm -f $O/stats/b_kernel_trace.csv
tail -1 $O/bench_default.json | cut -c1-400; tail -1 $O/bench_10M.json | cut -c1-200; head -8 $O/stats/b_kernel_stats.csv | cut -c1-110; cat $O/pmc_fetch.txt $O/pmc_write.txt $O/pmc_sq.txt | cut -c1-400
