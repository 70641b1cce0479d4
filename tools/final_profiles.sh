#!/bin/bash
# The round's evidence for profiles/ (TAG=r6 by default: TAG=r7 bash tools/final_profiles.sh ... in the next round): default bench (plain, with its extra legs; and under rocprofv3 kernel stats), PMC passes (separate,
# --kernel-trace only), the round timeline, the reference-arithmetic mode's kernel table, configs[4] at full size, the two-rank
# protocol runs on this one GPU.  Writes gpurun_out/${TAG}fin/; tools/pmc_summary.py $TAG turns it into profiles/${TAG}_*.
# PART=a: bench lines + kernel stats + timeline + PMC;  PART=b: reference arithmetic, configs[4] shape, 10M, two ranks.
exec </dev/null
TAG=${TAG:-r6}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}fin; mkdir -p $O
PART=${1:-ab}
cd /tmp && export TMPDIR=/tmp
if [[ $PART == *c* ]]; then     # full-size oracle tables of the two 8-GPU configs (CPU-bound: 15-20 minutes of oracle threads), in the background of the other parts
( cd $R && HMX_SLOW=1 timeout 3000 python -m pytest tests/test_gpu_parity2.py -q -m gpu -k "10M_against or 5M_against" > $O/slow_tables.log 2>&1 ) &
SLOWPID=$!
sleep 90      # (their GPU runs happen in the first minute: keep them out of the timed benches)
fi
if [[ $PART == *s* ]]; then     # the whole GPU suite + smoke, as the driver runs them
( cd $R && timeout 1800 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -20 > $O/suite.log; timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/suite.log ); tail -4 $O/suite.log | cut -c1-200
fi
if [[ $PART == *a* ]]; then
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python $R/bench.py --pmc --also none --cpu-sample 0 --no-e2e > $O/bench_pmc_selfcollected.json 2> $O/bench_pmc_selfcollected.err; cp $R/profiles/pmc_traffic_update_kernel.json $O/pmc_traffic_from_bench_pmc.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --cpu-sample 0 --no-e2e --also none > $O/bench_rocprof.json 2> $O/bench_rocprof.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > /dev/null 2> $O/trace.err
python $R/tools/trace_gaps.py $O/trace/t_kernel_trace.csv > $O/round_timeline.txt 2>&1; rm -rf $O/trace
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/tools/prof_update.py 1000000 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/tools/prof_update.py 1000000 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/tools/prof_update.py 1000000 > $O/pmc_sq.log 2>&1
cd $R
for p in fetch write sq; do f=$O/pmc_$p/p_counter_collection.csv; test -f $f && python tools/pmc_report.py $f "k_tile" "k_copy" "k_shuf" "k_moe" > $O/pmc_$p.txt; rm -f $O/pmc_$p/p_kernel_trace.csv; rm -f $O/pmc_$p/*.db; done
cp $O/stats/b_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats
tail -1 $O/bench_default.json | cut -c1-300; head -16 $O/kernel_stats.csv | cut -c1-130; cat $O/pmc_fetch.txt $O/pmc_write.txt | cut -c1-300; cat $O/round_timeline.txt | tail -8
fi
if [[ $PART == *b* ]]; then
cd /tmp
timeout 300 python $R/tools/ref_arith_profile.py > $O/ref_profile.json 2> $O/ref_profile.err
timeout 300 python $R/tools/ref_arith_profile.py --c5 > $O/ref_profile_c5.json 2> $O/ref_profile_c5.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/refstats -o r -- python $R/tools/ref_arith_profile.py --steps 1 > /dev/null 2> $O/ref_rocprof.err
cp $O/refstats/r_kernel_stats.csv $O/ref_kernel_stats.csv 2>/dev/null; rm -rf $O/refstats
timeout 600 python $R/bench.py --cpu-sample 0 --workload c5 --cells-per-gpu 5000000 --steps 2 --warmup 1 --no-e2e --also none > $O/bench_c5_5M.json 2> $O/bench_c5_5M.err
timeout 400 python $R/bench.py --cpu-sample 0 --total-cells 10000000 --batches 20 --steps 3 --also none > $O/bench_strong_1gpu_10M.json 2> $O/bench_strong_1gpu_10M.err
timeout 600 python $R/bench.py --cpu-sample 0 --no-e2e --steps 2 --also shares > $O/bench_shares.json 2> $O/bench_shares.err
# two ranks sharing this box's one GPU (gloo for the remaining collectives): the in-launch exchange of the block chain vs one all-reduce per block
timeout 600 python $R/bench.py --gpus 2 --backend gloo --cells-per-gpu 500000 --steps 3 --warmup 1 --no-e2e > $O/bench_2ranks_p2p.json 2> $O/bench_2ranks_p2p.err
HMX_BENCH_P2P=0 timeout 600 python $R/bench.py --gpus 2 --backend gloo --cells-per-gpu 500000 --steps 3 --warmup 1 --no-e2e > $O/bench_2ranks_allreduce.json 2> $O/bench_2ranks_allreduce.err
# configs[4] shape on two ranks sharing the GPU: every block step's K x B table an inbox all-reduce of its own, ridge statistics / old sums as reduce-scatter + all-gather windows
timeout 600 python $R/bench.py --gpus 2 --backend gloo --workload c5 --cells-per-gpu 500000 --steps 2 --warmup 1 --no-e2e > $O/bench_2ranks_c5.json 2> $O/bench_2ranks_c5.err
# (the torch-free bootstrap -- bench.py --bootstrap file -- uses the built-in RCCL communicator, which wants one GPU per rank: covered at world 1 by
#  tests/test_gpu_parity2.py::test_bench_bootstraps_without_torch, at world 2 only on a node)
cd $R
tail -1 $O/bench_c5_5M.json | cut -c1-200; tail -1 $O/bench_strong_1gpu_10M.json | cut -c1-200; tail -1 $O/bench_2ranks_p2p.json | cut -c1-200; head -12 $O/ref_kernel_stats.csv | cut -c1-130
fi
if [[ $PART == *c* ]]; then
wait $SLOWPID; tail -5 $O/slow_tables.log | cut -c1-300
fi
