#!/bin/bash
# rocprofv3 kernel table + HBM counters of configs[4]'s shape at 1M cells on the wave-pair chain (gpurun -- 'bash tools/gpu_runs/pair_chain_profile.sh'): gpurun_out/r6_c5_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5prof; mkdir -p $O
export PC_PAIR=1 PC_CELLS=1000000
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c -- python $R/tools/gpu_runs/pair_chain_check.py > $O/run.txt 2>&1
cp $O/stats/c_kernel_stats.csv $R/gpurun_out/r6_c5_shape_kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/tools/gpu_runs/pair_chain_check.py > /dev/null 2>&1
  python $R/tools/pmc_report.py $O/pmc_$c/p_counter_collection.csv "k_tile<7, 6" "k_oldsum" "k_moe" > $R/gpurun_out/r6_c5_shape_pmc_$c.txt 2>&1
done
rm -rf $O
head -8 $R/gpurun_out/r6_c5_shape_kernel_stats.csv | cut -c1-150; cat $R/gpurun_out/r6_c5_shape_pmc_FETCH_SIZE.txt $R/gpurun_out/r6_c5_shape_pmc_WRITE_SIZE.txt | cut -c1-200
