#!/bin/bash
# round 5: kernel table + phase table of the reference-arithmetic mode on the final tree
exec </dev/null
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5fin; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/tools/ref_arith_profile.py > $O/ref_profile.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/refstats -o r -- python $R/tools/ref_arith_profile.py --steps 1 > /dev/null 2>&1
cp $O/refstats/r_kernel_stats.csv $O/ref_kernel_stats.csv 2>/dev/null; rm -rf $O/refstats
cut -c1-300 $O/ref_profile.json; head -9 $O/ref_kernel_stats.csv | cut -c1-50,140-260
