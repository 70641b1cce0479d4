#!/bin/bash
# round 5: instruction / wait counters of the reference-arithmetic mode's kernels (one --pmc pass, --kernel-trace only)
exec </dev/null
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc -o p -- python $R/tools/ref_arith_profile.py --steps 1 > /dev/null 2> $O/pmc.err
cd $R
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_report.py $f k_seq k_obj > $O/ref_pmc.txt 2>&1; rm -rf $O/pmc; cat $O/ref_pmc.txt | cut -c1-400
