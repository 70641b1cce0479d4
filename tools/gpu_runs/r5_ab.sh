#!/bin/bash
# round 5: quick A/B of the reference-arithmetic mode (seq machinery tests, 100k probe, profile runs under two values of a switch)
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=$R/gpurun_out/${1:-r5j}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_seq.py -q -m gpu 2>&1 | tail -3
timeout 300 python tools/strict_probe.py --cells 100000 --settings default 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['settings'].items(): print(k,{a:v[a] for a in ('seconds','Z_rel','R_maxabs','objective_rel_max','argmax_diff_margin_ge_1e-5','passes_per_group_oe_obj_ridge_pairs')})"
for w in 16 12; do
HMX_SEQ_RIDGE_WPG=$w timeout 300 python tools/ref_arith_profile.py > $O/ref_profile_wpg$w.json 2>/dev/null; echo "wpg=$w"; cut -c1-420 $O/ref_profile_wpg$w.json
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/refstats -o r -- python $R/tools/ref_arith_profile.py --steps 1 > /dev/null 2>&1
cp $O/refstats/r_kernel_stats.csv $O/ref_kernel_stats.csv 2>/dev/null; rm -rf $O/refstats; head -12 $O/ref_kernel_stats.csv | cut -c1-60,150-260
