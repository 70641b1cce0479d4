#!/bin/bash
# round 5: quick A/B of the reference-arithmetic mode under two values of an environment switch (usage: r5_ab.sh VAR "v1 v2"): machinery tests, 100k probe, profile runs
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
VAR=${1:-HMX_NONE}; VALS=${2:-"0"}
timeout 600 python -m pytest tests/test_gpu_seq.py -q -m gpu 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "stand_alone" 2>&1 | tail -1
timeout 300 python tools/strict_probe.py --cells 100000 --settings default 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['settings'].items(): print(k,{a:v[a] for a in ('seconds','Z_rel','R_maxabs','objective_rel_max','argmax_diff_margin_ge_1e-5')})"
for w in $VALS; do
env $VAR=$w timeout 300 python tools/ref_arith_profile.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$w', round(d['ms_per_run_incl_egress'],1), d['gpu_phase_ms_per_run'], d['iterations'])"
done
