#!/bin/bash
# round 5: reference-arithmetic machinery tests, the probe against the faithful oracle at 1M cells (given settings), kernel timeline of one run
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=gpurun_out/${2:-r5c}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seq.py -q -m gpu -x --durations=6 2>&1 | tail -14 | tee $O/seq_tests.log
timeout 900 python tools/strict_probe.py --settings "${1:-default}" > $O/strict.json 2> $O/strict.err; tail -3 $O/strict.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/tools/ref_arith_profile.py --steps 1 --passes 2 > $R/$O/ref_profile.json 2> $R/$O/trace.err
cd $R
python tools/trace_gaps.py $O/trace/t_kernel_trace.csv > $O/ref_timeline.txt 2>&1; rm -rf $O/trace
head -34 $O/ref_timeline.txt | cut -c1-110; sed -n '/reference arithmetic: round/,/first 70/p' $O/ref_timeline.txt | head -14
python - ${2:-r5c} <<'P'
import json, sys
d = json.load(open("gpurun_out/%s/strict.json" % sys.argv[1]))
for k, v in d["settings"].items():
    print(k, {a: b for a, b in v.items() if a not in ("largest", "dR_by_cluster_top5", "cells_with_dR_above")})
P
cat $O/ref_profile.json | cut -c1-400
