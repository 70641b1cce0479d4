#!/bin/bash
# round 5, session 1: new tests, the strictness probe of the reference-arithmetic mode at 1M cells, kernel timeline of one ref_arith run
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=gpurun_out/${2:-r5a}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "getLambda or elided or fixed_lambda or cell_lines_small_full or stand_alone" 2>&1 | tail -6 | tee $O/tests.log
timeout 900 python tools/strict_probe.py --settings "${1:-default,strict,passes6}" > $O/strict.json 2> $O/strict.err; tail -3 $O/strict.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/tools/ref_arith_profile.py --steps 1 > $R/$O/ref_profile.json 2> $R/$O/trace.err
cd $R
python tools/trace_gaps.py $O/trace/t_kernel_trace.csv > $O/ref_timeline.txt 2>&1; rm -rf $O/trace
head -40 $O/ref_timeline.txt
python - ${2:-r5a} <<'P'
import json, sys
d = json.load(open("gpurun_out/%s/strict.json" % (sys.argv[1] if len(sys.argv) > 1 else "r5a")))
for k, v in d["settings"].items():
    print(k, {a: b for a, b in v.items() if a != "largest"})
    if "largest" in v: print("   largest", {a: b for a, b in v["largest"].items() if "row" not in a})
P
