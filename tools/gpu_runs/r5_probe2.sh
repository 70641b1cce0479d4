#!/bin/bash
# round 5: pass-count / tolerance settings of the restarted sums against the faithful oracle at two sizes; seq machinery tests; timeline
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=gpurun_out/${2:-r5d}; mkdir -p $O
S=${1:-passes2tol10000,passes2tol1000,passes3tol1000,passes3tol240}
timeout 600 python -m pytest tests/test_gpu_seq.py -q -m gpu --durations=4 2>&1 | tail -12 | tee $O/seq_tests.log
for n in 100000 1000000; do
timeout 900 python tools/strict_probe.py --cells $n --settings "$S" > $O/strict_$n.json 2> $O/strict_$n.err; tail -2 $O/strict_$n.err
python - $O/strict_$n.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("cells", d["cells"])
for k, v in d["settings"].items():
    print(" ", k, {a: (round(b, 9) if isinstance(b, float) else b) for a, b in v.items() if a in ("seconds", "passes_per_group_oe_obj_ridge_pairs", "Z_rel", "R_maxabs", "objective_rel_max", "argmax_diff", "argmax_diff_margin_ge_1e-5", "O_rel")})
P
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/tools/ref_arith_profile.py --steps 1 --passes 2 > $R/$O/ref_profile.json 2> $R/$O/trace.err
cd $R
python tools/trace_gaps.py $O/trace/t_kernel_trace.csv > $O/ref_timeline.txt 2>&1; rm -rf $O/trace
head -16 $O/ref_timeline.txt | cut -c1-110
