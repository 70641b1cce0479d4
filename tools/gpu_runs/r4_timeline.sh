#!/bin/bash
# round 4: timeline of one clustering round (kernel trace) for a value of an environment switch (usage: r4_timeline.sh VAR "v1 v2")
exec </dev/null
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4tl; mkdir -p $O
VAR=${1:-HMX_SORT_SCHED}; VALS=${2:-"2 1"}
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
env $VAR=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -o t -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > /dev/null 2> $O/trace_$v.err
python $R/tools/trace_gaps.py $O/trace_$v/t_kernel_trace.csv > $O/round_timeline_$v.txt 2>&1; rm -rf $O/trace_$v
echo "== $VAR=$v"; head -12 $O/round_timeline_$v.txt; tail -12 $O/round_timeline_$v.txt
done
