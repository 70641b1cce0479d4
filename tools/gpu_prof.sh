#!/bin/bash
# rocprofv3 kernel stats of a short default bench run -> gpurun_out/prof_<tag>/
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-x}
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.err
cd $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; find . -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} ../prof_${TAG}_kernel_stats.csv
find . -name "*_kernel_trace.csv" -delete; find . -name "*.db" -delete
python - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_kernel_stats.csv')))
for r in rows[:22]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), r['Percentage'].rjust(7))
PY
