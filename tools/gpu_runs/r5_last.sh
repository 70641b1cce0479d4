#!/bin/bash
# round 5, last call: the final tree (after the split of hmx_kernels.hip) through a parity subset + smoke, and the two-rank configs[4]-shape bench line
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
O=$R/gpurun_out/r5fin; mkdir -p $O
[ "$1" == "bench" ] || timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -3
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --gpus 2 --backend gloo --workload c5 --cells-per-gpu 500000 --steps 2 --warmup 1 --no-e2e > $O/bench_2ranks_c5.json 2> $O/bench_2ranks_c5.err
tail -1 $O/bench_2ranks_c5.json | cut -c1-300; grep -i "inconsistent\|error" $O/bench_2ranks_c5.err | head -3
timeout 300 python $R/bench.py --cpu-sample 0 --no-e2e --also none > $O/bench_final_tree.json 2>/dev/null; tail -1 $O/bench_final_tree.json | cut -c1-200
