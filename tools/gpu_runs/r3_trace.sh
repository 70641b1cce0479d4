#!/bin/bash
# round 3: kernel-trace timeline of the default bench (where do the microseconds between the kernels go?) + the new two-process test
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --also none > $GRAFT_REPO_ROOT/$O/trace_bench.json 2> $GRAFT_REPO_ROOT/$O/trace_bench.err
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $O/trace/t_kernel_trace.csv > $O/trace_gaps.txt 2>&1; head -70 $O/trace_gaps.txt
rm -rf $O/trace
timeout 600 python -m pytest tests/test_gpu_parity2.py -q -m gpu -k "unequal_shards" 2>&1 | tail -3
