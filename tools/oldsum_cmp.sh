# compare the three oldsum implementations (rocprofv3 kernel stats) -- run on the GPU box
exec </dev/null
cd /tmp && export TMPDIR=/tmp
for v in stream stream1 gather; do
  HMX_OLDSUM_IMPL=$v timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_os_$v -o b -- python $GRAFT_REPO_ROOT/tools/prof_update.py ${1:-1000000} > /dev/null 2>&1
  echo "$v: $(grep oldsum $GRAFT_REPO_ROOT/gpurun_out/prof_os_$v/b_kernel_stats.csv | cut -d, -f1-4)"
done
